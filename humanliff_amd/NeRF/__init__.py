from .renderer import Renderer, render, render_rays, render_view  # noqa: F401
