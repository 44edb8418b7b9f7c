from .renderer import Renderer, render, render_rays  # noqa: F401
