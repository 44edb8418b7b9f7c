"""MI355X-native mirror of human_diffusion/improved_diffusion (sampling path)."""
