"""Drop-in for the reference's model/sr3_modules package: exposes `unet.UNet` and
`diffusion.GaussianDiffusion` (the plugin seam of model/networks.py:85-109)."""
from . import diffusion, unet  # noqa: F401
