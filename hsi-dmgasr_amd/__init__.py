"""hsi_dmgasr_amd: the MI355X (gfx950) denoising hot path of HSI-DMGASR behind the reference's module interface.

This file is what an INSTALLED copy imports (setup.py maps this directory to the name ``hsi_dmgasr_amd``); inside the repository
the directory name carries a hyphen, and the alias package ``hsi_dmgasr_amd/`` at the repository root points the import system here.
"""
from .precision import get_default_precision, set_default_precision  # noqa: F401
