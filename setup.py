"""Installable form of the package: the sources live in `hsi-dmgasr_amd/` (the directory name the build contract prescribes, not a
valid Python identifier) and install as `hsi_dmgasr_amd`; the in-tree import goes through the alias package `hsi_dmgasr_amd/`.

    python -c "import __graft_entry__ as g; g.build()"     # builds hsi-dmgasr_amd/libhsidm.so for gfx950 first
    pip install --no-build-isolation .                       # or: python setup.py build_py -d DIR
"""
from setuptools import setup

setup(
    name="hsi-dmgasr-amd",
    version="0.3.0",
    description="MI355X-native (gfx950) denoising hot path of HSI-DMGASR: SR3 UNet sampler + group autoencoder on hand-written HIP kernels",
    packages=["hsi_dmgasr_amd", "hsi_dmgasr_amd.sr3_modules"],
    package_dir={"hsi_dmgasr_amd": "hsi-dmgasr_amd"},
    package_data={"hsi_dmgasr_amd": ["libhsidm.so"]},
    python_requires=">=3.10",
)
