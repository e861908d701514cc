import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def _usable_cpus():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def pytest_configure(config):
    import torch
    torch.set_num_threads(min(_usable_cpus(), 32))     # GPU boxes expose 256 CPUs behind a small cgroup quota
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "public_modes: the test checks module-level behaviour in the public precision modes "
                                       "(tests/test_gpu_unet.py: no kernels_as_named() around it)")
    # The suite measures every kernel set, the EXPERIMENTAL ones (bf16, fp16x1, fp16x2) included - as regression gates, never as
    # north-star claims; the product refuses them by name unless asked (tests/test_host_logic.py checks the refusal).
    from hsi_dmgasr_amd import precision
    precision.allow_experimental(True)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
