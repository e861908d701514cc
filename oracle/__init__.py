"""CPU oracle for the HSI-DMGASR denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / reported baseline.
The product path (``hsi-dmgasr_amd/``) never imports this package and fails
loudly when its HIP library is missing.

The oracle is a *functional* restatement: every function takes a plain
``state_dict`` (key names of the reference checkpoints) and
tensors, and evaluates the reference algorithm in fp32 PyTorch on the CPU.
Each function cites the reference file:line it follows (paths are relative to
the upstream repository root).

Parity pin: the reference ships no tests or golden vectors for this path
(SURVEY §4), so the oracle is pinned against outputs of the reference itself,
imported in the build container by ``tests/golden/make_golden.py``; the
resulting vectors are committed under ``tests/golden/`` and checked by
``tests/test_oracle_golden.py``.
"""
