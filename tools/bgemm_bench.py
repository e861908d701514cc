#!/usr/bin/env python3
"""Times the five strided batched GEMMs of one attention backward (hsidm_bgemm) at the training batch:  python tools/bgemm_bench.py [B] [N] [C]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsi_dmgasr_amd import train_ops as T  # noqa: E402

B, N, C = (int(a) for a in (sys.argv[1:4] + ["4", "256", "512"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
C3 = 3 * C
qkv = torch.randn(B, N, C3, device=dev).bfloat16()
do = torch.randn(B, N, C, device=dev).bfloat16()
P = torch.randn(B, N, N, device=dev)
dP = torch.randn(B, N, N, device=dev)
dqkv = torch.empty_like(qkv)
es = 2
q, k, v = qkv.data_ptr(), qkv.data_ptr() + C * es, qkv.data_ptr() + 2 * C * es
dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + C * es, dqkv.data_ptr() + 2 * C * es
f32 = False
cases = {
    "S=QK^T": lambda: T._gemm(q, f32, N * C3, C3, 1, k, f32, N * C3, 1, C3, P.data_ptr(), True, N * N, N, N, N, C, B, 0.1),
    "dP=dO V^T": lambda: T._gemm(do.data_ptr(), f32, N * C, C, 1, v, f32, N * C3, 1, C3, dP.data_ptr(), True, N * N, N, N, N, C, B, 1.0),
    "dQ=dS K": lambda: T._gemm(dP.data_ptr(), True, N * N, N, 1, k, f32, N * C3, C3, 1, dq, f32, N * C3, C3, N, C, N, B, 1.0),
    "dK=dS^T Q": lambda: T._gemm(dP.data_ptr(), True, N * N, 1, N, q, f32, N * C3, C3, 1, dk, f32, N * C3, C3, N, C, N, B, 1.0),
    "dV=P^T dO": lambda: T._gemm(P.data_ptr(), True, N * N, 1, N, do.data_ptr(), f32, N * C, C, 1, dv, f32, N * C3, C3, N, C, N, B, 1.0),
}
for name, fn in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-12s %7.1f us" % (name, e0.elapsed_time(e1) / 20 * 1e3))
