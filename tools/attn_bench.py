#!/usr/bin/env python3
"""Time hsidm_attention alone:  python tools/attn_bench.py [--batch 120]   (HSIDM_ATTENTION_V1=1 for the panel kernel)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsi_dmgasr_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=120)
ap.add_argument("--reps", type=int, default=8)
args = ap.parse_args()
dev = torch.device("cuda:0")
for mode, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16), ("fp32", torch.float32)):
    for hw, C in ((16, 512), (8, 512)):
        qkv = torch.randn(args.batch, hw, hw, 3 * C, device=dev).to(dt)
        best = 1e9
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.attention(qkv, mode)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        fl = 4.0 * args.batch * (hw * hw) ** 2 * C
        print("attention %s N=%d C=%d batch %d: %.1f us  %.1f TFLOP/s = %.3f of the 2500 TFLOP/s dense peak" %
              (mode, hw * hw, C, args.batch, best * 1e3, fl / best / 1e9, fl / best / 1e9 / 2500.0))
