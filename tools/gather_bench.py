#!/usr/bin/env python3
"""hsidm_gather_pack (the training step's re-pack) at the shipped UNet's size with a contiguous index map and with a map that takes 8
elements 36 bytes apart per vector (a [cout][cin][3][3] weight gathered along cin): the reason the trainer keeps its 3x3 weights in
channels-last memory order.  Run on the GPU box from the repo root:  python tools/gather_bench.py"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from hsi_dmgasr_amd import train_ops as T
dev = torch.device("cuda:0")
n_src, n_out = 97_800_000, 196_000_000 // 8 * 8
src = torch.randn(n_src, device=dev)
out = torch.empty(n_out, dtype=torch.bfloat16, device=dev)
def bench(idx, name):
    for _ in range(2): T.gather_pack(src, idx, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): T.gather_pack(src, idx, out)
    e1.record(); torch.cuda.synchronize()
    print(name, "%.1f us" % (e0.elapsed_time(e1) / 5 * 1e3))
# contiguous runs of 8
idx = (torch.arange(n_out, device=dev, dtype=torch.int64) % n_src).to(torch.int32)
bench(idx, "contiguous")
# stride-9 pattern like [cout][cin][3][3] gathered along cin for a fixed tap: 8 sources 9 apart, lanes along... emulate
base = torch.arange(n_out // 8, device=dev, dtype=torch.int64)
j = torch.arange(8, device=dev, dtype=torch.int64)
idx9 = ((base[:, None] * 72 + j[None, :] * 9) % (n_src - 80)).reshape(-1).to(torch.int32)
bench(idx9, "stride 9 within a vector, vectors 72 apart")
