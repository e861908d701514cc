#!/usr/bin/env python3
"""A/B of the attention kernel's workgroup shapes at the benchmark's shape (240 images, N = 256, C = 512): HSIDM_ATTENTION_V1 switch
values 0 the dispatch's choice (one 8-wave workgroup per image from 128 images on: K and V read once), 2 two 4-wave workgroups per
image, 1 the score-panel kernel.  Prints us per launch (best of 8) and the deviation from torch fp32 on the same fp16 inputs.
(Round 6's A/B also had attention_v3 - 64 queries per wave pair - under switch values 0 / 4: profiles/r06_attention/ab_forms.txt.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsi_dmgasr_amd import _lib, ops, precision  # noqa: E402

precision.allow_experimental(True)
dev = torch.device("cuda:0")
B, hw, C = int(os.environ.get("BATCH", 240)), 16, 512
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B, hw, hw, 3 * C, generator=g) * 1.5).to(torch.float16).to(dev)
q, k, v = qkv[:4].float().reshape(4, hw * hw, 3, C).unbind(2)
ref = torch.softmax(q @ k.transpose(1, 2) / C ** 0.5, dim=-1) @ v
for att in (0, 3, 2, 0, 3, 2):
    with _lib.debug_switch("ATTENTION_V1", att):
        best = 1e9
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = ops.attention(qkv, "fp16x1")
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
    err = float((out[:4].float().reshape(4, hw * hw, C) - ref).norm() / ref.norm())
    fl = 4.0 * B * (hw * hw) ** 2 * C
    print("switch %d: %.1f us  %.1f TFLOP/s  rel err vs torch %.2e" % (att, best * 1e3, fl / best / 1e9, err), flush=True)
