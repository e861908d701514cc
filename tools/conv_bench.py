#!/usr/bin/env python3
"""Micro-benchmark of representative hsidm_conv2d launches (bf16, batch 40) for kernel tuning.

    python tools/conv_bench.py [--reps 5] [--only NAME]
Prints one line per shape: name, microseconds (best of reps), TFLOP/s.  Used under rocprofv3 --pmc.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsi_dmgasr_amd import ops  # noqa: E402

SHAPES = [  # name, H, C0, C1, Cout, ksize, stride, ups, proj_cin
    ("l128_64_64", 128, 64, 0, 64, 3, 1, 0, 0),
    ("l128_192_64", 128, 128, 64, 64, 3, 1, 0, 0),
    ("l128_64_64_proj128", 128, 64, 0, 64, 3, 1, 0, 128),
    ("l64_128_128", 64, 128, 0, 128, 3, 1, 0, 0),
    ("l64_384_128", 64, 256, 128, 128, 3, 1, 0, 0),
    ("l32_256_256", 32, 256, 0, 256, 3, 1, 0, 0),
    ("l32_768_256", 32, 512, 256, 256, 3, 1, 0, 0),
    ("l16_512_512", 16, 512, 0, 512, 3, 1, 0, 0),
    ("l16_1024_512", 16, 512, 512, 512, 3, 1, 0, 0),
    ("l8_512_512", 8, 512, 0, 512, 3, 1, 0, 0),
    ("l8_1024_512", 8, 512, 512, 512, 3, 1, 0, 0),
    ("up_128_to128", 64, 128, 0, 128, 3, 1, 1, 0),
    ("up_256_to64", 32, 256, 0, 256, 3, 1, 1, 0),
    ("up_512_to32", 16, 512, 0, 512, 3, 1, 1, 0),
    ("up_512_to16", 8, 512, 0, 512, 3, 1, 1, 0),
    ("down_128_64", 128, 64, 0, 64, 3, 2, 0, 0),
    ("down_64_128", 64, 128, 0, 128, 3, 2, 0, 0),
    ("down_32_256", 32, 256, 0, 256, 3, 2, 0, 0),
    ("down_16_512", 16, 512, 0, 512, 3, 2, 0, 0),
    ("qkv_16_512", 16, 512, 0, 1536, 1, 1, 0, 0),
    ("stem_128_8_64", 128, 8, 0, 64, 3, 1, 0, 0),
    ("proj128_192_64", 128, 128, 64, 64, 1, 1, 0, 0),
    ("proj64_384_128", 64, 256, 128, 128, 1, 1, 0, 0),
    ("proj32_768_256", 32, 512, 256, 256, 1, 1, 0, 0),
    ("proj16_1024_512", 16, 512, 512, 512, 1, 1, 0, 0),
    ("proj8_1024_512", 8, 512, 512, 512, 1, 1, 0, 0),
    ("out16_512_512", 16, 512, 0, 512, 1, 1, 0, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=40)
    ap.add_argument("--only", default=None)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--sustain", type=float, default=0.0, help="after the timed repetitions keep launching the shape back to back for this many "
                                                               "seconds and report the sustained time per launch (power / clock A/B: tools/power_ab.sh)")
    ap.add_argument("--v1", action="store_true", help="force the v1 kernel (A/B against conv_v2)")
    ap.add_argument("--no-xf", action="store_true", help="3x3 convs without the GroupNorm+SiLU input transform (cost of the fused prologue)")
    ap.add_argument("--no-fold", action="store_true", help="upsample convs with HSIDM_UPS_ADDRESS instead of the parity-folded kernels")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ops.set_use_v2(not args.v1)
    prec = args.precision
    from hsi_dmgasr_amd import _lib
    dt = _lib.act_dtype(prec)
    g = torch.Generator(device="cpu").manual_seed(0)
    B = args.batch
    for name, H, C0, C1, Co, ks, st, up, pj in SHAPES:
        if args.only and args.only != name:
            continue
        w = torch.randn(Co, C0 + C1, ks, ks, generator=g) / (9 * (C0 + C1)) ** 0.5
        pwt = torch.randn(Co, pj, 1, 1, generator=g) / pj ** 0.5 if pj else None
        pk = ops.PackedConv(w.to(dev), torch.zeros(Co, device=dev), prec, proj_weight=None if pwt is None else pwt.to(dev),
                            proj_bias=None if pwt is None else torch.zeros(Co, device=dev), fold_ups=bool(up) and not args.no_fold, fold_dn=(st == 2) and not args.no_fold)
        x0 = torch.randn(B, H, H, C0, generator=g).to(dev, dt)
        x1 = torch.randn(B, H, H, C1, generator=g).to(dev, dt) if C1 else None
        px = torch.randn(B, H, H, pj, generator=g).to(dev, dt) if pj else None
        ab = ops.gn_table(torch.stack([torch.ones(B, C0 + C1), torch.zeros(B, C0 + C1)], dim=2).contiguous().to(dev))
        xf = ops.XF_AFFINE_SILU if (ks == 3 and st == 1 and not up and not args.no_xf and C0 > 8) else ops.XF_NONE
        best = 1e9
        for _ in range(args.reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = ops.conv2d(x0, pk, x1=x1, gn_ab=ab if xf else None, transform=xf, stride=st, ups=bool(up), proj_x0=px)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        Ho = y.shape[1]
        flops = 2.0 * B * Ho * Ho * Co * ((C0 + C1) * ks * ks + pj)
        print("%-22s %9.1f us %8.1f TFLOP/s (reference FLOP count)" % (name, best * 1e3, flops / (best * 1e-3) / 1e12), flush=True)
        if args.sustain > 0:
            import time
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < args.sustain:
                for _ in range(50):
                    ops.conv2d(x0, pk, x1=x1, gn_ab=ab if xf else None, transform=xf, stride=st, ups=bool(up), proj_x0=px)
                torch.cuda.synchronize()
                n += 50
            dt_s = time.perf_counter() - t0
            print("%-22s sustained %9.1f us per launch over %.1f s = %8.1f TFLOP/s" % (name, dt_s / n * 1e6, dt_s, flops * n / dt_s / 1e12), flush=True)
        del x0, x1, px, y, pk


if __name__ == "__main__":
    main()
