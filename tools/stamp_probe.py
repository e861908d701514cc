#!/usr/bin/env python3
"""In-kernel timeline of conv_v2 (diagnostic build with -DHSIDM_V2_STAMPS -> libhsidm_stamps.so).

    HSIDM_LIB=.../libhsidm_stamps.so python tools/stamp_probe.py SHAPE_NAME
Prints, per item index, the median cycles of: chunk loop, barrier waits, epilogue; over all blocks/waves.
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsi_dmgasr_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.environ.get("HSIDM_LIB", _lib.LIB_PATH)
from hsi_dmgasr_amd import ops  # noqa: E402
import tools.conv_bench as cb  # noqa: E402


def main():
    name = sys.argv[1]
    shp = [s for s in cb.SHAPES if s[0] == name][0]
    _, H, C0, C1, Co, ks, st, up, pj = shp
    dev = torch.device("cuda:0")
    B = int(os.environ.get("HSIDM_PROBE_BATCH", "40"))
    g = torch.Generator().manual_seed(0)
    w = torch.randn(Co, C0 + C1, ks, ks, generator=g) / (9 * (C0 + C1)) ** 0.5
    prec = os.environ.get("HSIDM_PROBE_PREC", "bf16")            # bf16 | fp16 (hi + lo weights where the policy says) | fp16x1
    adt = torch.bfloat16 if prec == "bf16" else torch.float16
    pjc = [int(v) for v in os.environ.get("HSIDM_PROBE_PROJ", "").split(",") if v]        # "128,64": a fused 1x1 projection of cat(128, 64) channels
    wp = torch.randn(Co, sum(pjc), 1, 1, generator=g) / max(sum(pjc), 1) ** 0.5 if pjc else None
    pk = ops.PackedConv(w.to(dev), torch.zeros(Co, device=dev), prec, proj_weight=None if wp is None else wp.to(dev))
    x0 = torch.randn(B, H, H, C0, generator=g).to(dev, adt)
    x1 = torch.randn(B, H, H, C1, generator=g).to(dev, adt) if C1 else None
    px = [torch.randn(B, H, H, c, generator=g).to(dev, adt) for c in pjc]
    ab = ops.gn_table(torch.stack([torch.ones(B, C0 + C1), torch.zeros(B, C0 + C1)], dim=2).contiguous().to(dev))
    xf = ops.XF_AFFINE_SILU if not up else ops.XF_NONE
    stamps = torch.zeros(512 * 4 * 8 * 16, dtype=torch.int64, device=dev)
    L = _lib.lib()
    L.hsidm_debug_set_stamps.argtypes = [ctypes.c_void_p]
    L.hsidm_debug_set_stamps.restype = None
    for rep in range(3):
        stamps.zero_()
        L.hsidm_debug_set_stamps(stamps.data_ptr())
        y = ops.conv2d(x0, pk, x1=x1, gn_ab=ab if xf else None, transform=xf, ups=bool(up), stats=True,
                       proj_x0=px[0] if pjc else None, proj_x1=px[1] if len(pjc) > 1 else None, fused_only=bool(pjc))
        assert y is not None
        torch.cuda.synchronize()
    L.hsidm_debug_set_stamps(None)
    s = stamps.cpu().numpy().reshape(512, 4, 8, 16).astype(np.float64)
    t0 = s[:, :, 0, 0].min()
    print("shape", name, "kernel span (cycles of the 100 MHz memtime clock x? ) first->last stamp:",
          (s[s > 0].max() - t0))
    for it in range(8):
        a = s[:, :, it, :]
        ok = a[:, :, 0] > 0
        if not ok.any():
            break
        start = a[:, :, 0][ok]
        ep0 = a[:, :, 12][ok]
        ep1 = a[:, :, 13][ok]
        pre = a[:, :, 10][ok]
        ch = [a[:, :, 1 + c][ok] for c in range(8) if (a[:, :, 1 + c][ok] > 0).all()]
        if os.environ.get("HSIDM_PROBE_V3"):
            e = lambda k: a[:, :, k][ok]
            print("item %d: commit=%6.0f issue=%6.0f bar1=%6.0f mfma=%6.0f bar2=%6.0f | epilogue=%6.0f bar3=%6.0f | total=%6.0f" % (
                it, np.median(e(1) - e(0)), np.median(e(2) - e(1)), np.median(e(3) - e(2)), np.median(e(4) - e(3)),
                np.median(e(5) - e(4)), np.median(e(14) - e(12)), np.median(e(13) - e(14)), np.median(e(13) - e(0))))
            if pjc:         # first projection chunk (slots 6..10) and whatever lies between it and the epilogue (further projection chunks)
                print("        proj chunk 1: commit=%6.0f issue=%6.0f bar1=%6.0f mfma=%6.0f bar2=%6.0f | further chunks=%6.0f" % (
                    np.median(e(6) - e(5)), np.median(e(7) - e(6)), np.median(e(8) - e(7)), np.median(e(9) - e(8)), np.median(e(10) - e(9)),
                    np.median(e(12) - e(10))))
            continue
        line = "item %d: n=%4d  start@%8.0f  loop=%7.0f  (last pre-barrier->barrier %6.0f)  epilogue=%7.0f" % (
            it, ok.sum(), np.median(start - t0), np.median(ep0 - start), np.median(ep0 - pre), np.median(ep1 - ep0))
        e = lambda k: a[:, :, k][ok]
        line += "  ep: scr=%5.0f vec=%5.0f stats=%5.0f barrier=%5.0f" % (np.median(e(9) - ep0), np.median(e(11) - e(9)),
                                                                          np.median(e(14) - e(11)), np.median(ep1 - e(14)))
        if len(ch) > 1:
            line += "  chunks=" + " ".join("%.0f" % np.median(ch[i] - (ch[i - 1] if i else start)) for i in range(len(ch)))
        print(line)


if __name__ == "__main__":
    main()
