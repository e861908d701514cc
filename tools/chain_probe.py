#!/usr/bin/env python3
"""Deviation of precision modes on members of the chain fixture set (GPU box):  python tools/chain_probe.py fp16x1,fp16 orth:2:1000 synth:0:20 ..."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_gpu_chain as tc  # noqa: E402

dev = torch.device("cuda:0")
for prec in sys.argv[1].split(","):
    for spec in sys.argv[2:]:
        w, d, t = spec.split(":")
        e_lat, e_y, dpsnr, dsam = tc._run_chain(dev, prec, (w, int(d), int(t)))
        r = tc.LAST_CHAIN_RECORD
        print(json.dumps(dict(precision=prec, fixture=spec, latents=e_lat, cube=e_y, dPSNR_dB=dpsnr, dSAM_deg=r["dSAM_deg"],
                              dSAM_common_support_deg=r["dSAM_common_support_deg"], zero_spectrum_crossings=r["zero_spectrum_crossings"])), flush=True)
