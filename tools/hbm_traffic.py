#!/usr/bin/env python3
"""HBM bytes per launch of each conv kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tools/hbm_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv [BATCH] > profiles/hbm_traffic.json

BATCH = latents per GPU of the profiled bench.py run (recorded as "batch_per_gpu"; bench.py only attaches the figure to a run of the same batch).

Correction per MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced
read stream -> bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Keys are bench.py's kernel labels.
"""
import collections
import csv
import json
import re
import sys


def label(name):
    """Kernel name -> the label ops.conv2d's probe gives the same launches (bench.py groups by it)."""
    m = re.search(r"conv_v2_kernel.*V2Cfg<(\d+), 8, (\d+), (\d), (\d), (\d)[,>]", name) or \
        re.search(r"conv_v2_kernelINS_5V2CfgILi(\d+)ELi8ELi(\d+)ELi(\d)ELi(\d)ELi(\d)", name)
    if m:
        bn, tw, ni, xf, up4 = map(int, m.groups())
        return "conv_v2 bn%d %s k3 s1%s" % (bn, "8x8x2" if ni == 2 else ("8x16" if tw == 16 else "8x8"), " gn+silu" if xf == 2 else (" up4" if up4 == 1 else (" dn4" if up4 == 2 else "")))
    m = re.search(r"conv1x1_g_kernel<(\d+), (\d), (\d)[,>]", name) or re.search(r"conv1x1_g_kernelILi(\d+)ELi(\d)ELi(\d)", name)
    if m:
        bn, xf, im = map(int, m.groups())
        return "conv1x1_g bn%d 8x16 k%d s1%s" % (bn, 3 if im else 1, " gn" if xf == 1 else "")
    m = re.search(r"conv_igemm_kernelINS_7ConvCfgI(?:f|DF16b|DF16_)Lb[01]ELi(\d+)ELi\d+ELi8ELi(\d+)ELi(\d)ELi(\d)ELi(\d)ELb([01])", name)
    if m:       # the LDS-tiled kernel takes its input transform at run time: one label per geometry (bench.py strips the suffix)
        bn, tw, ni, ks, st, nchw = map(int, m.groups())
        return "conv_igemm bn%d %s k%d s%d%s" % (bn, "8x8x2" if ni == 2 else "8x16", ks, st, " nchw" if nchw else "")
    if "conv_v3_kernel" in name:
        nchw = "<1, true" in name or "ILi1ELb1E" in name
        # the projection forms (PROJ, last template argument): mangled "...Lb0ELb1EEE" / "...Lb1ELb1EEE" = <..., SPL, PROJ = true>
        proj = bool(re.search(r"Lb[01]ELb1EEEvNS_12ConvV2Params", name)) or bool(re.search(r"(true|false), true>\s*\(", name))
        return "conv_v3 bn32 8x16 k3 s1 gn+silu nchw" if nchw else ("conv_v3 bn64 8x16 k3 s1 gn+silu" + (" +proj" if proj else ""))
    return None


def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            lb = label(r["Kernel_Name"])
            if lb:
                acc[lb].append(float(r["Counter_Value"]))
    return acc


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else None
    out = {}
    for k in sorted(set(f) | set(w)):
        fk = sum(f[k]) / max(len(f[k]), 1)
        wk = sum(w[k]) / max(len(w[k]), 1)
        out[k] = {"hbm_bytes_per_launch": (2 * fk + wk) * 1024, "fetch_kb_raw": fk, "write_kb": wk,
                  "launches_sampled": len(f[k]), "batch_per_gpu": batch, "note": "mean over all launches of this kernel in the sampled steps; "
                  "FETCH_SIZE doubled (gfx950 half-count of wide coalesced reads)"}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
