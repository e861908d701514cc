#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -S listing, per basic block, weighted view of where the vector-issue slots go.

    hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -S --cuda-device-only -o k.s hsi-dmgasr_amd/csrc/conv_v3.hip
    python tools/isa_hist.py k.s 'conv_v3_kernelILi2ELb0EDF16_Li2ELb1ELb0E' [--blocks]

Classes: mfma (v_mfma / v_smfmac), trans (v_exp / v_rcp / v_rsq / v_log / v_sqrt: 8 issue cycles), valu (other v_*: 4), salu, lds (ds_*),
vmem (global_ / buffer_ / flat_), scratch, wait (s_waitcnt), nop.  --blocks lists every basic block with >= 8 instructions."""
import collections
import re
import sys


def classify(op):
    if op.startswith(("v_mfma", "v_smfmac")):
        return "mfma"
    if re.match(r"v_(exp|rcp|rsq|log|sqrt|sin|cos)_", op):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    show = "--blocks" in sys.argv
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and pat in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\ts_endpgm") or lines[i].startswith(".Lfunc_end"))
    blocks, cur, name = [], collections.Counter(), "entry"
    ops = collections.Counter()
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            blocks.append((name, cur))
            cur, name = collections.Counter(), m.group(1)
            continue
        m = re.match(r"^\t([a-z_0-9]+)", l)
        if not m or l.startswith("\t."):
            continue
        op = m.group(1)
        cur[classify(op)] += 1
        if classify(op) == "valu":
            ops[re.sub(r"_e(32|64)$", "", op)] += 1
    blocks.append((name, cur))
    tot = collections.Counter()
    for _, c in blocks:
        tot.update(c)
    keys = ["mfma", "valu", "trans", "salu", "lds", "vmem", "scratch", "wait", "nop", "other"]
    print("kernel: %s  (%d instructions, %d blocks)" % (lines[start][:-1][:120], sum(tot.values()), len(blocks)))
    print("  total  " + "  ".join("%s %d" % (k, tot[k]) for k in keys))
    if tot["mfma"]:
        print("  per mfma: valu %.2f  trans %.2f  salu %.2f  lds %.2f  vmem %.2f" % tuple(tot[k] / tot["mfma"] for k in ("valu", "trans", "salu", "lds", "vmem")))
    print("  top valu ops: " + ", ".join("%s %d" % kv for kv in ops.most_common(24)))
    if show:
        for n, c in blocks:
            if sum(c.values()) >= 8:
                print("  %-14s " % n + "  ".join("%s %d" % (k, c[k]) for k in keys if c[k]))


if __name__ == "__main__":
    main()
