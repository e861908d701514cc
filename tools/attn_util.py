#!/usr/bin/env python3
"""MFMA utilisation of the attention kernels from the SQ counter passes of tools/collect_r03.sh:

    python tools/attn_util.py profiles/r03_attention/attention_b240_pass1.csv [pass2.csv]

SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over the SIMDs (= 32 x the number of v_mfma_f32_32x32x16 issued,
MI355X_MICROARCH.md); SQ_BUSY_CYCLES is the kernel's duration in shader cycles summed over the chip's 32 shader engines.  The
fraction of SIMD cycles in which the matrix pipe is busy is therefore  MFMA_BUSY / (1024 SIMDs x SQ_BUSY_CYCLES / 32).
"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if "attention" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    cyc = m["SQ_BUSY_CYCLES"] / 32.0
    print(k)
    print("  kernel duration              %10.0f shader cycles" % cyc)
    print("  MFMA instructions            %10.0f  (matrix-pipe busy %.0f SIMD-cycles = 32 per instruction: %s)" %
          (m["SQ_INSTS_MFMA"], m["SQ_VALU_MFMA_BUSY_CYCLES"], abs(m["SQ_VALU_MFMA_BUSY_CYCLES"] - 32 * m["SQ_INSTS_MFMA"]) < 1))
    print("  MFMA pipe busy               %10.3f of the SIMD cycles (1024 SIMDs)" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)))
    print("  per MFMA: %.1f VALU, %.1f LDS instructions%s" % (m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"], m["SQ_INSTS_LDS"] / m["SQ_INSTS_MFMA"],
          (", %.1f SALU; LDS bank conflicts %.1f %% of LDS cycles" % (m["SQ_INSTS_SALU"] / m["SQ_INSTS_MFMA"], 100 * m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1))) if "SQ_INSTS_SALU" in m else ""))
    print("  waves waiting on an instruction %.0f %% of wave cycles" % (100 * m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
