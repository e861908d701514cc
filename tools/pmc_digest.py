#!/usr/bin/env python3
"""One line per tools/pmc_conv.sh directory: instructions per matrix instruction, matrix-pipe busy fraction, wave wait fraction.

    python tools/pmc_digest.py profiles/r06_pmc/pmc_*        (reads pass1.txt / pass2.txt: the per-kernel means tools/pmc_summary.py printed)
matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES / 32 shader engines)  (tools/attn_util.py, MI355X_MICROARCH.md)."""
import os
import re
import sys

for d in sys.argv[1:]:
    m = {}
    for f in ("pass1.txt", "pass2.txt", "pass3.txt"):
        p = os.path.join(d, f)
        if os.path.exists(p):
            for line in open(p):
                g = re.match(r"\s+(SQ_\w+)\s+([\d.]+)", line)
                if g and g.group(1) not in m:
                    m[g.group(1)] = float(g.group(2))
    if "SQ_INSTS_MFMA" not in m:
        continue
    mf = m["SQ_INSTS_MFMA"]
    print("%-40s VALU %.2f SALU %.2f LDS %.2f VMEM_RD %.2f per MFMA | matrix pipe busy %.3f | waves waiting on an instruction %.0f %% | LDS bank conflicts %.1f %%" % (
        os.path.basename(d.rstrip("/")), m["SQ_INSTS_VALU"] / mf, m["SQ_INSTS_SALU"] / mf, m["SQ_INSTS_LDS"] / mf, m.get("SQ_INSTS_VMEM_RD", 0) / mf,
        m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * m["SQ_BUSY_CYCLES"] / 32.0), 100 * m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
        100 * m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1)))
