#!/bin/bash
# In-kernel timeline of conv_v3 on the 128x128 64 -> 64 layer (s_memtime stamps, diagnostic build).  Run on the GPU box from the repo root.
HSIDM_EXTRA_FLAGS="-DHSIDM_V2_STAMPS" HSIDM_OUT=$PWD/tools/_libstamps.so HSIDM_OBJ=/tmp/obj_stamps bash hsi-dmgasr_amd/csrc/build.sh > /dev/null 2>&1
HSIDM_LIB=$PWD/tools/_libstamps.so HSIDM_PROBE_V3=1 HSIDM_PROBE_BATCH=${BATCH:-240} python tools/stamp_probe.py ${1:-l128_64_64} 2>&1 | grep -v amdgpu
rm -f tools/_libstamps.so
