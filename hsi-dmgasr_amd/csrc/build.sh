#!/bin/bash
# Build libhsidm.so (gfx950 only) in-tree: one hipcc per translation unit in parallel, then link.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../libhsidm.so"
OBJ="$HERE/obj"
mkdir -p "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -fno-slp-vectorize: the SLP pass packs adjacent fp32 mul/add into v_pk_*_f32, which co-issues badly beside MFMAs
# (measured: the 3x3 convs with Cin >= 256 run 6-8 % faster without it; MI355X_MICROARCH.md, filler price table)
FLAGS="--offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -Wall -Wno-unused-variable -Wno-unused-but-set-variable ${HSIDM_EXTRA_FLAGS:-}"
OUT="${HSIDM_OUT:-$OUT}"
OBJ="${HSIDM_OBJ:-$OBJ}"
mkdir -p "$OBJ"
JOBS="${JOBS:-6}"
pids=()
objs=()
for src in "$HERE"/*.hip; do
  o="$OBJ/$(basename "${src%.hip}").o"
  objs+=("$o")            # link exactly the objects of the sources present (a stale object of a removed file would still link)
  if [ ! -f "$o" ] || [ "$src" -nt "$o" ] || [ -n "$(find "$HERE" -maxdepth 1 \( -name '*.h' -o -name '*.inc' \) -newer "$o")" ] || [ "$HERE/../../include/hsidm.h" -nt "$o" ]; then
    ( $HIPCC $FLAGS -c "$src" -o "$o" ) &
    pids+=($!)
    while [ "$(jobs -rp | wc -l)" -ge "$JOBS" ]; do sleep 0.2; done
  fi
done
fail=0
for p in "${pids[@]}"; do wait "$p" || fail=1; done
[ $fail -eq 0 ] || { echo "hsidm build: compile failed" >&2; exit 1; }
$HIPCC --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$OUT"
echo "built $OUT"
