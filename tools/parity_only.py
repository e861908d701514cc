"""Parity walk only (bench.chain_parity over the nine reference chains), one line per fixture:
   python tools/parity_only.py [mode ...]          (default: fp16; environment switches of precision.py apply, e.g. HSIDM_FULL_STEP_GAIN)"""
import sys
sys.path.insert(0, '.')
import torch
import bench
dev = torch.device('cuda:0')
modes = tuple(sys.argv[1:]) or ("fp16",)
out = bench.chain_parity(dev, modes=modes, long_modes=tuple(m for m in modes if m != "bf16"))
keys = ('latents_rel_err', 'latents_rel_err_unsaturated', 'cube_rel_err', 'dSAM_deg', 'dSAM_unclamped_deg')
for m in modes:
    print(m, "WORST", {k: out[m].get(k) for k in keys})
    for k, v in out[m]['fixtures'].items():
        print('   ', k, {kk: (round(v[kk], 7) if isinstance(v.get(kk), float) else v.get(kk)) for kk in keys})
