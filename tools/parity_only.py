import sys, json, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
out = bench.chain_parity(dev)
for m in ('fp32', 'fp16', 'bf16'):
    print(m, {k: out[m][k] for k in ('latents_rel_err', 'cube_rel_err', 'dPSNR_dB', 'dSAM_deg', 'meets_north_star', 'n_fixtures')})
    for k, v in out[m]['fixtures'].items():
        if 'chi' in k: print('   ', k, v)
