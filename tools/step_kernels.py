"""N reverse steps of the benchmark workload on ONE kernel set, for `rocprofv3 --kernel-trace --stats`:
   rocprofv3 --kernel-trace --stats -d gpurun_out/x -- python3 tools/step_kernels.py fp32h 12 [patches]
(eager launches: HSIDM_NO_GRAPH is set here so that every launch carries its kernel name in the trace)"""
import os
import sys
os.environ["HSIDM_NO_GRAPH"] = "1"
sys.path.insert(0, '.')
import torch
import bench
from hsi_dmgasr_amd import precision

mode, n = sys.argv[1], int(sys.argv[2])
patches = int(sys.argv[3]) if len(sys.argv) > 3 else 48
precision.allow_experimental(True)
dev = torch.device("cuda:0")
gd = bench.build_model(dev, "fp16")
cond = torch.randn((patches * bench.GROUPS, 3, 128, 128), generator=torch.Generator().manual_seed(1)).clamp(-2.5, 2.5).to(dev)
run = gd.make_run(cond, wrap=True, precision=mode)
with torch.no_grad():
    for _ in range(n):
        run.step()
torch.cuda.synchronize()
print("ran", n, "steps of", mode, "modes:", sorted(set(run.modes)))
