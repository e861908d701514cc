import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsi_dmgasr_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(23)
for (B, H, W, Ci, Co) in ((1, 16, 16, 64, 3), (3, 32, 48, 64, 3)):
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    bias = torch.randn(Co, generator=g)
    pk = ops.PackedConv(w.to(dev), bias.to(dev), "bf16", out_nchw=True)
    x = torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16)
    ab = torch.stack([1 + 0.1 * torch.randn(B, Ci, generator=g), 0.1 * torch.randn(B, Ci, generator=g)], 2).contiguous()
    y = ops.conv2d(x, pk, gn_ab=ops.gn_table(ab.to(dev)), transform=ops.XF_AFFINE_SILU)
    torch.cuda.synchronize()
    xf = x.float().cpu()
    act = torch.nn.functional.silu(xf * ab[:, None, None, :, 0] + ab[:, None, None, :, 1]).to(torch.bfloat16).float()
    want = torch.nn.functional.conv2d(act.permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), bias, padding=1)
    e = (y.cpu() - want).norm() / want.norm()
    print((B, H, W), "rel err", float(e), flush=True)
    d = (y.cpu() - want).abs()
    print(" per-channel err", [float(d[:, c].max()) for c in range(Co)], "row err", [round(float(d[0, 0, r].max()), 3) for r in range(H)], flush=True)
