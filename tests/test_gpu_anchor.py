"""Direct anchors: every tuned convolution kernel against a plain fp32 PyTorch reference of the SAME op (F.conv2d after the
GroupNorm affine map + SiLU, on the host), at benchmark-like shapes, in every precision mode - so that a regression is
localised to one kernel instead of surfacing three layers up in a chain of kernel-vs-kernel comparisons.

Inputs are taken as the kernel sees them (activations already rounded to the mode's storage type, the GroupNorm table as
fp32 pairs), so the measured error is the kernel's own: operand rounding after the transform, weight rounding, accumulation
order, the output store.  Bounds (relative Frobenius error): bf16 4e-3, fp16 6e-4 (one weight pass) / 4.5e-4 (hi + lo weights),
fp32 mode 1e-4 - about twice what the kernels measure (gpurun_out/parity.jsonl).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import assert_stats, check, log_err
from helpers import rel_err
from hsi_dmgasr_amd import _lib

pytestmark = pytest.mark.gpu

# ("fp32h": fp32 storage, the staged operand rounded once to fp16, hi + lo weights - the operand's rounding alone, ~2e-4)
TOL = {"bf16": 4e-3, "fp16x1": 6e-4, "fp16x2": 4.5e-4, "fp16": 6e-4, "fp32": 1e-4, "fp32h": 3.5e-4}
MODES = ["bf16", "fp16x1", "fp16x2", "fp32", "fp32h"]
# shapes the "fp32h" forms refuse (include/hsidm.h, HSIDM_F32H): the host runs them with the fp32 set's weights (ops.PackedConv.fallback)
F32H_FALLBACK = ("v2_8x8x2", "g1_qkv_gn")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _reference(x0, x1, ab, silu, w, b, film, res, stride=1, ups=False, ksize=3):
    """fp32 on the host: out = conv(T(cat(x0, x1))) + bias + film + res, NHWC in / NHWC out."""
    x = x0 if x1 is None else torch.cat([x0, x1], dim=3)
    x = x.float().cpu()
    if ab is not None:
        a = ab.float().cpu()
        x = x * a[:, None, None, :, 0] + a[:, None, None, :, 1]
        if silu:
            x = x * torch.sigmoid(x)
    x = x.permute(0, 3, 1, 2)
    if ups:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    y = F.conv2d(x, w.float().cpu(), None if b is None else b.float().cpu(), stride=stride, padding=ksize // 2)
    if film is not None:
        y = y + film.float().cpu()[:, :, None, None]
    y = y.permute(0, 2, 3, 1)
    if res is not None:
        y = y + res.float().cpu()
    return y.contiguous()


# name, B, H, W, C0, C1, Cout, ksize, stride, ups, GN+SiLU (2) / GN (1) / none (0), film, residual, expected kernel label
CASES = [
    ("v2_bn128_64x64", 4, 64, 64, 128, 0, 128, 3, 1, False, 2, True, False, "conv_v2 bn128 8x16"),
    ("v2_bn128_concat_res", 3, 32, 32, 128, 64, 128, 3, 1, False, 2, False, True, "conv_v2 bn128 8x16"),
    ("v2_bn256_32x32", 20, 32, 32, 256, 0, 256, 3, 1, False, 2, True, False, "conv_v2 bn256 8x16"),
    ("v2_bn256_res_16x16", 64, 16, 16, 256, 0, 512, 3, 1, False, 2, False, True, "conv_v2 bn256 8x16"),
    ("v2_8x8x2", 260, 8, 8, 128, 0, 256, 3, 1, False, 2, True, False, "conv_v2 bn256 8x8x2"),
    ("v3_64_64", 3, 32, 48, 64, 0, 64, 3, 1, False, 2, True, False, "conv_v3 bn64"),
    ("v3_concat_res", 2, 32, 32, 128, 64, 64, 3, 1, False, 2, False, True, "conv_v3 bn64"),
    ("up4_128", 3, 16, 16, 128, 0, 128, 3, 1, True, 0, False, False, "up4"),
    ("dn4_128", 3, 32, 32, 128, 0, 128, 3, 2, False, 0, False, False, "dn4"),
    ("dn4_64", 2, 64, 64, 64, 0, 64, 3, 2, False, 0, False, False, "dn4"),
    ("g1_proj_192_64", 2, 64, 64, 128, 64, 64, 1, 1, False, 0, False, False, "conv1x1_g"),
    ("g1_qkv_gn", 6, 16, 16, 512, 0, 1536, 1, 1, False, 1, False, False, "conv1x1_g"),
    ("g1_out_res", 6, 16, 16, 512, 0, 512, 1, 1, False, 0, False, True, "conv1x1_g"),
    ("stem_8_64", 2, 64, 64, 8, 0, 64, 3, 1, False, 0, False, False, "conv1x1_g"),
]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_kernel_against_fp32_torch(dev, mode, case):
    from hsi_dmgasr_amd import ops
    name, B, H, W, C0, C1, Co, ks, stride, ups, xf, with_film, with_res, label = case
    g = torch.Generator().manual_seed(B * 7 + C0 + Co + ks)
    cin = C0 + C1
    w = torch.randn(Co, cin, ks, ks, generator=g) / (ks * ks * cin) ** 0.5
    b = 0.1 * torch.randn(Co, generator=g)
    dt = _lib.act_dtype(mode)
    x0 = torch.randn(B, H, W, C0, generator=g).to(dt).to(dev)
    x1 = torch.randn(B, H, W, C1, generator=g).to(dt).to(dev) if C1 else None
    ab = torch.stack([1 + 0.1 * torch.randn(B, cin, generator=g), 0.2 * torch.randn(B, cin, generator=g)], 2).contiguous().to(dev) if xf else None
    Ho, Wo = (2 * H, 2 * W) if ups else ((H // 2, W // 2) if stride == 2 else (H, W))
    film = torch.randn(B, Co, generator=g).to(dev) if with_film else None
    res = torch.randn(B, Ho, Wo, Co, generator=g).to(dt).to(dev) if with_res else None
    pk = ops.PackedConv(w.to(dev), b.to(dev), mode, fold_ups=ups, fold_dn=stride == 2)
    recs = []
    ops.set_conv_probe(recs)
    try:
        y = ops.conv2d(x0, pk, x1=x1, gn_ab=None if ab is None else ops.gn_table(ab), transform=(ops.XF_NONE, ops.XF_AFFINE, ops.XF_AFFINE_SILU)[xf],
                       film=film, res=res, stride=stride, ups=ups, stats=True)
        torch.cuda.synchronize()
    finally:
        ops.set_conv_probe(None)
    got_label = recs[-1]["kernel"]
    if mode == "fp32h":
        assert pk.prec == _lib.F32H and (pk._fallback is not None) == (name in F32H_FALLBACK), (name, pk.prec, pk._fallback)
    if mode in ("fp32", "fp32h"):  # fp32 storage: the fp32 forms of the persistent 3x3 kernel (128- / 64-cout tiles) and of the 1x1 GEMM
        want = "conv_v2 bn64" if C0 + C1 == 8 else label.replace("bn256", "bn128").replace("conv_v3 bn64", "conv_v2 bn64")
    else:
        want = label.replace("bn256", "bn128") if mode == "fp16x2" else label      # hi + lo weights: no 256-cout items
    assert want in got_label, (want, got_label)
    slab, nsplit = y._hsidm_stats
    assert_stats(slab, y, name)
    want_y = _reference(x0, x1, ab, xf == 2, w, b, film, res, stride, ups, ks)
    check("anchor_" + name, mode, y, want_y, tol=1e-4 if (mode == "fp32h" and name in F32H_FALLBACK) else TOL[mode])
    if mode == "fp32h" and name not in F32H_FALLBACK:
        # ... and it IS the two-pass form: its distance from the fp32 set's result is the operand rounding (2^-12 rms), not zero
        y32 = ops.conv2d(x0, ops.PackedConv(w.to(dev), b.to(dev), "fp32", fold_ups=ups, fold_dn=stride == 2), x1=x1,
                         gn_ab=None if ab is None else ops.gn_table(ab), transform=(ops.XF_NONE, ops.XF_AFFINE, ops.XF_AFFINE_SILU)[xf],
                         film=film, res=res, stride=stride, ups=ups)
        d = rel_err(y.float().cpu().numpy(), y32.float().cpu().numpy())
        assert 2e-5 < d < TOL[mode], (name, d)


def test_low_weight_halves_reach_the_matrix_pipe(dev):
    """The hi + lo weight pass must deliver the low halves: they are fp16 SUBNORMALS (|w| ~ 0.02 -> w - fp16(w) ~ 5e-6 < 6.1e-5), so a
    matrix pipe that flushed subnormal operands would silently turn the second pass into nothing.  Activations exactly representable
    in fp16 and no transform: the only error left is the weights' - ~2^-12 with one pass, ~2^-18 with two."""
    from hsi_dmgasr_amd import ops
    g = torch.Generator().manual_seed(5)
    B, H, W, Ci, Co = 2, 32, 32, 64, 64
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    x = (torch.randint(0, 17, (B, H, W, Ci), generator=g).float() / 8).to(torch.float16).to(dev)     # mean 1: a weight bias survives the pixel average
    errs = {}
    for mode in ("fp16x1", "fp16x2"):
        pk = ops.PackedConv(w.to(dev), None, mode)
        y = ops.conv2d(x, pk)
        torch.cuda.synchronize()
        ref = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
        # compare before the fp16 store rounding matters: the store adds 2^-12 to both, so measure the MEAN error over 4096 outputs per
        # channel instead (rounding noise averages out, a weight bias does not)
        d = (y.double().cpu() - ref).mean(dim=(0, 1, 2)) / ref.abs().mean()
        errs[mode] = float(d.abs().max())
        log_err("low_weight_halves_" + mode, mode, errs[mode])
    assert errs["fp16x2"] < 0.25 * errs["fp16x1"] + 2e-6, errs


@pytest.mark.parametrize("shape", [(64, 64, 32, 32), (128, 128, 16, 32), (192, 64, 32, 32)], ids=["v3_64", "v2_128", "v3_192"])
def test_sparse_second_weight_pass_removes_the_weight_bias(dev, shape):
    """The GroupNorm + SiLU forms with the second weight pass on v_smfmac (conv_v3's 64-cout form, conv_v2's 128-cout form; low halves
    2:4-compressed, operand semantics measured by tools/ubench/smfmac_probe.hip) against the same convolution with the DENSE second pass
    (HSIDM_NO_SPARSE_LO switch) and with none: the three kernels stage identical operands, so their outputs differ by the weights alone.
    Per output channel, the MEAN over all pixels of (sparse - dense) - the contribution of the dropped smaller halves - must be well
    below that of (one pass - dense) - the contribution of ALL low halves.  A wrong index word, slot order or k-group mapping would make
    the sparse pass add an error of the low halves' full size instead - PROVIDED the channels differ in their mean activation: with
    one mean for all channels the shift sum_k mean_k * lo_k is invariant under any permutation of K, so every input channel gets its
    own scale here (1 + c % 7: different within every group of four, every 16-channel lane group and both 32-channel fragments)."""
    from hsi_dmgasr_amd import ops
    Ci, Co, H, W = shape
    g = torch.Generator().manual_seed(21)
    B = 4
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    x = (torch.rand(B, H, W, Ci, generator=g) * (1.0 + (torch.arange(Ci) % 7).float())).to(torch.float16)
    ab = ops.gn_table(torch.stack([torch.ones(B, Ci), torch.zeros(B, Ci)], dim=2).contiguous().to(dev))
    out = {}
    for tag, mode, dense in (("one", "fp16x1", 0), ("dense", "fp16x2", 1), ("sparse", "fp16x2", 0)):
        old = _lib.lib().hsidm_debug_switch(b"NO_SPARSE_LO", dense)
        try:
            recs = []
            ops.set_conv_probe(recs)
            y = ops.conv2d(x.to(dev), ops.PackedConv(w.to(dev), None, mode), gn_ab=ab, transform=ops.XF_AFFINE_SILU)
            torch.cuda.synchronize()
        finally:
            ops.set_conv_probe(None)
            _lib.lib().hsidm_debug_switch(b"NO_SPARSE_LO", old)
        assert ("conv_v3" if Co == 64 else "conv_v2 bn128") in recs[0]["kernel"], recs[0]["kernel"]
        out[tag] = y.double().cpu()
    scale = float(out["dense"].abs().mean())
    all_lo = float((out["one"] - out["dense"]).mean(dim=(0, 1, 2)).abs().mean()) / scale
    dropped = float((out["sparse"] - out["dense"]).mean(dim=(0, 1, 2)).abs().mean()) / scale
    log_err("sparse_lo_mean_shift_all_low_halves_%dx%d" % (Ci, Co), "fp16x1", all_lo)
    log_err("sparse_lo_mean_shift_dropped_halves_%dx%d" % (Ci, Co), "fp16x2", dropped)
    assert all_lo > 2e-5, all_lo                                # the low halves are visible in the channel means at all
    assert dropped < 0.6 * all_lo, (dropped, all_lo)
    assert rel_err(out["sparse"].numpy(), out["dense"].numpy()) < 3e-4
    # ... and the shift is the one a host model of the 2:4 rule predicts: per cout, the mean of conv(silu(x), the low halves the rule
    # drops - of every four consecutive input channels per cout and tap the two of smaller magnitude).  An operand-mapping error makes
    # the measured shift uncorrelated with the prediction (|got - pred| ~ 1.4 |pred|).
    lo = (w - w.to(torch.float16).float()).double()                                        # [Co][Ci][3][3]
    grp = lo.permute(0, 2, 3, 1).reshape(Co, 3, 3, Ci // 4, 4)
    keep = torch.zeros_like(grp).scatter_(4, grp.abs().topk(2, dim=4).indices, 1.0)
    lo_drop = (grp * (1.0 - keep)).reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2)
    act = F.silu(x.double()).permute(0, 3, 1, 2)
    pred = F.conv2d(act, lo_drop, padding=1).mean(dim=(0, 2, 3))                           # dense - sparse, per cout
    got = (out["dense"] - out["sparse"]).mean(dim=(0, 1, 2))
    miss = float((got - pred).abs().mean() / pred.abs().mean())
    log_err("sparse_lo_mean_shift_vs_host_model_%dx%d" % (Ci, Co), "fp16x2", miss)
    assert miss < 0.5, miss


@pytest.mark.parametrize("shape", [(64, 128, 64, 2, 32, 48), (64, 64, 0, 3, 16, 16), (64, 64, 64, 2, 32, 32), (128, 64, 8, 2, 32, 32)],
                         ids=["cat_192", "single_64", "cat_128", "ragged_72"])
def test_projection_fused_into_the_persistent_kernel(dev, shape):
    """ResnetBlock's tail, h2 = conv3x3(silu(GN(h))) + conv1x1(cat(x0, x1)) + biases (reference unet.py:105-111), on the
    64-cout layers: the projection as one-tap chunks of the 3x3 launch (conv_v3.hip, PROJ: raw staging, centre tap, projection
    weights scaled by log2(e) on the host, three static slots behind the 3-step weight ring, zero-padded steps) against fp32 torch on
    the host and against the two-launch form (1x1 GEMM, then the 3x3 kernel with its result as the residual) - in the ONE-PASS kernel
    sets the benchmark's chain steps run ("fp16d1": a dithered set of the fp16 policy; "fp16x1"; "bf16").  Layers with hi + lo weights
    (the experimental two-pass sets) are not offered the form: the caller keeps its two launches.  Projection widths: 192 = two tensors
    (three chunks), 64 (one), 128 = two tensors (two), and 72 (a last chunk of 8 live channels: the staging zeroes the rest)."""
    from hsi_dmgasr_amd import ops
    Ci, P0, P1, B, H, W = shape
    g = torch.Generator().manual_seed(Ci + P0 + P1)
    Co = 64
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    b = 0.1 * torch.randn(Co, generator=g)
    wp = torch.randn(Co, P0 + P1, 1, 1, generator=g) / (P0 + P1) ** 0.5
    bp = 0.1 * torch.randn(Co, generator=g)
    h16 = torch.randn(B, H, W, Ci, generator=g).to(torch.float16)
    x016 = torch.randn(B, H, W, P0, generator=g).to(torch.float16)
    x116 = torch.randn(B, H, W, P1, generator=g).to(torch.float16) if P1 else None
    ab = torch.stack([1 + 0.1 * torch.randn(B, Ci, generator=g), 0.2 * torch.randn(B, Ci, generator=g)], 2).contiguous().to(dev)
    tab = ops.gn_table(ab)
    for mode in ("fp16d1", "fp16x1", "bf16", "fp16", "fp16x2"):
        dt = torch.bfloat16 if mode == "bf16" else torch.float16
        h, x0 = h16.to(dt).to(dev), x016.to(dt).to(dev)
        x1 = None if x116 is None else x116.to(dt).to(dev)
        want = _reference(h, None, ab, True, w, b, None, None) + _reference(x0, x1, None, False, wp, bp, None, None, ksize=1)
        pk = ops.PackedConv(w.to(dev), b.to(dev), mode, proj_weight=wp.to(dev), proj_bias=bp.to(dev))
        if mode in ("fp16", "fp16x2"):            # hi + lo weights: no persistent projection form, the offer is declined
            assert pk.wide and pk.w_v2 is None
            assert ops.conv2d(h, pk, gn_ab=tab, transform=ops.XF_AFFINE_SILU, proj_x0=x0, proj_x1=x1, fused_only=True) is None
            continue
        assert pk.w_v2 is not None and pk.w_v2_lo is None
        recs = []
        ops.set_conv_probe(recs)
        try:
            y = ops.conv2d(h, pk, gn_ab=tab, transform=ops.XF_AFFINE_SILU, proj_x0=x0, proj_x1=x1, stats=True, fused_only=True)
            torch.cuda.synchronize()
        finally:
            ops.set_conv_probe(None)
        assert y is not None and "conv_v3" in recs[-1]["kernel"], recs
        assert_stats(y._hsidm_stats[0], y, "proj_fused")
        check("anchor_proj_fused_%d_%d" % (Ci, P0 + P1), mode, y, want, tol=TOL["bf16"] if mode == "bf16" else (8e-4 if mode == "fp16d1" else TOL["fp16"]))
        # the two-launch form on the same operands
        r = ops.conv2d(x0, ops.PackedConv(wp.to(dev), bp.to(dev), mode), x1=x1)
        y2 = ops.conv2d(h, ops.PackedConv(w.to(dev), b.to(dev), mode), gn_ab=tab, transform=ops.XF_AFFINE_SILU, res=r)
        torch.cuda.synchronize()
        e1, e2 = rel_err(y.float().cpu().numpy(), want.numpy()), rel_err(y2.float().cpu().numpy(), want.numpy())
        log_err("proj_two_launch_%d_%d" % (Ci, P0 + P1), mode, e2)
        assert e1 < 1.1 * e2 + 2e-5, (e1, e2)          # (the fused form skips one fp16 rounding of the projection's result)
    # the switch: with it the descriptor is not taken by a persistent kernel and the caller keeps its two launches
    old = _lib.lib().hsidm_debug_switch(b"NO_FUSED_PROJ", 1)
    try:
        assert ops.conv2d(h, pk, gn_ab=tab, transform=ops.XF_AFFINE_SILU, proj_x0=x0, proj_x1=x1, fused_only=True) is None
    finally:
        _lib.lib().hsidm_debug_switch(b"NO_FUSED_PROJ", old)


@pytest.mark.parametrize("mode", ["fp16x1", "fp16x2"])
def test_subnormal_only_weights_are_multiplied(dev, mode):
    """Every weight an fp16 SUBNORMAL (|w| <= 4e-5 < 6.1e-5; in the two-pass form the low halves are multiples of 6e-8): a matrix
    pipe that flushed subnormal operands would return bias only.  Activations exactly representable and large (so that the outputs
    are ordinary fp16 numbers): the result must be the fp32 convolution of the ROUNDED weights to the store's 2^-11, and in the
    one-pass form it IS what the subnormal high halves alone give."""
    from hsi_dmgasr_amd import ops
    g = torch.Generator().manual_seed(9)
    B, H, W, Ci, Co = 2, 32, 32, 64, 64
    w = (torch.rand(Co, Ci, 3, 3, generator=g) * 2 - 1) * 4e-5
    assert float(w.abs().max()) < 6.1e-5
    x = (torch.randint(-64, 65, (B, H, W, Ci), generator=g).float() * 16).to(torch.float16).to(dev)
    pk = ops.PackedConv(w.to(dev), None, mode)
    y = ops.conv2d(x, pk)
    torch.cuda.synchronize()
    hi = w.to(torch.float16)
    w_eff = hi.double() + ((w - hi.float()).to(torch.float16).double() if mode == "fp16x2" else 0.0)
    ref = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w_eff, padding=1).permute(0, 2, 3, 1)
    e = rel_err(y.double().cpu().numpy(), ref.numpy())
    log_err("subnormal_only_weights", mode, e)
    assert float(ref.abs().mean()) > 0.1 and e < 6e-4, (e, float(ref.abs().mean()))


@pytest.mark.parametrize("mode", ["bf16", "fp16", "fp32"])
def test_group_norm_table_against_torch(dev, mode):
    """hsidm_gn_partial + hsidm_gn_finalize against F.group_norm's affine form on the host (fp32 pairs and both fp16x2 copies)."""
    from hsi_dmgasr_amd import ops
    g = torch.Generator().manual_seed(3)
    B, H, W, Cc, groups = 3, 16, 16, 96, 32
    x = (2.0 + torch.randn(B, H, W, Cc, generator=g)).to(_lib.act_dtype(mode)).to(dev)
    gamma = (1 + 0.1 * torch.randn(Cc, generator=g)).to(dev)
    beta = (0.1 * torch.randn(Cc, generator=g)).to(dev)
    tab = ops.gn_scale_shift(x, None, gamma, beta, groups, mode)
    torch.cuda.synchronize()
    n = B * Cc
    pairs = tab[:2 * n].reshape(B, Cc, 2).cpu()
    xf = x.float().cpu().permute(0, 3, 1, 2)
    want = F.group_norm(xf, groups, gamma.cpu(), beta.cpu(), eps=1e-5)
    got = xf * pairs[:, :, 0, None, None] + pairs[:, :, 1, None, None]
    check("gn_table_fp32_pairs", mode, got, want, tol=1e-5)
    h = tab[2 * n:3 * n].view(torch.int32).cpu().view(torch.float16).reshape(B, Cc, 2).float()
    got_h = xf * h[:, :, 0, None, None] + h[:, :, 1, None, None]
    check("gn_table_fp16_pairs", mode, got_h, want, tol=6e-4)
    hs = tab[3 * n:4 * n].view(torch.int32).cpu().view(torch.float16).reshape(B, Cc, 2).float() / 1.44269504
    got_s = xf * hs[:, :, 0, None, None] + hs[:, :, 1, None, None]
    check("gn_table_fp16_log2e_pairs", mode, got_s, want, tol=6e-4)


F32_CASES = [  # B, H, W, C0, C1, Cout, stride, ups, GN+SiLU, film, residual, res_scale, leaky : the fp32 form of the persistent kernel
    (3, 16, 16, 64, 0, 64, 1, False, True, True, False, 1.0, False),       # odd batch, 64-cout tiles
    (3, 8, 8, 32, 0, 32, 1, False, True, True, True, 1.0, False),          # two-image 8x8 tiles, odd batch (a partial tile), 32 couts
    (5, 8, 8, 512, 0, 512, 1, False, True, False, True, 1.0, False),       # one-image 8x8 tiles (tile_kind 2), four cout slices
    (1, 8, 8, 1024, 0, 512, 1, False, True, True, False, 1.0, False),      # ... batch 1, sixteen chunks
    (2, 13, 21, 40, 0, 24, 1, False, True, False, False, 1.0, False),      # ragged map: every tile partial (scalar epilogue), bn 32
    (4, 32, 32, 64, 0, 64, 1, False, False, False, True, 0.1, True),       # the autoencoder's ResBlock tail: LeakyReLU, 0.1 * conv + x
    (2, 24, 40, 72, 24, 48, 1, False, True, True, True, 1.0, False),       # concat input, channel counts off the chunk grid
    (3, 16, 16, 64, 0, 64, 2, False, False, False, False, 1.0, False),     # stride 2 over the parity planes
    (3, 8, 8, 128, 0, 128, 1, True, False, False, False, 1.0, False),      # folded nearest-x2
]


@pytest.mark.parametrize("case", F32_CASES)
def test_fp32_form_of_the_persistent_kernel_matches_the_lds_tiled_kernel(dev, case):
    """fp32 mode: conv_v2's fp32 form (fp32 storage, bf16 hi + lo operands, three MFMAs per product) against the LDS-tiled kernel that
    carried the mode until round 3 (same arithmetic, other tiling) and against fp32 torch, on the shapes the benchmark-size anchors do
    not reach: odd batches, partial tiles, one-image 8x8 tiles, LeakyReLU + scaled residual, concat inputs."""
    from hsi_dmgasr_amd import ops
    B, H, W, C0, C1, Co, stride, ups, xf, with_film, with_res, res_scale, leaky = case
    g = torch.Generator().manual_seed(sum(int(v) for v in case[:6]))
    cin = C0 + C1
    w = torch.randn(Co, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    b = 0.1 * torch.randn(Co, generator=g)
    x0 = torch.randn(B, H, W, C0, generator=g).to(dev)
    x1 = torch.randn(B, H, W, C1, generator=g).to(dev) if C1 else None
    ab = torch.stack([1 + 0.1 * torch.randn(B, cin, generator=g), 0.2 * torch.randn(B, cin, generator=g)], 2).contiguous().to(dev) if xf else None
    Ho, Wo = (2 * H, 2 * W) if ups else (((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W))
    film = torch.randn(B, Co, generator=g).to(dev) if with_film else None
    res = torch.randn(B, Ho, Wo, Co, generator=g).to(dev) if with_res else None
    pk = ops.PackedConv(w.to(dev), b.to(dev), "fp32", fold_ups=ups, fold_dn=stride == 2)
    outs, labels = [], []
    for use_v2 in (False, True):
        ops.set_use_v2(use_v2)
        recs = []
        ops.set_conv_probe(recs)
        try:
            y = ops.conv2d(x0, pk, x1=x1, gn_ab=None if ab is None else ops.gn_table(ab), transform=ops.XF_AFFINE_SILU if xf else ops.XF_NONE,
                           film=film, res=res, res_scale=res_scale, act=ops.ACT_LEAKY if leaky else ops.ACT_NONE, stride=stride, ups=ups, stats=True)
            torch.cuda.synchronize()
        finally:
            ops.set_conv_probe(None)
            ops.set_use_v2(True)
        assert_stats(y._hsidm_stats[0], y, case)
        outs.append(y.float().cpu())
        labels.append(recs[-1]["kernel"])
    # (slices of 32 couts have no fp32 form on the persistent kernel: those two cases compare the LDS-tiled kernel with torch only)
    assert labels[0].startswith("conv_igemm") and labels[1].startswith("conv_v2" if Co > 32 else "conv_igemm"), labels
    check("fp32_v2_vs_igemm%s" % (case,), "fp32", outs[1], outs[0], tol=2e-5)
    xr = x0 if x1 is None else torch.cat([x0, x1], dim=3)
    xr = xr.float().cpu()
    if ab is not None:
        a = ab.float().cpu()
        xr = xr * a[:, None, None, :, 0] + a[:, None, None, :, 1]
        xr = xr * torch.sigmoid(xr)
    xr = xr.permute(0, 3, 1, 2)
    if ups:
        xr = F.interpolate(xr, scale_factor=2, mode="nearest")
    want = F.conv2d(xr, w, b, stride=stride, padding=1)
    if film is not None:
        want = want + film.cpu()[:, :, None, None]
    if leaky:
        want = F.leaky_relu(want, 0.01)
    want = res_scale * want.permute(0, 2, 3, 1)
    if res is not None:
        want = want + res.cpu()
    check("fp32_v2_vs_torch%s" % (case,), "fp32", outs[1], want, tol=1e-4)


F32H_CASES = F32_CASES + [
    (3, 19, 37, 128, 64, 128, 1, False, True, True, True, 1.0, False),     # ragged 128-cout map with concat, FiLM and residual: partial tiles on the native form
    (2, 20, 24, 64, 0, 128, 1, True, False, False, False, 1.0, False),     # folded nearest-x2 on a ragged input grid
    (2, 22, 38, 128, 0, 64, 2, False, False, False, True, 1.0, False),     # stride 2, even ragged map, residual
]


@pytest.mark.parametrize("case", F32H_CASES)
def test_fp32h_forms_on_ragged_and_odd_shapes(dev, case):
    """The "fp32h" kernel set (fp32 storage, one fp16 activation operand, fp16 hi + lo weights: HSIDM_F32H) on the shapes the benchmark-size
    anchors do not reach - odd batches, partial tiles (the scalar epilogue), LeakyReLU + scaled residual, concat inputs off the chunk
    grid, 8x8 maps, 32-cout slices: whatever the dispatch does with the shape - its own two-pass form, or the fp32 set's kernel on the
    layer's fp32-set weights (ops.PackedConv.fallback) - the result is the fp32 set's to the operand rounding (native) or bit for bit
    (fall-back), the statistics slab describes the stored tensor, and torch's fp32 convolution is 3.5e-4 away at most."""
    from hsi_dmgasr_amd import ops
    B, H, W, C0, C1, Co, stride, ups, xf, with_film, with_res, res_scale, leaky = case
    g = torch.Generator().manual_seed(sum(int(v) for v in case[:6]) + 1)
    cin = C0 + C1
    w = torch.randn(Co, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    b = 0.1 * torch.randn(Co, generator=g)
    x0 = torch.randn(B, H, W, C0, generator=g).to(dev)
    x1 = torch.randn(B, H, W, C1, generator=g).to(dev) if C1 else None
    ab = torch.stack([1 + 0.1 * torch.randn(B, cin, generator=g), 0.2 * torch.randn(B, cin, generator=g)], 2).contiguous().to(dev) if xf else None
    Ho, Wo = (2 * H, 2 * W) if ups else (((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W))
    film = torch.randn(B, Co, generator=g).to(dev) if with_film else None
    res = torch.randn(B, Ho, Wo, Co, generator=g).to(dev) if with_res else None
    kw = dict(x1=x1, gn_ab=None if ab is None else ops.gn_table(ab), transform=ops.XF_AFFINE_SILU if xf else ops.XF_NONE, film=film, res=res,
              res_scale=res_scale, act=ops.ACT_LEAKY if leaky else ops.ACT_NONE, stride=stride, ups=ups, stats=True)
    pk = ops.PackedConv(w.to(dev), b.to(dev), "fp32h", fold_ups=ups, fold_dn=stride == 2)
    pk32 = ops.PackedConv(w.to(dev), b.to(dev), "fp32", fold_ups=ups, fold_dn=stride == 2)
    y = ops.conv2d(x0, pk, **kw)
    y32 = ops.conv2d(x0, pk32, **kw)
    torch.cuda.synchronize()
    assert y.dtype == torch.float32 and y.shape == y32.shape
    assert_stats(y._hsidm_stats[0], y, case)
    native = pk.prec == _lib.F32H and pk._fallback is None
    d = rel_err(y.cpu().numpy(), y32.cpu().numpy())
    log_err("fp32h_vs_fp32%s" % (case,), "fp32h", d, {"native": native})
    if native:
        assert 5e-6 < d < 3.5e-4, (case, d)          # (0.1 * conv + x: the residual carries most of the norm)
    else:
        assert torch.equal(y, y32), (case, d)
    # the shapes of this list the two-pass forms take: output maps >= 16 wide, cout slices of 64 / 128 (Cout > 32)
    assert native == ((W if ups else Wo) >= 16 and Co > 32), (case, native, pk.prec)
    xr = x0 if x1 is None else torch.cat([x0, x1], dim=3)
    xr = xr.float().cpu()
    if ab is not None:
        a = ab.float().cpu()
        xr = xr * a[:, None, None, :, 0] + a[:, None, None, :, 1]
        xr = xr * torch.sigmoid(xr)
    xr = xr.permute(0, 3, 1, 2)
    if ups:
        xr = F.interpolate(xr, scale_factor=2, mode="nearest")
    want = F.conv2d(xr, w, b, stride=stride, padding=1)
    if film is not None:
        want = want + film.cpu()[:, :, None, None]
    if leaky:
        want = F.leaky_relu(want, 0.01)
    want = res_scale * want.permute(0, 2, 3, 1)
    if res is not None:
        want = want + res.cpu()
    check("fp32h_vs_torch%s" % (case,), "fp32h", y, want, tol=3.5e-4)


def test_fp16_stores_saturate_instead_of_overflowing(dev):
    """fp16's range ends at 65 504: a convolution whose result exceeds it must store +-65504, not inf (an inf would turn the next
    GroupNorm's statistics, and with them the rest of the chain, into NaN).  Every store path: the vector epilogue (with and without a
    residual), the scalar epilogue of partial tiles, the 1x1 GEMM, the split-K finish."""
    from hsi_dmgasr_amd import ops
    g = torch.Generator().manual_seed(9)
    for (B, H, W, Ci, Co, ks) in ((2, 16, 16, 64, 64, 3), (2, 16, 16, 128, 128, 3), (1, 13, 9, 64, 48, 3), (2, 16, 16, 128, 128, 1), (1, 8, 8, 512, 512, 3)):
        w = torch.full((Co, Ci, ks, ks), 40.0)                        # 64 x 9 x 40 x 30 ~ 7e5 >> 65504
        w[::2] *= -1.0
        x = torch.full((B, H, W, Ci), 30.0).to(torch.float16).to(dev)
        res = torch.randn(B, H, W, Co, generator=g).to(torch.float16).to(dev)
        # (with the GroupNorm + SiLU prologue - identity pairs, silu(30) = 30 - the 64- and 128-cout shapes take conv_v3 / conv_v2's
        # forms with the sparse second weight pass: transposed accumulators, 8-byte patch writes)
        ab = ops.gn_table(torch.stack([torch.ones(B, Ci), torch.zeros(B, Ci)], dim=2).contiguous().to(dev))
        for r, xf in ((None, False), (res, False)) + (((None, True), (res, True)) if ks == 3 and Ci % 64 == 0 else ()):
            pk = ops.PackedConv(w.to(dev), None, "fp16")
            y = ops.conv2d(x, pk, res=r, stats=True, gn_ab=ab if xf else None, transform=ops.XF_AFFINE_SILU if xf else ops.XF_NONE)
            torch.cuda.synchronize()
            yf = y.float().cpu()
            assert torch.isfinite(yf).all(), (B, H, W, Ci, Co, ks, r is not None, xf)
            inner = yf[:, 2:-2, 2:-2, :] if ks == 3 else yf
            assert float(inner[..., 1::2].min()) == 65504.0 and float(inner[..., 0::2].max()) == -65504.0, (B, H, W, Ci, Co, ks, xf)
