"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the committed golden vectors,
for every UNet building block, whole networks and the sampler."""
import numpy as np
import pytest
import torch

from gpu_util import assert_stats, check, fill_synth
from hsi_dmgasr_amd import _lib
from helpers import jload, load_npz, rel_err, sub_shapes, synth_sd, synth_tensor

pytestmark = pytest.mark.gpu

# Mode names in this file: "fp32", "fp16", "bf16" parametrise KERNEL SETS in the per-kernel tests (the autouse fixture below enters
# precision.kernels_as_named(), so "fp16" means the fp16 kernels themselves); the tests marked `public_modes` run WITHOUT it and
# check what a user of the modules gets: in the package-default "fp16" policy a bare forward's output is the result (gain 1) and
# runs on the fp32 kernel set (precision.forward_precision), so UNet.forward / Block.forward / ResnetBlock.forward meet 1e-3.
#
# ONE forward of a whole UNet on the fp16 KERNEL SET (fp16 storage and operands, hi + lo weights on Cout <= 128) measures 1.07e-3 ...
# 1.16e-3 against the reference (tiny / mid / full / non-square): 11-bit storage of ~70 tensors, of which the residual stream's
# is the largest part (tests/precision_emul.py on the full-size forward: stream tensors in fp32 -27 %, fp32 GroupNorm pairs -10 %,
# fp32 projection outputs -10 %, centred rounding -5 %).  The policy never hands that to a caller: the kernel set only runs inside a
# reverse chain at steps whose gain is <= 0.45 (seven reference chains at 1e-3 / 0.01 dB / 0.001 deg in tests/test_gpu_chain.py).
# The bound below is a REGRESSION gate on the kernel set (measured worst x 1.08), used by the kernel-set test only.
KERNEL_SET_FWD_FP16 = 1.25e-3
PRECS = ["fp32", "fp16", "fp16d2", "bf16"]      # ("fp16": the hi + lo set; "fp16d2": one of the four dithered one-pass sets the policy's chain steps run)


@pytest.fixture(autouse=True)
def _kernel_sets_as_named(request):
    from hsi_dmgasr_amd import precision
    if request.node.get_closest_marker("public_modes"):
        yield
    else:
        with precision.kernels_as_named():
            yield


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops_npz():
    return load_npz("ops.npz")


def G(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_library_loaded_in_process(dev):
    from hsi_dmgasr_amd import _lib
    assert _lib.lib().hsidm_version() >= 1
    assert any("libhsidm.so" in l for l in open("/proc/self/maps"))


@pytest.mark.parametrize("prec", PRECS)
def test_block(dev, ops_npz, prec):
    from hsi_dmgasr_amd.sr3_modules import unet
    m = unet.Block(64, 48, groups=32).to(dev).eval()
    m.precision = prec
    fill_synth(m, "block.")
    check("block", prec, m(G(ops_npz["block.x"], dev)), ops_npz["block.y"])


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("tag,cin,cout", [("res_same", 64, 64), ("res_proj", 32, 64), ("res_cat", 96, 32)])
def test_resnet_block(dev, ops_npz, prec, tag, cin, cout):
    from hsi_dmgasr_amd.sr3_modules import unet
    m = unet.ResnetBlock(cin, cout, noise_level_emb_dim=32, norm_groups=32).to(dev).eval()
    m.precision = prec
    fill_synth(m, tag + ".")
    y = m(G(ops_npz[tag + ".x"], dev), G(ops_npz[tag + ".t"], dev))
    check(tag, prec, y, ops_npz[tag + ".y"])


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("tag,cin,cout", [("aff_same", 64, 64), ("aff_proj", 32, 64)])
def test_resnet_block_with_affine_noise_level(dev, prec, tag, cin, cout):
    """ResnetBlock(use_affine_level=True): h = (1 + gamma) * block1(x) + beta (reference unet.py:44-47), against the reference's output."""
    from hsi_dmgasr_amd.sr3_modules import unet
    g = load_npz("variants.npz")
    m = unet.ResnetBlock(cin, cout, noise_level_emb_dim=32, use_affine_level=True, norm_groups=32).to(dev).eval()
    m.precision = prec
    fill_synth(m, tag + ".")
    y = m(G(synth_tensor(tag + ".x", (2, cin, 8, 8)), dev), G(synth_tensor(tag + ".t", (2, 1, 32)), dev))
    check(tag, prec, y, g[tag + ".y"])
    with pytest.raises(TypeError):          # the variant the reference cannot construct either (nn.Linear(None, ...))
        unet.UNet(in_channel=6, out_channel=3, inner_channel=32, channel_mults=[1, 2], attn_res=[8], res_blocks=1,
                  with_noise_level_emb=False, image_size=16)


@pytest.mark.parametrize("prec", PRECS)
def test_self_attention_with_four_heads(dev, prec):
    """SelfAttention(n_head=4) (reference unet.py:114-143; unreachable from its UNet) against the reference's output."""
    from hsi_dmgasr_amd.sr3_modules import unet
    g = load_npz("variants.npz")
    m = unet.SelfAttention(64, n_head=4, norm_groups=32).to(dev).eval()
    m.precision = prec
    fill_synth(m, "attn4.")
    check("attn_4_heads", prec, m(G(synth_tensor("attn4.x", (2, 64, 8, 8)), dev)), g["attn4.y"])


@pytest.mark.parametrize("prec", PRECS)
def test_resnet_block_two_pointer_concat(dev, ops_npz, prec):
    """Skip-concat input read in place from two tensors (GroupNorm group 21 straddles the seam at 64)."""
    from hsi_dmgasr_amd import ops
    from hsi_dmgasr_amd.sr3_modules import unet
    m = unet.ResnetBlock(96, 32, noise_level_emb_dim=32, norm_groups=32).to(dev).eval()
    fill_synth(m, "res_cat.")
    x = G(ops_npz["res_cat.x"], dev)
    t = G(ops_npz["res_cat.t"], dev).reshape(2, 32).contiguous()
    lin = m.noise_func.noise_func[0]
    film = ops.noise_film(2, 32, None, lin.weight, lin.bias, t_emb=t)
    a = ops.to_nhwc(x[:, :64].contiguous(), prec)
    b = ops.to_nhwc(x[:, 64:].contiguous(), prec)
    y = ops.to_nchw(m._run(a, b, film, prec), prec)
    check("res_cat_two_pointer", prec, y, ops_npz["res_cat.y"])


@pytest.mark.parametrize("prec", PRECS)
def test_self_attention(dev, ops_npz, prec):
    from hsi_dmgasr_amd.sr3_modules import unet
    m = unet.SelfAttention(64, norm_groups=32).to(dev).eval()
    m.precision = prec
    fill_synth(m, "attn.")
    check("attn", prec, m(G(ops_npz["attn.x"], dev)), ops_npz["attn.y"])


@pytest.mark.parametrize("prec", PRECS)
def test_up_down(dev, ops_npz, prec):
    from hsi_dmgasr_amd.sr3_modules import unet
    m = unet.Upsample(32).to(dev).eval()
    m.precision = prec
    fill_synth(m, "up.")
    check("up", prec, m(G(ops_npz["up.x"], dev)), ops_npz["up.y"])
    m = unet.Downsample(32).to(dev).eval()
    m.precision = prec
    fill_synth(m, "down.")
    check("down", prec, m(G(ops_npz["down.x"], dev)), ops_npz["down.y"])


def test_noise_embedding(dev, ops_npz):
    from hsi_dmgasr_amd.sr3_modules import unet
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=32, channel_mults=[1, 2], attn_res=[8], res_blocks=1,
                  image_size=16).to(dev).eval()
    sd = synth_sd(sub_shapes(jload(ops_npz["shapes_json"]), "mlp."), "mlp.")
    u.load_state_dict(sd, strict=False)
    t = u.noise_embedding(G(ops_npz["mlp.gamma"], dev))
    check("noise_level_mlp", "fp32", t.reshape(3, 32), ops_npz["mlp.y"].reshape(3, 32), tol=1e-4)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("shape", [(1, 40, 13, 21), (3, 64, 5, 5), (2, 72, 24, 40)])
def test_block_ragged_shapes(dev, prec, shape):
    """Edge cases: sizes that are not multiples of the 8x16 tile, partial tiles, odd batch (two-image tiles)."""
    from hsi_dmgasr_amd.sr3_modules import unet
    from oracle import sr3_unet
    b, c, h, w = shape
    m = unet.Block(c, 24, groups=8).to(dev).eval()
    m.precision = prec
    sd = fill_synth(m, "ragged.")
    x = synth_tensor("ragged.x%s" % (shape,), shape)
    want = sr3_unet.block(sd, "", torch.from_numpy(x), 8)
    check("block_ragged%s" % (shape,), prec, m(G(x, dev)), want)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("hw", [(7, 9), (16, 16), (33, 18)])
def test_down_up_ragged(dev, prec, hw):
    from hsi_dmgasr_amd.sr3_modules import unet
    from oracle import sr3_unet
    x = synth_tensor("du.x%s" % (hw,), (2, 32) + hw)
    for cls, fn, tag in ((unet.Downsample, sr3_unet.downsample, "down"), (unet.Upsample, sr3_unet.upsample, "up")):
        m = cls(32).to(dev).eval()
        m.precision = prec
        sd = fill_synth(m, "du_%s." % tag)
        check("%s_ragged%s" % (tag, hw), prec, m(G(x, dev)), fn(sd, "", torch.from_numpy(x)))


def _unet_case(name, prec, dev):
    """(network, input, noise level, expected output, config) of a whole-UNet parity case: tiny / mid / full vs outputs captured from
    the reference (unets.npz), wide_nonsquare vs the oracle (pinned to the reference on the other three)."""
    from hsi_dmgasr_amd.sr3_modules import unet
    from oracle import sr3_unet
    g = load_npz("unets.npz")
    if name == "wide_nonsquare":
        # the shipped channel plan (64-128-256-512-512, attention where the map is image_size/8) on a 3 x 6 x 64 x 96 batch: every
        # specialised kernel sees shapes other than the benchmark's (16x16 tiles on a 4x6 grid, parity-folded up/down sampling of
        # non-square maps, attention over 96 tokens on the panel kernel, two-image tiles with an odd batch)
        cfg = dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8], attn_res=[8],
                   res_blocks=1, image_size=64)
        u = unet.UNet(dropout=0.2, precision=prec, **cfg).to(dev).eval()
        sd = fill_synth(u, "unet_wide.")
        x = synth_tensor("unet_wide.x", (3, 6, 64, 96))
        gam = np.array([[0.8], [0.3], [0.02]], dtype=np.float32)
        return u, x, gam, sr3_unet.unet_forward(sd, cfg, torch.from_numpy(x), torch.from_numpy(gam)).numpy()
    cfg = jload(g[name + ".cfg_json"])
    u = unet.UNet(in_channel=cfg["in_channel"], out_channel=cfg["out_channel"], inner_channel=cfg["inner_channel"],
                  norm_groups=cfg["norm_groups"], channel_mults=cfg["channel_mults"], attn_res=cfg["attn_res"],
                  res_blocks=cfg["res_blocks"], dropout=0.2, image_size=cfg["image_size"], precision=prec).to(dev).eval()
    fill_synth(u, "unet_%s." % name)
    x = synth_tensor("unet_full.x", (1, 6, 128, 128)) if name == "full" else g[name + ".x"]
    return u, x, g[name + ".gamma"], g[name + ".y"]


@pytest.mark.public_modes
@pytest.mark.parametrize("prec", [None, "fp16", "fp32"])
@pytest.mark.parametrize("name", ["tiny", "mid", "full", "wide_nonsquare"])
def test_unet_forward_golden(dev, prec, name):
    """UNet.forward (reference unet.py:239-263) as a user of the drop-in calls it - `netG.denoise_fn(x, t)` - in the package default
    (precision=None -> the "fp16" policy), in "fp16" by name and in "fp32": within north_star's 1e-3 of the reference's output on
    the tiny / mid / shipped 97.8 M networks and of the oracle on the non-square batch.  (The policy runs a bare forward on the
    fp32 kernel set: its output is the result; the fp16 kernel set is for chain steps with an error gain below 0.5.)"""
    u, x, gam, want = _unet_case(name, prec, dev)
    check("unet_" + name, "default" if prec is None else prec, u(G(x, dev), G(gam, dev)), want, tol=1e-3)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", ["tiny", "mid", "full", "wide_nonsquare"])
def test_unet_forward_per_kernel_set(dev, prec, name):
    """The same four cases on each KERNEL SET by name (regression gates of the sets themselves: the fp16 set's 1.07e-3 ... 1.16e-3
    is why the policy does not hand a bare forward to it; bf16 is experimental)."""
    u, x, gam, want = _unet_case(name, prec, dev)
    check("unet_%s_kernel_set" % name, prec, u(G(x, dev), G(gam, dev)), want, tol={"fp32": 1e-3, "fp16": KERNEL_SET_FWD_FP16, "fp16d2": 2e-3, "bf16": 8e-2}[prec])


@pytest.mark.public_modes
def test_bare_module_calls_in_the_default_mode(dev, ops_npz):
    """Block.forward / ResnetBlock.forward / SelfAttention.forward / Up- and Downsample.forward in the package-default mode: each
    within 1e-3 of the reference's vectors, and bit-identical to the same module in "fp32" (the kernel set the policy picks for a
    call whose output is the result); inside precision.kernels_as_named() the fp16 kernel set runs instead (a different result)."""
    from hsi_dmgasr_amd import precision
    from hsi_dmgasr_amd.sr3_modules import unet
    assert precision.get_default_precision() == "fp16" and precision.forward_precision("fp16") == "fp32"
    cases = [("block", lambda: unet.Block(64, 48, groups=32), ("block.x",)),
             ("res_proj", lambda: unet.ResnetBlock(32, 64, noise_level_emb_dim=32, norm_groups=32), ("res_proj.x", "res_proj.t")),
             ("attn", lambda: unet.SelfAttention(64, norm_groups=32), ("attn.x",)),
             ("up", lambda: unet.Upsample(32), ("up.x",)), ("down", lambda: unet.Downsample(32), ("down.x",))]
    for tag, make, keys in cases:
        m = make().to(dev).eval()
        assert m.precision is None
        fill_synth(m, tag + ".")
        args = [G(ops_npz[k], dev) for k in keys]
        y = m(*args)
        check(tag + "_default_mode", "default", y, ops_npz[tag + ".y"], tol=1e-3)
        m.precision = "fp32"
        assert torch.equal(y, m(*args)), tag
        m.precision = "fp16"
        assert torch.equal(y, m(*args)), tag
        with precision.kernels_as_named():
            assert not torch.equal(y, m(*args)), tag


@pytest.mark.parametrize("name", ["tiny", "mid", "wide_nonsquare"])
def test_headline_mode_chain_on_the_other_networks(dev, name):
    """north_star's quantity - the OUTPUT of the reverse chain - in the headline (fp16) mode on the network configurations whose single
    forward on the fp16 kernel set sits at 1.07e-3 ... 1.17e-3 (KERNEL_SET_FWD_FP16 above): a 20-step cosine chain (Philox noise; the policy's eight
    high-gain steps on the fp32 kernel set, twelve on the dithered one-pass fp16 sets) against the oracle's chain, held to 1e-3.  The oracle's
    forward is pinned to the reference's for these configurations (unets.npz; the non-square one is the shipped channel plan)."""
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    from oracle import diffusion as odiff, sr3_unet
    if name == "wide_nonsquare":
        cfg = dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8], attn_res=[8], res_blocks=1, image_size=64)
        prefix, shape = "unet_wide.", (2, 3, 64, 96)
    else:
        cfg = jload(load_npz("unets.npz")[name + ".cfg_json"])
        prefix, shape = "unet_%s." % name, (3, 3, cfg["image_size"], cfg["image_size"])
    u = unet.UNet(dropout=0.2, precision="fp16", **cfg).to(dev).eval()
    sd = fill_synth(u, prefix)
    T = 20
    opt = dict(schedule="cosine", n_timestep=T, linear_start=1e-6, linear_end=1e-2)
    gd = diffusion.GaussianDiffusion(u, image_size=cfg["image_size"], channels=3, conditional=True)
    gd.set_loss(dev)
    gd.set_new_noise_schedule(opt, dev)
    gd.noise, gd.seed = "philox", 2024
    cond = synth_tensor("headline_chain.%s.cond" % name, shape)
    got = gd.p_sample_loop_batched(G(cond, dev))
    torch.cuda.synchronize()
    sched = odiff.noise_schedule(opt)
    den = lambda x, gam: sr3_unet.unet_forward(sd, cfg, x, gam)
    nf = odiff.philox_noise_fn(2024, shape)
    xo = nf(T)
    with torch.no_grad():
        for i in reversed(range(T)):
            xo = odiff.p_sample_step(den, sched, xo, torch.from_numpy(cond), i, nf(i) if i > 0 else None)
    check("headline_chain_T20_" + name, "fp16", got, xo, tol=1e-3)


def test_full_size_sampler_is_deterministic_and_finite(dev):
    """Two runs of an 8-step chain on the shipped UNet (bf16 mode, HIP-graph replay, Philox noise, batch 7): bit-identical and
    finite.  Guards the kernels' ordering assumptions (loads the compiler does not track, statistics written exactly once,
    no atomics) against anything timing-dependent."""
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    cfg = jload(load_npz("unets.npz")["full.cfg_json"])
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=cfg["channel_mults"],
                  attn_res=cfg["attn_res"], res_blocks=2, dropout=0.2, image_size=128, precision="bf16").to(dev).eval()
    fill_synth(u, "unet_full.")
    gd = diffusion.GaussianDiffusion(u, image_size=128, channels=3, conditional=True)
    gd.set_loss(dev)
    gd.set_new_noise_schedule(dict(schedule="cosine", n_timestep=8, linear_start=1e-6, linear_end=1e-2), dev)
    gd.noise, gd.seed = "philox", 99
    cond = G(synth_tensor("det.cond", (7, 3, 128, 128)), dev)
    a = gd.p_sample_loop_batched(cond).clone()
    b = gd.p_sample_loop_batched(cond)
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and torch.equal(a, b)
    assert float(a.abs().max()) <= 1.0 + 1e-6                      # clip_denoised at the last step


def test_philox_matches_oracle(dev):
    from hsi_dmgasr_amd import ops
    from oracle import philox
    for n, seed, stream in ((1000, 7, 3), (4097, (1 << 40) + 5, 999)):
        z = ops.philox_normal((n,), seed, stream, dev).cpu().numpy()
        ref = philox.normal(seed, stream, n)
        assert np.max(np.abs(z - ref)) < 1e-5


@pytest.mark.parametrize("prec", ["fp32"])
@pytest.mark.parametrize("name", ["T4", "T24"])
def test_sampler_stored_noise_golden(dev, prec, name):
    """p_sample_loop with the x_T / per-step noise captured from the reference run: continous frames and the
    ret_img[-1] convention."""
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    g = load_npz("sampler.npz")
    cfg = jload(load_npz("unets.npz")["tiny.cfg_json"])
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=32, channel_mults=[1, 2], attn_res=[8], res_blocks=1,
                  image_size=16, precision=prec).to(dev).eval()
    fill_synth(u, "unet_tiny.")
    gd = diffusion.GaussianDiffusion(u, image_size=16, channels=3, conditional=True)
    gd.set_loss(dev)
    gd.set_new_noise_schedule(jload(g[name + ".opt_json"]), dev)
    cond = G(g[name + ".cond"], dev)
    x, snap, _ = gd._reverse(cond, tuple(cond.shape), True, x_T=G(g[name + ".x_T"], dev), noise=G(g[name + ".noise"], dev))
    frames = torch.cat([cond, snap.reshape((-1,) + tuple(cond.shape[1:]))], dim=0)
    check("sampler_%s_continous" % name, prec, frames, g[name + ".continous"], tol=1e-3 if prec == "fp32" else 2e-3)
    check("sampler_%s_last" % name, prec, frames[-1], g[name + ".last"], tol=1e-3 if prec == "fp32" else 2e-3)


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-3), ("bf16", 2e-2)])
@pytest.mark.parametrize("kind", ["l1", "l2"])
def test_training_objective_golden(dev, prec, tol, kind):
    """GaussianDiffusion.forward(dict) = p_losses value, with t / gamma drawn from numpy's seed as the reference does."""
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    g = load_npz("losses.npz")
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=32, channel_mults=[1, 2], attn_res=[8], res_blocks=1,
                  image_size=16, precision=prec).to(dev).eval()
    fill_synth(u, "unet_tiny.")
    k = "loss_%s." % kind
    gd = diffusion.GaussianDiffusion(u, image_size=16, channels=3, loss_type=kind, conditional=True)
    gd.set_loss(dev)
    gd.set_new_noise_schedule(jload(g[k + "opt_json"]), dev)
    xq = gd.q_sample(G(g[k + "hr"], dev), G(g[k + "gamma"], dev).view(-1, 1, 1, 1), G(g[k + "noise"], dev))
    check("q_sample_%s" % kind, "fp32", xq, g[k + "x_noisy"], tol=1e-6)
    np.random.seed(int(g[k + "np_seed"]))
    loss = gd({"HR": G(g[k + "hr"], dev), "SR": G(g[k + "sr"], dev)}, noise=G(g[k + "noise"], dev))
    assert loss.ndim == 0 and loss.dtype == torch.float32
    check("p_losses_%s" % kind, prec, loss.reshape(1), g[k + "loss"].reshape(1), tol=tol)


def test_loss_sum_large_and_bad_args(dev):
    from hsi_dmgasr_amd import ops
    a = torch.randn(3_000_001, device=dev)
    b = torch.randn(3_000_001, device=dev)
    ref1 = (a.double() - b.double()).abs().sum().item()
    ref2 = ((a.double() - b.double()) ** 2).sum().item()
    assert abs(ops.loss_sum(a, b, "l1").item() - ref1) < 1e-6 * ref1
    assert abs(ops.loss_sum(a, b, "l2").item() - ref2) < 1e-6 * ref2
    assert ops.loss_sum(a, b, "l1").item() == ops.loss_sum(a, b, "l1").item()      # deterministic
    with pytest.raises(KeyError):
        ops.loss_sum(a, b, "huber")


def test_sampler_api_shapes_and_philox_mode(dev):
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    from oracle import diffusion as odiff, sr3_unet
    cfg = jload(load_npz("unets.npz")["tiny.cfg_json"])
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=32, channel_mults=[1, 2], attn_res=[8], res_blocks=1,
                  image_size=16, precision="fp32").to(dev).eval()
    sd = fill_synth(u, "unet_tiny.")
    opt = dict(schedule="cosine", n_timestep=12, linear_start=1e-6, linear_end=1e-2)
    gd = diffusion.GaussianDiffusion(u, image_size=16, channels=3, conditional=True)
    gd.set_loss(dev)
    gd.set_new_noise_schedule(opt, dev)
    cond = G(synth_tensor("api.cond", (3, 3, 16, 16)), dev)
    assert gd.super_resolution(cond, continous=False).shape == (3, 16, 16)
    inter = 1 | (12 // 10)
    assert gd.super_resolution(cond, continous=True).shape == ((1 + (11 // inter + 1)) * 3, 3, 16, 16)
    # Philox mode: device noise == oracle noise, so the whole chain is comparable
    gd.noise, gd.seed = "philox", 2024
    got = gd.p_sample_loop_batched(cond)
    sched = odiff.noise_schedule(opt)
    den = lambda x, gam: sr3_unet.unet_forward(sd, cfg, x, gam)
    nf = odiff.philox_noise_fn(2024, (3, 3, 16, 16))
    x_T = nf(12)
    xo = x_T
    for i in reversed(range(12)):
        xo = odiff.p_sample_step(den, sched, xo, cond.cpu(), i, nf(i) if i > 0 else None)
    check("sampler_philox_T12", "fp32", got, xo, tol=1e-3)


@pytest.mark.parametrize("steps,eta", [(6, 0.0), (9, 0.7), (12, 1.0)])
def test_strided_ddim_sampler_matches_oracle(dev, steps, eta):
    """K-step DDIM on the device (same kernels, alternative coefficient table) against the oracle's restatement with the
    same Philox noise; eta = 1 with all steps is the reference's ancestral sampler."""
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    from oracle import diffusion as odiff, sr3_unet
    cfg = jload(load_npz("unets.npz")["tiny.cfg_json"])
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=32, channel_mults=[1, 2], attn_res=[8], res_blocks=1,
                  image_size=16, precision="fp32").to(dev).eval()
    sd = fill_synth(u, "unet_tiny.")
    opt = dict(schedule="cosine", n_timestep=12, linear_start=1e-6, linear_end=1e-2)
    gd = diffusion.GaussianDiffusion(u, image_size=16, channels=3, conditional=True)
    gd.set_loss(dev)
    gd.set_new_noise_schedule(opt, dev)
    gd.set_sampler("ddim", steps=steps, eta=eta)
    gd.noise, gd.seed = "philox", 77
    cond = G(synth_tensor("ddim.cond", (2, 3, 16, 16)), dev)
    got = gd.p_sample_loop_batched(cond)
    tab = odiff.ddim_schedule(opt, steps, eta)
    den = lambda x, gam: sr3_unet.unet_forward(sd, cfg, x, gam)
    nf = odiff.philox_noise_fn(77, (2, 3, 16, 16))
    want = odiff.ddim_sample_loop(den, tab, cond.cpu(), nf(steps), nf)
    check("ddim_K%d_eta%g" % (steps, eta), "fp32", got, want, tol=1e-3)
    if steps == 12 and eta == 1.0:
        gd.set_sampler("ddpm")
        check("ddim_full_eta1_is_ddpm", "fp32", got, gd.p_sample_loop_batched(cond), tol=1e-4)
    assert gd.super_resolution(cond, continous=False).shape == (3, 16, 16)


@pytest.mark.public_modes
@pytest.mark.parametrize("steps,eta", [(10, 0.0), (20, 0.5)])
def test_strided_ddim_sampler_in_the_default_mode(dev, steps, eta):
    """The strided sampler under the "fp16" policy (the package default): its coefficient table carries its own update gains
    (GaussianDiffusion.set_sampler: _run_eps_gain), so the policy's choice of the fp32-set steps follows the DDIM table, the other
    steps run on the dithered fp16 sets; a K-step chain of a 40-step cosine schedule on the mid network against the oracle at 1e-3."""
    from hsi_dmgasr_amd import precision
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    from oracle import diffusion as odiff, sr3_unet
    cfg = jload(load_npz("unets.npz")["mid.cfg_json"])
    u = unet.UNet(dropout=0.2, **cfg).to(dev).eval()              # precision=None: the package default
    sd = fill_synth(u, "unet_mid.")
    opt = dict(schedule="cosine", n_timestep=40, linear_start=1e-6, linear_end=1e-2)
    gd = diffusion.GaussianDiffusion(u, image_size=cfg["image_size"], channels=3, conditional=True)
    gd.set_loss(dev)
    gd.set_new_noise_schedule(opt, dev)
    gd.set_sampler("ddim", steps=steps, eta=eta)
    gd.noise, gd.seed = "philox", 91
    shape = (2, 3, cfg["image_size"], cfg["image_size"])
    cond = G(synth_tensor("ddim.default.cond", shape), dev)
    run = gd.make_run(cond)
    n_hi = sum(1 for m in run.modes if m in ("fp32", "fp32h"))
    assert 1 <= n_hi < steps and all(precision.family(m) == "fp16" for m in run.modes[n_hi:]) and set(run.modes[:n_hi]) <= {"fp32", "fp32h"}
    got = run.run_all()
    tab = odiff.ddim_schedule(opt, steps, eta)
    den = lambda x, gam: sr3_unet.unet_forward(sd, cfg, x, gam)
    nf = odiff.philox_noise_fn(91, shape)
    want = odiff.ddim_sample_loop(den, tab, cond.cpu(), nf(steps), nf)
    check("ddim_default_mode_K%d_eta%g" % (steps, eta), "default", got, want, tol=1e-3)


@pytest.mark.parametrize("shape", [(3, 16, 16, 512), (2, 8, 8, 128), (5, 8, 8, 64), (2, 16, 16, 64), (1, 4, 8, 96), (136, 16, 16, 128),
                                   (260, 8, 8, 64), (3, 16, 16, 64)])
def test_attention_core_matches_torch_and_v1(dev, shape, monkeypatch):
    """softmax(q k^T / sqrt(C)) v on random qkv: the register-resident kernel (attention_v2: N = 64 / 256, C % 64 == 0; at N = 256 one
    8-wave workgroup per image from 128 images on - 136 images - else two of 4 waves, HSIDM_ATTENTION_V1=2; 260 images at N = 64: more
    workgroups than CUs), the panel kernel (other shapes, and HSIDM_ATTENTION_V1=1) and torch fp32 on the same 16-bit inputs, in
    both element types; the two workgroup shapes compute the same products in the same order: bit-identical."""
    from hsi_dmgasr_amd import ops
    B, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape))
    raw = torch.randn(B, H, W, 3 * C, generator=g) * 1.5
    for mode, dt, tol in (("bf16", torch.bfloat16, 1e-2), ("fp16x1", torch.float16, 1.5e-3)):
        qkv = raw.to(dt)
        q, k, v = qkv.float().reshape(B, H * W, 3, C).unbind(2)
        ref = torch.softmax(q @ k.transpose(1, 2) / C ** 0.5, dim=-1) @ v
        got = ops.attention(qkv.to(dev), mode)
        torch.cuda.synchronize()
        with _lib.debug_switch("ATTENTION_V1", 1):
            old = ops.attention(qkv.to(dev), mode)
            torch.cuda.synchronize()
        with _lib.debug_switch("ATTENTION_V1", 2):
            v2 = ops.attention(qkv.to(dev), mode)
            torch.cuda.synchronize()
        check("attention%s" % (shape,), mode, got.reshape(B, H * W, C), ref, tol=tol)
        check("attention_panel%s" % (shape,), mode, old.reshape(B, H * W, C), ref, tol=tol)
        check("attention_4wave%s" % (shape,), mode, v2.reshape(B, H * W, C), ref, tol=tol)
        assert torch.equal(got, v2)



@pytest.mark.parametrize("shape", [(3, 16, 16, 512), (5, 8, 8, 64), (2, 16, 16, 32), (260, 8, 8, 64), (136, 16, 16, 96), (1, 4, 8, 96)])
def test_attention_core_fp32_mode(dev, shape):
    """fp32 mode (fp32 storage, bf16 hi + lo operands, three MFMAs per product): the register-resident form (N = 64 / 256, C % 32 == 0)
    and the score-panel kernel (other shapes, HSIDM_ATTENTION_V1=1) against torch in float64 on the same fp32 inputs; the two kernels
    agree with each other to the rounding of the split."""
    from hsi_dmgasr_amd import ops
    B, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    qkv = torch.randn(B, H, W, 3 * C, generator=g) * 1.5
    q, k, v = qkv.double().reshape(B, H * W, 3, C).unbind(2)
    ref = (torch.softmax(q @ k.transpose(1, 2) / C ** 0.5, dim=-1) @ v).float()
    got = ops.attention(qkv.to(dev), "fp32")
    torch.cuda.synchronize()
    with _lib.debug_switch("ATTENTION_V1", 1):
        old = ops.attention(qkv.to(dev), "fp32")
        torch.cuda.synchronize()
    check("attention_f32%s" % (shape,), "fp32", got.reshape(B, H * W, C), ref, tol=2e-5)
    check("attention_f32_panel%s" % (shape,), "fp32", old.reshape(B, H * W, C), ref, tol=2e-5)


CONV_CASES = [  # B, H, W, C0, C1, Cout, ups, proj_cin, xf
    (3, 16, 32, 64, 0, 128, False, 0, True),      # 8x16 tiles, BN=128, several items per block
    (2, 24, 40, 64, 32, 64, False, 0, True),      # concat input, BN=64, partial tiles
    (5, 8, 8, 128, 0, 96, False, 0, True),        # two-image tiles (odd batch), cout not a multiple of 32
    (2, 8, 16, 32, 0, 24, True, 0, False),        # nearest-x2 folded in, BN=32, no transform
    (40, 16, 16, 64, 0, 64, False, 0, True),      # many items: persistent loop over tiles (16x16 tiles: conv_v3)
    (3, 32, 48, 64, 0, 64, False, 0, True),       # conv_v3: 256-pixel tiles, one chunk
    (2, 32, 32, 128, 64, 64, False, 0, True),     # conv_v3: concat input, three chunks
    (1, 16, 32, 72, 0, 64, False, 0, True),       # conv_v3: channel count off the chunk grid
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_v2_matches_v1_and_emits_statistics(dev, case):
    """The persistent bf16 kernel (conv_v2) against the v1 kernel on identical inputs, and the statistics slab
    of both against sums of the stored output (for conv_v2 / conv_v3 up to the store's rounding noise, see below)."""
    from hsi_dmgasr_amd import ops
    B, H, W, C0, C1, Co, ups, pj, xf = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    w = torch.randn(Co, C0 + C1, 3, 3, generator=g) / (9 * (C0 + C1)) ** 0.5
    pwt = (torch.randn(Co, pj, 1, 1, generator=g) / pj ** 0.5).to(dev) if pj else None
    pk = ops.PackedConv(w.to(dev), torch.randn(Co, generator=g).to(dev), "bf16", proj_weight=pwt,
                        proj_bias=None if pwt is None else torch.zeros(Co, device=dev))
    assert pk.w_v2 is not None
    x0 = torch.randn(B, H, W, C0, generator=g).to(dev, torch.bfloat16)
    x1 = torch.randn(B, H, W, C1, generator=g).to(dev, torch.bfloat16) if C1 else None
    px = torch.randn(B, H, W, pj, generator=g).to(dev, torch.bfloat16) if pj else None
    ab = torch.stack([1 + 0.1 * torch.randn(B, C0 + C1, generator=g), 0.1 * torch.randn(B, C0 + C1, generator=g)], 2).contiguous().to(dev)
    film = torch.randn(B, Co, generator=g).to(dev)
    outs = []
    for use_v2 in (False, True):
        ops.set_use_v2(use_v2)
        y = ops.conv2d(x0, pk, x1=x1, gn_ab=ops.gn_table(ab) if xf else None, transform=ops.XF_AFFINE_SILU if xf else ops.XF_NONE,
                       film=film, ups=ups, proj_x0=px, stats=True)
        torch.cuda.synchronize()
        slab, nsplit = y._hsidm_stats
        yf = y.float()
        assert slab.shape == (B, nsplit, Co, 2)
        assert_stats(slab, y, use_v2)
        outs.append(yf.cpu())
    ops.set_use_v2(True)
    check("conv_v2_vs_v1%s" % (case,), "bf16", outs[1], outs[0], tol=2e-3)


def test_conv_v3_residual_and_dispatch(dev, monkeypatch):
    """The 256-pixel kernel with a residual operand (ResnetBlock.block2 + x) against conv_v2 on the same inputs
    (HSIDM_NO_V3=1), which the previous test ties to v1."""
    from hsi_dmgasr_amd import ops
    g = torch.Generator().manual_seed(11)
    B, H, W, Ci, Co = 5, 32, 32, 64, 64
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    pk = ops.PackedConv(w.to(dev), torch.randn(Co, generator=g).to(dev), "bf16")
    x = torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16)
    res = torch.randn(B, H, W, Co, generator=g).to(dev, torch.bfloat16)
    ab = torch.stack([1 + 0.1 * torch.randn(B, Ci, generator=g), 0.1 * torch.randn(B, Ci, generator=g)], 2).contiguous().to(dev)
    film = torch.randn(B, Co, generator=g).to(dev)
    outs, slabs = [], []
    for no_v3 in (True, False):
        with _lib.debug_switch("NO_V3", int(no_v3)):
            y = ops.conv2d(x, pk, gn_ab=ops.gn_table(ab), transform=ops.XF_AFFINE_SILU, film=film, res=res, stats=True)
            torch.cuda.synchronize()
        slab, nsplit = y._hsidm_stats
        yf = y.float()
        assert_stats(slab, y, no_v3)
        outs.append(yf.cpu())
        slabs.append(nsplit)
    assert slabs == [8 * 2, 4 * 2]                       # 8x16 tiles x 2 wave rows vs 16x16 tiles x 2 pixel halves
    check("conv_v3_vs_v2_residual", "bf16", outs[1], outs[0], tol=2e-3)


WIDE_CASES = [  # B, H, W, Cin, Cout, residual : GroupNorm+SiLU convs whose Cout is a multiple of 256
    (40, 16, 16, 256, 512, False),     # 8x16 tiles, two 256-cout slices
    (36, 16, 32, 128, 256, True),      # one slice, residual + statistics in the vector domain
    (260, 8, 8, 128, 256, False),      # two-image 8x8 tiles
    (4, 16, 16, 128, 256, False),      # too few items for one workgroup per CU: stays on the 128-cout form
]


@pytest.mark.parametrize("case", WIDE_CASES)
def test_256_cout_items_on_8_waves_match_the_128_cout_form(dev, monkeypatch, case):
    """conv_v2's NW = 8 form (256 couts per item, one workgroup per CU) against the 4-wave 128-cout form on identical inputs
    (HSIDM_V2_BN256=0): same staging, same K order, same epilogue per wave -> bit-identical output and statistics; and the
    dispatch rule (hsidm_conv_kernel_id through the probe label)."""
    from hsi_dmgasr_amd import ops
    B, H, W, Ci, Co, with_res = case
    g = torch.Generator().manual_seed(B + Ci + Co)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    pk = ops.PackedConv(w.to(dev), torch.randn(Co, generator=g).to(dev), "bf16")
    x = torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16)
    res = torch.randn(B, H, W, Co, generator=g).to(dev, torch.bfloat16) if with_res else None
    ab = torch.stack([1 + 0.1 * torch.randn(B, Ci, generator=g), 0.1 * torch.randn(B, Ci, generator=g)], 2).contiguous().to(dev)
    film = torch.randn(B, Co, generator=g).to(dev)
    outs, slabs, labels = [], [], []
    for wide in (False, True):
        recs = []
        with _lib.debug_switch("V2_BN256", int(wide)):
            ops.set_conv_probe(recs)
            y = ops.conv2d(x, pk, gn_ab=ops.gn_table(ab), transform=ops.XF_AFFINE_SILU, film=film, res=res, stats=True)
            ops.set_conv_probe(None)
            torch.cuda.synchronize()
        slab, _ = y._hsidm_stats
        assert_stats(slab, y, wide)
        outs.append(y.float().cpu())
        slabs.append(slab.float().cpu())
        labels.append(recs[-1]["kernel"])
    assert "bn128" in labels[0]
    assert ("bn256" if B >= 36 else "bn128") in labels[1], labels
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(slabs[0], slabs[1])


SK_CASES = [  # B, H, W, C0, C1, Cout, xf, residual
    (5, 8, 8, 512, 512, 512, True, True),      # the 8x8 level's 1024 -> 512 block at one CAVE image: 16 chunks over 16 parts
    (5, 16, 16, 512, 0, 512, True, False),     # 16x16 level
    (4, 8, 8, 512, 0, 1024, False, False),     # an input-gradient convolution of the training step (no transform)
    (3, 8, 16, 256, 72, 128, True, True),      # ragged last chunk (328 channels), non-square map, one cout slice
    (16, 8, 8, 512, 0, 512, True, False),      # 8 two-image tiles x 4 slices = 32 items: still an eighth of the slots
]


@pytest.mark.parametrize("case", SK_CASES)
def test_split_k_convolution_matches_the_persistent_kernel(dev, case):
    """conv_sk (split over the contraction, fp32 partial sums, finishing kernel) against conv_v2 on the same inputs (debug switch
    NO_SPLIT_K) and against torch; output, dispatch label and statistics slab."""
    from hsi_dmgasr_amd import ops
    B, H, W, C0, C1, Co, xf, with_res = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    w = torch.randn(Co, C0 + C1, 3, 3, generator=g) / (9 * (C0 + C1)) ** 0.5
    bias = torch.randn(Co, generator=g)
    pk = ops.PackedConv(w.to(dev), bias.to(dev), "bf16")
    x0 = torch.randn(B, H, W, C0, generator=g).to(dev, torch.bfloat16)
    x1 = torch.randn(B, H, W, C1, generator=g).to(dev, torch.bfloat16) if C1 else None
    res = torch.randn(B, H, W, Co, generator=g).to(dev, torch.bfloat16) if with_res else None
    ab = torch.stack([1 + 0.1 * torch.randn(B, C0 + C1, generator=g), 0.1 * torch.randn(B, C0 + C1, generator=g)], 2).contiguous()
    film = torch.randn(B, Co, generator=g).to(dev)
    outs, labels = [], []
    for no_sk in (1, 0):
        recs = []
        with _lib.debug_switch("NO_SPLIT_K", no_sk):
            ops.set_conv_probe(recs)
            y = ops.conv2d(x0, pk, x1=x1, gn_ab=ops.gn_table(ab.to(dev)) if xf else None,
                           transform=ops.XF_AFFINE_SILU if xf else ops.XF_NONE, film=film, res=res, stats=True)
            ops.set_conv_probe(None)
            torch.cuda.synchronize()
        slab, nsplit = y._hsidm_stats
        assert_stats(slab, y, no_sk)
        outs.append(y.float().cpu())
        labels.append(recs[-1]["kernel"])
    assert labels[0].startswith("conv_v2") and labels[1].startswith("conv_sk"), labels
    if case == SK_CASES[0]:                 # at 40 latents (the 8-GPU shard of configs[3]) the persistent kernel is the faster one: no split
        recs = []
        ops.set_conv_probe(recs)
        ops.conv2d(x0.repeat(8, 1, 1, 1), pk, x1=x1.repeat(8, 1, 1, 1), stats=True)
        ops.set_conv_probe(None)
        assert recs[-1]["kernel"].startswith("conv_v2"), recs[-1]["kernel"]
    check("conv_sk_vs_v2%s" % (case,), "bf16", outs[1], outs[0], tol=4e-3)
    xin = torch.cat([x0, x1], dim=3).float().cpu() if C1 else x0.float().cpu()
    a = torch.nn.functional.silu(xin * ab[:, None, None, :, 0] + ab[:, None, None, :, 1]).to(torch.bfloat16).float() if xf else xin
    want = torch.nn.functional.conv2d(a.permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), bias, padding=1) + film.cpu()[:, :, None, None]
    want = want.permute(0, 2, 3, 1) + (res.float().cpu() if with_res else 0)
    check("conv_sk_vs_torch%s" % (case,), "bf16", outs[1], want, tol=6e-3)


SKP_CASES = [  # B, H, W, C (block2's width), P0, P1 (the projection's concat input)
    (5, 8, 8, 512, 512, 512),       # ups[0]-like block at one CAVE image: 8 + 16 chunks
    (5, 16, 16, 512, 512, 256),     # 768-channel concat, 16x16 level
    (2, 32, 32, 256, 200, 0),       # ragged projection chunk (200 channels), 32x32 level
]


@pytest.mark.parametrize("case", SKP_CASES)
def test_split_k_convolution_with_fused_projection(dev, case):
    """ResnetBlock tail (reference unet.py:105-111) in one launch: block2's GroupNorm + SiLU + 3x3 conv plus res_conv's 1x1 projection of
    the block input as extra one-tap chunks of the split-K kernel, against the two-launch form (projection on the GEMM kernel, entering
    as the residual) and against torch."""
    from hsi_dmgasr_amd import ops
    B, H, W, Cc, P0, P1 = case
    g = torch.Generator().manual_seed(sum(case))
    w = torch.randn(Cc, Cc, 3, 3, generator=g) / (9 * Cc) ** 0.5
    bias = torch.randn(Cc, generator=g)
    wp = torch.randn(Cc, P0 + P1, 1, 1, generator=g) / (P0 + P1) ** 0.5
    bp = torch.randn(Cc, generator=g)
    fused = ops.PackedConv(w.to(dev), bias.to(dev), "bf16", proj_weight=wp.to(dev), proj_bias=bp.to(dev))
    assert fused.w_v2 is not None and fused.proj_cin == (P0 + P1 + 7) // 8 * 8
    pk2, pkp = ops.PackedConv(w.to(dev), bias.to(dev), "bf16"), ops.PackedConv(wp.to(dev), bp.to(dev), "bf16")
    hh = torch.randn(B, H, W, Cc, generator=g).to(dev, torch.bfloat16)
    x0 = torch.randn(B, H, W, P0, generator=g).to(dev, torch.bfloat16)
    x1 = torch.randn(B, H, W, P1, generator=g).to(dev, torch.bfloat16) if P1 else None
    ab = torch.stack([1 + 0.1 * torch.randn(B, Cc, generator=g), 0.1 * torch.randn(B, Cc, generator=g)], 2).contiguous()
    tab = ops.gn_table(ab.to(dev))
    recs = []
    ops.set_conv_probe(recs)
    y = ops.conv2d(hh, fused, gn_ab=tab, transform=ops.XF_AFFINE_SILU, proj_x0=x0, proj_x1=x1, stats=True, sk_only=True)
    ops.set_conv_probe(None)
    assert y is not None and recs[-1]["kernel"].startswith("conv_sk"), recs
    torch.cuda.synchronize()
    assert_stats(y._hsidm_stats[0], y, 0)
    r = ops.conv2d(x0, pkp, x1=x1)
    y2 = ops.conv2d(hh, pk2, gn_ab=tab, transform=ops.XF_AFFINE_SILU, res=r, stats=True)
    check("conv_sk_proj_vs_two_launches%s" % (case,), "bf16", y, y2.float().cpu(), tol=5e-3)
    a = torch.nn.functional.silu(hh.float().cpu() * ab[:, None, None, :, 0] + ab[:, None, None, :, 1]).to(torch.bfloat16).float()
    xin = (torch.cat([x0, x1], dim=3) if P1 else x0).float().cpu()
    want = torch.nn.functional.conv2d(a.permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), bias, padding=1)
    want = want + torch.nn.functional.conv2d(xin.permute(0, 3, 1, 2), wp.to(torch.bfloat16).float(), bp)
    check("conv_sk_proj_vs_torch%s" % (case,), "bf16", y, want.permute(0, 2, 3, 1), tol=6e-3)
    # a batch that fills the chip: the descriptor is refused (the caller launches the projection on its own)
    big = ops.conv2d(hh.repeat(16, 1, 1, 1), fused, gn_ab=ops.gn_table(ab.repeat(16, 1, 1).to(dev)), transform=ops.XF_AFFINE_SILU,
                     proj_x0=x0.repeat(16, 1, 1, 1), proj_x1=None if x1 is None else x1.repeat(16, 1, 1, 1), sk_only=True)
    assert big is None


def test_final_block_conv_on_the_256_pixel_kernel(dev, monkeypatch):
    """The UNet's last conv (GroupNorm + SiLU + 3x3, 64 -> 3, fp32 NCHW out; reference unet.py:231,262) on conv_v3<WN = 1, NCHW>
    against the generic kernel (HSIDM_NO_V3=1) and against torch fp32."""
    from hsi_dmgasr_amd import ops
    g = torch.Generator().manual_seed(23)
    B, H, W, Ci, Co = 3, 32, 48, 64, 3
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    bias = torch.randn(Co, generator=g)
    pk = ops.PackedConv(w.to(dev), bias.to(dev), "bf16", out_nchw=True)
    assert pk.w_v2 is not None
    x = torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16)
    ab = torch.stack([1 + 0.1 * torch.randn(B, Ci, generator=g), 0.1 * torch.randn(B, Ci, generator=g)], 2).contiguous()
    outs, labels = [], []
    for no_v3 in (True, False):
        recs = []
        with _lib.debug_switch("NO_V3", int(no_v3)):
            ops.set_conv_probe(recs)
            y = ops.conv2d(x, pk, gn_ab=ops.gn_table(ab.to(dev)), transform=ops.XF_AFFINE_SILU)
            ops.set_conv_probe(None)
            torch.cuda.synchronize()
        assert y.shape == (B, Co, H, W) and y.dtype == torch.float32
        outs.append(y.cpu())
        labels.append(recs[-1]["kernel"])
    assert labels[0].startswith("conv_igemm") and labels[1].startswith("conv_v3") and labels[1].endswith("nchw"), labels
    xf = x.float().cpu()
    act = torch.nn.functional.silu(xf * ab[:, None, None, :, 0] + ab[:, None, None, :, 1]).to(torch.bfloat16).float()
    want = torch.nn.functional.conv2d(act.permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), bias, padding=1)
    check("final_conv_v3_vs_torch", "bf16", outs[1], want, tol=3e-3)
    check("final_conv_v3_vs_v1", "bf16", outs[1], outs[0], tol=3e-3)


UP4_CASES = [  # B, H, W, Cin, Cout  (input grid)
    (4, 16, 32, 128, 128),     # 8x16 tiles, one cout slice, L2-friendly (tile, parity) order (16 tiles % 8 == 0)
    (3, 16, 32, 64, 128),      # 12 tiles: plain order
    (8, 8, 8, 128, 256),       # two-image 8x8 tiles, two cout slices, 4 XCDs per slice
    (5, 8, 8, 64, 256),        # odd batch: a half-empty two-image tile
    (2, 12, 20, 64, 128),      # partial tiles on both axes
    (1, 8, 16, 72, 100),       # channel counts that are not multiples of the chunk / slice
]


@pytest.mark.parametrize("case", UP4_CASES)
def test_folded_upsample_conv_matches_addressed_upsample(dev, case):
    """HSIDM_UPS_FOLDED (four parity 2x2 kernels, K = 4*Cin) against HSIDM_UPS_ADDRESS (3x3 on in[y>>1][x>>1]) and against
    torch's fp32 conv3x3(nearest_x2(x)); statistics slab against sums of the stored output."""
    from hsi_dmgasr_amd import ops
    B, H, W, Ci, Co = case
    g = torch.Generator().manual_seed(sum(case))
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    bias = torch.randn(Co, generator=g)
    pk = ops.PackedConv(w.to(dev), bias.to(dev), "bf16", fold_ups=True)
    assert pk.w_up4 is not None and pk.w_up4.shape[0] == 16 * ((Ci + 63) // 64)
    x = torch.randn(B, H, W, Ci, generator=g).to(torch.bfloat16)
    outs = []
    for fold in (False, True):
        ops.set_fold_ups(fold)
        try:
            y = ops.conv2d(x.to(dev), pk, ups=True, stats=True)
            torch.cuda.synchronize()
        finally:
            ops.set_fold_ups(True)
        slab, nsplit = y._hsidm_stats
        yf = y.float()
        assert y.shape == (B, 2 * H, 2 * W, Co) and slab.shape == (B, nsplit, Co, 2)
        assert_stats(slab, y, fold)
        outs.append(yf.cpu())
    ref = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="nearest"),
                                     w, bias, padding=1).permute(0, 2, 3, 1)
    check("up4_vs_addressed%s" % (case,), "bf16", outs[1], outs[0], tol=6e-3)
    check("up4_vs_torch%s" % (case,), "bf16", outs[1], ref, tol=1e-2)
    check("ups_vs_torch%s" % (case,), "bf16", outs[0], ref, tol=1e-2)


DN4_CASES = [  # B, H, W, Cin, Cout  (input grid, even)
    (4, 32, 32, 64, 64),       # 64-cout slices, 8x16 output tiles
    (3, 16, 16, 128, 128),     # 8x8 output map: two-image tiles
    (2, 24, 40, 72, 100),      # partial tiles, channel counts off the chunk / slice grid
    (5, 16, 32, 256, 256),     # two cout slices, four channel chunks per plane
]


@pytest.mark.parametrize("case", DN4_CASES)
def test_stride2_conv_over_parity_planes_matches_v1_and_torch(dev, case):
    """Downsample conv (3x3, stride 2, pad 1) on the conv_v2 schedule over the four input-parity planes against the generic
    kernel and torch fp32; statistics slab against sums of the stored output."""
    from hsi_dmgasr_amd import ops
    B, H, W, Ci, Co = case
    g = torch.Generator().manual_seed(sum(case))
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    bias = torch.randn(Co, generator=g)
    pk = ops.PackedConv(w.to(dev), bias.to(dev), "bf16", fold_dn=True)
    assert pk.w_dn4 is not None and pk.w_dn4.shape[0] == 16 * ((Ci + 63) // 64)
    x = torch.randn(B, H, W, Ci, generator=g).to(torch.bfloat16)
    outs = []
    for use_v2 in (False, True):
        ops.set_use_v2(use_v2)
        try:
            y = ops.conv2d(x.to(dev), pk, stride=2, stats=True)
            torch.cuda.synchronize()
        finally:
            ops.set_use_v2(True)
        slab, nsplit = y._hsidm_stats
        yf = y.float()
        assert y.shape == (B, H // 2, W // 2, Co)
        assert_stats(slab, y, use_v2)
        outs.append(yf.cpu())
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w, bias, stride=2, padding=1).permute(0, 2, 3, 1)
    check("dn4_vs_v1%s" % (case,), "bf16", outs[1], outs[0], tol=4e-3)
    check("dn4_vs_torch%s" % (case,), "bf16", outs[1], ref, tol=1e-2)


@pytest.mark.parametrize("case", [(5, 16, 16, 512, 512), (5, 32, 32, 256, 256), (3, 16, 32, 328, 128)])
def test_split_k_stride2_convolution(dev, case):
    """Downsample conv (3x3, stride 2, pad 1) at one CAVE image per GPU: the split-K kernel's stride-2 form (17x17 input halo, the
    parity-plane weights addressed per 3x3 tap) against the persistent plane-wise kernel (debug switch NO_SPLIT_K) and torch."""
    from hsi_dmgasr_amd import ops
    B, H, W, Ci, Co = case
    g = torch.Generator().manual_seed(sum(case))
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    bias = torch.randn(Co, generator=g)
    pk = ops.PackedConv(w.to(dev), bias.to(dev), "bf16", fold_dn=True)
    x = torch.randn(B, H, W, Ci, generator=g).to(dev, torch.bfloat16)
    res = torch.randn(B, H // 2, W // 2, Co, generator=g).to(dev, torch.bfloat16)       # the training step's input gradient adds one
    outs, labels = [], []
    for no_sk in (1, 0):
        recs = []
        with _lib.debug_switch("NO_SPLIT_K", no_sk):
            ops.set_conv_probe(recs)
            y = ops.conv2d(x, pk, stride=2, res=res, stats=True)
            ops.set_conv_probe(None)
            torch.cuda.synchronize()
        assert_stats(y._hsidm_stats[0], y, no_sk)
        outs.append(y.float().cpu())
        labels.append(recs[-1]["kernel"])
    assert labels[0].startswith("conv_v2") and labels[1].startswith("conv_sk"), labels
    ref = torch.nn.functional.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), bias, stride=2, padding=1)
    ref = ref.permute(0, 2, 3, 1) + res.float().cpu()
    check("conv_sk_s2_vs_v2%s" % (case,), "bf16", outs[1], outs[0], tol=4e-3)
    check("conv_sk_s2_vs_torch%s" % (case,), "bf16", outs[1], ref, tol=6e-3)


@pytest.mark.parametrize("case", [(3, 24, 16, 6, 64), (3, 16, 24, 6, 64), (2, 16, 8, 8, 128), (1, 16, 16, 6, 32), (2, 32, 32, 5, 64), (3, 8, 8, 8, 64)])
def test_eight_channel_conv_as_tap_major_gemm(dev, case):
    """3x3 convs with <= 8 input channels (the UNet stem) run as a K = 72 GEMM (conv1x1_g, im2col on the fly); shapes the
    GEMM kernel refuses (32 couts) stay on the generic kernel.  Against v1 and torch fp32, with statistics."""
    from hsi_dmgasr_amd import ops
    B, H, W, Ci, Co = case
    g = torch.Generator().manual_seed(sum(case))
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    bias = torch.randn(Co, generator=g)
    pk = ops.PackedConv(w.to(dev), bias.to(dev), "bf16")
    assert pk.tap_major and pk.cin == 8
    x = torch.zeros(B, H, W, 8)
    x[..., :Ci] = torch.randn(B, H, W, Ci, generator=g)
    x = x.to(torch.bfloat16)
    outs = []
    for use_v2 in (False, True):
        ops.set_use_v2(use_v2)
        try:
            y = ops.conv2d(x.to(dev), pk, stats=True)
            torch.cuda.synchronize()
        finally:
            ops.set_use_v2(True)
        slab, nsplit = y._hsidm_stats
        yf = y.float()
        assert_stats(slab, y, use_v2)
        outs.append(yf.cpu())
    ref = torch.nn.functional.conv2d(x.float()[..., :Ci].permute(0, 3, 1, 2), w, bias, padding=1).permute(0, 2, 3, 1)
    check("stem_gemm_vs_v1%s" % (case,), "bf16", outs[1], outs[0], tol=4e-3)
    check("stem_gemm_vs_torch%s" % (case,), "bf16", outs[1], ref, tol=1e-2)


C1_CASES = [  # B, H, W, C0, C1, Cout, with_res, gn_affine
    (2, 16, 16, 64, 0, 128, False, False),       # K=64 padded to 128, one 128-cout slice
    (3, 8, 8, 128, 64, 64, True, False),         # concat input, K=192 -> 256, residual + statistics; 192 pixels: half-empty last tile
    (2, 16, 8, 256, 256, 512, False, False),     # K=512, four cout slices
    (5, 8, 16, 512, 512, 96, True, False),       # Cout not a multiple of the slice: stays on the v1 kernel (dispatch check)
    (3, 16, 16, 512, 0, 1536, False, True),      # attention qkv: GroupNorm affine prologue, 12 cout slices
    (5, 8, 8, 64, 0, 192, False, True),          # two images per 128-pixel tile: per-half GroupNorm parameters; Cout 192 -> v1
    (7, 8, 8, 128, 0, 256, True, True),          # same with whole slices (448 pixels: half-empty last tile)
    (2, 32, 32, 72, 40, 128, True, False),       # channel counts that are not multiples of the 64-channel chunk
]


@pytest.mark.parametrize("case", C1_CASES)
def test_conv1x1_gemm_kernels_match_v1(dev, case):
    """1x1 convolutions on the LDS-staged GEMM kernel (conv1x1_g) against the generic v1 kernel, statistics included."""
    from hsi_dmgasr_amd import ops
    B, H, W, C0, C1, Co, with_res, affine = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    w = torch.randn(Co, C0 + C1, 1, 1, generator=g) / (C0 + C1) ** 0.5
    pk = ops.PackedConv(w.to(dev), torch.randn(Co, generator=g).to(dev), "bf16")
    assert pk.w_v2 is not None and pk.w_v2.shape[0] % 2 == 0
    x0 = torch.randn(B, H, W, C0, generator=g).to(dev, torch.bfloat16)
    x1 = torch.randn(B, H, W, C1, generator=g).to(dev, torch.bfloat16) if C1 else None
    res = torch.randn(B, H, W, Co, generator=g).to(dev, torch.bfloat16) if with_res else None
    ab = torch.stack([1 + 0.2 * torch.randn(B, C0 + C1, generator=g), 0.3 * torch.randn(B, C0 + C1, generator=g)], 2).contiguous().to(dev)
    outs = []
    for use_v2 in (False, True):
        ops.set_use_v2(use_v2)
        y = ops.conv2d(x0, pk, x1=x1, res=res, stats=True, gn_ab=ops.gn_table(ab) if affine else None,
                       transform=ops.XF_AFFINE if affine else ops.XF_NONE)
        torch.cuda.synchronize()
        slab, nsplit = y._hsidm_stats
        yf = y.float()
        assert_stats(slab, y, use_v2)
        outs.append(yf.cpu())
    ops.set_use_v2(True)
    # with a residual the vector epilogue rounds the conv result to bf16 before the add (v1 adds in fp32): <= 1 bf16 ulp
    check("conv1x1_g_vs_v1%s" % (case,), "bf16", outs[1], outs[0], tol=4e-3)


def test_split_k_scratch_outlives_the_graphs_that_point_into_it(dev):
    """The split-K scratch buffer's address is baked into captured steps, and the size it needs is not monotonic in the batch: a later
    launch that needs more must not free the buffer a live graph still writes to.  Capture a run at 5 latents, make the scratch
    grow, drop every other reference, replay: same result as an uninterrupted run."""
    import gc
    from hsi_dmgasr_amd import ops
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    cfg = dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8], attn_res=[4], res_blocks=1, image_size=32)
    u = unet.UNet(precision="bf16", **cfg).to(dev).eval()
    fill_synth(u, "sk_life.")
    gd = diffusion.GaussianDiffusion(u, image_size=32, channels=3, conditional=True)
    gd.set_loss(dev)
    gd.set_new_noise_schedule(dict(schedule="cosine", n_timestep=12, linear_start=1e-6, linear_end=1e-2), dev)
    gd.noise, gd.seed = "philox", 77
    cond5 = G(synth_tensor("sk_life.cond", (5, 3, 32, 32)), dev)
    outs = []
    for disturb in (False, True):
        ops._sk_ws.clear()
        run = gd.make_run(cond5)
        with torch.no_grad():
            for _ in range(4):
                run.step()                                     # eager, capture, replays
            assert run.graph is not None
            before = ops._sk_ws.get(dev.index)
            assert before is not None, "the 4x4 / 8x8 levels at 5 latents must take the split-K form"
            if disturb:
                # (on this chip the scratch never outgrows its first 32 MiB - parts x tiles x slices <= 2 x CUs workgroups of 32 KiB
                # each - so a larger request is made directly, as a launch on a bigger device or a future kernel would)
                grown = ops._sk_workspace(before.numel() + 1, dev)
                assert grown is not before and ops._sk_ws[dev.index] is grown and any(b is before for b in ops._sk_retired)
                del before
                gc.collect()
                torch.cuda.empty_cache()
                junk = torch.full((grown.numel() // 8,), float("nan"), device=dev)          # would land in a freed block
                del junk
            for _ in range(8):
                run.step()
            torch.cuda.synchronize()
        outs.append(run.x.clone())
    assert torch.isfinite(outs[1]).all() and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_unconditional_sample_matches_oracle(dev, prec):
    """GaussianDiffusion(conditional=False).sample() (reference diffusion.py:182-189, 203-207): the UNet sees x_t alone (in_channel = 3),
    ret_img starts with x_T; Philox noise, continous frames and the ret_img[-1] convention against the oracle's restatement."""
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    from oracle import diffusion as odiff, sr3_unet
    cfg = dict(in_channel=3, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=[1, 2], attn_res=[8], res_blocks=1, image_size=16)
    u = unet.UNet(precision=prec, **cfg).to(dev).eval()
    sd = fill_synth(u, "unet_uncond.")
    opt = dict(schedule="linear", n_timestep=12, linear_start=1e-4, linear_end=2e-2)
    gd = diffusion.GaussianDiffusion(u, image_size=16, channels=3, conditional=False)
    gd.set_loss(dev)
    gd.set_new_noise_schedule(opt, dev)
    gd.noise, gd.seed = "philox", 4242
    B = 3
    frames = gd.sample(batch_size=B, continous=True)
    last = gd.sample(batch_size=B, continous=False)
    torch.cuda.synchronize()
    sched = odiff.noise_schedule(opt)
    den = lambda x, gam: sr3_unet.unet_forward(sd, cfg, x, gam)
    nf = odiff.philox_noise_fn(4242, (B, 3, 16, 16))
    with torch.no_grad():
        want = odiff.p_sample_loop_unconditional(den, sched, nf(12), nf, continous=True)
    assert frames.shape == want.shape == ((1 + 12) * B, 3, 16, 16)          # inter = 1 | (12 // 10) = 1: x_T and every step
    tol = {"fp32": 1e-3, "fp16": 1e-3}[prec]               # north_star (measured 3e-6 / 1.5e-4)
    check("uncond_sample_frames", prec, frames, want, tol=tol)
    check("uncond_sample_last", prec, last, want[-1], tol=tol)
