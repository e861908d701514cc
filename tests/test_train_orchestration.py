"""CPU: the training step's orchestration (hsi_dmgasr_amd/training.py) with the kernel wrappers replaced by the torch doubles
of tests/cpu_double.py, against the oracle's autograd gradients (oracle/train.py, pinned to the reference by grads.npz).
Covers the tape / skip-connection bookkeeping, the flat buffers, the pack-map gather and the Adam step - not the kernels
(those: tests/test_gpu_train.py)."""
import numpy as np
import pytest
import torch

import cpu_double
from gpu_util import fill_synth
from helpers import jload, load_npz, synth_tensor

CFGS = {
    "tiny": dict(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=[1, 2], attn_res=[8], res_blocks=1,
                 image_size=16),
    # 3 levels, 2 blocks per level, 2 channels per group, attention at the lowest level: every kind of skip / concat seam
    "mid": dict(in_channel=6, out_channel=3, inner_channel=32, norm_groups=16, channel_mults=[1, 2, 2], attn_res=[4], res_blocks=2,
                image_size=16),
}


def build(cfg_name, kind, train_mode):
    from hsi_dmgasr_amd import training
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet

    class CpuTrainer(training.Trainer):
        def _check_device(self):
            pass

    cfg = CFGS[cfg_name]
    u = unet.UNet(dropout=0.2, precision="fp32", **cfg)
    sd = fill_synth(u, "unet_%s." % cfg_name)
    u.train(train_mode)
    gd = diffusion.GaussianDiffusion(u, image_size=16, channels=3, loss_type=kind, conditional=True)
    gd.set_new_noise_schedule(dict(schedule="cosine", n_timestep=20, linear_start=1e-6, linear_end=1e-2), "cpu")
    return cfg, sd, gd, CpuTrainer(gd, lr=1e-3, dropout_seed=9)


def compare(tr, grads, tol=2e-3):
    scale = max(float(g.norm()) for g in grads.values())
    worst = 0.0
    for name, p in tr.net.named_parameters():
        got, want = tr.G(p), grads[name]
        # (gradients that are mathematically zero - a per-channel constant ahead of a one-channel-per-group GroupNorm - are
        # rounding noise of relative size 1e-8: the floor keeps them out of the ratio)
        err = float((got - want).norm()) / (float(want.norm()) + 1e-4 * scale)
        worst = max(worst, err)
        assert err < tol, (name, err)
    return worst


@pytest.mark.parametrize("cfg_name,kind,train_mode", [("tiny", "l1", False), ("tiny", "l2", True), ("mid", "l1", True)])
def test_training_step_orchestration_matches_oracle_autograd(monkeypatch, cfg_name, kind, train_mode):
    from oracle import train as otrain
    cpu_double.install(monkeypatch)
    cfg, sd, gd, tr = build(cfg_name, kind, train_mode)
    # every parameter now lives in the flat buffer, FiLM tables contiguous
    assert all(p.data_ptr() >= tr.flat.data_ptr() and p.data_ptr() < tr.flat.data_ptr() + 4 * tr.flat.numel() for p in tr.net.parameters())
    hr, sr, noise = (torch.from_numpy(synth_tensor("orch.%s" % n, (2, 3, 16, 16))) for n in ("hr", "sr", "noise"))
    gamma = torch.tensor([0.83, 0.31])
    loss = tr.loss_and_grads({"HR": hr, "SR": sr}, noise=noise, gamma=gamma)
    want_loss, grads = otrain.loss_and_grads(sd, cfg, hr, sr, noise, gamma, kind, 0.2 if train_mode else 0.0, otrain.drop_key(9, 0))
    assert abs(float(loss) - want_loss) < 1e-4 * abs(want_loss)
    compare(tr, grads)
    assert all(p.grad is tr.G(p) for p in tr.net.parameters())


def test_three_optimiser_steps_match_torch_adam(monkeypatch):
    """Trainer.optimize_parameters x3 (dropout on: a new mask per step) against torch.optim.Adam driven by the oracle's gradients."""
    from oracle import train as otrain
    cpu_double.install(monkeypatch)
    cfg, sd, gd, tr = build("tiny", "l1", True)
    hr, sr, noise = (torch.from_numpy(synth_tensor("orch.%s" % n, (2, 3, 16, 16))) for n in ("hr", "sr", "noise"))
    gamma = torch.tensor([0.6, 0.2])
    for _ in range(3):
        tr.optimize_parameters({"HR": hr, "SR": sr}, noise=noise, gamma=gamma)
    want = otrain.adam_steps(sd, lambda s, ps: otrain.loss_and_grads(ps, cfg, hr, sr, noise, gamma, "l1", 0.2, otrain.drop_key(9, s))[1],
                             3, lr=1e-3)
    # Adam normalises the gradient, so parameters whose gradient is mathematically zero (rounding noise of either
    # implementation, see compare()) move by +-lr in noise-determined directions: not comparable, left out
    g0 = otrain.loss_and_grads(sd, cfg, hr, sr, noise, gamma, "l1", 0.2, otrain.drop_key(9, 0))[1]
    scale = max(float(g.norm()) for g in g0.values())
    checked = 0
    for name, p in tr.net.named_parameters():
        if float(g0[name].norm()) < 1e-4 * scale:
            continue
        moved = float((want[name] - sd[name]).norm())
        assert float((p.detach() - want[name]).norm()) < 2e-2 * moved + 1e-7, name
        checked += 1
    assert tr.step_count == 3 and checked > 100


def test_pack_maps_reproduce_the_inference_packing():
    """The gather maps (ops.pack_layouts on index-valued tensors) give bit-identical buffers to ops.PackedConv on the weights."""
    import types
    from hsi_dmgasr_amd import ops, training
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet

    class CpuTrainer(training.Trainer):
        def _check_device(self):
            pass

        def repack(self):
            cpu_double.gather_pack(self.flat, self._pack_idx, self._pack_hi, self._pack_lo)

    for prec in ("bf16", "fp32"):
        u = unet.UNet(dropout=0.2, precision=prec, **CFGS["tiny"])
        fill_synth(u, "unet_tiny.")
        gd = diffusion.GaussianDiffusion(u, image_size=16, channels=3, conditional=True)
        tr = CpuTrainer(gd)
        for conv, out_nchw, fold_dn, need_dg in tr._convs():
            ref = ops.PackedConv(conv.weight, conv.bias, prec, out_nchw=out_nchw, fold_dn=fold_dn)
            got = tr.pk(conv)
            assert torch.equal(got.w_hi, ref.w_hi) and got.bn == ref.bn and got.cin == ref.cin
            assert (got.w_lo is None) == (ref.w_lo is None) and (got.w_lo is None or torch.equal(got.w_lo, ref.w_lo))
            assert (got.w_v2 is None) == (ref.w_v2 is None) and (got.w_v2 is None or torch.equal(got.w_v2, ref.w_v2))
            assert (got.w_dn4 is None) == (ref.w_dn4 is None) and (got.w_dn4 is None or torch.equal(got.w_dn4, ref.w_dn4))
            if need_dg:
                wt = conv.weight.detach().transpose(0, 1).flip(2, 3).contiguous()
                refd = ops.PackedConv(wt, None, prec)
                gd_ = tr.dpk(conv)
                assert torch.equal(gd_.w_hi, refd.w_hi) and (refd.w_v2 is None or torch.equal(gd_.w_v2, refd.w_v2))


def test_reference_training_loop_runs_unchanged_with_a_torch_optimiser(monkeypatch):
    """model/model.py:49-59 verbatim - optG.zero_grad(); l_pix = netG(data); l_pix = l_pix.sum() / n; l_pix.backward(); optG.step() -
    with torch.optim.Adam over the module's parameters: one autograd node wraps the hand-written backward pass.  Two steps, dropout
    on, t / gamma drawn from numpy as the reference draws them; against torch.optim.Adam driven by the oracle's gradients."""
    from hsi_dmgasr_amd import training
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    from oracle import train as otrain
    cpu_double.install(monkeypatch)
    monkeypatch.setattr(training.Trainer, "_check_device", lambda self: None)
    cfg = CFGS["tiny"]
    u = unet.UNet(dropout=0.2, precision="fp32", **cfg)
    sd = fill_synth(u, "unet_tiny.")
    gd = diffusion.GaussianDiffusion(u, image_size=16, channels=3, loss_type="l1", conditional=True)
    gd.set_new_noise_schedule(dict(schedule="cosine", n_timestep=20, linear_start=1e-6, linear_end=1e-2), "cpu")
    gd.train()
    optG = torch.optim.Adam(list(gd.parameters()), lr=1e-3)                    # model/model.py:37-41
    hr, sr, noise = (torch.from_numpy(synth_tensor("orch.%s" % n, (2, 3, 16, 16))) for n in ("hr", "sr", "noise"))
    data = {"HR": hr, "SR": sr}
    gammas = []
    for step in range(2):
        np.random.seed(100 + step)
        t = np.random.randint(1, gd.num_timesteps + 1)
        gammas.append(torch.FloatTensor(np.random.uniform(gd.sqrt_alphas_cumprod_prev[t - 1], gd.sqrt_alphas_cumprod_prev[t], size=2)))
        np.random.seed(100 + step)
        optG.zero_grad()
        l_pix = gd(data, noise=noise)
        b, c, h, w = hr.shape
        l_pix = l_pix.sum() / int(b * c * h * w)
        l_pix.backward()
        optG.step()
        if step == 0:
            want_loss = otrain.l_pix(sd, cfg, hr, sr, noise, gammas[0], "l1", 0.2, otrain.drop_key(0, 0))
            assert abs(float(l_pix.detach()) - float(want_loss)) < 1e-4 * abs(float(want_loss))
    want = otrain.adam_steps(sd, lambda s, ps: otrain.loss_and_grads(ps, cfg, hr, sr, noise, gammas[s], "l1", 0.2, otrain.drop_key(0, s))[1],
                             2, lr=1e-3)
    g0 = otrain.loss_and_grads(sd, cfg, hr, sr, noise, gammas[0], "l1", 0.2, otrain.drop_key(0, 0))[1]
    scale = max(float(g.norm()) for g in g0.values())
    checked = 0
    for name, p in u.named_parameters():
        if float(g0[name].norm()) < 1e-4 * scale:
            continue
        moved = float((want[name] - sd[name]).norm())
        assert float((p.detach() - want[name]).norm()) < 2e-2 * moved + 1e-7, name
        checked += 1
    assert checked > 100
    # evaluation in train() mode under no_grad: a value with Dropout active; eval(): the fused inference kernels' value
    with torch.no_grad():
        assert gd(data, noise=noise).ndim == 0


def _ddp_worker(rank, world, port, out_dir):
    import os
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import pytest as _pytest
    import cpu_double as cd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    mp_ = _pytest.MonkeyPatch()
    cd.install(mp_)
    cfg, sd, gd, tr = build("tiny", "l1", True)
    tr.bucket_bytes = 256 << 10                     # several buckets: exercises the overlap bookkeeping
    hr, sr, noise = (torch.from_numpy(synth_tensor("ddp.r%d.%s" % (rank, n), (2, 3, 16, 16))) for n in ("hr", "sr", "noise"))
    gamma = torch.tensor([0.7, 0.25]) - 0.1 * rank
    loss = tr.optimize_parameters({"HR": hr, "SR": sr}, noise=noise, gamma=gamma)
    fired = len(tr.reducer.buckets)
    torch.save({"flat": tr.flat.clone(), "loss": float(loss), "buckets": fired}, os.path.join(out_dir, "r%d.pt" % rank))
    mp_.undo()
    dist.destroy_process_group()


def test_two_rank_data_parallel_training_step_on_gloo(tmp_path):
    """BASELINE configs[4] on two CPU ranks (gloo, kernel doubles): every rank computes gradients on its own data, the bucketed
    all-reduce (parallel.GradReducer, launched from inside the backward pass) sums them, Adam applies their mean: both ranks end
    with identical parameters, equal to torch.optim.Adam on the oracle's averaged gradients."""
    import os
    import torch.multiprocessing as mp
    from oracle import train as otrain
    port = 29500 + (os.getpid() + 7) % 2000
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["flat"], r1["flat"]) and r0["buckets"] > 3
    cfg = CFGS["tiny"]
    from helpers import synth_sd
    from oracle import sr3_unet
    sd = synth_sd(sr3_unet.unet_param_shapes(cfg), "unet_tiny.")
    gs = []
    for rank in range(2):
        hr, sr, noise = (torch.from_numpy(synth_tensor("ddp.r%d.%s" % (rank, n), (2, 3, 16, 16))) for n in ("hr", "sr", "noise"))
        gs.append(otrain.loss_and_grads(sd, cfg, hr, sr, noise, torch.tensor([0.7, 0.25]) - 0.1 * rank, "l1", 0.2, otrain.drop_key(9, 0, rank)))
    assert abs(r0["loss"] - gs[0][0]) < 1e-4 * abs(gs[0][0]) and abs(r1["loss"] - gs[1][0]) < 1e-4 * abs(gs[1][0])
    mean = {k: 0.5 * (gs[0][1][k] + gs[1][1][k]) for k in sd}
    want = otrain.adam_steps(sd, lambda s, ps: mean, 1, lr=1e-3)
    # rebuild the name -> flat offset map the workers used
    cfg2, sd2, gd2, tr2 = None, None, None, None
    import cpu_double as cd
    mp_ = pytest.MonkeyPatch()
    cd.install(mp_)
    try:
        _, _, _, tr = build("tiny", "l1", True)
    finally:
        mp_.undo()
    scale = max(float(g.norm()) for g in mean.values())
    checked = 0
    for name, p in tr.net.named_parameters():
        if float(mean[name].norm()) < 1e-4 * scale:
            continue
        o = tr._off[id(p)]
        got = torch.as_strided(r0["flat"], p.shape, p.stride(), o)      # (3x3 weights sit in channels-last order in the flat buffer)
        moved = float((want[name] - sd[name]).norm())
        assert float((got - want[name]).norm()) < 2e-2 * moved + 1e-7, name
        checked += 1
    assert checked > 100


def test_trainer_rejects_what_its_backward_pass_does_not_cover(monkeypatch):
    """The constructor variants with forward kernels only (SelfAttention(n_head > 1), FeatureWiseAffine(use_affine_level=True))
    and the fp16 inference mode must be refused by the training engine instead of producing wrong gradients."""
    from hsi_dmgasr_amd import training
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    cpu_double.install(monkeypatch)

    class CpuTrainer(training.Trainer):
        def _check_device(self):
            pass

    def gd_of(u):
        gd = diffusion.GaussianDiffusion(u, image_size=16, channels=3, conditional=True)
        gd.set_new_noise_schedule(dict(schedule="cosine", n_timestep=20, linear_start=1e-6, linear_end=1e-2), "cpu")
        return gd
    u = unet.UNet(precision="fp32", **CFGS["tiny"])
    u.mid[0].attn.n_head = 4
    with pytest.raises(NotImplementedError, match="n_head"):
        CpuTrainer(gd_of(u))
    u = unet.UNet(precision="fp32", **CFGS["tiny"])
    u.downs[1].res_block.noise_func.use_affine_level = True
    with pytest.raises(NotImplementedError, match="use_affine_level"):
        CpuTrainer(gd_of(u))
    u = unet.UNet(precision="fp16", **CFGS["tiny"])
    with pytest.raises(NotImplementedError, match="bf16 or the fp32"):
        CpuTrainer(gd_of(u), precision="fp16")
    assert CpuTrainer(gd_of(u)).precision == "fp32"          # inherited from an inference-only mode: trains as the reference does
    assert CpuTrainer(gd_of(unet.UNet(**CFGS["tiny"]))).precision == "fp32"          # ... and so does the package default


def test_optimizer_state_round_trips_through_torch_adam_format(monkeypatch, tmp_path):
    """Trainer.state_dict() is a torch.optim.Adam state_dict over netG.parameters() (what the reference stores in *_opt.pth,
    model/model.py:140-143): after two steps it loads into a real torch.optim.Adam, that optimiser's own state_dict loads back,
    and a resumed trainer continues exactly like the one that never stopped; save_network / load_network write and read the
    reference's two files with its key-drop filter (model/model.py:187-191)."""
    from oracle import train as otrain
    cpu_double.install(monkeypatch)
    hr, sr, noise = (torch.from_numpy(synth_tensor("ckpt.%s" % n, (2, 3, 16, 16))) for n in ("hr", "sr", "noise"))
    gamma = torch.tensor([0.6, 0.3])
    data = {"HR": hr, "SR": sr}

    def step(tr):
        tr.loss_and_grads(data, noise=noise, gamma=gamma)
        tr.optimizer_step()

    _, _, gd_a, a = build("tiny", "l1", True)
    step(a); step(a)
    sd = a.state_dict()
    names = [n for n, _ in a.net.named_parameters()]
    assert sorted(sd["state"]) == list(range(len(names))) and sd["param_groups"][0]["params"] == list(range(len(names)))
    assert all(float(st["step"]) == 2.0 for st in sd["state"].values())
    # ... into torch's own Adam over clones of the parameters, and back out
    clones = [torch.nn.Parameter(p.detach().clone()) for p in a.net.parameters()]
    opt = torch.optim.Adam(clones, lr=1e-3)
    opt.load_state_dict({"state": sd["state"], "param_groups": sd["param_groups"]})
    back = opt.state_dict()
    for i, p in enumerate(a.net.parameters()):
        assert torch.equal(back["state"][i]["exp_avg"], a._strided_like(p, a.m).contiguous())
    # resume: files on disk, a fresh model, the optimiser restored -> the third step equals the uninterrupted run's
    gen, optp = a.save_network(str(tmp_path / "I2_E0"), epoch=0, iter_step=2)
    assert gen.endswith("_gen.pth") and optp.endswith("_opt.pth")
    on_disk = torch.load(optp, weights_only=False)
    assert on_disk["epoch"] == 0 and on_disk["iter"] == 2 and on_disk["scheduler"] is None
    _, _, gd_b, b = build("tiny", "l1", True)
    with torch.no_grad():
        for p in b.net.parameters():
            p.mul_(0.5)                                                          # anything but the saved weights
    assert b.load_network(str(tmp_path / "I2_E0"), drop_stem_and_final=False, load_optimizer=True) == (0, 2)
    assert b.step_count == 2 and b._iter == a._iter
    step(a); step(b)
    assert torch.allclose(a.flat, b.flat, rtol=0, atol=1e-7)
    # the reference's shipped filter: stem weight and final conv are NOT taken from the file, the optimiser is not restored
    _, sd0, gd_c, c = build("tiny", "l1", True)
    stem0 = c.net.downs[0].weight.detach().clone()
    assert c.load_network(str(tmp_path / "I2_E0")) == (0, 0)
    assert torch.equal(c.net.downs[0].weight, stem0) and c.step_count == 0
    assert torch.allclose(c.net.downs[1].res_block.block1.block[3].weight, a_weight_at_save(on_disk, gen), atol=0)


def a_weight_at_save(_opt, gen_path):
    return torch.load(gen_path)["denoise_fn.downs.1.res_block.block1.block.3.weight"]


def test_graphed_step_captures_only_right_after_an_eager_step_of_the_same_shape(monkeypatch):
    """The reference's DataLoader has no drop_last: full, full, partial, full, full.  A capture must directly follow an eager step
    at the same batch shape (it sizes every workspace the captured kernels point into); here the capture itself is stubbed and
    the sequence of eager / capture / replay decisions is what is checked."""
    cpu_double.install(monkeypatch)
    _, _, gd, tr = build("tiny", "l1", True)
    log = []
    monkeypatch.setattr(type(tr), "loss_and_grads", lambda self, data, **kw: (log.append(("eager", data["HR"].shape[0])), setattr(self, "_eager_shape", None), torch.zeros(()))[-1])
    monkeypatch.setattr(type(tr), "optimizer_step", lambda self: None)

    def fake_graph_path(self, data):                      # what _graphed_step does once it has decided NOT to run eagerly
        log.append(("graph", data["HR"].shape[0]))
        self._g = dict(shape=(tuple(data["HR"].shape), tuple(data["SR"].shape)))
        return torch.zeros(())
    orig = type(tr)._graphed_step

    def wrapped(self, data):
        shape_key = (tuple(data["HR"].shape), tuple(data["SR"].shape))
        if self._g is not None and self._g["shape"] != shape_key:
            self._g = None
        if self._g is None and self._eager_shape != shape_key:
            return orig(self, data)                       # the real decision code: eager branch
        return fake_graph_path(self, data)
    monkeypatch.setattr(type(tr), "_graphed_step", wrapped)
    full = {"HR": torch.zeros(4, 3, 16, 16), "SR": torch.zeros(4, 3, 16, 16)}
    part = {"HR": torch.zeros(1, 3, 16, 16), "SR": torch.zeros(1, 3, 16, 16)}
    for d in (full, full, full, part, full, full):
        tr._graphed_step(d)
    assert log == [("eager", 4), ("graph", 4), ("graph", 4), ("eager", 1), ("eager", 4), ("graph", 4)], log
