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
