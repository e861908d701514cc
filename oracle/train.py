"""Oracle: the training objective, its gradient and the optimiser step (fp32, CPU, torch.autograd over the functional
restatement of the UNet).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates the training side of the reference:
  * GaussianDiffusion.p_losses (model/sr3_modules/diffusion.py:222-250) with injected (gamma, noise);
  * Block.forward in training mode (model/sr3_modules/unet.py:83-91): Dropout(p) between Swish and the convolution of every
    ResnetBlock's block2 (unet.py:100-101).  torch's dropout stream cannot be reproduced on another device, so - like the
    sampler noise - the product draws its masks from the stateless Philox generator of oracle/philox.py and this oracle
    regenerates the very same masks;
  * DDPM.optimize_parameters (model/model.py:49-59): l_pix = loss.sum() / (b*c*h*w); backward; Adam step (model.py:37-41).
Pinned by tests/golden/grads.npz (reference autograd gradients, dropout off).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import philox, sr3_unet


def drop_key(dropout_seed, step_count, rank=0):
    """Philox key of the dropout masks of the `step_count`-th (0-based) training forward pass on data-parallel rank `rank`, as
    training.Trainer._drop_key (replicas draw independent masks, as the reference's nn.DataParallel replicas do)."""
    return (int(dropout_seed) + 0x9E3779B97F4A7C15 * (int(step_count) + 1) + 0xD1B54A32D192ED03 * int(rank)) & 0xFFFFFFFFFFFFFFFF


def dropout_factor(key, layer, shape_nchw, p):
    """Keep/scale factors (0 or 1/(1-p)) of an activation tensor of NCHW shape (B, C, H, W): element e = flat NHWC index
    keeps its value iff word (e & 3) of Philox4x32-10(key, counter = (e >> 2, 0, layer, 0)) >= p * 2^32 (p as float32)."""
    B, C, H, W = shape_nchw
    n = B * H * W * C
    nq = (n + 3) // 4
    q = np.arange(nq, dtype=np.uint64)
    w = philox.philox4x32((q & np.uint64(0xFFFFFFFF)).astype(np.uint32), (q >> np.uint64(32)).astype(np.uint32),
                          np.full(nq, layer, dtype=np.uint32), np.zeros(nq, dtype=np.uint32), key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF)
    words = np.stack(w, axis=1).reshape(-1)[:n]
    p32 = np.float32(p)
    thresh = min(int(float(p32) * 4294967296.0), 4294967295)
    inv_keep = np.float32(1.0) / (np.float32(1.0) - p32)
    f = np.where(words >= np.uint32(thresh), inv_keep, np.float32(0.0)).astype(np.float32)
    return torch.from_numpy(f.reshape(B, H, W, C).transpose(0, 3, 1, 2).copy())


def block(sd, prefix, x, groups, drop=None):
    """Block.forward unet.py:83-91; drop = keep/scale factors of the Dropout layer (None: identity)."""
    h = F.group_norm(x, groups, sd[prefix + "block.0.weight"], sd[prefix + "block.0.bias"], eps=1e-5)
    h = sr3_unet.swish(h)
    if drop is not None:
        h = h * drop
    return F.conv2d(h, sd[prefix + "block.3.weight"], sd[prefix + "block.3.bias"], padding=1)


def resnet_block(sd, prefix, x, t_emb, groups, drop=None):
    h = block(sd, prefix + "block1.", x, groups)
    film = F.linear(t_emb, sd[prefix + "noise_func.noise_func.0.weight"], sd[prefix + "noise_func.noise_func.0.bias"])
    h = h + film.view(x.shape[0], -1, 1, 1)
    h = block(sd, prefix + "block2.", h, groups, drop)
    if (prefix + "res_conv.weight") in sd:
        x = F.conv2d(x, sd[prefix + "res_conv.weight"], sd[prefix + "res_conv.bias"])
    return h + x


def unet_forward_train(sd, cfg, x, gamma, dropout=0.0, key=0):
    """UNet.forward unet.py:239-263 in training mode.  Dropout masks: layer id 2k+1 for the block2 of the k-th
    ResnetBlock (downs, mid, ups order)."""
    groups = cfg.get("norm_groups", 32)
    downs, mid, ups = sr3_unet.unet_layout(cfg["in_channel"], cfg["out_channel"], cfg["inner_channel"], cfg["channel_mults"],
                                           cfg["attn_res"], cfg["res_blocks"], cfg["image_size"])
    t_emb = sr3_unet.noise_level_mlp(sd, gamma, cfg["inner_channel"])
    k = [0]

    def unit(prefix, e, xx):
        cout = e[2]
        drop = None
        if dropout > 0:
            drop = dropout_factor(key, 2 * k[0] + 1, (xx.shape[0], cout, xx.shape[2], xx.shape[3]), dropout)
        k[0] += 1
        xx = resnet_block(sd, prefix + "res_block.", xx, t_emb, groups, drop)
        if e[3]:
            xx = sr3_unet.self_attention(sd, prefix + "attn.", xx, groups)
        return xx

    feats = []
    for i, e in enumerate(downs):
        p = "downs.%d." % i
        if e[0] == "conv":
            x = F.conv2d(x, sd[p + "weight"], sd[p + "bias"], padding=1)
        elif e[0] == "res":
            x = unit(p, e, x)
        else:
            x = sr3_unet.downsample(sd, p, x)
        feats.append(x)
    for i, e in enumerate(mid):
        x = unit("mid.%d." % i, e, x)
    for i, e in enumerate(ups):
        p = "ups.%d." % i
        if e[0] == "res":
            x = unit(p, e, torch.cat((x, feats.pop()), dim=1))
        else:
            x = sr3_unet.upsample(sd, p, x)
    return block(sd, "final_conv.", x, groups)


def l_pix(sd, cfg, hr, sr, noise, gamma, loss_type="l1", dropout=0.0, key=0):
    """model/model.py:50-54 over diffusion.py:222-250 with injected gamma [B] and noise: sum-loss / (b*c*h*w)."""
    b, c, h, w = hr.shape
    g = gamma.view(b, 1, 1, 1)
    x_noisy = g * hr + (1 - g ** 2).sqrt() * noise                                   # q_sample, diffusion.py:213-220
    eps = unet_forward_train(sd, cfg, torch.cat([sr, x_noisy], dim=1), gamma.view(b, 1), dropout, key)
    loss = (noise - eps).abs().sum() if loss_type == "l1" else ((noise - eps) ** 2).sum()
    return loss / float(b * c * h * w)


def loss_and_grads(sd, cfg, hr, sr, noise, gamma, loss_type="l1", dropout=0.0, key=0):
    """-> (l_pix value, {name: gradient}) by torch.autograd over the restatement."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    loss = l_pix(leaf, cfg, hr, sr, noise, gamma, loss_type, dropout, key)
    names = list(leaf)
    grads = torch.autograd.grad(loss, [leaf[k] for k in names], allow_unused=True)
    return float(loss.detach()), {k: (g if g is not None else torch.zeros_like(leaf[k])) for k, g in zip(names, grads)}


def adam_steps(params, grad_fn, steps, lr=1e-5, betas=(0.9, 0.999), eps=1e-8):
    """`steps` steps of torch.optim.Adam (the optimiser model/model.py:37-41 builds) on {name: tensor}; grad_fn(step, params) ->
    {name: gradient}.  Returns the final parameters."""
    ps = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    opt = torch.optim.Adam(list(ps.values()), lr=lr, betas=betas, eps=eps)
    for s in range(steps):
        g = grad_fn(s, {k: v.detach() for k, v in ps.items()})
        for k, p in ps.items():
            p.grad = g[k].detach().clone()
        opt.step()
    return {k: v.detach().clone() for k, v in ps.items()}
