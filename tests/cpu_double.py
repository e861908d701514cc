"""TEST DOUBLE (tests only): torch/CPU stand-ins for the kernel wrappers in hsi_dmgasr_amd.ops / train_ops, with the same
signatures and tensor conventions (NHWC activations, the GroupNorm table layout, packed weight buffers).

Purpose: the training step's ORCHESTRATION - tape and skip-connection bookkeeping, gradient routing, the flat parameter /
gradient buffers, the FiLM table ordering, the pack-map gather - can be checked against the oracle's autograd gradients in
the CPU suite, without a GPU.  What each kernel computes is checked on the GPU against torch formulas (tests/test_gpu_train.py);
this file makes no claim about that and is never imported by the product.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import train as otrain

XF_NONE, XF_AFFINE, XF_AFFINE_SILU = 0, 1, 2


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def cat(x0, x1):
    return x0 if x1 is None else torch.cat([x0, x1], dim=3)


def unpack_weight(pw):
    """Inverse of ops.PackedConv._steps on the packed buffers (hi + lo): [Cout, Cin(8-padded), k, k] fp32."""
    w = pw.w_hi.float() + (pw.w_lo.float() if pw.w_lo is not None else 0.0)
    taps = pw.ksize * pw.ksize
    steps, cpad, bk = w.shape
    nch = steps // taps
    w = w.reshape(nch, taps, cpad, bk).permute(2, 0, 3, 1).reshape(cpad, nch * bk, taps)
    return w[:pw.cout, :pw.cin].reshape(pw.cout, pw.cin, pw.ksize, pw.ksize).contiguous()


# ----------------------------------------------------------------------------------------------------------- ops.*
def conv2d(x0, pw, *, x1=None, gn_ab=None, transform=XF_NONE, film=None, res=None, res_scale=1.0, act=0, stride=1, ups=False,
           proj_x0=None, proj_x1=None, stats=False):
    assert transform == XF_NONE and act == 0 and proj_x0 is None, "the training path materialises its operands"
    x = nchw(cat(x0, x1))
    assert x.shape[1] == pw.cin
    if ups:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    y = F.conv2d(x, unpack_weight(pw), pw.bias, stride=stride, padding=pw.ksize // 2)
    if film is not None:
        y = y + film.reshape(film.shape[0], -1, 1, 1)
    y = res_scale * y
    if pw.out_nchw:
        assert res is None
        return y.contiguous()
    y = nhwc(y)
    return y + res if res is not None else y


def gn_scale_shift(x0, x1, gamma, beta, groups, precision, eps=1e-5):
    x = cat(x0, x1)
    B, H, W, C = x.shape
    g = x.reshape(B, H * W, groups, C // groups)
    mean = g.mean(dim=(1, 3))
    var = g.var(dim=(1, 3), unbiased=False)
    rstd = (var + eps).rsqrt()
    cpg = C // groups
    sc = rstd.repeat_interleave(cpg, dim=1) * gamma[None]
    sh = beta[None] - mean.repeat_interleave(cpg, dim=1) * sc
    table = torch.cat([torch.stack([sc, sh], dim=2).reshape(-1), torch.zeros(2 * B * C), torch.stack([mean, rstd], dim=2).reshape(-1)])
    return table


def _table(ab, B, C, groups):
    pairs = ab[:2 * B * C].reshape(B, C, 2)
    mr = ab[4 * B * C:4 * B * C + 2 * B * groups].reshape(B, groups, 2)
    return pairs, mr


def noise_film(B, dim, mlp, wf, bf, *, gamma=None, level_table=None, t_ptr=None, t_emb=None, want_t=False):
    w1, b1, w2, b2 = mlp
    half = dim // 2
    step = torch.arange(half, dtype=torch.float32) / half
    enc = gamma.reshape(B, 1) * torch.exp(-math.log(1e4) * step)[None]
    pe = torch.cat([enc.sin(), enc.cos()], dim=1)
    h = F.linear(pe, w1, b1)
    t = F.linear(h * torch.sigmoid(h), w2, b2)
    film = F.linear(t, wf, bf)
    return (film, t) if want_t else film


def to_nhwc(x, precision, x1=None, **kw):
    x = x if x1 is None else torch.cat([x, x1], dim=1)
    pad = (-x.shape[1]) % 8
    if pad:
        x = F.pad(x, (0, 0, 0, 0, 0, pad))
    return nhwc(x)


def attention(qkv, precision):
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    q, k, v = qkv.reshape(B, H * W, 3, C).unbind(2)
    p = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), dim=-1)
    return (p @ v).reshape(B, H, W, C)


def loss_sum(a, b, kind):
    return (a - b).abs().sum() if kind == "l1" else ((a - b) ** 2).sum()


def q_sample(x0, noise, gamma):
    g = gamma.reshape(-1, 1, 1, 1)
    return g * x0 + (1 - g ** 2).sqrt() * noise


# ----------------------------------------------------------------------------------------------------------- train_ops.*
def _key(seed):
    return int(seed.item()) & 0xFFFFFFFFFFFFFFFF if torch.is_tensor(seed) else int(seed)


def _factor(shape_nhwc, p_drop, seed, layer):
    B, H, W, C = shape_nhwc
    return nhwc(otrain.dropout_factor(_key(seed), int(layer), (B, C, H, W), p_drop))


def gn_act_apply(x0, x1, gn_ab, silu, precision, p_drop=0.0, seed=0, layer=0):
    x = cat(x0, x1)
    B, H, W, C = x.shape
    pairs = gn_ab[:2 * B * C].reshape(B, 1, 1, C, 2)
    u = x * pairs[..., 0] + pairs[..., 1]
    a = u * torch.sigmoid(u) if silu else u
    if p_drop > 0:
        a = a * _factor(a.shape, p_drop, seed, layer)
    return a


def gn_act_bwd(da, x0, x1, gn_ab, gamma, groups, silu, precision, dgamma, dbeta, p_drop=0.0, seed=0, layer=0, add=None):
    x = cat(x0, x1).detach().clone().requires_grad_(True)
    B, H, W, C = x.shape
    pairs, mr = _table(gn_ab, B, C, groups)
    cpg = C // groups
    beta = (pairs[0, :, 1] + mr[0, :, 0].repeat_interleave(cpg) * pairs[0, :, 0]).detach().clone().requires_grad_(True)
    gam = gamma.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        y = F.group_norm(nchw(x), groups, gam, beta, eps=1e-5)
        a = y * torch.sigmoid(y) if silu else y
        if p_drop > 0:
            a = a * otrain.dropout_factor(_key(seed), int(layer), tuple(a.shape), p_drop)
        gx, gg, gb = torch.autograd.grad(nhwc(a), [x, gam, beta], grad_outputs=da)
    dgamma.copy_(gg)
    dbeta.copy_(gb)
    if add is not None:
        gx = gx + add
    C0 = x0.shape[3]
    return gx[..., :C0].contiguous(), (None if x1 is None else gx[..., C0:].contiguous())


def conv_wgrad(a0, a1, dy, dw, precision, stride=1, ups=False, deferred=None, db=None, db_images=None):
    if db is not None:
        db.copy_(dy.float().sum(dim=(0, 1, 2))[:db.numel()])
    if db_images is not None:
        db_images.zero_()
        db_images[:, :dy.shape[3]] = dy.float().sum(dim=(1, 2))
    x = nchw(cat(a0, a1))
    if ups:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    k = dw.shape[2]
    g = torch.nn.grad.conv2d_weight(x, (dy.shape[3], x.shape[1], k, k), nchw(dy), stride=stride, padding=k // 2)
    dw.copy_(g[:dw.shape[0], :dw.shape[1]])


def add(a, b, precision):
    return a + b


def zero_insert2(x, Ho, Wo, precision):
    B, Hi, Wi, C = x.shape
    out = torch.zeros(B, Ho, Wo, C)
    out[:, ::2, ::2] = x
    return out


def sum2x2(x, precision):
    B, H2, W2, C = x.shape
    return x.reshape(B, H2 // 2, 2, W2 // 2, 2, C).sum(dim=(2, 4))


def channel_sums(x, precision, out_c=None, want_bc=False, cout=None):
    cout = x.shape[3] if cout is None else cout
    bc = x.sum(dim=(1, 2))[:, :cout]
    if out_c is not None:
        out_c.copy_(bc.sum(dim=0))
    return bc.contiguous() if want_bc else None


def loss_grad(noise, eps, kind, scale, precision):
    d = noise - eps
    g = -scale * torch.sign(d) if kind == "l1" else -2.0 * scale * d
    return to_nhwc(g, precision)


def noise_film_bwd(gamma, t_emb, dfilm, mlp, wf, grads):
    w1, b1, w2, b2 = (t.detach().clone().requires_grad_(True) for t in mlp)
    wfl = wf.detach().clone().requires_grad_(True)
    bfl = torch.zeros(wf.shape[0], requires_grad=True)
    with torch.enable_grad():
        film = noise_film(dfilm.shape[0], t_emb.shape[1], (w1, b1, w2, b2), wfl, bfl, gamma=gamma)
        g = torch.autograd.grad(film, [w1, b1, w2, b2, wfl, bfl], grad_outputs=dfilm)
    for dst, src in zip(grads, g):
        dst.copy_(src)


def attention_bwd(qkv, do, precision):
    x = qkv.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        o = attention(x, precision)
        (g,) = torch.autograd.grad(o, [x], grad_outputs=do)
    return g


def gather_pack(src, idx, out_hi, out_lo=None):
    w = torch.where(idx >= 0, src[idx.clamp(min=0).long()], torch.zeros(()))
    hi = w.to(torch.bfloat16)
    out_hi.copy_(hi)
    if out_lo is not None:
        out_lo.copy_((w - hi.float()).to(torch.bfloat16))


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, coef=None):
    gg = g * grad_scale
    m.mul_(beta1).add_(gg, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
    step_size, inv = (float(coef[0]), float(coef[1])) if coef is not None else (lr / (1 - beta1 ** step), 1 / math.sqrt(1 - beta2 ** step))
    p.sub_(step_size * (m / (v.sqrt() * inv + eps)))


OPS = dict(conv2d=conv2d, gn_scale_shift=gn_scale_shift, noise_film=noise_film, to_nhwc=to_nhwc, attention=attention,
           loss_sum=loss_sum, q_sample=q_sample)
TRAIN_OPS = dict(gn_act_apply=gn_act_apply, gn_act_bwd=gn_act_bwd, conv_wgrad=conv_wgrad, add=add, zero_insert2=zero_insert2,
                 sum2x2=sum2x2, channel_sums=channel_sums, loss_grad=loss_grad, noise_film_bwd=noise_film_bwd,
                 attention_bwd=attention_bwd, gather_pack=gather_pack, adam_step=adam_step)


def install(monkeypatch):
    """Replace the kernel wrappers by the doubles for the duration of one test."""
    from hsi_dmgasr_amd import ops, train_ops
    for k, f in OPS.items():
        monkeypatch.setattr(ops, k, f)
    for k, f in TRAIN_OPS.items():
        monkeypatch.setattr(train_ops, k, f)
