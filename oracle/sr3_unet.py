"""Oracle: SR3 UNet forward as pure functions of a state_dict (fp32, CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates model/sr3_modules/unet.py of the reference.  Keys follow the
reference checkpoints: ``downs.N...``, ``mid.N...``, ``ups.N...``,
``final_conv.block.{0,3}``, ``noise_level_mlp.{1,3}``.
"""
import math

import torch
import torch.nn.functional as F


def unet_layout(in_channel, out_channel, inner_channel, channel_mults, attn_res,
                res_blocks, image_size):
    """Module list of UNet.__init__ (unet.py:163-237) as plain tuples.

    Returns (downs, mid, ups) where an entry is
      ("conv", cin, cout) | ("res", cin, cout, with_attn) | ("down", c) | ("up", c).
    Attention placement follows the *constructor* image_size (unet.py:195,200,223),
    not the runtime input size.
    """
    attn_res = tuple(attn_res) if not isinstance(attn_res, int) else (attn_res,)
    n = len(channel_mults)
    pre = inner_channel
    skip = [pre]
    res = image_size
    downs = [("conv", in_channel, inner_channel)]
    for lvl, mult in enumerate(channel_mults):
        ch = inner_channel * mult
        for _ in range(res_blocks):
            downs.append(("res", pre, ch, res in attn_res))
            skip.append(ch)
            pre = ch
        if lvl != n - 1:
            downs.append(("down", pre))
            skip.append(pre)
            res //= 2
    mid = [("res", pre, pre, True), ("res", pre, pre, False)]
    ups = []
    for lvl in reversed(range(n)):
        ch = inner_channel * channel_mults[lvl]
        for _ in range(res_blocks + 1):
            ups.append(("res", pre + skip.pop(), ch, res in attn_res))
            pre = ch
        if lvl >= 1:
            ups.append(("up", pre))
            res *= 2
    return downs, mid, ups


def positional_encoding(gamma, dim):
    """unet.py:23-31.  gamma (B,1) -> (B,1,dim)."""
    half = dim // 2
    step = torch.arange(half, dtype=gamma.dtype) / half
    enc = gamma.unsqueeze(1) * torch.exp(-math.log(1e4) * step.unsqueeze(0))
    return torch.cat([torch.sin(enc), torch.cos(enc)], dim=-1)


def swish(x):
    """unet.py:54-55."""
    return x * torch.sigmoid(x)


def noise_level_mlp(sd, gamma, inner_channel, prefix="noise_level_mlp."):
    """unet.py:182-187: PE -> Linear(c,4c) -> Swish -> Linear(4c,c)."""
    e = positional_encoding(gamma, inner_channel)
    e = F.linear(e, sd[prefix + "1.weight"], sd[prefix + "1.bias"])
    e = swish(e)
    return F.linear(e, sd[prefix + "3.weight"], sd[prefix + "3.bias"])


def block(sd, prefix, x, groups):
    """Block.forward unet.py:83-91 in eval mode (dropout = identity)."""
    h = F.group_norm(x, groups, sd[prefix + "block.0.weight"], sd[prefix + "block.0.bias"], eps=1e-5)
    h = swish(h)
    return F.conv2d(h, sd[prefix + "block.3.weight"], sd[prefix + "block.3.bias"], padding=1)


def resnet_block(sd, prefix, x, t_emb, groups):
    """ResnetBlock.forward unet.py:105-111 (+ FeatureWiseAffine non-affine branch :49)."""
    h = block(sd, prefix + "block1.", x, groups)
    film = F.linear(t_emb, sd[prefix + "noise_func.noise_func.0.weight"],
                    sd[prefix + "noise_func.noise_func.0.bias"])
    h = h + film.view(x.shape[0], -1, 1, 1)
    h = block(sd, prefix + "block2.", h, groups)
    if (prefix + "res_conv.weight") in sd:
        x = F.conv2d(x, sd[prefix + "res_conv.weight"], sd[prefix + "res_conv.bias"])
    return h + x


def self_attention(sd, prefix, x, groups):
    """SelfAttention.forward unet.py:124-143, n_head = 1."""
    b, c, hh, ww = x.shape
    n = F.group_norm(x, groups, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], eps=1e-5)
    qkv = F.conv2d(n, sd[prefix + "qkv.weight"])
    q, k, v = qkv.reshape(b, 3, c, hh * ww).unbind(1)          # (b, c, N) each
    score = torch.einsum("bci,bcj->bij", q, k) / math.sqrt(c)   # rows = query pixel
    score = torch.softmax(score, dim=-1)
    o = torch.einsum("bij,bcj->bci", score, v).reshape(b, c, hh, ww)
    o = F.conv2d(o, sd[prefix + "out.weight"], sd[prefix + "out.bias"])
    return o + x


def upsample(sd, prefix, x):
    """Upsample.forward unet.py:64-65: nearest x2 then 3x3 conv."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return F.conv2d(x, sd[prefix + "conv.weight"], sd[prefix + "conv.bias"], padding=1)


def downsample(sd, prefix, x):
    """Downsample.forward unet.py:73-74: 3x3 conv stride 2 pad 1."""
    return F.conv2d(x, sd[prefix + "conv.weight"], sd[prefix + "conv.bias"], stride=2, padding=1)


def _res_unit(sd, prefix, entry, x, t_emb, groups):
    x = resnet_block(sd, prefix + "res_block.", x, t_emb, groups)
    if entry[3]:
        x = self_attention(sd, prefix + "attn.", x, groups)
    return x


def unet_forward(sd, cfg, x, gamma):
    """UNet.forward unet.py:239-263.

    cfg: dict(in_channel, out_channel, inner_channel, norm_groups, channel_mults,
              attn_res, res_blocks, image_size).  x (B,in,H,W), gamma (B,1).
    """
    groups = cfg.get("norm_groups", 32)
    downs, mid, ups = unet_layout(cfg["in_channel"], cfg["out_channel"], cfg["inner_channel"],
                                  cfg["channel_mults"], cfg["attn_res"], cfg["res_blocks"],
                                  cfg["image_size"])
    t_emb = noise_level_mlp(sd, gamma, cfg["inner_channel"])
    feats = []
    for i, e in enumerate(downs):
        p = "downs.%d." % i
        if e[0] == "conv":
            x = F.conv2d(x, sd[p + "weight"], sd[p + "bias"], padding=1)
        elif e[0] == "res":
            x = _res_unit(sd, p, e, x, t_emb, groups)
        else:
            x = downsample(sd, p, x)
        feats.append(x)
    for i, e in enumerate(mid):
        x = _res_unit(sd, "mid.%d." % i, e, x, t_emb, groups)
    for i, e in enumerate(ups):
        p = "ups.%d." % i
        if e[0] == "res":
            x = _res_unit(sd, p, e, torch.cat((x, feats.pop()), dim=1), t_emb, groups)
        else:
            x = upsample(sd, p, x)
    return block(sd, "final_conv.", x, groups)


def unet_param_shapes(cfg):
    """Shapes of every parameter of the reference UNet for cfg."""
    c = cfg["inner_channel"]
    downs, mid, ups = unet_layout(cfg["in_channel"], cfg["out_channel"], c, cfg["channel_mults"],
                                  cfg["attn_res"], cfg["res_blocks"], cfg["image_size"])
    shp = {
        "noise_level_mlp.1.weight": (4 * c, c), "noise_level_mlp.1.bias": (4 * c,),
        "noise_level_mlp.3.weight": (c, 4 * c), "noise_level_mlp.3.bias": (c,),
    }

    def res(p, cin, cout, attn):
        r = p + "res_block."
        shp[r + "noise_func.noise_func.0.weight"] = (cout, c)
        shp[r + "noise_func.noise_func.0.bias"] = (cout,)
        shp[r + "block1.block.0.weight"] = (cin,)
        shp[r + "block1.block.0.bias"] = (cin,)
        shp[r + "block1.block.3.weight"] = (cout, cin, 3, 3)
        shp[r + "block1.block.3.bias"] = (cout,)
        shp[r + "block2.block.0.weight"] = (cout,)
        shp[r + "block2.block.0.bias"] = (cout,)
        shp[r + "block2.block.3.weight"] = (cout, cout, 3, 3)
        shp[r + "block2.block.3.bias"] = (cout,)
        if cin != cout:
            shp[r + "res_conv.weight"] = (cout, cin, 1, 1)
            shp[r + "res_conv.bias"] = (cout,)
        if attn:
            a = p + "attn."
            shp[a + "norm.weight"] = (cout,)
            shp[a + "norm.bias"] = (cout,)
            shp[a + "qkv.weight"] = (3 * cout, cout, 1, 1)
            shp[a + "out.weight"] = (cout, cout, 1, 1)
            shp[a + "out.bias"] = (cout,)

    for name, lst in (("downs", downs), ("mid", mid), ("ups", ups)):
        for i, e in enumerate(lst):
            p = "%s.%d." % (name, i)
            if e[0] == "conv":
                shp[p + "weight"] = (e[2], e[1], 3, 3)
                shp[p + "bias"] = (e[2],)
            elif e[0] == "res":
                res(p, e[1], e[2], e[3])
            else:
                shp[p + "conv.weight"] = (e[1], e[1], 3, 3)
                shp[p + "conv.bias"] = (e[1],)
    last = ups[-1][2]
    shp["final_conv.block.0.weight"] = (last,)
    shp["final_conv.block.0.bias"] = (last,)
    shp["final_conv.block.3.weight"] = (cfg["out_channel"], last, 3, 3)
    shp["final_conv.block.3.bias"] = (cfg["out_channel"],)
    return shp
