"""Oracle: Group-Autoencoder (GAE) encode / decode as functions of a state_dict.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates AE.py (GAE:256-360, Encoder:168-199, Decoder:202-242, BranchUnit:145-165,
SSPN:120-141, SSB:102-109) and the live definitions in common.py (ResBlock:163-182,
ResAttentionBlock:250-271, CALayer:231-247, Upsampler(scale=1) = identity :184-211).
The reference's hard-coded 'cuda:0' scratch tensors (AE.py:285-287,313-314) are
not reproduced: everything lives on the input's device (CPU).
"""
import math

import torch
import torch.nn.functional as F

RES_SCALE = 0.1          # AE.py:192,225,270 (res_scale=0.1)
LEAKY = 0.01             # nn.LeakyReLU() default slope


def group_indices(n_colors, n_subs, n_ovls):
    """GAE.__init__ AE.py:263-280: overlapping spectral groups, tail group clamped."""
    g = math.ceil((n_colors - n_ovls) / (n_subs - n_ovls))
    start, end = [], []
    for i in range(g):
        s = (n_subs - n_ovls) * i
        e = s + n_subs
        if e > n_colors:
            e = n_colors
            s = n_colors - n_subs
        start.append(s)
        end.append(e)
    return start, end


def _conv(sd, p, x, pad):
    return F.conv2d(x, sd[p + "weight"], sd[p + "bias"], padding=pad)


def ssb(sd, p, x):
    """SSB.forward AE.py:108-109 = spc(spa(x))."""
    # spa: ResBlock (common.py:163-182): conv3 -> LeakyReLU -> conv3, *0.1, + x
    r = _conv(sd, p + "spa.body.0.", x, 1)
    r = F.leaky_relu(r, LEAKY)
    r = _conv(sd, p + "spa.body.2.", r, 1)
    x = r * RES_SCALE + x
    # spc: ResAttentionBlock (common.py:250-271): conv1 -> LeakyReLU -> conv1 -> CALayer, *0.1, + x
    r = _conv(sd, p + "spc.body.0.", x, 0)
    r = F.leaky_relu(r, LEAKY)
    r = _conv(sd, p + "spc.body.2.", r, 0)
    # CALayer (common.py:231-247): GAP -> 1x1 -> ReLU -> 1x1 -> sigmoid -> scale
    y = r.mean(dim=(2, 3), keepdim=True)
    y = F.relu(_conv(sd, p + "spc.body.3.conv_du.0.", y, 0))
    y = torch.sigmoid(_conv(sd, p + "spc.body.3.conv_du.2.", y, 0))
    r = r * y
    return r * RES_SCALE + x


def branch_unit(sd, p, x, n_blocks):
    """BranchUnit.forward AE.py:158-165 with use_tail=False, up_scale=1."""
    y = _conv(sd, p + "head.", x, 1)
    r = y
    for i in range(n_blocks):               # SSPN.forward AE.py:136-141
        r = ssb(sd, p + "body.net.%d." % i, r)
    return r + y


def encoder(sd, x, p="Encoder."):
    """Encoder.forward AE.py:195-199: branch(3 SSB) -> final conv3 (n_feats -> 3)."""
    return _conv(sd, p + "final.", branch_unit(sd, p + "branch.", x, 3), 1)


def decoder(sd, z, p="Decoder."):
    """Decoder.forward AE.py:233-242."""
    return _conv(sd, p + "final.", branch_unit(sd, p + "branch.", z, 3), 1)


def gae_encode(sd, x, n_subs, n_ovls):
    """GAE.encode AE.py:310-324 -> list of G latents (B,3,H,W)."""
    start, end = group_indices(x.shape[1], n_subs, n_ovls)
    return [encoder(sd, x[:, s:e]) for s, e in zip(start, end)]


def gae_decode(sd, n_colors, z_list, n_subs, n_ovls):
    """GAE.decode AE.py:283-308: overlap-add decoder outputs, average, trunk+final residual."""
    start, end = group_indices(n_colors, n_subs, n_ovls)
    b, _, h, w = z_list[0].shape
    y = torch.zeros(b, n_colors, h, w)
    cnt = torch.zeros(n_colors)
    for z, s, e in zip(z_list, start, end):
        y[:, s:e] += decoder(sd, z)
        cnt[s:e] += 1
    y = y / cnt.view(1, -1, 1, 1)
    y1 = _conv(sd, "final.", branch_unit(sd, "trunk.", y, 2), 1)
    return y1 + y


def gae_forward(sd, x, n_subs, n_ovls):
    """GAE.forward AE.py:326-360 -> (reconstruction, z_list)."""
    z = gae_encode(sd, x, n_subs, n_ovls)
    return gae_decode(sd, x.shape[1], z, n_subs, n_ovls), z


def gae_param_shapes(n_subs, n_colors, n_feats=64, trunk_feats=32):
    """Shapes of the 108 GAE tensors."""
    shp = {}

    def branch(p, cin, f, nb):
        shp[p + "head.weight"] = (f, cin, 3, 3)
        shp[p + "head.bias"] = (f,)
        red = f // 3
        for i in range(nb):
            q = p + "body.net.%d." % i
            for j in (0, 2):
                shp[q + "spa.body.%d.weight" % j] = (f, f, 3, 3)
                shp[q + "spa.body.%d.bias" % j] = (f,)
                shp[q + "spc.body.%d.weight" % j] = (f, f, 1, 1)
                shp[q + "spc.body.%d.bias" % j] = (f,)
            shp[q + "spc.body.3.conv_du.0.weight"] = (red, f, 1, 1)
            shp[q + "spc.body.3.conv_du.0.bias"] = (red,)
            shp[q + "spc.body.3.conv_du.2.weight"] = (f, red, 1, 1)
            shp[q + "spc.body.3.conv_du.2.bias"] = (f,)

    branch("Encoder.branch.", n_subs, n_feats, 3)
    shp["Encoder.final.weight"] = (3, n_feats, 3, 3)
    shp["Encoder.final.bias"] = (3,)
    branch("Decoder.branch.", 3, n_feats, 3)
    shp["Decoder.final.weight"] = (n_subs, n_feats, 3, 3)
    shp["Decoder.final.bias"] = (n_subs,)
    branch("trunk.", n_colors, trunk_feats, 2)
    shp["final.weight"] = (n_colors, trunk_feats, 3, 3)
    shp["final.bias"] = (n_colors,)
    return shp
