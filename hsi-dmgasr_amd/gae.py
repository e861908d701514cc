"""Group-Autoencoder (GAE) behind the reference's module interface, executed by the HIP kernels.

Mirrors the live classes of the reference's AE.py (GAE:256, Encoder:168, Decoder:202, BranchUnit:145, SSPN:120,
SSB:102) and common.py (ResBlock:163, ResAttentionBlock:250, CALayer:231, Upsampler:184 with scale 1 = empty):
same constructor arguments, attribute tree and state_dict keys (108 tensors), and the
same public methods encode(x) / decode(x, z_list) / forward(x).

Differences that are deliberate:
  * the G spectral groups are stacked on the batch axis and run through the shared Encoder / Decoder in ONE
    pass (bit-identical per sample; the reference loops over groups at batch 1, AE.py:316-323);
  * the overlap-add + count + divide of decode (AE.py:286-295) is one gather kernel;
  * tensors live on the input's device (the reference hard-codes 'cuda:0', AE.py:285,313,328).
Every convolution is hsidm_conv2d (LeakyReLU, 0.1*res + x and the final residual fused in its epilogue);
CALayer = per-channel sums + a tiny squeeze-excite kernel + one fused scale/residual kernel.
"""
import math
import pickle

import torch
from torch import nn

from . import ops
from .precision import resolve_precision
from .sr3_modules.unet import _PackCache

RES_SCALE = 0.1


def default_conv(in_channels, out_channels, kernel_size, bias=True, dilation=1):
    if dilation != 1:
        raise NotImplementedError("hsidm: dilated convolutions are not used by the GAE path")
    return nn.Conv2d(in_channels, out_channels, kernel_size, padding=(kernel_size // 2), bias=bias)


class Upsampler(nn.Sequential):
    """common.py:184-211; the GAE only instantiates scale=1, i.e. an empty Sequential."""

    def __init__(self, conv, scale, n_feats, bn=False, act=False, bias=True):
        if scale != 1:
            raise NotImplementedError("hsidm: GAE uses up_scale=1 (AE.py:192,225)")
        super().__init__()


class CALayer(nn.Module):
    def __init__(self, channel, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv_du = nn.Sequential(
            nn.Conv2d(channel, channel // reduction, 1, padding=0, bias=True),
            nn.ReLU(inplace=False),
            nn.Conv2d(channel // reduction, channel, 1, padding=0, bias=True),
            nn.Sigmoid(),
        )

    def vector(self, r, precision):
        """Squeeze-excite vector [B, C] of an NHWC tensor r."""
        part, nsplit = ops._partials(r, precision)
        c0, c2 = self.conv_du[0], self.conv_du[2]
        return ops.ca_vector(part, nsplit, r.shape[1] * r.shape[2],
                             c0.weight.reshape(c0.weight.shape[0], -1), c0.bias,
                             c2.weight.reshape(c2.weight.shape[0], -1), c2.bias)


class _ConvPair(nn.Module):
    """body = [conv, act, conv (, CALayer)] as in common.ResBlock / ResAttentionBlock."""

    def __init__(self, conv, n_feats, kernel_size, bias, bn, act, res_scale, with_ca):
        super().__init__()
        if bn:
            raise NotImplementedError("hsidm: the GAE never enables batch-norm")
        if not isinstance(act, nn.LeakyReLU):
            raise NotImplementedError("hsidm: GAE blocks use nn.LeakyReLU() (AE.py:192,225,270)")
        m = [conv(n_feats, n_feats, kernel_size, bias=bias), act, conv(n_feats, n_feats, kernel_size, bias=bias)]
        if with_ca:
            m.append(CALayer(n_feats, 3))
        self.body = nn.Sequential(*m)
        self.res_scale = res_scale
        self._cache = _PackCache()

    def _pk(self, idx, precision):
        c = self.body[idx]
        return self._cache.get((idx, precision), [c.weight, c.bias], lambda: ops.PackedConv(c.weight, c.bias, precision))


class ResBlock(_ConvPair):
    def __init__(self, conv, n_feats, kernel_size, bias=True, bn=False, act=nn.ReLU(True), res_scale=1):
        super().__init__(conv, n_feats, kernel_size, bias, bn, act, res_scale, with_ca=False)

    def _run(self, x, precision):
        h = ops.conv2d(x, self._pk(0, precision), act=ops.ACT_LEAKY)
        return ops.conv2d(h, self._pk(2, precision), res=x, res_scale=self.res_scale)


class ResAttentionBlock(_ConvPair):
    def __init__(self, conv, n_feats, kernel_size, bias=True, bn=False, act=nn.ReLU(True), res_scale=1):
        super().__init__(conv, n_feats, kernel_size, bias, bn, act, res_scale, with_ca=True)

    def _pair(self, precision):
        c0, c2 = self.body[0], self.body[2]
        return self._cache.get(("pair", precision), [c0.weight, c0.bias, c2.weight, c2.bias],
                               lambda: ops.PackedPair(c0.weight, c0.bias, c2.weight, c2.bias, precision))

    def _run(self, x, precision, skip2=None):
        c0 = self.body[0]
        B, H, W, Cc = x.shape
        if c0.kernel_size == (1, 1) and Cc == 64 and c0.out_channels == 64 and (H * W) % 64 == 0:
            # the spectral block (SSB.spc, AE.py:102-109: kernel_size 1): conv -> LeakyReLU -> conv in ONE launch, h stays on the chip
            r = ops.conv1x1_pair(x, self._pair(precision), act=ops.ACT_LEAKY, stats=True)
        else:
            h = ops.conv2d(x, self._pk(0, precision), act=ops.ACT_LEAKY)
            r = ops.conv2d(h, self._pk(2, precision), stats=True)
        ca = self.body[3].vector(r, precision)
        return ops.ca_apply(r, ca, x, self.res_scale, precision, skip2=skip2)


class SSB(nn.Module):
    def __init__(self, n_feats, kernel_size, act, res_scale, conv=default_conv):
        super().__init__()
        self.spa = ResBlock(conv, n_feats, kernel_size, act=act, res_scale=res_scale)
        self.spc = ResAttentionBlock(conv, n_feats, 1, act=act, res_scale=res_scale)

    def _run(self, x, precision, skip2=None):
        return self.spc._run(self.spa._run(x, precision), precision, skip2=skip2)


class SSPN(nn.Module):
    def __init__(self, n_feats, n_blocks, act, res_scale):
        super().__init__()
        self.net = nn.Sequential(*[SSB(n_feats, 3, act=act, res_scale=res_scale) for _ in range(n_blocks)])

    def _run(self, x, precision):
        r = x
        n = len(self.net)
        for i, blk in enumerate(self.net):          # the outer skip (AE.py:137-139) rides on the last block
            r = blk._run(r, precision, skip2=x if i == n - 1 else None)
        return r


class BranchUnit(nn.Module):
    def __init__(self, n_colors, n_feats, n_blocks, act, res_scale, up_scale, use_tail=True, conv=default_conv):
        super().__init__()
        if use_tail:
            raise NotImplementedError("hsidm: the GAE builds its branches with use_tail=False")
        self.head = nn.Conv2d(n_colors, n_feats, kernel_size=3, padding=1)
        self.body = SSPN(n_feats, n_blocks, act, res_scale)
        self.upsample = Upsampler(conv, up_scale, n_feats)
        self.tail = None
        self._cache = _PackCache()

    def _run(self, x, precision):
        pk = self._cache.get(precision, [self.head.weight, self.head.bias],
                             lambda: ops.PackedConv(self.head.weight, self.head.bias, precision))
        return self.body._run(ops.conv2d(x, pk), precision)


class _Codec(nn.Module):
    precision = None

    def __init__(self, input_channel, out_channel, n_feats):
        super().__init__()
        self.input_channel = input_channel
        self.out_channel = out_channel
        self.branch = BranchUnit(input_channel, n_feats=n_feats, n_blocks=3, act=nn.LeakyReLU(), res_scale=RES_SCALE,
                                 use_tail=False, up_scale=1, conv=default_conv)
        self.final = nn.Conv2d(n_feats, out_channel, kernel_size=3, padding=1)
        self._cache = _PackCache()

    def _run(self, x, precision):
        """NHWC in -> NCHW fp32 out."""
        pk = self._cache.get(precision, [self.final.weight, self.final.bias],
                             lambda: ops.PackedConv(self.final.weight, self.final.bias, precision, out_nchw=True))
        return ops.conv2d(self.branch._run(x, precision), pk)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("hsidm: input is on %s; this module only runs on a ROCm device (no CPU fallback)" % x.device)
        p = resolve_precision(self.precision if self.precision is not None else "fp32")
        if p == "bf16":
            raise ValueError("hsidm: the group autoencoder's codecs run in precision 'fp32' or 'fp16' (see GAE)")
        return self._run(ops.to_nhwc(x.float(), p), p)


class Encoder(_Codec):
    def __init__(self, input_channel, out_channel, n_feats=128):
        super().__init__(input_channel, out_channel, n_feats)


class Decoder(_Codec):
    def __init__(self, input_channel, out_channel, n_feats=128):
        super().__init__(input_channel, out_channel, n_feats)


class GAE(nn.Module):
    """Group autoencoder.  precision: "fp32" (default: the autoencoder is ~0.02 % of the path's FLOPs and its decode sets the final
    PSNR) or "fp16" (measured 2.5e-4 ... 7.8e-4 per tensor, < 1e-3 dB / 1e-4 deg from the reference on the pretrained CAVE
    autoencoder).  A bf16 form existed until round 3: 0.08 dB / 0.011 deg from the reference on the same cube - outside the path's
    0.01 dB / 0.001 deg - and is no longer offered."""

    def __init__(self, Encoder=Encoder, Decoder=Decoder, n_subs=8, n_ovls=2, n_colors=31, n_feats=128, precision="fp32"):
        super().__init__()
        if precision not in (None, "fp32", "fp16"):         # (checked again per call: the attribute is assignable)
            raise ValueError("hsidm: the group autoencoder runs in precision 'fp32' (default) or 'fp16', got %r" % (precision,))
        self.Encoder = Encoder(n_subs, 3, n_feats)
        self.Decoder = Decoder(3, n_subs, n_feats)
        self.precision = precision
        self.n_subs, self.n_ovls, self.n_colors = n_subs, n_ovls, n_colors
        self.G = math.ceil((n_colors - n_ovls) / (n_subs - n_ovls))
        self.start_idx, self.end_idx = [], []
        self.trunk = BranchUnit(n_colors, n_feats=32, n_blocks=2, act=nn.LeakyReLU(), res_scale=RES_SCALE, up_scale=1,
                                conv=default_conv, use_tail=False)
        self.final = nn.Conv2d(32, n_colors, kernel_size=3, padding=1)
        for g in range(self.G):
            s = (n_subs - n_ovls) * g
            e = s + n_subs
            if e > n_colors:
                e, s = n_colors, n_colors - n_subs
            self.start_idx.append(s)
            self.end_idx.append(e)
        self._cache = _PackCache()
        self._dev_tables = {}

    # ------------------------------------------------------------------------------------------------
    def _prec(self):
        p = resolve_precision(self.precision if self.precision is not None else "fp32")
        if p == "bf16":
            raise ValueError("hsidm: the group autoencoder runs in precision 'fp32' (default) or 'fp16'; its bf16 form measured 0.08 dB / "
                             "0.011 deg from the reference's reconstruction (bounds: 0.01 dB / 0.001 deg) and is not offered")
        return p

    def _tables(self, B, C, HW, dev):
        key = (B, C, HW, str(dev))
        t = self._dev_tables.get(key)
        if t is None:
            start = torch.tensor(self.start_idx, dtype=torch.int64)
            off = (torch.arange(B, dtype=torch.int64).view(B, 1) * C + start.view(1, -1)) * HW     # [B, G]
            t = (off.reshape(-1).to(dev), start.to(torch.int32).to(dev))
            self._dev_tables[key] = t
        return t

    @staticmethod
    def _check(x):
        if not x.is_cuda:
            raise RuntimeError("hsidm: input is on %s; GAE only runs on a ROCm device (no CPU fallback)" % x.device)

    @torch.no_grad()
    def encode_batched(self, x):
        """x [B,C,H,W] -> latents [B, G, 3, H, W] (all spectral groups in one pass)."""
        self._check(x)
        p = self._prec()
        x = x.float().contiguous()
        B, C, H, W = x.shape
        off, _ = self._tables(B, C, H * W, x.device)
        xin = ops.to_nhwc(x, p, offsets0=off, c0=self.n_subs, n_out=B * self.G)
        return self.Encoder._run(xin, p).view(B, self.G, 3, H, W)

    @torch.no_grad()
    def decode_batched(self, z, n_colors=None):
        """z [B, G, 3, H, W] -> cube [B, C, H, W]."""
        self._check(z)
        p = self._prec()
        B, G, _, H, W = z.shape
        C = self.n_colors if n_colors is None else n_colors
        zin = ops.to_nhwc(z.float().contiguous().view(B * G, 3, H, W), p)
        dec = self.Decoder._run(zin, p)                                            # [B*G, n_subs, H, W] fp32
        _, start = self._tables(B, C, H * W, z.device)
        y = ops.overlap_average(dec, start, G, self.n_subs, B, C)                  # AE.py:286-295
        pk = self._cache.get(p, [self.final.weight, self.final.bias],
                             lambda: ops.PackedConv(self.final.weight, self.final.bias, p, out_nchw=True))
        t = self.trunk._run(ops.to_nhwc(y, p), p)
        return ops.conv2d(t, pk, res=y, res_scale=1.0)                             # y1 + y (AE.py:301-308)

    # ---- reference interface -------------------------------------------------------------------------
    def encode(self, x):
        z = self.encode_batched(x)
        return [z[:, g] for g in range(self.G)]

    def decode(self, x, z_list):
        z = torch.stack([t.to(x.device) for t in z_list], dim=1)
        return self.decode_batched(z, x.shape[1])

    def forward(self, x):
        z = self.encode_batched(x)
        return self.decode_batched(z, x.shape[1]), [z[:, g] for g in range(self.G)]


# ---------------------------------------------------------------------------------------------------------
def load_reference_checkpoint(path, precision="fp32", map_location="cpu"):
    """Read one of the reference's whole-module GAE pickles (torch.save(model), AE.py:637; resolved against
    __main__ / common there) WITHOUT the reference's code: every pickled class is replaced by a bare
    nn.Module stub, the state_dict and group layout are read off it, and a GAE of this package is built.

    A whole-module pickle is code: unpickling resolves arbitrary globals.  Everything outside the reference's own modules
    (stubbed) is therefore held to an allowlist - torch tensor/parameter rebuilders, torch.nn module classes, OrderedDict,
    torch.device / dtype / Size / storage types - and anything else raises UnpicklingError.  Only load files you trust as much as
    the reference's own GAE_pretrained/*.pth all the same."""
    stubs = {}
    allowed = {("collections", "OrderedDict"), ("torch", "device"), ("torch", "Size"), ("torch", "dtype"),
               ("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_tensor_v2"),
               ("torch._utils", "_rebuild_parameter_with_state"), ("torch.serialization", "_get_layout"),
               ("numpy.core.multiarray", "scalar"), ("numpy", "dtype"), ("__builtin__", "set"), ("builtins", "set")}

    class _Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module.split(".")[0] in ("__main__", "AE", "common", "SSPSR", "GELIN", "quantize"):
                if name not in stubs:
                    stubs[name] = type(name, (nn.Module,), {"forward": lambda self, *a, **k: None})
                return stubs[name]
            ok = (module, name) in allowed or (module == "torch" and name.endswith("Storage")) or \
                (module.startswith("torch.nn.modules.") and name[:1].isupper())
            if not ok:
                raise pickle.UnpicklingError("hsidm: refusing to resolve %s.%s from a GAE checkpoint (not on the allowlist)"
                                             % (module, name))
            return super().find_class(module, name)

    class _PickleModule:
        Unpickler = _Unpickler
        __name__ = "pickle"

        @staticmethod
        def load(f, **kw):
            return _Unpickler(f, **kw).load()

    obj = torch.load(path, map_location=map_location, pickle_module=_PickleModule, weights_only=False)
    sd = obj.state_dict()
    n_subs = sd["Encoder.branch.head.weight"].shape[1]
    n_feats = sd["Encoder.branch.head.weight"].shape[0]
    n_colors = sd["final.weight"].shape[0]
    start = list(obj.start_idx)
    n_ovls = n_subs - (start[1] - start[0]) if len(start) > 1 else 0
    g = GAE(Encoder, Decoder, n_subs=n_subs, n_ovls=n_ovls, n_colors=n_colors, n_feats=n_feats, precision=precision)
    assert g.start_idx == start and g.end_idx == list(obj.end_idx)
    g.load_state_dict(sd)
    return g.eval()
