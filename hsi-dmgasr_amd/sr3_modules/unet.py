"""SR3 denoiser network behind the reference's module interface, executed by the HIP kernels.

Drop-in for the reference's ``model/sr3_modules/unet.py``: same class names, constructor arguments,
attribute tree and state_dict keys, so reference checkpoints load with
``load_state_dict`` and ``networks.define_G`` (model/networks.py:91-101) can build it unchanged.
What differs is everything behind ``forward``: parameters are held by ordinary torch modules only
for naming / loading / initialisation; the arithmetic runs in libhsidm.so on NHWC tensors:

  ResnetBlock  = gn_stats -> conv3x3[GN+SiLU fused in, +bias +FiLM] -> gn_stats
                 -> conv3x3[GN+SiLU fused in, +bias, + res_conv(x) as extra K steps | + x]
  SelfAttention = gn_stats -> conv1x1[GN fused in] -> MFMA attention core -> conv1x1[+bias +x]
  Upsample / Downsample = conv3x3 with the nearest-x2 / stride-2 folded into the addressing
  skip concat = two-pointer conv input (never materialised)

There is no eager fallback: inputs must be ROCm tensors and the library must be built.
"""
import torch
from torch import nn

from .. import ops
from .._lib import act_dtype
from ..precision import forward_precision, is_16bit, resolve_precision, wide_weights


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def _need_eval(mod, dropout_p):
    if mod.training and dropout_p:
        raise RuntimeError(
            "hsidm: this module-level forward runs the fused inference kernels, which have no Dropout; in training mode go through "
            "GaussianDiffusion.forward / .p_losses (differentiable) or GaussianDiffusion.trainer(), or call .eval() first")


class _PackCache:
    """Re-packs weights when a parameter changed (load_state_dict, .to(), optimiser step)."""

    def __init__(self):
        self._store = {}

    def get(self, key, params, build):
        ver = tuple((p.data_ptr(), p._version) for p in params if p is not None)
        hit = self._store.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, build())
            self._store[key] = hit
        return hit[1]


def _kernel_only(what):
    raise RuntimeError("hsidm: %s holds parameters / names only; its arithmetic runs inside a fused kernel of the enclosing "
                       "module (there is no eager path) - call the enclosing module" % what)


class PositionalEncoding(nn.Module):
    """Parameter-free sinusoidal embedding of the noise level (reference unet.py:18-31): a placeholder in
    ``noise_level_mlp`` that keeps the reference's module indices; evaluated by hsidm_noise_film (UNet.noise_embedding)."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, noise_level):
        _kernel_only("PositionalEncoding")


class Swish(nn.Module):
    """x * sigmoid(x) (reference unet.py:53-55): fused into the operand staging of the following convolution."""

    def forward(self, x):
        _kernel_only("Swish")


class FeatureWiseAffine(nn.Module):
    """Holds the FiLM projection Linear(emb -> C) (reference unet.py:34-50).  The addition itself is the
    `film` operand of the block's first convolution."""

    def __init__(self, in_channels, out_channels, use_affine_level=False):
        super().__init__()
        self.use_affine_level = use_affine_level
        self.noise_func = nn.Sequential(nn.Linear(in_channels, out_channels * (1 + self.use_affine_level)))

    def forward(self, x, noise_embed):
        _kernel_only("FeatureWiseAffine")


class _HipModule(nn.Module):
    precision = None           # None -> package default (hsi_dmgasr_amd.set_default_precision)

    def _prec(self, precision=None):
        """Kernel set of this call.  An explicit `precision` (the sampler's per-step choice, the training step) is taken as named; a
        bare module call runs the set its mode prescribes for an output that IS the result (precision.forward_precision: the
        "fp16" policy -> the fp32 kernel set)."""
        if precision is not None:
            return resolve_precision(precision)
        return forward_precision(resolve_precision(self.precision))

    @staticmethod
    def _check_input(x):
        if not x.is_cuda:
            raise RuntimeError("hsidm: input is on %s; this module only runs on a ROCm device (no CPU fallback)" % x.device)
        if x.dtype != torch.float32:
            raise TypeError("hsidm: module inputs are float32 NCHW like the reference's, got %s" % x.dtype)


class Upsample(_HipModule):
    def __init__(self, dim):
        super().__init__()
        self.up = nn.Upsample(scale_factor=2, mode="nearest")
        self.conv = nn.Conv2d(dim, dim, 3, padding=1)
        self._cache = _PackCache()

    def _packed(self, precision):
        return self._cache.get(precision, [self.conv.weight, self.conv.bias],
                               lambda: ops.PackedConv(self.conv.weight, self.conv.bias, precision, fold_ups=True))

    def _run(self, x, precision):
        return ops.conv2d(x, self._packed(precision), ups=True, stats=True)

    def forward(self, x):
        self._check_input(x)
        p = self._prec()
        return ops.to_nchw(self._run(ops.to_nhwc(x, p), p), p)


class Downsample(_HipModule):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv2d(dim, dim, 3, 2, 1)
        self._cache = _PackCache()

    def _packed(self, precision):
        return self._cache.get(precision, [self.conv.weight, self.conv.bias],
                               lambda: ops.PackedConv(self.conv.weight, self.conv.bias, precision, fold_dn=True))

    def _run(self, x, precision):
        return ops.conv2d(x, self._packed(precision), stride=2, stats=True)

    def forward(self, x):
        self._check_input(x)
        p = self._prec()
        return ops.to_nchw(self._run(ops.to_nhwc(x, p), p), p)


class Block(_HipModule):
    """GroupNorm -> Swish -> Dropout -> Conv3x3 (reference unet.py:80-91) as one fused convolution."""

    def __init__(self, dim, dim_out, groups=32, dropout=0):
        super().__init__()
        self.block = nn.Sequential(
            nn.GroupNorm(groups, dim),
            Swish(),
            nn.Dropout(dropout) if dropout != 0 else nn.Identity(),
            nn.Conv2d(dim, dim_out, 3, padding=1),
        )
        self._dropout = dropout
        self._cache = _PackCache()

    def _packed(self, precision, out_nchw=False, proj=None):
        conv = self.block[3]
        params = [conv.weight, conv.bias] + ([proj.weight, proj.bias] if proj is not None else [])
        return self._cache.get((precision, out_nchw, proj is not None), params, lambda: ops.PackedConv(
            conv.weight, conv.bias, precision,
            proj_weight=None if proj is None else proj.weight, proj_bias=None if proj is None else proj.bias,
            out_nchw=out_nchw))

    def _run(self, x0, precision, x1=None, film=None, res=None, proj=None, proj_x0=None, proj_x1=None, out_nchw=False, sk_only=False,
             fused_only=False):
        _need_eval(self, self._dropout)
        gn = self.block[0]
        ab = ops.gn_scale_shift(x0, x1, gn.weight, gn.bias, gn.num_groups, precision, gn.eps)
        return ops.conv2d(x0, self._packed(precision, out_nchw, proj), x1=x1, gn_ab=ab, transform=ops.XF_AFFINE_SILU,
                          film=film, res=res, proj_x0=proj_x0, proj_x1=proj_x1, stats=not out_nchw, sk_only=sk_only, fused_only=fused_only)

    def forward(self, x):
        self._check_input(x)
        p = self._prec()
        return self._run(ops.to_nhwc(x, p), p, out_nchw=True)


class ResnetBlock(_HipModule):
    def __init__(self, dim, dim_out, noise_level_emb_dim=None, dropout=0, use_affine_level=False, norm_groups=32):
        super().__init__()
        self.noise_func = FeatureWiseAffine(noise_level_emb_dim, dim_out, use_affine_level)
        self.block1 = Block(dim, dim_out, groups=norm_groups)
        self.block2 = Block(dim_out, dim_out, groups=norm_groups, dropout=dropout)
        self.res_conv = nn.Conv2d(dim, dim_out, 1) if dim != dim_out else nn.Identity()
        self._cache = _PackCache()

    def _run(self, x0, x1, film, precision):
        """x = cat(x0, x1) on channels (x1 may be None); film: [B, dim_out] slice of the FiLM table ([B, 2*dim_out] = (gamma | beta)
        with use_affine_level: h = (1 + gamma) * block1(x) + beta, reference unet.py:44-47, as its own pass - the reference's UNet
        never enables it, so it is not fused into the convolution's epilogue)."""
        if self.noise_func.use_affine_level:
            h = ops.film_affine(self.block1._run(x0, precision, x1=x1), film.contiguous(), precision)
        else:
            h = self.block1._run(x0, precision, x1=x1, film=film)
        if isinstance(self.res_conv, nn.Conv2d):
            if ops.use_v2():
                # the persistent 3x3 kernel is single-phase (in every mode: the fp32 mode has its forms of it and of the 1x1 GEMM since
                # round 3); the 1x1 projection is its own launch and enters block2's epilogue as the residual.  (Measured and dropped, round 2: running it on a second stream beside block1's
                # convolution - a parallel branch of the captured step - at batches that leave workgroup slots free: the fork / join
                # costs more than the overlap returns, 2.98 vs 2.83 ms per step at 5 latents, 5.03 vs 4.93 at 40, in-box A/B.)
                # Few pixel tiles and a long contraction (one or two CAVE images per GPU on the 32x32 ... 8x8 levels): block2 runs in
                # its split-K form, where the projection is a few more one-tap chunks of the same launch.
                B, H, W, _ = h.shape
                # (not for a layer with hi + lo weights: the split-K kernel has no second weight pass and refuses the descriptor,
                # so the offer would only pack dead layouts and pay a workspace query per forward)
                if is_16bit(precision) and B * H * W <= 16384 and h.shape[3] % 128 == 0 and h.shape[3] >= 256 and \
                        not wide_weights(precision, h.shape[3], h.shape[3], 3):
                    out = self.block2._run(h, precision, proj=self.res_conv, proj_x0=x0, proj_x1=x1, sk_only=True)
                    if out is not None:
                        return out
                # 64-cout layers on whole 16x16 tiles (the 128x128 level): the projection as one-tap chunks of block2's own launch
                # (conv_v3.hip, PROJ) - no separate GEMM, no round trip of its result through HBM.  One-pass weight sets
                # (bf16, the fp16 policy's dithered sets, fp16x1), at most three 64-channel chunks: every projection of the shipped
                # UNet's 128x128 level (192 -> 64, 128 -> 64).  A layer with hi + lo weights (the experimental two-pass sets) keeps the GEMM.
                # (every condition the dispatch checks is checked HERE first - api.hip, v3_proj - so a refused offer never costs a
                # second GroupNorm table or dead packed layouts)
                cx = x0.shape[3] + (0 if x1 is None else x1.shape[3])
                if h.dtype in (torch.float16, torch.bfloat16) and h.shape[3] == 64 and H % 16 == 0 and W % 16 == 0 and cx % 8 == 0 and \
                        cx <= 192 and not wide_weights(precision, h.shape[3], h.shape[3], 3) and ops.use_fused_proj():
                    out = self.block2._run(h, precision, proj=self.res_conv, proj_x0=x0, proj_x1=x1, fused_only=True)
                    if out is not None:
                        return out
                rc = self.res_conv
                pk = self._cache.get(("proj", precision), [rc.weight, rc.bias], lambda: ops.PackedConv(rc.weight, rc.bias, precision))
                r = ops.conv2d(x0, pk, x1=x1)
                return self.block2._run(h, precision, res=r)
            return self.block2._run(h, precision, proj=self.res_conv, proj_x0=x0, proj_x1=x1)
        assert x1 is None
        return self.block2._run(h, precision, res=x0)

    def forward(self, x, time_emb):
        self._check_input(x)
        p = self._prec()
        lin = self.noise_func.noise_func[0]
        t = time_emb.reshape(x.shape[0], -1).contiguous()
        film = ops.noise_film(x.shape[0], t.shape[1], None, lin.weight, lin.bias, t_emb=t)
        return ops.to_nchw(self._run(ops.to_nhwc(x, p), None, film, p), p)


class SelfAttention(_HipModule):
    def __init__(self, in_channel, n_head=1, norm_groups=32):
        super().__init__()
        if in_channel % n_head:
            raise ValueError("in_channel must be a multiple of n_head")
        self.n_head = n_head
        self.norm = nn.GroupNorm(norm_groups, in_channel)
        self.qkv = nn.Conv2d(in_channel, in_channel * 3, 1, bias=False)
        self.out = nn.Conv2d(in_channel, in_channel, 1)
        self._cache = _PackCache()

    def _run(self, x, precision):
        pq = self._cache.get(("qkv", precision), [self.qkv.weight], lambda: ops.PackedConv(self.qkv.weight, None, precision))
        po = self._cache.get(("out", precision), [self.out.weight, self.out.bias],
                             lambda: ops.PackedConv(self.out.weight, self.out.bias, precision))
        ab = ops.gn_scale_shift(x, None, self.norm.weight, self.norm.bias, self.norm.num_groups, precision, self.norm.eps)
        qkv = ops.conv2d(x, pq, gn_ab=ab, transform=ops.XF_AFFINE)
        # (the reference's UNet only ever builds n_head = 1, unet.py:152: the fused MFMA kernel; more heads go through the
        # strided GEMM + softmax kernels)
        o = ops.attention(qkv, precision) if self.n_head == 1 else ops.attention_multihead(qkv, self.n_head, precision)
        return ops.conv2d(o, po, res=x, stats=True)

    def forward(self, input):
        self._check_input(input)
        p = self._prec()
        return ops.to_nchw(self._run(ops.to_nhwc(input, p), p), p)


class ResnetBlocWithAttn(_HipModule):
    def __init__(self, dim, dim_out, *, noise_level_emb_dim=None, norm_groups=32, dropout=0, with_attn=False):
        super().__init__()
        self.with_attn = with_attn
        self.res_block = ResnetBlock(dim, dim_out, noise_level_emb_dim, norm_groups=norm_groups, dropout=dropout)
        if with_attn:
            self.attn = SelfAttention(dim_out, norm_groups=norm_groups)

    def _run(self, x0, x1, film, precision):
        x = self.res_block._run(x0, x1, film, precision)
        return self.attn._run(x, precision) if self.with_attn else x

    def forward(self, x, time_emb):
        self._check_input(x)
        p = self._prec()
        lin = self.res_block.noise_func.noise_func[0]
        t = time_emb.reshape(x.shape[0], -1).contiguous()
        film = ops.noise_film(x.shape[0], t.shape[1], None, lin.weight, lin.bias, t_emb=t)
        return ops.to_nchw(self._run(ops.to_nhwc(x, p), None, film, p), p)


class UNet(_HipModule):
    def __init__(self, in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 4, 8, 8),
                 attn_res=(8), res_blocks=3, dropout=0, with_noise_level_emb=True, image_size=128, precision=None):
        super().__init__()
        if not with_noise_level_emb:
            # the reference cannot build this variant either: its ResnetBlock constructs nn.Linear(None, dim_out)
            # (unet.py:97-98 with noise_level_emb_dim=None) and raises the same TypeError
            raise TypeError("hsidm: with_noise_level_emb=False leaves the FiLM projections without an input width "
                            "(the reference's constructor fails on nn.Linear(None, ...) in the same way)")
        self.precision = precision
        attn_res = (attn_res,) if isinstance(attn_res, int) else tuple(attn_res)
        emb = inner_channel
        self.noise_level_mlp = nn.Sequential(
            PositionalEncoding(inner_channel),
            nn.Linear(inner_channel, inner_channel * 4),
            Swish(),
            nn.Linear(inner_channel * 4, inner_channel),
        )
        self._emb_dim = emb
        self._in_channel = in_channel

        def unit(cin, cout, attn):
            return ResnetBlocWithAttn(cin, cout, noise_level_emb_dim=emb, norm_groups=norm_groups, dropout=dropout,
                                      with_attn=attn)

        # encoder: stem conv, then per level `res_blocks` units (+ a stride-2 conv between levels)
        width = inner_channel
        skip_widths = [width]
        res = image_size
        downs = [nn.Conv2d(in_channel, inner_channel, kernel_size=3, padding=1)]
        for level, mult in enumerate(channel_mults):
            for _ in range(res_blocks):
                downs.append(unit(width, inner_channel * mult, res in attn_res))
                width = inner_channel * mult
                skip_widths.append(width)
            if level != len(channel_mults) - 1:
                downs.append(Downsample(width))
                skip_widths.append(width)
                res //= 2
        self.downs = nn.ModuleList(downs)
        self.mid = nn.ModuleList([unit(width, width, True), unit(width, width, False)])
        # decoder: per level `res_blocks + 1` units fed by (x ++ skip), nearest-x2 conv between levels
        ups = []
        for level in reversed(range(len(channel_mults))):
            for _ in range(res_blocks + 1):
                ups.append(unit(width + skip_widths.pop(), inner_channel * channel_mults[level], res in attn_res))
                width = inner_channel * channel_mults[level]
            if level > 0:
                ups.append(Upsample(width))
                res *= 2
        self.ups = nn.ModuleList(ups)
        self.final_conv = Block(width, default(out_channel, in_channel), groups=norm_groups)
        self._cache = _PackCache()

    # ------------------------------------------------------------------------------------------------
    def _res_units(self):
        return [m for m in list(self.downs) + list(self.mid) + list(self.ups) if isinstance(m, ResnetBlocWithAttn)]

    def _film_pack(self):
        units = self._res_units()
        lins = [u.res_block.noise_func.noise_func[0] for u in units]
        params = [p for l in lins for p in (l.weight, l.bias)]

        def build():
            wf = torch.cat([l.weight.detach().float() for l in lins], dim=0).contiguous()
            bf = torch.cat([l.bias.detach().float() for l in lins], dim=0).contiguous()
            offs, o = [], 0
            for l in lins:
                offs.append((o, o + l.weight.shape[0]))
                o += l.weight.shape[0]
            return wf, bf, offs
        return self._cache.get("film", params, build)

    def _stem_pack(self, precision):
        c = self.downs[0]
        return self._cache.get(("stem", precision), [c.weight, c.bias], lambda: ops.PackedConv(c.weight, c.bias, precision))

    def _mlp(self):
        l1, l2 = self.noise_level_mlp[1], self.noise_level_mlp[3]
        return (l1.weight, l1.bias, l2.weight, l2.bias)

    def noise_embedding(self, time):
        """noise_level_mlp(time) evaluated by the HIP kernel: (B,1) -> (B,1,emb)  (reference unet.py:240-241)."""
        self._check_input(time)
        wf, bf, _ = self._film_pack()
        B = time.shape[0]
        _, t = ops.noise_film(B, self._emb_dim, self._mlp(), wf, bf, gamma=time.reshape(B).contiguous(), want_t=True)
        return t.view(B, 1, -1)

    def run_nhwc(self, stem_in, film, precision):
        """Whole network on an NHWC stem input (cat(cond, x), zero-padded to 8 channels).  Returns NCHW fp32."""
        _, _, offs = self._film_pack()
        k = 0
        skips = []
        x = None
        for layer in self.downs:
            if isinstance(layer, ResnetBlocWithAttn):
                lo, hi = offs[k]
                k += 1
                x = layer._run(x, None, film[:, lo:hi], precision)
            elif isinstance(layer, Downsample):
                x = layer._run(x, precision)
            else:
                x = ops.conv2d(stem_in, self._stem_pack(precision), stats=True)
            skips.append(x)
        for layer in self.mid:
            lo, hi = offs[k]
            k += 1
            x = layer._run(x, None, film[:, lo:hi], precision)
        for layer in self.ups:
            if isinstance(layer, ResnetBlocWithAttn):
                lo, hi = offs[k]
                k += 1
                x = layer._run(x, skips.pop(), film[:, lo:hi], precision)
            else:
                x = layer._run(x, precision)
        return self.final_conv._run(x, precision, out_nchw=True)

    def forward_pair(self, cond, x, *, gamma=None, level_table=None, t_ptr=None, precision=None):
        """eps = UNet(cat(cond, x), gamma) without materialising the concat.  gamma: [B] device tensor, or
        (level_table, t_ptr) for a graph-replayable device-side lookup (reference diffusion.py:154-158)."""
        p = self._prec(precision)
        B = x.shape[0]
        wf, bf, _ = self._film_pack()
        if gamma is None:
            # sampler mode: every sample of the batch is at the same noise level (reference diffusion.py:154-155), so the
            # embedding MLP and the 27 FiLM projections are evaluated once and read with batch stride 0
            film = ops.noise_film(1, self._emb_dim, self._mlp(), wf, bf, level_table=level_table, t_ptr=t_ptr).expand(B, -1)
        else:
            film = ops.noise_film(B, self._emb_dim, self._mlp(), wf, bf, gamma=gamma)
        stem_in = ops.to_nhwc(cond, p, x1=x) if cond is not None else ops.to_nhwc(x, p)
        return self.run_nhwc(stem_in, film, p)

    def forward(self, x, time):
        self._check_input(x)
        self._check_input(time)
        if x.shape[1] != self._in_channel:
            raise ValueError("expected %d input channels, got %d" % (self._in_channel, x.shape[1]))
        B = x.shape[0]
        return self.forward_pair(None, x.contiguous(), gamma=time.reshape(B).contiguous())
