"""Host-side operator layer: weight packing and thin wrappers that enqueue the HIP kernels.

Tensors here are the path's internal representation: NHWC activations (torch shape [B, H, W, C]) in
the storage type of the precision mode (bf16 / fp32), fp32 everywhere else.  torch only supplies
device memory and the current stream.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ACT_LEAKY, ACT_NONE, XF_AFFINE, XF_AFFINE_SILU, XF_NONE  # noqa: F401 (re-exported)
from .precision import dither_offset, dither_phase, wide_weights


# ------------------------------------------------------------------------------------------- packing
def pick_bn(cout, out_nchw=False):
    """Output-channel slice per workgroup (must match the packed weights)."""
    if out_nchw:
        return 32 if cout <= 32 else 128
    if cout <= 32:
        return 32
    if cout <= 64:
        return 64
    return 128


LOG2E = 1.4426950408889634


def pack_layouts(weight, precision, proj_weight=None, out_nchw=False, fold_ups=False, fold_dn=False, proj_scale=1.0):
    """The packed layouts of one convolution as tensors of `weight`'s dtype (values are only moved, or - for the folded
    upsample kernels - summed): {"w": [step][Cout_pad][BK], "w_v2": ..., "w_up4": ..., "w_dn4": ...} plus the meta data
    PackedConv carries.  Called with the real fp32 weights (PackedConv) and with float64 index-valued tensors
    (training.PackMap: the layouts become gather maps into the flat parameter buffer)."""
    prec = _lib.prec_id(precision)
    bk = 32 if prec == _lib.F32X3 else 64
    b16 = prec != _lib.F32X3               # bf16 / fp16: the same layouts
    cout, cin, kh, kw = weight.shape
    assert kh == kw and kh in (1, 3)
    if cin % 8:     # NHWC tensors carry channels in 16-B vectors: pad the K axis with zero weights
        weight = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, 8 - cin % 8))
        cin = weight.shape[1]
    bn = pick_bn(cout, out_nchw)
    cpad = (cout + bn - 1) // bn * bn
    parts = [PackedConv._steps(weight, cpad, bk)]
    proj_cin = 0
    if proj_weight is not None:
        assert proj_weight.shape[2] == 1 and proj_weight.shape[0] == cout
        if proj_weight.shape[1] % 8:
            proj_weight = torch.nn.functional.pad(proj_weight, (0, 0, 0, 0, 0, 8 - proj_weight.shape[1] % 8))
        proj_cin = proj_weight.shape[1]
        parts.append(PackedConv._steps(proj_weight, cpad, bk))
    lay = dict(w=torch.cat(parts, dim=0).contiguous(), w_v2=None, w_up4=None, w_dn4=None)
    meta = dict(ksize=kh, cin=cin, cout=cout, bn=bn, cpad=cpad, proj_cin=proj_cin, prec=prec, tap_major=False)
    # register-streaming order for the persistent bf16 3x3 kernel: [step][Cout_pad/32][kk][lane][8] with
    # lane = (k-half h, cout r): element j = W[cout = 32*slice + r][k = 16*kk + 8*h + j]
    # bn == 64 (conv_v3's PROJ forms; proj_scale = log2(e)): the projection steps of the register-streaming order carry log2(e) - the
    # persistent kernels stage log2(e) * silu(.) and undo the factor on the accumulators, i.e. on the projection's products too - and
    # are padded with zero steps to THREE per item (conv_v3.hip: the one-pass form pulls three steps through its 3-step weight ring
    # whatever the projection's width, so that the ring's phase is the same at every item); the "w" order, which the LDS-tiled kernel
    # reads, stays unscaled and unpadded
    if b16 and proj_weight is not None and kh == 3 and not out_nchw and (bn == 128 or (bn == 64 and proj_scale != 1.0)):
        steps = lay["w"]
        if proj_scale != 1.0:
            pj = parts[1] * proj_scale
            if pj.shape[0] < 3:
                pj = torch.cat([pj, pj.new_zeros((3 - pj.shape[0],) + tuple(pj.shape[1:]))], dim=0)
            steps = torch.cat([parts[0], pj], dim=0).contiguous()
        lay["w_v2_steps"] = steps
        lay["w_v2"] = PackedConv._lanes(steps, cpad)            # read by the split-K kernel (conv_sk.hip: projection chunks) or conv_v3's PROJ forms
    # fp32 mode: the persistent 3x3 kernel has an fp32 form too (conv_v2.h, AP = 2: fp32 storage, hi + lo operands); it reads the same
    # register-streaming order with 64-channel steps
    f32_v2 = (not b16) and proj_weight is None and not out_nchw and bn in (64, 128) and (kh == 3 or cout % bn == 0)
    if f32_v2:
        wv = PackedConv._steps(weight, cpad, 64).contiguous()
        if kh == 1 and wv.shape[0] % 2:          # 1x1 GEMM kernel (conv1x1_g.hip): K padded to a multiple of 128
            wv = torch.cat([wv, torch.zeros_like(wv[:1])], dim=0)
        lay["w_v2"] = PackedConv._lanes(wv, cpad)
    if b16 and proj_weight is None and (not out_nchw or (kh == 3 and bn == 32)):
        wv = lay["w"]
        meta["tap_major"] = kh == 3 and cin == 8
        if meta["tap_major"]:                    # stem-like convs: tap-major GEMM (conv1x1_g IM mode), k = 8*tap + c -> 128
            w2 = weight.reshape(cout, 8, 9).permute(0, 2, 1).reshape(cout, 72, 1, 1)
            wv = PackedConv._steps(w2, cpad, bk).contiguous()
        if kh == 1 and wv.shape[0] % 2:          # 1x1 GEMM kernel (conv1x1_g.hip): K padded to a multiple of 128
            wv = torch.cat([wv, torch.zeros_like(wv[:1])], dim=0)
        lay["w_v2"] = PackedConv._lanes(wv, cpad)
    # nearest-x2 + conv3x3 as four 2x2 convs on the input grid (include/hsidm.h, HSIDM_UPS_FOLDED): taps that read
    # the same input pixel are summed, then rounded to bf16 once
    if fold_ups and lay["w_v2"] is not None and kh == 3 and bn == 128:
        rows = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}          # parity -> 3x3 taps behind each of the two 2x2 taps
        pars = []
        for py in (0, 1):
            for px in (0, 1):
                f = torch.stack([torch.stack([sum(weight[:, :, dy, dx] for dy in rows[py][ty] for dx in rows[px][tx])
                                              for tx in (0, 1)], dim=-1) for ty in (0, 1)], dim=-2)
                pars.append(PackedConv._steps(f, cpad, 64))
        lay["w_up4_steps"] = torch.cat(pars, dim=0).contiguous()            # [step][Cout_pad][64]: what the sparse low halves are cut from
        lay["w_up4"] = PackedConv._lanes(lay["w_up4_steps"], cpad)
    # stride-2 conv over the four input-parity planes (include/hsidm.h, hsidm_conv_desc.stride)
    if fold_dn and lay["w_v2"] is not None and kh == 3 and bn in (64, 128):
        tapmap = {0: (None, 1), 1: (0, 2)}                      # plane parity -> 3x3 tap behind each of the two window taps
        planes = []
        for ry in (0, 1):
            for rx in (0, 1):
                f = torch.zeros(cout, cin, 2, 2, dtype=weight.dtype, device=weight.device)
                for ty in (0, 1):
                    for tx in (0, 1):
                        dy, dx = tapmap[ry][ty], tapmap[rx][tx]
                        if dy is not None and dx is not None:
                            f[:, :, ty, tx] = weight[:, :, dy, dx]
                planes.append(PackedConv._steps(f, cpad, 64))
        lay["w_dn4_steps"] = torch.cat(planes, dim=0).contiguous()
        lay["w_dn4"] = PackedConv._lanes(lay["w_dn4_steps"], cpad)
    return lay, meta


class PackedConv:
    """Weights of one hsidm_conv2d launch in the kernel's streaming order.

    Layout (include/hsidm.h): bf16 [step][Cout_pad][BK] with step = (phase, cin-chunk, tap); phase 1 is
    the optional fused 1x1 projection (ResnetBlock.res_conv, reference unet.py:102-103).
    """

    def __init__(self, weight, bias, precision, proj_weight=None, proj_bias=None, out_nchw=False, fold_ups=False, fold_dn=False):
        dev = weight.device
        cout, cin, ksz = weight.shape[0], weight.shape[1], weight.shape[2]
        # the "fp32h" kernel set (precision.py; include/hsidm.h, HSIDM_F32H): fp32 tensors; a layer the persistent kernels take multiplies
        # ONE fp16 activation operand by fp16 hi + lo weights; every other layer (fused projection, NCHW output, odd slices) and every
        # SHAPE the dispatch refuses (conv2d asks) is the fp32 set's own - same tensors in and out
        self._fallback, self._fallback_src = None, None
        if precision == "fp32h":
            self._fallback_src = (weight, bias, proj_weight, proj_bias, out_nchw, fold_ups, fold_dn)
        wide16 = _lib.prec_id(precision) == _lib.F16 and wide_weights(precision, cout, cin + (-cin) % 8, ksz)
        # (a fused projection on a 64-cout slice of a one-pass layer is conv_v3's: its steps carry log2(e), see pack_layouts)
        v3_proj = proj_weight is not None and _lib.prec_id(precision) != _lib.F32X3 and pick_bn(cout, out_nchw) == 64 and not wide16
        lay, meta = pack_layouts(weight.detach().float(), precision, None if proj_weight is None else proj_weight.detach().float(),
                                 out_nchw, fold_ups, fold_dn, proj_scale=LOG2E if v3_proj else 1.0)
        self._set_meta(meta, precision, out_nchw)
        half = precision == "fp32h" and lay["w_v2"] is not None and proj_weight is None
        if precision == "fp32h" and not half:
            self.precision = "fp32"
        if half:
            self.prec = _lib.F32H
        b = None if bias is None else bias.detach().float().clone()
        if proj_weight is not None and proj_bias is not None:
            b = proj_bias.detach().float().clone() if b is None else b + proj_bias.detach().float()
        w = lay["w"]
        et = torch.float16 if self.prec in (_lib.F16, _lib.F32H) else torch.bfloat16    # element type of the packed weights (fp32 mode: bf16 hi + lo)
        # (F32H: the LDS-tiled kernel has no form of it - its layout is not packed)
        self.w_hi = None if half else w.to(et).contiguous()
        # low halves: fp32 mode (three MFMAs per product) and the generic kernel's fp16 form (always hi + lo weights)
        self.w_lo = (w - self.w_hi.float()).to(et).contiguous() if (self.prec != _lib.BF16 and not half) else None
        self.bias = None if b is None else b.to(dev).contiguous()
        # fp16 mode: does this layer multiply by hi + lo weights on the persistent kernels too (precision.wide_weights)?
        self.wide = (self.prec == _lib.F16 and wide_weights(precision, self.cout, self.cin, self.ksize)) or self.prec in (_lib.F32X3, _lib.F32H)
        # one-pass layouts of a dithered kernel set ("fp16d<k>", precision.py): step k of a chain multiplies by fp16(w + d_k * ulp(w)), the
        # K offsets d_k spread over (-1/2, 1/2) ulp - the mean weight over K steps is w to 1 / (2K) ulp, so the weight rounding stops
        # being a bias of the chain.
        ph = dither_phase(precision)
        dith = ph is not None and not self.wide

        def rnd(t):
            if not dith:
                return t.to(et).contiguous()
            ulp = torch.exp2(torch.floor(torch.log2(t.abs().clamp_min(2.0 ** -14))) - 10.0)      # fp16: 10 stored significand bits; subnormals share 2^-24
            return (t + dither_offset(*ph) * ulp).to(et).contiguous()
        for name in ("w_v2", "w_up4", "w_dn4"):
            t = lay[name]
            hi = None if t is None else rnd(t)
            setattr(self, name, hi)
            setattr(self, name + "_lo", (t - hi.float()).to(et).contiguous() if (hi is not None and self.wide) else None)
        # fp16 hi + lo layers on the plain 3x3 schedule: the low halves once more, 2:4 structured-sparse, for the kernels that run the
        # second pass on v_smfmac (include/hsidm.h: w_v2_ls / w_v2_li; today conv_v3's 64-cout form)
        self.w_v2_ls = self.w_v2_li = self.w_up4_ls = self.w_up4_li = self.w_dn4_ls = self.w_dn4_li = None
        if (self.prec == _lib.F16 and self.wide and self.w_v2_lo is not None and self.ksize == 3 and
                not out_nchw and not self.tap_major and self.cin % 64 == 0):
            st = w if proj_weight is None else lay["w_v2_steps"]          # (with a projection: its steps scaled, see pack_layouts)
            self.w_v2_ls, self.w_v2_li = PackedConv._sparse_lo(st - st.to(et).float(), meta["cpad"])
            for name in ("w_up4", "w_dn4"):                     # the folded up / down-sampling layouts (their own steps: summed / re-ordered taps)
                st = lay.get(name + "_steps")
                if st is not None and getattr(self, name + "_lo") is not None:
                    ls, li = PackedConv._sparse_lo(st - st.to(et).float(), meta["cpad"])
                    setattr(self, name + "_ls", ls)
                    setattr(self, name + "_li", li)

    def fallback(self):
        """The fp32 set's packed weights of an "fp32h" layer (built on first use: a shape the HSIDM_F32H forms refuse)."""
        if self._fallback is None:
            weight, bias, proj_weight, proj_bias, out_nchw, fold_ups, fold_dn = self._fallback_src
            self._fallback = PackedConv(weight, bias, "fp32", proj_weight=proj_weight, proj_bias=proj_bias, out_nchw=out_nchw,
                                        fold_ups=fold_ups, fold_dn=fold_dn)
        return self._fallback

    def _set_meta(self, meta, precision, out_nchw):
        self.ksize, self.cin, self.cout, self.bn = meta["ksize"], meta["cin"], meta["cout"], meta["bn"]
        self.proj_cin, self.prec, self.tap_major = meta["proj_cin"], meta["prec"], meta["tap_major"]
        self.precision, self.out_nchw = precision, out_nchw

    @classmethod
    def from_buffers(cls, meta, precision, out_nchw, w_hi, w_lo, w_v2, w_dn4, bias, w_v2_lo=None, w_dn4_lo=None):
        """A PackedConv over caller-owned packed buffers (the training step refreshes them in place every iteration)."""
        self = cls.__new__(cls)
        self._set_meta(meta, precision, out_nchw)
        self.w_hi, self.w_lo, self.w_v2, self.w_up4, self.w_dn4, self.bias = w_hi, w_lo, w_v2, None, w_dn4, bias
        # fp32 mode: the low halves of the register-streaming layouts (the persistent kernel's fp32 form reads both)
        self.wide, self.w_v2_lo, self.w_up4_lo, self.w_dn4_lo = w_v2_lo is not None, w_v2_lo, None, w_dn4_lo
        self.w_v2_ls = self.w_v2_li = self.w_up4_ls = self.w_up4_li = self.w_dn4_ls = self.w_dn4_li = None
        return self

    @staticmethod
    def _lanes(w_steps, cpad):
        """[step][Cout_pad][64] -> [step][Cout_pad/32][kk 4][lane = (k-half, cout r)][8]: one wave-load per MFMA B fragment."""
        st = w_steps.shape[0]
        return w_steps.reshape(st, cpad // 32, 32, 4, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous()

    @staticmethod
    def _sparse_lo(lo, cpad):
        """[step][Cout_pad][64] fp32 low halves -> (values fp16 [step][Cout_pad/32][half 2][lane 64][8], index words int32
        [step][Cout_pad/32][lane 64]): of every four consecutive input channels the two of larger magnitude, in the operand order of
        v_smfmac_f32_16x16x64_f16's A side (include/hsidm.h: w_v2_ls; measured by tools/ubench/smfmac_probe.hip): lane 16 * kgroup +
        cout % 16 holds channels 16 * kgroup .. + 15; its stored slots 2m, 2m + 1 are the kept values of channels 16 * kgroup + 4m .. + 3
        in ascending position, bits [2s+1 : 2s] of the index half-word the position of slot s; low half-word = first 16-cout half."""
        st = lo.shape[0]
        g = lo.reshape(st, cpad, 16, 4)
        pos = g.abs().topk(2, dim=3).indices.sort(dim=3).values                     # [st, cpad, 16 groups, 2] positions, ascending
        vals = g.gather(3, pos)
        # cout = 32 * slice + 16 * half + i; group index = 4 * kgroup + m; slot = 2 * m + s
        vals = vals.reshape(st, cpad // 32, 2, 16, 4, 4, 2).permute(0, 1, 2, 4, 3, 5, 6).reshape(st, cpad // 32, 2, 64, 8)
        shifts = (2 * (2 * torch.arange(4, device=lo.device).view(4, 1) + torch.arange(2, device=lo.device).view(1, 2))).to(torch.int64)   # [m, s]
        word = (pos.reshape(st, cpad // 32, 2, 16, 4, 4, 2).to(torch.int64) << shifts).sum(dim=(5, 6))       # [st, slice, half, i, kgroup] 16-bit words
        word = word.permute(0, 1, 2, 4, 3).reshape(st, cpad // 32, 2, 64)
        idx = (word[:, :, 0] | (word[:, :, 1] << 16))
        idx = torch.where(idx >= 2 ** 31, idx - 2 ** 32, idx).to(torch.int32)
        return vals.to(torch.float16).contiguous(), idx.contiguous()

    @staticmethod
    def _steps(w, cpad, bk):
        cout, cin, kh, kw = w.shape
        taps = kh * kw
        nch = (cin + bk - 1) // bk
        wp = torch.zeros(cpad, nch * bk, taps, dtype=w.dtype, device=w.device)
        wp[:cout, :cin] = w.reshape(cout, cin, taps)
        # [cpad][chunk][k][tap] -> [chunk][tap][cpad][k]
        return wp.reshape(cpad, nch, bk, taps).permute(1, 3, 0, 2).reshape(nch * taps, cpad, bk)


# ------------------------------------------------------------------------------------------- kernels
def conv2d(x0, pw, *, x1=None, gn_ab=None, transform=XF_NONE, film=None, res=None, res_scale=1.0,
           act=ACT_NONE, stride=1, ups=False, proj_x0=None, proj_x1=None, stats=False, sk_only=False, fused_only=False):
    """out = res_scale * act(conv(T(cat(x0, x1))) [+ proj(cat(proj_x0, proj_x1))] + bias + film) + res.

    sk_only: launch only if the dispatch takes its split-K form (few pixel tiles, long contraction), else return None - how a
    ResnetBlock offers its fused-projection descriptor to the kernels of the throughput modes that accept one; fused_only: likewise
    for the persistent kernel's projection form (one-pass 64-cout layers: conv_v3.hip, PROJ).

    stats=True: the kernel also writes per-(image, tile part, channel) sums of `out`; they ride on the returned
    tensor as ``out._hsidm_stats = (slab [B, nsplit, C, 2], nsplit)`` and feed gn_scale_shift / ca_vector."""
    B, H, W, C0 = x0.shape
    C1 = 0 if x1 is None else x1.shape[3]
    assert C0 + C1 == pw.cin, "conv input channels %d+%d != packed %d" % (C0, C1, pw.cin)
    Ho, Wo = (2 * H, 2 * W) if ups else (((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W))
    dt = x0.dtype
    if pw.out_nchw:
        out = torch.empty((B, pw.cout, Ho, Wo), dtype=torch.float32, device=x0.device)
    else:
        out = torch.empty((B, Ho, Wo, pw.cout), dtype=dt, device=x0.device)
    d = _lib.ConvDesc()
    p0 = d.ph[0]
    p0.src0, p0.src1, p0.gn_ab = _lib.ptr(x0), _lib.ptr(x1), _lib.ptr(gn_ab)
    p0.C0, p0.C1, p0.transform, p0.ntaps = C0, C1, transform, pw.ksize * pw.ksize
    d.nphase = 1
    if pw.proj_cin:
        assert proj_x0 is not None
        q0 = proj_x0.shape[3]
        q1 = 0 if proj_x1 is None else proj_x1.shape[3]
        assert q0 + q1 == pw.proj_cin and proj_x0.shape[:3] == x0.shape[:3]
        p1 = d.ph[1]
        p1.src0, p1.src1, p1.gn_ab = _lib.ptr(proj_x0), _lib.ptr(proj_x1), None
        p1.C0, p1.C1, p1.transform, p1.ntaps = q0, q1, XF_NONE, 1
        d.nphase = 2
    d.w_hi, d.w_lo, d.bias = _lib.ptr(pw.w_hi), _lib.ptr(pw.w_lo), _lib.ptr(pw.bias)
    d.w_v2 = _lib.ptr(pw.w_v2) if _use_v2 else None
    w_v2_lo = pw.w_v2_lo
    if pw.tap_major and (ups or stride != 1 or transform != XF_NONE or act != ACT_NONE or film is not None or x1 is not None or
                         pw.cout % pw.bn or pw.bn < 64 or (Ho * Wo) % 64 or Ho * Wo < 128 or (Wo & (Wo - 1))):
        d.w_v2 = None                 # the tap-major layout is only read by the GEMM kernel (include/hsidm.h)
    folded = bool(ups) and _use_v2 and _fold_ups and pw.w_up4 is not None
    if folded:
        d.w_v2, w_v2_lo = _lib.ptr(pw.w_up4), pw.w_up4_lo
    planes = stride == 2 and _use_v2 and _fold_ups and pw.w_dn4 is not None and H % 2 == 0 and W % 2 == 0
    if stride == 2:
        d.w_v2, w_v2_lo = (_lib.ptr(pw.w_dn4), pw.w_dn4_lo) if planes else (None, None)
    d.w_v2_lo = _lib.ptr(w_v2_lo) if d.w_v2 else None
    ls = li = None                  # the low halves 2:4-compressed, in the steps of whichever register-streaming layout this launch reads
    if d.w_v2 and d.w_v2_lo:
        which = "w_up4" if folded else ("w_dn4" if stride == 2 else "w_v2")
        ls, li = getattr(pw, which + "_ls", None), getattr(pw, which + "_li", None)
    d.w_v2_ls, d.w_v2_li = (_lib.ptr(ls), _lib.ptr(li)) if ls is not None else (None, None)
    if film is not None:          # a column slice of the [B, F] FiLM table
        assert film.stride(1) == 1 and film.shape == (B, pw.cout)
        d.film, d.film_stride = film.data_ptr(), film.stride(0)
    d.res, d.res_scale, d.out, d.stats = _lib.ptr(res), float(res_scale), _lib.ptr(out), None
    d.B, d.Hin, d.Win, d.Hout, d.Wout, d.Cout = B, H, W, Ho, Wo, pw.cout
    d.ksize, d.stride, d.ups, d.act = pw.ksize, stride, (UPS_FOLDED if folded else int(bool(ups))), act
    d.out_nchw, d.prec, d.bn = int(pw.out_nchw), pw.prec, pw.bn
    if pw.prec == _lib.F32H and _lib.lib().hsidm_conv_kernel_id(C.byref(d)) < 0:
        # a shape the "fp32h" forms do not take (two-image tiles of narrow maps, the GEMM's GroupNorm prologue): the fp32 set's kernels
        return conv2d(x0, pw.fallback(), x1=x1, gn_ab=gn_ab, transform=transform, film=film, res=res, res_scale=res_scale, act=act,
                      stride=stride, ups=ups, proj_x0=proj_x0, proj_x1=proj_x1, stats=stats, sk_only=sk_only, fused_only=fused_only)
    nb = 0
    if d.w_v2 and not pw.out_nchw and pw.ksize == 3 and not ups and pw.bn == 128 and pw.cin >= 200:
        # few pixel tiles x a long contraction (the 8x8 / 16x16 levels at small batches): the split-K form needs scratch
        # (set before the statistics query: the slab's split count depends on the kernel the dispatch picks)
        nb = _lib.lib().hsidm_conv_workspace_bytes(C.byref(d))
        if nb > 0:
            ws = _sk_workspace(int(nb), x0.device)
            d.workspace, d.workspace_bytes = _lib.ptr(ws), ws.numel()
    if sk_only and nb <= 0:
        return None
    if fused_only:              # launch only if a persistent kernel takes the projection as part of its own contraction (today: conv_v3, PROJ)
        kid = _lib.lib().hsidm_conv_kernel_id(C.byref(d))
        if kid < 0 or (kid & 15) != 4:
            return None
    if stats and not pw.out_nchw:
        nsplit = _lib.lib().hsidm_conv_stats_nsplit(C.byref(d))
        if nsplit <= 0:
            _lib.check(nsplit, "conv_stats_nsplit")
        slab = torch.empty((B, nsplit, pw.cout, 2), dtype=torch.float32, device=x0.device)
        d.stats = _lib.ptr(slab)
        out._hsidm_stats = (slab, nsplit)
    trk = getattr(pw, "_track", None)
    if trk is not None:     # the training step records which packed layout the dispatch reads (training.Trainer._prune_layouts)
        kid = _lib.lib().hsidm_conv_kernel_id(C.byref(d))
        trk.add("w" if (kid & 15) == 0 else ("w_dn4" if planes else "w_v2"))
    if _conv_probe is None:
        _lib.check(_lib.lib().hsidm_conv2d(C.byref(d), _lib.stream_ptr()), "conv2d")
    else:   # measurement hook (bench.py): HIP events on the launch stream around this one kernel
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(_lib.lib().hsidm_conv2d(C.byref(d), _lib.stream_ptr()), "conv2d")
        e1.record()
        k_total = pw.cin * (4 if folded else (16 if planes else pw.ksize * pw.ksize)) + pw.proj_cin      # multiplications actually executed
        kid = _lib.lib().hsidm_conv_kernel_id(C.byref(d))
        label = "%s bn%d %s k%d s%d%s%s" % (("conv_igemm", "conv_v2", "-", "conv1x1_g", "conv_v3", "conv_sk")[kid & 15], kid >> 8,
                                            ("8x16", "8x8x2", "8x8")[(kid >> 4) & 3], pw.ksize, stride,
                                            " gn+silu" if transform == XF_AFFINE_SILU else (" gn" if transform == XF_AFFINE else "") +
                                            (" up4" if folded else (" ups" if ups else (" dn4" if planes else ""))),
                                            (" nchw" if pw.out_nchw else "") + (" +proj" if pw.proj_cin else ""))
        _conv_probe.append(dict(e0=e0, e1=e1, flops=2.0 * B * Ho * Wo * pw.cout * k_total, bn=pw.bn, ksize=pw.ksize, kernel=label,
                                bytes=(B * H * W * (C0 + C1) + B * Ho * Wo * pw.cout) * x0.element_size() + (pw.w_hi if pw.w_hi is not None else pw.w_v2).numel() * 2,
                                stride=stride, ups=bool(ups), cin=pw.cin, cout=pw.cout, hw=(Ho, Wo),
                                out_nchw=pw.out_nchw, tile=0 if Wo >= 16 else 1))
    return out


class PackedPair:
    """Weights of one hsidm_conv1x1_pair launch: two 64 -> 64 1x1 convolutions (W1, b1), (W2, b2) as the two steps of the
    register-streaming order (include/hsidm.h), high and low halves in the mode's operand type (fp32 mode: bf16; fp16 mode: fp16)."""

    def __init__(self, w1, b1, w2, b2, precision):
        assert tuple(w1.shape) == (64, 64, 1, 1) and tuple(w2.shape) == (64, 64, 1, 1)
        self.prec = _lib.prec_id(precision)
        assert self.prec in (_lib.F32X3, _lib.F16)
        et = torch.float16 if self.prec == _lib.F16 else torch.bfloat16
        st = torch.cat([PackedConv._steps(w1.detach().float(), 64, 64), PackedConv._steps(w2.detach().float(), 64, 64)], dim=0)
        lay = PackedConv._lanes(st.contiguous(), 64)
        self.w = lay.to(et).contiguous()
        self.w_lo = (lay - self.w.float()).to(et).contiguous()
        self.b1 = None if b1 is None else b1.detach().float().contiguous()
        self.b2 = None if b2 is None else b2.detach().float().contiguous()


def conv1x1_pair(x, pp, act=ACT_NONE, stats=False):
    """out = W2 act(W1 x + b1) + b2 on an NHWC [B, H, W, 64] tensor in one launch (hsidm_conv1x1_pair: the spectral ResAttentionBlock's
    body, reference common.py:250-271 / AE.py:102-109); stats=True: the output's per-64-pixel-group sums ride on the result like
    conv2d's (``out._hsidm_stats``)."""
    B, H, W, Cc = x.shape
    assert Cc == 64 and (H * W) % 64 == 0 and x.is_contiguous()
    out = torch.empty_like(x)
    slab = None
    if stats:
        nsplit = H * W // 64
        slab = torch.empty((B, nsplit, 64, 2), dtype=torch.float32, device=x.device)
        out._hsidm_stats = (slab, nsplit)
    _lib.check(_lib.lib().hsidm_conv1x1_pair(pp.prec, _lib.ptr(x), _lib.ptr(pp.w), _lib.ptr(pp.w_lo), _lib.ptr(pp.b1), int(act), _lib.ptr(pp.b2),
                                             _lib.ptr(out), _lib.ptr(slab), B * H * W, H * W, _lib.stream_ptr()), "conv1x1_pair")
    return out


_sk_ws = {}
_sk_retired = []


def _sk_workspace(nbytes, dev):
    """One scratch buffer per device for the split-K partial sums (written and consumed inside one hsidm_conv2d call, so stream
    order makes sharing it between launches safe).  Captured hipGraphs (ReverseRun, Trainer) bake its ADDRESS in, and the size
    needed is not monotonic in the batch (18.35 MB at 5 latents, 18.9 MB at 6 on the 16x16 level): a buffer that has to grow is
    therefore never freed - the superseded one is parked in _sk_retired, where a live graph may keep writing to it - and sizes
    are rounded up to powers of two (>= 32 MiB), so that at most log2 of them ever exist and their sum stays below twice the
    largest."""
    buf = _sk_ws.get(dev.index)
    if buf is None or buf.numel() < nbytes:
        size = 32 << 20
        while size < nbytes:
            size <<= 1
        if buf is not None:
            _sk_retired.append(buf)
        buf = torch.empty(size, dtype=torch.uint8, device=dev)
        _sk_ws[dev.index] = buf
    return buf


_conv_probe = None
_use_v2 = True          # set False to force the v1 kernel everywhere (A/B measurements)
_fold_ups = True        # set False to run upsample convs with HSIDM_UPS_ADDRESS (A/B measurements)
UPS_FOLDED = 2          # include/hsidm.h HSIDM_UPS_FOLDED


import os as _os
# set False (or HSIDM_NO_FUSED_PROJ=1, read ONCE here) to keep every residual projection a launch of its own (A/B measurements)
_fused_proj = not _os.environ.get("HSIDM_NO_FUSED_PROJ")


def set_fused_proj(flag):
    global _fused_proj
    _fused_proj = bool(flag)


def use_fused_proj():
    """Is the persistent kernel's projection form (conv_v3, PROJ: one-pass 64-cout layers) on offer?  The host switch AND the library's
    own switches (hsidm_debug_query: NO_FUSED_PROJ / NO_V3): asked BEFORE a ResnetBlock computes a GroupNorm table or packs the
    projection layouts for the offer, so that a refused offer costs nothing (A/B runs through the debug switches stay clean)."""
    if not _fused_proj:
        return False
    L = _lib.lib()
    return not any(L.hsidm_debug_query(name) for name in (b"NO_FUSED_PROJ", b"NO_V3"))


def set_fold_ups(flag):
    global _fold_ups
    _fold_ups = bool(flag)


def set_use_v2(flag):
    global _use_v2
    _use_v2 = bool(flag)


def use_v2():
    return _use_v2


def set_conv_probe(records):
    """records: a list to append per-launch {events, flops, shape} dicts to, or None to switch the hook off."""
    global _conv_probe
    _conv_probe = records


def _partials(x, precision):
    """(slab [B, nsplit, C, 2], nsplit) of an NHWC tensor: from its producer's epilogue when present, else one
    streaming pass (tensors that did not come out of hsidm_conv2d)."""
    st = getattr(x, "_hsidm_stats", None)
    if st is not None:
        return st
    return channel_partials(x, precision)


def gn_scale_shift(x0, x1, gamma, beta, groups, precision, eps=1e-5):
    """GroupNorm statistics of cat(x0, x1) folded with (gamma, beta): [B, C, 2] (scale, shift)."""
    B, H, W, C0 = x0.shape
    C1 = 0 if x1 is None else x1.shape[3]
    p0, n0 = _partials(x0, precision)
    p1, n1 = _partials(x1, precision) if x1 is not None else (None, 0)
    # fp32 pairs + two fp16x2 parts + (mean, rstd) per (image, group) (hsidm.h)
    ab = torch.empty(B * (C0 + C1) * 4 + B * groups * 2, dtype=torch.float32, device=x0.device)
    _lib.check(_lib.lib().hsidm_gn_finalize(_lib.ptr(p0), n0, C0, _lib.ptr(p1), n1, C1, B, H * W, groups,
                                            _lib.ptr(gamma), _lib.ptr(beta), float(eps), _lib.ptr(ab),
                                            _lib.stream_ptr()), "gn_finalize")
    return ab


def gn_table(scale_shift):
    """[B, C, 2] fp32 (scale, shift) -> the table layout hsidm_gn_finalize produces and the conv kernels read: the fp32 pairs
    followed by their fp16x2 copies and the fp16x2 copies of log2(e) * (scale, shift) (for callers that bring their own
    normalisation parameters, e.g. tests)."""
    ss = scale_shift.to(torch.float32).contiguous()
    packed = ss.to(torch.float16).contiguous().view(torch.int32).reshape(-1)          # (scale, shift) -> one 32-bit word
    scaled = (ss * 1.44269504).to(torch.float16).contiguous().view(torch.int32).reshape(-1)
    return torch.cat([ss.reshape(-1), packed.view(torch.float32), scaled.view(torch.float32)]).contiguous()


def channel_partials(x, precision, nsplit=None):
    """Per-(image, split, channel) (sum, sumsq) of an NHWC tensor: [B, nsplit, C, 2]."""
    B, H, W, Cc = x.shape
    HW = H * W
    if nsplit is None:
        nsplit = max(1, min(32, HW // 512))
    part = torch.empty((B, nsplit, Cc, 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsidm_gn_partial(_lib.prec_id(precision), _lib.ptr(x), None, Cc, 0, B, HW, nsplit,
                                           _lib.ptr(part), _lib.stream_ptr()), "gn_partial")
    return part, nsplit


def noise_film(B, dim, mlp, wf, bf, *, gamma=None, level_table=None, t_ptr=None, t_emb=None, want_t=False):
    """FiLM table [B, F] (+ optionally the noise embedding [B, dim]).  mlp = (w1, b1, w2, b2) or None."""
    dev = wf.device
    F = wf.shape[0]
    film = torch.empty((B, F), dtype=torch.float32, device=dev)
    t_out = torch.empty((B, dim), dtype=torch.float32, device=dev) if want_t else None
    w1, b1, w2, b2 = mlp if mlp is not None else (None, None, None, None)
    _lib.check(_lib.lib().hsidm_noise_film(_lib.ptr(gamma), _lib.ptr(level_table), _lib.ptr(t_ptr), _lib.ptr(t_emb),
                                           B, dim, _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2),
                                           _lib.ptr(wf), _lib.ptr(bf), F, _lib.ptr(film), _lib.ptr(t_out),
                                           _lib.stream_ptr()), "noise_film")
    return (film, t_out) if want_t else film


def film_affine(x, gamma_beta, precision):
    """(1 + gamma[b, c]) * x + beta[b, c] (FeatureWiseAffine with use_affine_level, reference unet.py:44-47); gamma_beta [B, 2C]."""
    B, H, W, Cc = x.shape
    assert gamma_beta.shape == (B, 2 * Cc) and gamma_beta.is_contiguous()
    out = torch.empty_like(x)
    _lib.check(_lib.lib().hsidm_film_affine(_lib.prec_id(precision), _lib.ptr(x), _lib.ptr(gamma_beta), _lib.ptr(out), B, H * W, Cc,
                                            _lib.stream_ptr()), "film_affine")
    return out


def attention(qkv, precision):
    B, H, W, C3 = qkv.shape
    Cc = C3 // 3
    out = torch.empty((B, H, W, Cc), dtype=qkv.dtype, device=qkv.device)
    _lib.check(_lib.lib().hsidm_attention(_lib.prec_id(precision), _lib.ptr(qkv), _lib.ptr(out), B, H * W, Cc,
                                          _lib.stream_ptr()), "attention")
    return out


def attention_multihead(qkv, n_head, precision):
    """SelfAttention's core for n_head > 1 (reference unet.py:131-141; its UNet only ever builds n_head = 1, the fused kernel
    above): the qkv channels are n_head blocks of (q | k | v) of head_dim each, the scale stays 1/sqrt(C).  Three launches per
    head on the strided batched GEMM + row softmax kernels (hsidm_bgemm, hsidm_softmax_rows)."""
    B, H, W, C3 = qkv.shape
    Cc, N = C3 // 3, H * W
    hd = Cc // n_head
    f32 = {torch.bfloat16: 0, torch.float32: 1, torch.float16: 2}[qkv.dtype]        # hsidm_bgemm's element type code
    es = qkv.element_size()
    out = torch.empty((B, H, W, Cc), dtype=qkv.dtype, device=qkv.device)
    P = torch.empty((B, N, N), dtype=torch.float32, device=qkv.device)
    L = _lib.lib()
    st = _lib.stream_ptr()
    for h in range(n_head):
        q = qkv.data_ptr() + (h * 3 * hd) * es
        k, v = q + hd * es, q + 2 * hd * es
        o = out.data_ptr() + h * hd * es
        _lib.check(L.hsidm_bgemm(q, int(f32), N * C3, C3, 1, k, int(f32), N * C3, 1, C3, P.data_ptr(), 1, N * N, N, N, N, hd, B,
                                 1.0 / Cc ** 0.5, st), "bgemm")
        _lib.check(L.hsidm_softmax_rows(_lib.ptr(P), B * N, N, st), "softmax_rows")
        _lib.check(L.hsidm_bgemm(P.data_ptr(), 1, N * N, N, 1, v, int(f32), N * C3, C3, 1, o, int(f32), N * Cc, Cc, N, hd, N, B, 1.0, st), "bgemm")
    return out


_offset_cache = {}


def _offsets(n_img, img_stride, dev):
    """Device int64 [n_img] = i * img_stride, cached (a host->device copy cannot be graph-captured)."""
    key = (n_img, img_stride, str(dev))
    off = _offset_cache.get(key)
    if off is None:
        off = (torch.arange(n_img, dtype=torch.int64) * img_stride).to(dev)
        _offset_cache[key] = off
    return off


def to_nhwc(x, precision, x1=None, offsets0=None, offsets1=None, c0=None, n_out=None):
    """NCHW fp32 -> NHWC storage type, channels zero-padded to a multiple of 8; optional concat (x, x1).

    offsets0 / c0 / n_out allow channel slicing with a per-output-image element offset (GAE groups).
    """
    x = x.contiguous()
    B, Cx, H, W = x.shape
    HW = H * W
    C0 = Cx if c0 is None else c0
    n_out = B if n_out is None else n_out
    if offsets0 is None:
        offsets0 = _offsets(B, Cx * HW, x.device)
    C1 = 0
    if x1 is not None:
        x1 = x1.contiguous()
        C1 = x1.shape[1]
        if offsets1 is None:
            offsets1 = _offsets(B, C1 * HW, x.device)
    cpad = (C0 + C1 + 7) // 8 * 8
    out = torch.empty((n_out, H, W, cpad), dtype=_lib.act_dtype(precision), device=x.device)
    _lib.check(_lib.lib().hsidm_nchw_to_nhwc(_lib.prec_id(precision), _lib.ptr(x), _lib.ptr(offsets0), C0,
                                             _lib.ptr(x1), _lib.ptr(offsets1), C1, _lib.ptr(out), n_out, HW, cpad,
                                             _lib.stream_ptr()), "nchw_to_nhwc")
    return out


def to_nchw(x, precision):
    B, H, W, Cc = x.shape
    out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsidm_nhwc_to_nchw(_lib.prec_id(precision), _lib.ptr(x), _lib.ptr(out), B, H * W, Cc,
                                             _lib.stream_ptr()), "nhwc_to_nchw")
    return out


def ca_vector(part, nsplit, HW, w1, b1, w2, b2):
    B, _, Cc, _ = part.shape
    R = w1.shape[0]
    ca = torch.empty((B, Cc), dtype=torch.float32, device=part.device)
    _lib.check(_lib.lib().hsidm_ca_vector(_lib.ptr(part), nsplit, B, Cc, HW, R, _lib.ptr(w1), _lib.ptr(b1),
                                          _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(ca), _lib.stream_ptr()), "ca_vector")
    return ca


def ca_apply(r, ca, skip, res_scale, precision, skip2=None):
    B, H, W, Cc = r.shape
    out = torch.empty_like(r)
    _lib.check(_lib.lib().hsidm_ca_apply(_lib.prec_id(precision), _lib.ptr(r), _lib.ptr(ca), _lib.ptr(skip),
                                         _lib.ptr(skip2), float(res_scale), _lib.ptr(out), B, H * W, Cc,
                                         _lib.stream_ptr()), "ca_apply")
    return out


def overlap_average(dec, start, G, n_subs, B, Cc):
    _, _, H, W = dec.shape
    y = torch.empty((B, Cc, H, W), dtype=torch.float32, device=dec.device)
    _lib.check(_lib.lib().hsidm_overlap_average(_lib.ptr(dec), _lib.ptr(start), G, n_subs, B, Cc, H * W,
                                                _lib.ptr(y), _lib.stream_ptr()), "overlap_average")
    return y


def philox_normal(shape, seed, stream_id, device):
    out = torch.empty(shape, dtype=torch.float32, device=device)
    _lib.check(_lib.lib().hsidm_philox_normal(_lib.ptr(out), out.numel(), int(seed), int(stream_id),
                                              _lib.stream_ptr()), "philox_normal")
    return out


def p_sample_update(x, eps, coef, t_ptr, T, *, noise=None, noise_stride=0, seed=0, snap=None, inter=1):
    _lib.check(_lib.lib().hsidm_p_sample_update(_lib.ptr(x), _lib.ptr(eps), _lib.ptr(coef), _lib.ptr(t_ptr), T,
                                                _lib.ptr(noise), int(noise_stride), int(seed), x.numel(),
                                                _lib.ptr(snap), int(inter),
                                                _lib.stream_ptr()), "p_sample_update")


def q_sample(x0, noise, gamma):
    """gamma[b]*x0 + sqrt(1-gamma[b]^2)*noise on fp32 [B,...] tensors (reference diffusion.py:213-220)."""
    out = torch.empty_like(x0)
    B = x0.shape[0]
    _lib.check(_lib.lib().hsidm_q_sample(_lib.ptr(x0), _lib.ptr(noise), _lib.ptr(gamma), _lib.ptr(out), B,
                                         x0.numel() // B, _lib.stream_ptr()), "q_sample")
    return out


def loss_sum(a, b, kind):
    """sum |a-b| (kind "l1") or sum (a-b)^2 ("l2") as a 0-dim fp32 device tensor (reference set_loss, diffusion.py:85-91)."""
    L = _lib.lib()
    ws = torch.empty(L.hsidm_loss_workspace_bytes() // 8, dtype=torch.float64, device=a.device)
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    _lib.check(L.hsidm_loss_sum(_lib.ptr(a), _lib.ptr(b), a.numel(), {"l1": 0, "l2": 1}[kind], _lib.ptr(ws),
                                _lib.ptr(out), _lib.stream_ptr()), "loss_sum")
    return out[0]


def step_advance(t_ptr, wrap_T=0):
    _lib.check(_lib.lib().hsidm_step_advance(_lib.ptr(t_ptr), int(wrap_T), _lib.stream_ptr()), "step_advance")
