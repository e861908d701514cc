"""Training step of the diffusion model on the HIP kernels (BASELINE configs[4]).

Counterpart of the reference's ``DDPM.optimize_parameters`` (model/model.py:49-59):

    optG.zero_grad(); l_pix = netG(data); l_pix = l_pix.sum() / (b*c*h*w); l_pix.backward(); optG.step()

with ``netG(data) = GaussianDiffusion.p_losses`` (diffusion.py:222-250), Adam(lr) over all UNet parameters
(model/model.py:37-41) and, for several GPUs, gradient averaging (the reference: nn.DataParallel's reduce-add,
model/networks.py:113-115; here a bucketed RCCL all-reduce overlapped with the backward pass).

Nothing here is autograd: the forward pass runs the inference kernels with every Block's operand
``a = dropout(silu(GroupNorm(x)))`` materialised (hsidm_gn_act_apply) - it is needed again by the weight gradient - and
the backward pass chains the hand-written adjoints in reverse order:

    conv        -> hsidm_conv_wgrad (weights), hsidm_conv2d with transposed+flipped weights (input), channel sums (bias)
    GN+SiLU(+dropout) -> hsidm_gn_act_bwd            attention core -> hsidm_bgemm x5 + row softmax forward/backward
    stride 2 / nearest x2 -> hsidm_zero_insert2 / hsidm_sum2x2      FiLM + noise MLP -> hsidm_noise_film_bwd

Parameters, gradients and Adam moments live in flat fp32 buffers (the nn.Parameters are views), so the optimiser is one
launch (hsidm_adam_step), the all-reduce works on a few large contiguous buckets, and the packed bf16 (or hi+lo) kernel
weights of ALL convolutions - forward and transposed - are refreshed by one gather launch (hsidm_gather_pack) through index
maps built once from the same layout code inference uses (ops.pack_layouts).
"""
import numpy as np
import torch
import torch.distributed as dist
from torch import nn

from . import ops
from . import train_ops as T
from .precision import resolve_precision
from .sr3_modules.unet import Block, Downsample, ResnetBlocWithAttn, UNet, Upsample


_NOTICE_DONE = False


class Trainer:
    def __init__(self, gd, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, precision=None, dropout_seed=0, bucket_bytes=64 << 20):
        net = gd.denoise_fn
        if not isinstance(net, UNet):
            raise TypeError("hsidm: Trainer needs a GaussianDiffusion over hsi_dmgasr_amd's UNet")
        self.gd, self.net = gd, net
        # what the hand-written backward pass covers is what the reference's UNet builds (unet.py:152,206: one attention head,
        # additive FiLM); the two constructor variants its classes also accept have forward kernels only
        for u in net._res_units():
            if u.res_block.noise_func.use_affine_level:
                raise NotImplementedError("hsidm: the training step has no adjoint for FeatureWiseAffine(use_affine_level=True)")
            if u.with_attn and u.attn.n_head != 1:
                raise NotImplementedError("hsidm: the training step has no adjoint for SelfAttention(n_head=%d); n_head must be 1"
                                          % u.attn.n_head)
        self.precision = resolve_precision(precision if precision is not None else net.precision)
        if self.precision not in ("bf16", "fp32"):
            if precision is not None:
                raise NotImplementedError("hsidm: the training step runs in the bf16 or the fp32 mode (got %r)" % self.precision)
            # A network built in an inference-only mode (the fp16 family, the package default) and no mode asked for: train as the
            # reference trains - in fp32 arithmetic (gradients within 1e-5 of autograd); Trainer(gd, precision="bf16") is the fast form.
            global _NOTICE_DONE
            if not _NOTICE_DONE:
                import sys
                print("hsidm: the network's precision mode %r is an inference mode; this Trainer runs the training step in the fp32 mode "
                      "(about 1.7x the bf16 step's time; pass Trainer(gd, precision=\"bf16\") / gd.trainer(precision=\"bf16\") for the fast form)"
                      % self.precision, file=sys.stderr)
                _NOTICE_DONE = True
            self.precision = "fp32"
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.dropout_seed = int(dropout_seed)
        self.bucket_bytes = int(bucket_bytes)
        self.step_count = 0            # optimiser steps taken (Adam's bias correction)
        self._iter = 0                 # training forward passes started (dropout mask key)
        self.reducer, self._reduced, self._low_cache = None, False, {}
        self._film_bufs, self._b1_idx = {}, None
        self._shape_seen, self._tracking = None, False
        self._defer_obj, self._defer = None, None
        self._g, self._g_active = None, False          # captured step (optimize_parameters): state, capturing / replaying
        self._eager_shape = None                       # batch shape of the LAST eager step (what every workspace is sized for)
        self.dev = next(net.parameters()).device
        self._check_device()
        self._flatten()
        self._build_pack_maps()
        self.repack()

    def _check_device(self):
        if self.dev.type != "cuda":
            raise RuntimeError("hsidm: the training step only runs on a ROCm device (no CPU fallback)")

    # ------------------------------------------------------------------------------------------------ flat buffers
    def _flatten(self):
        """Move every parameter into one flat fp32 buffer (the nn.Parameters become views): FiLM projection weights first
        (contiguous = the [F, dim] matrix hsidm_noise_film reads), then their biases, then everything else in module order."""
        net = self.net
        units = net._res_units()
        lins = [u.res_block.noise_func.noise_func[0] for u in units]
        first = [l.weight for l in lins] + [l.bias for l in lins]
        seen = {id(p) for p in first}
        rest = [p for p in net.parameters() if id(p) not in seen]
        order = first + rest
        # FiLM weights, then FiLM biases, back to back (each block a multiple of 4 floats: F is a multiple of the group count);
        # every other parameter starts on a 16-byte boundary
        offs, total = {}, 0
        for p in order:
            if id(p) not in seen:
                total = (total + 3) // 4 * 4
            offs[id(p)] = total
            total += p.numel()
        total = (total + 3) // 4 * 4
        self.flat = torch.zeros(total, dtype=torch.float32, device=self.dev)
        self.grad = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self._off = {}
        self._gview = {}
        self._params = order
        with torch.no_grad():
            for p in order:
                o, n = offs[id(p)], p.numel()
                self._off[id(p)] = o
                if p.dim() == 4 and p.shape[2] * p.shape[3] > 1:
                    # 3x3 weights live in channels-last memory order [cout][ky][kx][cin] (same tensor, other strides): the packed
                    # layouts take 8 consecutive input channels of one tap per vector, so the re-pack gathers contiguous runs
                    # instead of elements 36 bytes apart (805 -> ~430 us per step, tools/gather_bench.py)
                    co, ci, kh, kw = p.shape
                    self.flat[o:o + n].view(co, kh, kw, ci).copy_(p.detach().float().permute(0, 2, 3, 1))
                    p.data = self.flat[o:o + n].view(co, kh, kw, ci).permute(0, 3, 1, 2)
                    self._gview[id(p)] = self.grad[o:o + n].view(co, kh, kw, ci).permute(0, 3, 1, 2)
                    continue
                self.flat[o:o + n].copy_(p.detach().reshape(-1).float())
                p.data = self.flat[o:o + n].view(p.shape)
                self._gview[id(p)] = self.grad[o:o + n].view(p.shape)
        F = sum(l.weight.shape[0] for l in lins)
        dim = lins[0].weight.shape[1]
        o_w, o_b = offs[id(lins[0].weight)], offs[id(lins[0].bias)]
        assert offs[id(lins[-1].weight)] + lins[-1].weight.numel() == o_w + F * dim, "FiLM weights must be contiguous"
        assert offs[id(lins[-1].bias)] + lins[-1].bias.numel() == o_b + F, "FiLM biases must be contiguous"
        self._wf, self._bf = self.flat[o_w:o_w + F * dim].view(F, dim), self.flat[o_b:o_b + F]
        self._dwf, self._dbf = self.grad[o_w:o_w + F * dim].view(F, dim), self.grad[o_b:o_b + F]
        self._film_offs, o = [], 0
        for l in lins:
            self._film_offs.append((o, o + l.weight.shape[0]))
            o += l.weight.shape[0]
        self._emb_dim = dim
        self._head_floats = o_b + F           # everything below this offset (FiLM) gets its gradient last

    def G(self, p):
        """fp32 gradient view of a parameter (a slice of the flat gradient buffer)."""
        return self._gview[id(p)]

    # ------------------------------------------------------------------------------------------------ packed weights
    def _convs(self):
        """(module, role flags) of every convolution: (conv, out_nchw, fold_dn, needs_dgrad)."""
        net, out = self.net, []
        out.append((net.downs[0], False, False, False))
        for layer in list(net.downs)[1:] + list(net.mid) + list(net.ups):
            if isinstance(layer, ResnetBlocWithAttn):
                rb = layer.res_block
                out.append((rb.block1.block[3], False, False, True))
                out.append((rb.block2.block[3], False, False, True))
                if isinstance(rb.res_conv, nn.Conv2d):
                    out.append((rb.res_conv, False, False, True))
                if layer.with_attn:
                    out.append((layer.attn.qkv, False, False, True))
                    out.append((layer.attn.out, False, False, True))
            elif isinstance(layer, Downsample):
                out.append((layer.conv, False, True, True))
            elif isinstance(layer, Upsample):
                out.append((layer.conv, False, False, True))
        out.append((net.final_conv.block[3], True, False, True))
        return out

    def _build_pack_maps(self):
        """Gather maps (packed element -> index into the flat parameter buffer) of every layout of every convolution, built by
        running the inference packing code on index-valued tensors.  Which of them are materialised is decided by _assemble."""
        prec = self.precision
        self._seg = {}                       # (conv id, role, layout name) -> (idx int32 [n padded to 8], n, shape)
        self._pk_meta = {}                   # (conv id, role) -> (meta, out_nchw)
        for conv, out_nchw, fold_dn, need_dg in self._convs():
            w = conv.weight
            base = self._off[id(w)]
            iw = torch.arange(w.numel(), dtype=torch.float64, device=self.dev) + (base + 1)
            if w.shape[2] * w.shape[3] > 1:              # memory order of a 3x3 weight: [cout][ky][kx][cin] (_flatten)
                iw = iw.view(w.shape[0], w.shape[2], w.shape[3], w.shape[1]).permute(0, 3, 1, 2)
            else:
                iw = iw.view(w.shape)
            roles = [("fwd", iw, out_nchw, fold_dn)]
            if need_dg:
                roles.append(("dgrad", iw.transpose(0, 1).flip(2, 3).contiguous(), False, False))
            for role, iwt, nchw, fdn in roles:
                lay, meta = ops.pack_layouts(iwt, prec, out_nchw=nchw, fold_dn=fdn)
                self._pk_meta[(id(conv), role)] = (meta, nchw)
                for name in ("w", "w_v2", "w_dn4"):
                    if lay[name] is None:
                        continue
                    idx = (lay[name].round().to(torch.int64) - 1).to(torch.int32).reshape(-1)
                    n = idx.numel()
                    if n % 8:
                        idx = torch.cat([idx, torch.full((8 - n % 8,), -1, dtype=torch.int32, device=idx.device)])
                    self._seg[(id(conv), role, name)] = (idx, n, tuple(lay[name].shape))
        self._assemble(set(self._seg))

    def _assemble(self, active):
        """One index map + one packed buffer over the `active` layouts, and PackedConv views over it."""
        prec = self.precision
        keys = [k for k in self._seg if k in active]
        self._active = set(keys)
        self._pack_idx = torch.cat([self._seg[k][0] for k in keys])
        total = self._pack_idx.numel()
        self._pack_hi = torch.zeros(total, dtype=torch.bfloat16, device=self.dev)
        self._pack_lo = torch.zeros(total, dtype=torch.bfloat16, device=self.dev) if prec == "fp32" else None
        views, o = {}, 0
        for k in keys:
            idx, n, shape = self._seg[k]
            views[k] = (self._pack_hi[o:o + n].view(shape), None if self._pack_lo is None else self._pack_lo[o:o + n].view(shape))
            o += idx.numel()
        self._pk = {}
        for (cid, role), (meta, nchw) in self._pk_meta.items():
            w = views.get((cid, role, "w"))
            v2 = views.get((cid, role, "w_v2"))
            dn4 = views.get((cid, role, "w_dn4"))
            any_hi = next(v[0] for v in (w, v2, dn4) if v is not None)
            # hsidm_conv2d wants a non-null w_hi even when the dispatch reads another layout: a pruned one aliases a live buffer
            pk = ops.PackedConv.from_buffers(meta, prec, nchw, w[0] if w is not None else any_hi, None if w is None else w[1],
                                             None if v2 is None else v2[0], None if dn4 is None else dn4[0], None,
                                             w_v2_lo=None if v2 is None else v2[1], w_dn4_lo=None if dn4 is None else dn4[1])
            pk._track = None
            self._pk[(cid, role)] = pk
        for conv, _, _, _ in self._convs():          # forward convolutions read their bias straight from the master copy
            self._pk[(id(conv), "fwd")].bias = None if conv.bias is None else conv.bias.detach()
        self._g, self._eager_shape = None, None      # a captured step holds pointers into the old buffers

    def _track_layouts(self, on):
        for pk in self._pk.values():
            pk._track = set() if on else None

    def _prune_layouts(self):
        """After a tracked step: keep only the layouts the kernel dispatch read for this batch shape (in the bf16 mode most
        convolutions read the register-streaming order only; the LDS-tiled order is packed for the rest)."""
        keep = set()
        for (cid, role), pk in self._pk.items():
            used = pk._track or set()
            names = [n for n in ("w", "w_v2", "w_dn4") if (cid, role, n) in self._seg]
            live = [n for n in names if n in used] or names        # never drop everything
            keep.update((cid, role, n) for n in live)
        self._track_layouts(False)
        if keep != self._active:
            self._assemble(keep)
            self.repack()

    def repack(self):
        """Kernel-order bf16 (hi [+ lo]) weights of every convolution, forward and transposed, from the fp32 master copy."""
        T.gather_pack(self.flat, self._pack_idx, self._pack_hi, self._pack_lo)
        self._packed_at = self._versions()

    def pk(self, conv):
        return self._pk[(id(conv), "fwd")]

    def dpk(self, conv):
        return self._pk[(id(conv), "dgrad")]

    # ------------------------------------------------------------------------------------------------ forward (training mode)
    def _drop_key_value(self):
        # one key per forward pass AND per rank: data-parallel replicas draw independent dropout masks, as the reference's do
        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        return (self.dropout_seed + 0x9E3779B97F4A7C15 * self._iter + 0xD1B54A32D192ED03 * rank) & 0xFFFFFFFFFFFFFFFF

    def _drop_key(self):
        """Philox key of the current forward pass's dropout masks (a new one per forward_loss call; backward re-derives it):
        a Python int, or - inside a captured step - the device word the host refreshes before every replay."""
        return self._g["key"] if self._g_active else self._drop_key_value()

    def _block_fwd(self, blk, x0, x1, lid, film=None, res=None, out_nchw=False, p_drop=0.0):
        gn, p = blk.block[0], self.precision
        ab = ops.gn_scale_shift(x0, x1, gn.weight, gn.bias, gn.num_groups, p, gn.eps)
        a = T.gn_act_apply(x0, x1, ab, True, p, p_drop, self._drop_key(), lid)
        y = ops.conv2d(a, self.pk(blk.block[3]), film=film, res=res, stats=not out_nchw)
        return y, (x0, x1, ab, a, lid, p_drop)

    def _block_bwd(self, blk, ctx, dy, add=None, bias=True, db_images=None):
        """dy: gradient at the block's conv output.  bias: the conv's bias gradient (the channel sums of dy) comes out of the weight
        gradient launch; False when the caller takes the per-image sums (db_images, or channel_sums).  -> (dx0, dx1)"""
        x0, x1, ab, a, lid, p_drop = ctx
        gn, conv, p = blk.block[0], blk.block[3], self.precision
        T.conv_wgrad(a, None, dy, self.G(conv.weight), p, deferred=self._defer, db=self.G(conv.bias) if bias else None,
                     db_images=db_images)
        da = ops.conv2d(dy, self.dpk(conv))
        return T.gn_act_bwd(da, x0, x1, ab, gn.weight, gn.num_groups, True, p, self.G(gn.weight), self.G(gn.bias), p_drop,
                            self._drop_key(), lid, add=add)

    def _res_fwd(self, rb, x0, x1, film, lid):
        h1, c1 = self._block_fwd(rb.block1, x0, x1, 2 * lid, film=film)
        proj = isinstance(rb.res_conv, nn.Conv2d)
        r = ops.conv2d(x0, self.pk(rb.res_conv), x1=x1) if proj else x0
        p_drop = rb.block2._dropout if self.net.training else 0.0
        out, c2 = self._block_fwd(rb.block2, h1, None, 2 * lid + 1, res=r, p_drop=p_drop)
        return out, (c1, c2)

    def _res_bwd(self, rb, ctx, d_out, skip_add=None):
        """-> (dx0, dx1, dfilm [B, Cout])"""
        c1, c2 = ctx
        x0, x1 = c1[0], c1[1]
        p = self.precision
        conv1, conv2 = rb.block1.block[3], rb.block2.block[3]
        proj = isinstance(rb.res_conv, nn.Conv2d)
        dh1, _ = self._block_bwd(rb.block2, c2, d_out)
        # FiLM's gradient is the per-image channel sum of dh1, block1's bias gradient their total.  With the deferred reductions
        # (one GPU) both come out of block1's weight-gradient launch: per-image sums into a buffer of this block's own, the total
        # copied from the FiLM projection's bias gradient - the same number - at the end of the pass; with a gradient reducer the
        # bias gradient must be final when the layer is, so the sums are taken here by the streaming reduction.
        film_buf = None
        if self._defer is not None:
            key = (id(rb), dh1.shape[0])
            film_buf = self._film_bufs.get(key)
            if film_buf is None:
                film_buf = torch.zeros((dh1.shape[0], T.bias_image_cols(dh1.shape[3])), dtype=torch.float32, device=self.dev)
                self._film_bufs[key] = film_buf
            dfilm = film_buf[:, :conv1.bias.shape[0]]
        else:
            dfilm = T.channel_sums(dh1, p, out_c=self.G(conv1.bias), want_bc=True)
        if proj:                                                      # (both biases see the same gradient sum)
            T.conv_wgrad(x0, x1, d_out, self.G(rb.res_conv.weight), p, deferred=self._defer, db=self.G(rb.res_conv.bias))
            add = ops.conv2d(d_out, self.dpk(rb.res_conv), res=skip_add)
        else:
            add = d_out if skip_add is None else T.add(d_out, skip_add, p)
        dx0, dx1 = self._block_bwd(rb.block1, c1, dh1, add=add, bias=False, db_images=film_buf)
        return dx0, dx1, dfilm

    def _attn_fwd(self, at, x):
        p = self.precision
        ab = ops.gn_scale_shift(x, None, at.norm.weight, at.norm.bias, at.norm.num_groups, p, at.norm.eps)
        n = T.gn_act_apply(x, None, ab, False, p)
        qkv = ops.conv2d(n, self.pk(at.qkv))
        o = ops.attention(qkv, p)
        y = ops.conv2d(o, self.pk(at.out), res=x, stats=True)
        return y, (x, ab, n, qkv, o)

    def _attn_bwd(self, at, ctx, dy):
        x, ab, n, qkv, o = ctx
        p = self.precision
        T.conv_wgrad(o, None, dy, self.G(at.out.weight), p, deferred=self._defer, db=self.G(at.out.bias))
        do = ops.conv2d(dy, self.dpk(at.out))
        dqkv = T.attention_bwd(qkv, do, p)
        T.conv_wgrad(n, None, dqkv, self.G(at.qkv.weight), p, deferred=self._defer)
        dn = ops.conv2d(dqkv, self.dpk(at.qkv))
        dx, _ = T.gn_act_bwd(dn, x, None, ab, at.norm.weight, at.norm.num_groups, False, p, self.G(at.norm.weight),
                             self.G(at.norm.bias), add=dy)
        return dx

    def _unit_fwd(self, unit, x0, x1, film, lid):
        x, c = self._res_fwd(unit.res_block, x0, x1, film, lid)
        if unit.with_attn:
            x, ca = self._attn_fwd(unit.attn, x)
            return x, (c, ca)
        return x, (c, None)

    def _unit_bwd(self, unit, ctx, d_out, skip_add=None):
        c, ca = ctx
        if ca is not None:
            d_out = self._attn_bwd(unit.attn, ca, d_out)
        return self._res_bwd(unit.res_block, c, d_out, skip_add)

    def forward(self, cond, x, gamma):
        """eps = UNet(cat(cond, x), gamma) in training mode, recording what the backward pass needs.  NCHW fp32 in and out."""
        net, p = self.net, self.precision
        B = x.shape[0]
        mlp = net._mlp()
        film, t_emb = ops.noise_film(B, self._emb_dim, mlp, self._wf, self._bf, gamma=gamma, want_t=True)
        stem_in = ops.to_nhwc(cond, p, x1=x) if cond is not None else ops.to_nhwc(x, p)
        tape, skips, k, lid = [], [], 0, 0
        h = None
        for layer in net.downs:
            if isinstance(layer, ResnetBlocWithAttn):
                lo, hi = self._film_offs[k]
                h, c = self._unit_fwd(layer, h, None, film[:, lo:hi], lid)
                tape.append(("unit", layer, c, k))
                k += 1
                lid += 1
            elif isinstance(layer, Downsample):
                xin = h
                h = ops.conv2d(xin, self.pk(layer.conv), stride=2, stats=True)
                tape.append(("down", layer, xin, None))
            else:
                h = ops.conv2d(stem_in, self.pk(layer), stats=True)
                tape.append(("stem", layer, stem_in, None))
            skips.append(h)
        for layer in net.mid:
            lo, hi = self._film_offs[k]
            h, c = self._unit_fwd(layer, h, None, film[:, lo:hi], lid)
            tape.append(("unit", layer, c, k))
            k += 1
            lid += 1
        for layer in net.ups:
            if isinstance(layer, ResnetBlocWithAttn):
                lo, hi = self._film_offs[k]
                h, c = self._unit_fwd(layer, h, skips.pop(), film[:, lo:hi], lid)
                tape.append(("unit_cat", layer, c, k))
                k += 1
                lid += 1
            else:
                xin = h
                h = ops.conv2d(xin, self.pk(layer.conv), ups=True, stats=True)
                tape.append(("up", layer, xin, None))
        eps, cf = self._block_fwd(net.final_conv, h, None, 2 * lid, out_nchw=True)
        self._tape = dict(tape=tape, final=cf, gamma=gamma, t_emb=t_emb, n_units=k)
        return eps

    # ------------------------------------------------------------------------------------------------ backward
    def backward(self, d_eps):
        """d_eps: gradient at the network output as NHWC [B, H, W, 8] (train_ops.loss_grad).  Fills the flat gradient buffer."""
        net, p = self.net, self.precision
        tp = self._tape
        fconv = net.final_conv.block[3]
        red = self._reducer()
        # one GPU: the 94 split-K reductions of the weight gradients are deferred to ONE launch at the end (each is 10-20 us of
        # latency on its own); several GPUs: every layer reduces at once, so that its bucket can go on the wire
        if red is None and self._defer_obj is None and self.dev.type == "cuda":
            self._defer_obj = T.DeferredReductions(self.dev)
        self._defer = self._defer_obj if red is None else None
        d, _ = self._block_bwd(net.final_conv, tp["final"], d_eps)
        if red is not None:
            red.ready(self._low(net.final_conv))
        dfilms = [None] * tp["n_units"]
        skip_grads = []                   # gradients of the encoder outputs, in the order the decoder consumed them
        tape = tp["tape"]
        n_enc = len(net.downs)
        for i in reversed(range(len(tape))):
            kind, layer, ctx, k = tape[i]
            # tape[1 .. n_enc] (the encoder layers after the stem, and mid[0]) consumed an encoder output, which the decoder
            # consumed too (skip connection): the decoder's gradient for it joins this layer's input gradient.  The decoder
            # popped the outputs last-first, so its gradients were appended first-last: pop() pairs them up again.
            consumes_skip = 1 <= i <= n_enc
            skip_add = skip_grads.pop() if consumes_skip else None
            if kind == "unit_cat":
                d, dskip, dfilms[k] = self._unit_bwd(layer, ctx, d)
                skip_grads.append(dskip)
            elif kind == "unit":
                d, _, dfilms[k] = self._unit_bwd(layer, ctx, d, skip_add)
            elif kind == "up":
                conv = layer.conv
                T.conv_wgrad(ctx, None, d, self.G(conv.weight), p, ups=True, deferred=self._defer, db=self.G(conv.bias))
                d = T.sum2x2(ops.conv2d(d, self.dpk(conv)), p)
            elif kind == "down":
                conv = layer.conv
                T.conv_wgrad(ctx, None, d, self.G(conv.weight), p, stride=2, deferred=self._defer, db=self.G(conv.bias))
                z = T.zero_insert2(d, ctx.shape[1], ctx.shape[2], p)
                d = ops.conv2d(z, self.dpk(conv), res=skip_add)
            else:                          # stem: parameters only
                T.conv_wgrad(ctx, None, d, self.G(layer.weight), p, deferred=self._defer, db=self.G(layer.bias))
            # the flat gradient buffer follows the module order and the backward pass fills it from the end: every bucket
            # above this layer's lowest offset is final and can go on the wire while the earlier layers are still computing
            # (FiLM projections - first in the buffer - and the noise MLP are written after the loop)
            if red is not None and kind != "stem":
                red.ready(self._low(layer))
        assert not skip_grads
        mlp = net._mlp()
        l1, l2 = net.noise_level_mlp[1], net.noise_level_mlp[3]
        if self._defer is not None:
            self._defer.reduce()                            # every weight / bias gradient, and the per-image sums FiLM needs
        T.noise_film_bwd(tp["gamma"], tp["t_emb"], torch.cat(dfilms, dim=1).contiguous(), mlp, self._wf,
                         (self.G(l1.weight), self.G(l1.bias), self.G(l2.weight), self.G(l2.bias), self._dwf, self._dbf))
        if self._defer is not None:                         # block1's bias gradients = the FiLM projections' bias gradients
            self.grad.index_copy_(0, self._conv1_bias_index(), self._dbf.clone())
        self._tape = None
        if red is not None:
            red.finish()
            self._reduced = True

    def _conv1_bias_index(self):
        """Flat offsets of every ResnetBlock's block1 conv bias, in the order of the FiLM bias vector (= the order of the blocks)."""
        if self._b1_idx is None:
            idx = []
            for u in self.net._res_units():
                b = u.res_block.block1.block[3].bias
                o = self._off[id(b)]
                idx.append(torch.arange(o, o + b.numel(), dtype=torch.int64))
            self._b1_idx = torch.cat(idx).to(self.dev)
            assert self._b1_idx.numel() == self._dbf.numel()
        return self._b1_idx

    def _low(self, module):
        """Lowest flat offset of a module's parameters: after its backward everything from there up is final."""
        key = id(module)
        v = self._low_cache.get(key)
        if v is None:
            # (the FiLM projections of a ResnetBlock live in the head of the buffer and are written last: not part of the layer's range)
            v = min(self._off[id(p)] for p in module.parameters() if self._off[id(p)] >= self._head_floats)
            self._low_cache[key] = v
        return v

    def _reducer(self):
        if self.reducer is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from . import parallel
            self.reducer = parallel.GradReducer(self.grad, self.bucket_bytes)
        return self.reducer

    # ------------------------------------------------------------------------------------------------ the step
    def _versions(self):
        return sum(p._version for p in self._params)

    @torch.no_grad()
    def forward_loss(self, x_in, noise=None, t=None, gamma=None):
        """p_losses (diffusion.py:222-250) in training mode: the SUM-reduced loss as a 0-dim device tensor and the state
        backward_loss needs.  t / gamma are drawn from numpy's global generator exactly as the reference does unless given."""
        gd = self.gd
        self._iter += 1
        x_start = x_in["HR"].contiguous()
        if not self._g_active:
            self._eager_shape = None                   # (set again by _graphed_step once this eager step is complete)
        self._note_shape(tuple(x_start.shape))
        if self._versions() != self._packed_at:        # someone else (a torch optimiser, load_state_dict) changed the weights
            self.repack()
        b, c, h, w = x_start.shape
        if gamma is None:
            if t is None:
                t = np.random.randint(1, gd.num_timesteps + 1)
            gamma = torch.FloatTensor(np.random.uniform(gd.sqrt_alphas_cumprod_prev[t - 1], gd.sqrt_alphas_cumprod_prev[t], size=b))
        gamma = gamma.to(x_start.device, torch.float32).reshape(b).contiguous()
        noise = torch.randn_like(x_start) if noise is None else noise.contiguous()
        x_noisy = gd.q_sample(x_start, gamma, noise)
        cond = x_in["SR"].contiguous() if gd.conditional else None
        eps = self.forward(cond, x_noisy, gamma)
        return ops.loss_sum(noise, eps, gd.loss_type), (noise, eps, b * c * h * w)

    def _note_shape(self, shape):
        """Layout pruning is per batch shape (the dispatch depends on it): a new shape restores every layout and tracks again."""
        if shape != self._shape_seen:
            self._shape_seen = shape
            if self._active != set(self._seg):
                self._assemble(set(self._seg))
                self.repack()
            self._track_layouts(True)
            self._tracking = True

    @torch.no_grad()
    def backward_loss(self, state, scale):
        """Gradient of scale * (sum-reduced loss) into the flat gradient buffer."""
        noise, eps, _ = state
        self.backward(T.loss_grad(noise, eps, self.gd.loss_type, scale, self.precision))
        if self._tracking and not self._g_active:
            self._tracking = False
            self._prune_layouts()

    @torch.no_grad()
    def loss_and_grads(self, x_in, noise=None, t=None, gamma=None):
        """l_pix = sum-loss / (b*c*h*w) (model/model.py:53-54) as a 0-dim device tensor; its gradient is left in the flat gradient
        buffer (``Trainer.grad`` / ``p.grad``)."""
        loss, state = self.forward_loss(x_in, noise, t, gamma)
        scale = 1.0 / float(state[2])
        self.backward_loss(state, scale)
        for p in self._params:              # (not before: autograd would ACCUMULATE into a pre-set .grad on the differentiable path)
            if p.grad is not self._gview[id(p)]:
                p.grad = self._gview[id(p)]
        return loss * scale

    @torch.no_grad()
    def optimizer_step(self):
        """Adam over the flat buffers (after averaging the gradients over the ranks of the default process group), then
        one gather launch re-packs the kernel weights."""
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        if world > 1 and not self._reduced:
            from . import parallel
            parallel.allreduce_grads_(self.grad, self.bucket_bytes)
        self._reduced = False
        self.step_count += 1
        T.adam_step(self.flat, self.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.step_count, 1.0 / world)
        # the kernel wrote through raw pointers: tell torch (and the inference path's packed-weight caches, which key on it)
        torch.autograd.graph.increment_version(self._params)
        self.repack()

    # ------------------------------------------------------------------------------------------------ checkpoint / resume
    def _adam_params(self):
        """The parameters in the order ``torch.optim.Adam(netG.parameters())`` numbers them (model/model.py:37-41)."""
        return list(self.net.parameters())

    def _strided_like(self, p, flat):
        """View of a flat buffer laid out like parameter p's slot (3x3 weights are channels-last in memory, see _flatten)."""
        o, n = self._off[id(p)], p.numel()
        if p.dim() == 4 and p.shape[2] * p.shape[3] > 1:
            co, ci, kh, kw = p.shape
            return flat[o:o + n].view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return flat[o:o + n].view(p.shape)

    @torch.no_grad()
    def state_dict(self):
        """The optimiser state in the format of ``torch.optim.Adam.state_dict()`` over ``netG.parameters()`` - what the reference
        stores under 'optimizer' in ``*_opt.pth`` (model/model.py:140-143) - so either side can resume the other's run."""
        state = {}
        if self.step_count > 0:
            for i, p in enumerate(self._adam_params()):
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self._strided_like(p, self.m).detach().clone().contiguous(),
                            "exp_avg_sq": self._strided_like(p, self.v).detach().clone().contiguous()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self._adam_params())))}
        return {"state": state, "param_groups": [group], "hsidm": {"iter": self._iter, "dropout_seed": self.dropout_seed}}

    @torch.no_grad()
    def load_state_dict(self, sd):
        """Inverse of state_dict(); also accepts the state_dict of a torch.optim.Adam built over the same parameters."""
        params = self._adam_params()
        group = sd["param_groups"][0]
        if len(group["params"]) != len(params):
            raise ValueError("hsidm: optimizer state has %d parameters, the network %d" % (len(group["params"]), len(params)))
        if group.get("weight_decay", 0) or group.get("amsgrad", False) or group.get("maximize", False):
            raise NotImplementedError("hsidm: Adam with weight_decay / amsgrad / maximize is not what model/model.py:37-41 builds")
        self.lr, self.betas, self.eps = float(group["lr"]), (float(group["betas"][0]), float(group["betas"][1])), float(group["eps"])
        self.m.zero_(); self.v.zero_()
        steps = set()
        for i, p in enumerate(params):
            st = sd["state"].get(group["params"][i])
            if st is None:
                continue
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("hsidm: optimizer state %d has shape %s, the parameter %s" % (i, tuple(st["exp_avg"].shape), tuple(p.shape)))
            self._strided_like(p, self.m).copy_(st["exp_avg"].to(self.dev, torch.float32))
            self._strided_like(p, self.v).copy_(st["exp_avg_sq"].to(self.dev, torch.float32))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("hsidm: per-parameter step counts differ (%s); the fused Adam keeps one" % sorted(steps))
        self.step_count = steps.pop() if steps else 0
        extra = sd.get("hsidm", {})
        self._iter = int(extra.get("iter", self.step_count))
        self.dropout_seed = int(extra.get("dropout_seed", self.dropout_seed))
        self._g = None          # REQUIRED: a captured step has Adam's betas / eps baked in as kernel scalars - the next call re-captures

    def save_network(self, prefix, epoch, iter_step):
        """DDPM.save_network (model/model.py:125-145): ``<prefix>_gen.pth`` = the GaussianDiffusion's state_dict on the host,
        ``<prefix>_opt.pth`` = {'epoch', 'iter', 'scheduler': None, 'optimizer': Adam state}.  Returns the two paths."""
        gen_path, opt_path = "%s_gen.pth" % prefix, "%s_opt.pth" % prefix
        torch.save({k: v.detach().cpu().contiguous() for k, v in self.gd.state_dict().items()}, gen_path)
        torch.save({"epoch": epoch, "iter": iter_step, "scheduler": None, "optimizer": self.state_dict()}, opt_path)
        return gen_path, opt_path

    def load_network(self, prefix, drop_stem_and_final=True, load_optimizer=False, map_location="cpu"):
        """DDPM.load_network (model/model.py:177-202).  As shipped the reference drops the stem weight and the final convolution's
        weight and bias (so that a checkpoint trained for another channel count seeds this one, :187-191), loads non-strictly and
        does NOT restore the optimiser (its lines are commented out, :198-202): the defaults here.  load_optimizer=True is the
        resume the commented code describes.  Returns (epoch, iter) of the optimiser file, or (0, 0)."""
        ckpt = torch.load("%s_gen.pth" % prefix, map_location=map_location)
        if drop_stem_and_final:
            drop = ("denoise_fn.downs.0.weight", "denoise_fn.final_conv.block.3.weight", "denoise_fn.final_conv.block.3.bias")
            ckpt = {k: v for k, v in ckpt.items() if k not in drop}
        with torch.no_grad():
            self.gd.load_state_dict(ckpt, strict=False)         # copies INTO the flat views (the parameters stay views)
        self.repack()
        self._g = None
        if not load_optimizer:
            return 0, 0
        opt = torch.load("%s_opt.pth" % prefix, map_location=map_location, weights_only=False)
        self.load_state_dict(opt["optimizer"])
        return opt.get("epoch", 0), opt.get("iter", 0)

    def optimize_parameters(self, data, use_graph=True, **kw):
        """model/model.py:49-59.  -> l_pix (0-dim device tensor; the reference logs .item()).

        use_graph (and no injected noise / t / gamma, one rank): from the SECOND call at a batch shape on (the first one runs eagerly and
        sizes every workspace; the second captures and replays) the whole step - forward, backward, Adam, re-pack: ~850 launches at
        B = 4 - is ONE hipGraph replay.  What changes per iteration lives in device memory the host
        refreshes before the replay: the batch, the drawn noise levels, the dropout key and Adam's bias corrections; the noise
        comes from torch's graph-safe generator."""
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        if kw or not use_graph or world > 1 or self.dev.type != "cuda":
            loss = self.loss_and_grads(data, **kw)
            self.optimizer_step()
            return loss
        return self._graphed_step(data)

    def _host_draws(self, b):
        """(t, gamma) exactly as diffusion.py:226-234 draws them."""
        gd = self.gd
        t = np.random.randint(1, gd.num_timesteps + 1)
        return np.random.uniform(gd.sqrt_alphas_cumprod_prev[t - 1], gd.sqrt_alphas_cumprod_prev[t], size=b).astype(np.float32)

    @torch.no_grad()
    def _graphed_step(self, data):
        hr, sr = data["HR"], data.get("SR")
        b = hr.shape[0]
        shape_key = (tuple(hr.shape), None if sr is None else tuple(sr.shape))
        if self._g is not None and self._g["shape"] != shape_key:
            self._g = None                                      # another batch shape: capture again
        if self._g is None and self._eager_shape != shape_key:
            # A capture must directly follow an EAGER step at the same shape: that step sizes every cache the captured kernels
            # take pointers to (split-K and deferred-reduction workspaces, FiLM buffers, offset tables, LDS caps) and prunes the
            # packed layouts for this shape.  The reference's DataLoader has no drop_last: a partial last batch drops the graph,
            # runs eagerly, and the next full batch runs eagerly once more before it is captured again.
            loss = self.loss_and_grads(data)
            self.optimizer_step()
            self._eager_shape = shape_key
            return loss
        if self._g is None:
            g = dict(shape=shape_key, hr=hr.clone(), sr=None if sr is None else sr.clone(),
                     gamma=torch.zeros(b, dtype=torch.float32, device=self.dev), key=torch.zeros(1, dtype=torch.int64, device=self.dev),
                     coef=torch.zeros(2, dtype=torch.float32, device=self.dev),
                     h_gamma=torch.zeros(b, dtype=torch.float32).pin_memory(), h_key=torch.zeros(1, dtype=torch.int64).pin_memory(),
                     h_coef=torch.zeros(2, dtype=torch.float32).pin_memory(), graph=None, loss=None)
            self._g = g
        g = self._g
        # ---- host side of the iteration: draws, counters, staging copies (stream-ordered ahead of the replay)
        if g.get("ev") is not None:
            g["ev"].synchronize()              # the previous call's staging copies have left the pinned buffers
        self._iter += 1
        self.step_count += 1
        g["h_gamma"].copy_(torch.from_numpy(self._host_draws(b)))
        key = self._drop_key_value()
        g["h_key"][0] = key - (1 << 64) if key >= (1 << 63) else key
        c0, c1 = T.adam_coefs(self.lr, self.betas[0], self.betas[1], self.step_count)
        g["h_coef"][0], g["h_coef"][1] = c0, c1
        g["gamma"].copy_(g["h_gamma"], non_blocking=True)
        g["key"].copy_(g["h_key"], non_blocking=True)
        g["coef"].copy_(g["h_coef"], non_blocking=True)
        g["ev"] = torch.cuda.Event()
        g["ev"].record()
        if g["hr"].data_ptr() != hr.data_ptr():
            g["hr"].copy_(hr)
        if sr is not None and g["sr"].data_ptr() != sr.data_ptr():
            g["sr"].copy_(sr)
        if g["graph"] is None:
            if self._versions() != self._packed_at:
                self.repack()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            self._g_active = True
            try:
                with torch.cuda.graph(graph):
                    g["loss"] = self._captured_body(g)
            finally:
                self._g_active = False
            g["graph"] = graph
        elif self._versions() != self._packed_at:               # load_state_dict / a broadcast / an init wrote the master copy
            self.repack()                                       # between two replays: the packed kernel weights follow
        g["graph"].replay()
        torch.autograd.graph.increment_version(self._params)    # the replay rewrote the parameters through raw pointers
        self._packed_at = self._versions()
        return g["loss"].clone()                                # (the graph's own tensor is overwritten by the next replay)

    def _captured_body(self, g):
        """One training step with every per-iteration quantity read from device memory (recorded once, replayed)."""
        gd = self.gd
        x_start = g["hr"]
        b, c, h, w = x_start.shape
        noise = torch.randn_like(x_start)
        g["noise"] = noise                                       # (kept for inspection: tests re-derive the step from it)
        x_noisy = gd.q_sample(x_start, g["gamma"], noise)
        eps = self.forward(g["sr"] if gd.conditional else None, x_noisy, g["gamma"])
        scale = 1.0 / float(b * c * h * w)
        loss = ops.loss_sum(noise, eps, gd.loss_type) * scale
        self.backward(T.loss_grad(noise, eps, gd.loss_type, scale, self.precision))
        T.adam_step(self.flat, self.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, 0, 1.0, coef=g["coef"])
        T.gather_pack(self.flat, self._pack_idx, self._pack_hi, self._pack_lo)
        return loss
