#!/usr/bin/env python3
"""Precision-mode study on the CPU (test infrastructure, not a test): the reference's T = 20 validation chain
(tests/golden/chain.npz) re-run on the oracle with the roundings a device precision mode applies inserted where the
kernels apply them - stored activations, MFMA operands (after GroupNorm + SiLU), weights, GroupNorm (scale, shift) pairs,
softmax probabilities - and the north-star quantities (latents relative, dPSNR, dSAM) reported per mode.

    python tests/precision_emul.py bf16 fp16 fp16:w=x2 ...

A mode is  <type>[:key=value,...]  with type in {fp32, bf16, fp16}; keys
    w   = fp32 | x2     weights exact / as hi + lo of the type (two MFMA passes)
    st  = fp32          activations stored in fp32
    op  = fp32 | x2     operand exact / hi + lo
    gn  = fp32 | fp16 | fp16c   type of the (scale, shift) pairs (default: fp32); fp16c: the shift formed with the rounded scale
    w=x2@c128           hi + lo weights on layers with at most 128 output channels (the product's "fp16" mode)
    w=dK                one-pass weights, DITHERED over the steps of the chain: step s uses round(w + d[s % K] ulp) with K offsets
                        spread over (-1/2, 1/2) ulp (bit-reversed order), so that the weight error - the one rounding that is the
                        same in every step - averages to 1/K of its size over K consecutive steps
    a value of w / st / op may carry "@128+64": the map sizes (of the conv's INPUT) on which the override applies
    is=K                the FIRST K steps of the chain on the exact UNet (the policy's fp32-set steps); fs=K: the last K
    mid=N  with  SPEC|SPEC2   the N steps behind the first `is` ones run in mode SPEC2 (e.g. "fp16:w=d4,is=1,mid=7|fp16:w=x2,st=fp32":
                        one exact step, seven with hi + lo weights and fp32 storage, the rest on dithered one-pass weights)
"""
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import diffusion, gae, metrics, sr3_unet  # noqa: E402
from synth import CHAIN_T, chain_cubes, chain_noise, synth_param, synth_tensor  # noqa: E402

FULL = dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8, 8], attn_res=[16],
            res_blocks=2, image_size=128)
DT = {"bf16": torch.bfloat16, "fp16": torch.float16}


def rnd(x, t):
    return x if t == "fp32" else x.to(DT[t]).float()


def split2(x, t):
    hi = rnd(x, t)
    return hi + rnd(x - hi, t)


def round_zero_sum(w, t):
    """Round to type t so that the rounding errors of the taps of one (cout, cin) pair (3x3) - or of the cins of one cout
    (1x1) - sum to ~0: nearest rounding first, then up to four roundings flipped to the other neighbour, greedily."""
    shp = w.shape
    v = w.reshape(shp[0] * shp[1], -1) if shp[-1] == 3 else w.reshape(shp[0], -1)
    hi = rnd(v, t)
    mant = 10 if t == "fp16" else 7
    for _ in range(4 if shp[-1] == 3 else 64):
        e = v - hi
        E = e.sum(dim=1, keepdim=True)
        ulp = torch.exp2(torch.floor(torch.log2(hi.abs().clamp_min(1e-30))) - mant)
        step = torch.sign(e) * ulp                         # moving hi by +step flips the rounding
        already = e.abs() > 0.5 * ulp * 1.0001             # flipped before
        newE = (E - step).abs()
        newE[already | (e == 0)] = float("inf")
        best = newE.argmin(dim=1, keepdim=True)
        gain = newE.gather(1, best) < E.abs()
        upd = torch.zeros_like(hi).scatter_(1, best, step.gather(1, best) * gain)
        hi = rnd(hi + upd, t)
    return hi.reshape(shp)


class Mode:
    def __init__(self, spec):
        self.spec = spec
        spec, _, spec2 = spec.partition("|")
        self.second = Mode(spec2) if spec2 else None
        parts = spec.split(":")
        self.t = parts[0]
        kv = dict(p.split("=") for p in parts[1].split(",")) if len(parts) > 1 else {}
        self.lvl = {}
        self.cout_max = self.cin_max = None
        for k in ("w", "st", "op"):
            v = kv.get(k, "")
            if "@" in v:
                v, lv = v.split("@")
                if lv.startswith("c"):                      # w=x2@c128: layers with at most 128 output channels (the product's rule)
                    cc = lv[1:].split("i")                  # w=x2@c128i128: ... and at most 128 input channels
                    self.cout_max = int(cc[0])
                    self.cin_max = int(cc[1]) if len(cc) > 1 else None
                else:
                    self.lvl[k] = [int(u) for u in lv.split("+")]
            setattr(self, k, v)
        # wd=4c128: layers the `w` rule leaves at one pass (e.g. w=x2s@c64 -> couts 65 ...) and with at most 128 output channels get
        # one-pass weights DITHERED over the chain's steps (K = 4 offsets; see w=dK)
        self.wd_k, self.wd_cmax = (int(kv["wd"].split("c")[0]), int(kv["wd"].split("c")[1])) if "wd" in kv else (0, 0)
        self.dseq = kv.get("dseq", "")
        self.gn = kv.get("gn", "fp32")
        self.sk = kv.get("sk", "sSuhra")
        self.wl = kv["wl"].split("+") if "wl" in kv else None      # wl=final+ups.18: hi + lo weights on these units only
        self.stl = kv["stl"].split("+") if "stl" in kv else []     # stl=ups.18+ups.17: fp32 storage for the tensors of these units
        self.gnx = kv["gnx"].split("+") if "gnx" in kv else []     # gnx=final_conv: fp32 GroupNorm pairs in these units
        self.cur = ""
        self.phase = 0                                             # step of the chain (w=dK)
        self.sm = kv.get("sm", "")
        self.fs = int(kv.get("fs", 0))                             # fs=K: the LAST K steps of the chain (t < K) on the exact UNet
        self.first = int(kv.get("is", 0))                          # is=K: the FIRST K steps of the chain on the exact UNet
        self.mid = int(kv.get("mid", 0))                           # mid=N: the N steps behind them in the mode behind "|"

    def _ovr(self, k, hw):
        return k not in self.lvl or hw in self.lvl[k]

    def store(self, x, kind="s"):
        """kind: S = residual stream behind an identity skip (block2 of a Cin = Cout unit, attention output), s = behind a projection,
        u = stem / up / down conv outputs, h = block1 output, r = projection output, a = attention internals (qkv, core output)."""
        if self.stl and any(self.cur.startswith(q) for q in self.stl) and kind in self.sk:
            return x
        if self.st and kind in self.sk and self._ovr("st", x.shape[-1]):
            if self.st == "ctr":                            # rounded about its per-(image, channel) mean (diagnostic)
                mu = x.mean(dim=(2, 3), keepdim=True)
                return rnd(x - mu, self.t) + mu
            return x if self.st == "fp32" else split2(x, self.t)
        return rnd(x, self.t)

    def operand(self, x):
        hw = x.shape[-1]
        if self.op == "fp32" and self._ovr("op", hw):
            return x
        if self.op == "x2" and self._ovr("op", hw):
            return split2(x, self.t)
        return rnd(x, self.t)

    def weight(self, w, hw):
        if self.wl is not None and any(self.cur.startswith(q) for q in self.wl):     # wl=final_conv+ups.18: hi + lo weights on these units, the other rules elsewhere
            return split2(w, self.t)
        if self.cout_max is not None and (w.shape[0] > self.cout_max or (self.cin_max is not None and w.shape[1] > self.cin_max)):
            if self.wd_k and w.shape[0] <= self.wd_cmax:
                return self._dither(w, self.wd_k)
            return rnd(w, self.t)
        if self.w == "fp32" and self._ovr("w", hw):
            return w
        if self.w == "x2" and self._ovr("w", hw):
            return split2(w, self.t)
        if self.w == "x2s" and self._ovr("w", hw):          # hi + a 2:4 structured-sparse lo: of every four consecutive input channels
            hi = rnd(w, self.t)                             # (per cout and tap) the two larger low halves are kept (v_smfmac operand)
            lo = w - hi
            co, ci = lo.shape[0], lo.shape[1]
            if ci % 4:
                return hi + rnd(lo, self.t)
            g = lo.reshape(co, ci // 4, 4, -1)
            keep = torch.zeros_like(g)
            keep.scatter_(2, g.abs().topk(2, dim=2).indices, 1.0)
            return hi + rnd((g * keep).reshape(lo.shape), self.t)
        if self.w == "ed" and self._ovr("w", hw):
            return round_zero_sum(w, self.t)
        if self.w.startswith("d") and self._ovr("w", hw):
            return self._dither(w, int(self.w[1:]))
        return rnd(w, self.t)

    def _dither(self, w, K):
        ph = int(format(self.phase % K, "0%db" % max(1, (K - 1).bit_length()))[::-1], 2) if K & (K - 1) == 0 else self.phase % K
        if self.dseq:                                       # dseq=0321 / 03213012: the ORDER of the K offsets over the steps (indices into the sorted offsets)
            ph = int(self.dseq[(self.phase - self.first) % len(self.dseq)])
        mant = 10 if self.t == "fp16" else 7
        ulp = torch.exp2(torch.floor(torch.log2(w.abs().clamp_min(2.0 ** -14))) - mant)
        return rnd(w + ((ph + 0.5) / K - 0.5) * ulp, self.t)

    def conv_in(self, x):
        """What a convolution reads of a stored tensor: with st=x2 (hi + lo planes) only the hi plane."""
        return rnd(x, self.t) if self.st == "x2" else x

    def prob(self, p):
        return p if self.sm == "fp32" else rnd(p, self.t)


def gn_affine(m, x, groups, gamma, beta, xin=None):
    """GroupNorm as the kernels apply it: fp32 statistics, then one fma per element with per-(image, channel) pairs."""
    b, c, h, w = x.shape
    xg = x.reshape(b, groups, -1).double()
    mean = xg.mean(dim=2)
    var = xg.var(dim=2, unbiased=False)
    rstd = (1.0 / torch.sqrt(var + 1e-5)).float()
    mean = mean.float()
    cg = c // groups
    scale = gamma.view(1, c) * rstd.repeat_interleave(cg, dim=1)
    shift = beta.view(1, c) - mean.repeat_interleave(cg, dim=1) * scale
    if m.gnx and any(m.cur.startswith(q) for q in m.gnx):
        pass
    elif m.gn == "fp16c":     # what hsidm_gn_finalize writes: the shift formed with the ROUNDED scale
        scale = rnd(scale, "fp16")
        shift = rnd(beta.view(1, c) - mean.repeat_interleave(cg, dim=1) * scale, "fp16")
    elif m.gn != "fp32":
        scale, shift = rnd(scale, m.gn), rnd(shift, m.gn)
    return (x if xin is None else xin) * scale.view(b, c, 1, 1) + shift.view(b, c, 1, 1)


def block(m, sd, p, x, groups, film=None, res=None, last=False, kind="s"):
    a = gn_affine(m, x, groups, sd[p + "block.0.weight"], sd[p + "block.0.bias"], m.conv_in(x))
    a = m.operand(a * torch.sigmoid(a))
    y = F.conv2d(a, m.weight(sd[p + "block.3.weight"], x.shape[-1]), sd[p + "block.3.bias"], padding=1)
    if film is not None:
        y = y + film.view(x.shape[0], -1, 1, 1)
    if res is not None:
        y = y + res
    return y if last else m.store(y, kind)


def resnet_block(m, sd, p, x, t_emb, groups):
    film = F.linear(t_emb, sd[p + "noise_func.noise_func.0.weight"], sd[p + "noise_func.noise_func.0.bias"])
    h = block(m, sd, p + "block1.", x, groups, film=film, kind="h")
    if (p + "res_conv.weight") in sd:
        r = m.store(F.conv2d(m.conv_in(x), m.weight(sd[p + "res_conv.weight"], x.shape[-1]), sd[p + "res_conv.bias"]), "r")
    else:
        r = x
    # kind S: the stream leaves a block whose residual operand IS the stream (identity skip); s: behind a projection
    return block(m, sd, p + "block2.", h, groups, res=r, kind="s" if (p + "res_conv.weight") in sd else "S")


def self_attention(m, sd, p, x, groups):
    b, c, hh, ww = x.shape
    n = m.operand(gn_affine(m, x, groups, sd[p + "norm.weight"], sd[p + "norm.bias"], m.conv_in(x)))
    qkv = m.store(F.conv2d(n, m.weight(sd[p + "qkv.weight"], hh)), "a")
    q, k, v = qkv.reshape(b, 3, c, hh * ww).unbind(1)
    score = torch.softmax(torch.einsum("bci,bcj->bij", q, k) / math.sqrt(c), dim=-1)
    o = m.store(torch.einsum("bij,bcj->bci", m.prob(score), v).reshape(b, c, hh, ww), "a")
    o = F.conv2d(o, m.weight(sd[p + "out.weight"], hh), sd[p + "out.bias"])
    return m.store(o + x, "S")


def unet_forward(m, sd, cfg, x, gamma):
    groups = cfg["norm_groups"]
    downs, mid, ups = sr3_unet.unet_layout(cfg["in_channel"], cfg["out_channel"], cfg["inner_channel"], cfg["channel_mults"],
                                           cfg["attn_res"], cfg["res_blocks"], cfg["image_size"])
    t_emb = sr3_unet.noise_level_mlp(sd, gamma, cfg["inner_channel"])
    x = m.store(x, "u")
    feats = []

    def unit(p, e, x):
        m.cur = p
        x = resnet_block(m, sd, p + "res_block.", x, t_emb, groups)
        return self_attention(m, sd, p + "attn.", x, groups) if e[3] else x

    for i, e in enumerate(downs):
        p = "downs.%d." % i
        m.cur = p
        if e[0] == "conv":
            x = m.store(F.conv2d(x, m.weight(sd[p + "weight"], x.shape[-1]), sd[p + "bias"], padding=1), "u")
        elif e[0] == "res":
            x = unit(p, e, x)
        else:
            x = m.store(F.conv2d(m.conv_in(x), m.weight(sd[p + "conv.weight"], x.shape[-1]), sd[p + "conv.bias"], stride=2, padding=1), "u")
        feats.append(x)
    for i, e in enumerate(mid):
        x = unit("mid.%d." % i, e, x)
    for i, e in enumerate(ups):
        p = "ups.%d." % i
        if e[0] == "res":
            x = unit(p, e, torch.cat((x, feats.pop()), dim=1))
        else:
            m.cur = p
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = m.store(F.conv2d(m.conv_in(x), m.weight(sd[p + "conv.weight"], x.shape[-1]), sd[p + "conv.bias"], padding=1), "u")
    m.cur = "final_conv."
    return block(m, sd, "final_conv.", x, groups, last=True)


_REF = {}


def run(spec, steps=CHAIN_T, seed=0, fixture=None):
    """seed = 0: the committed chain (reference outputs from chain.npz); other seeds: other noise draws, checked against the
    fp32 oracle run on the same draws (the oracle matches the reference to 6e-7 on the committed chain).
    fixture = (weights, draw, T): a member of the chain fixture set (reference outputs; tests/helpers.py: chain_fixture)."""
    m = Mode(spec)
    if fixture is not None:
        from helpers import chain_fixture
        g, sd, hr, sr, cn = chain_fixture(*fixture)
        seed = 0
    else:
        cn = (lambda gi, k: chain_noise(gi, k)) if seed == 0 else (lambda gi, k: synth_tensor("chain.noise.g%d.k%d" % (gi, k), (1, 3, 128, 128), seed=seed))
        g = np.load(os.path.join(HERE, "golden", "chain.npz"))
        hr, sr = chain_cubes()
        shapes = sr3_unet.unet_param_shapes(FULL)
        sd = {k: torch.from_numpy(synth_param("unet_full." + k, s)) for k, s in shapes.items()}
    gsd = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(HERE, "golden", "gae_cav_state.npz")).items()}
    sched = diffusion.noise_schedule(dict(schedule="cosine", n_timestep=CHAIN_T, linear_start=1e-6, linear_end=1e-2))
    ngr = g["x0"].shape[0]
    nfull = ngr
    ngr = min(ngr, int(os.environ.get("EMUL_GROUPS", ngr)))                           # EMUL_GROUPS=1: a quick look on one group (latents only)
    with torch.no_grad():
        z = torch.from_numpy(g["z"][:ngr])                                            # the reference's own encoder output
        x = torch.from_numpy(np.concatenate([cn(gi, 0) for gi in range(ngr)]))
        exact = lambda xc, gam: sr3_unet.unet_forward(sd, FULL, xc, gam)
        den = exact if spec == "fp32" else (lambda xc, gam: unet_forward(m, sd, FULL, xc, gam))
        den2 = None if m.second is None else (exact if m.second.spec == "fp32" else (lambda xc, gam: unet_forward(m.second, sd, FULL, xc, gam)))
        for i in reversed(range(CHAIN_T)):
            zn = torch.from_numpy(np.concatenate([cn(gi, CHAIN_T - i) for gi in range(ngr)])) if i > 0 else None
            m.phase = CHAIN_T - 1 - i
            if m.second is not None:
                m.second.phase = m.phase
            fn = exact if (i < m.fs or i >= CHAIN_T - m.first) else (den2 if (den2 is not None and m.phase < m.first + m.mid) else den)
            x = diffusion.p_sample_step(fn, sched, x, z, i, zn)
            if CHAIN_T - i >= steps:
                break
        lat = x.numpy()
        out = {"mode": spec}
        if ngr < nfull:
            out.update(groups=ngr, latents_rel=float(np.linalg.norm(lat.astype(np.float64) - g["x0"][:ngr]) / np.linalg.norm(g["x0"][:ngr].astype(np.float64))))
        elif steps >= CHAIN_T:
            y_raw = gae.gae_decode(gsd, 31, [x[i:i + 1] for i in range(ngr)], 8, 2)
            y = y_raw.clamp(0, 1).numpy()
            a = hr[0].transpose(1, 2, 0)
            if seed == 0:
                ref_lat, ref_y = g["x0"], g["y"]
            else:
                if spec == "fp32":
                    _REF[seed] = (lat, y)
                ref_lat, ref_y = _REF[seed]
            ref = ref_y[0].transpose(1, 2, 0)
            got = y[0].transpose(1, 2, 0)
            nrm = lambda u, v: float(np.linalg.norm(u.astype(np.float64) - v) / np.linalg.norm(v.astype(np.float64)))
            free = np.abs(ref_lat) < 1.0
            out.update(seed=seed, latents_rel=nrm(lat, ref_lat), latents_rel_unsaturated=nrm(lat[free], ref_lat[free]), cube_rel=nrm(y, ref_y),
                       dPSNR_dB=abs(metrics.mpsnr(a, got) - metrics.mpsnr(a, ref)),
                       dSAM_deg=abs(metrics.sam_degrees(a, got) - metrics.sam_degrees(a, ref)))
            if seed == 0:       # the continuous companion of the SAM index: the angle on the UN-clamped cubes (tests/helpers.py: sam_continuous)
                from helpers import sam_continuous
                zr = torch.from_numpy(np.ascontiguousarray(g["x0"]))
                ref_raw = gae.gae_decode(gsd, 31, [zr[i:i + 1] for i in range(ngr)], 8, 2).numpy()
                out["dSAM_unclamped_deg"] = abs(sam_continuous(a, y_raw.numpy()[0].transpose(1, 2, 0)) - sam_continuous(a, ref_raw[0].transpose(1, 2, 0)))
            out["_y"], out["_lat"] = y, lat
            out["meets_north_star"] = bool(out["latents_rel"] <= 1e-3 and out["dPSNR_dB"] <= 0.01 and out["dSAM_deg"] <= 1e-3)
    return out


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("EMUL_THREADS", os.cpu_count())))
    args = sys.argv[1:]
    seeds = [0]
    fixtures = [None]
    if args and args[0].startswith("--seeds="):
        seeds = [int(v) for v in args.pop(0)[8:].split(",")]
    if args and args[0].startswith("--fixtures="):          # --fixtures=synth:0:20,orth:0:20,...
        fixtures = [(a, int(b), int(c)) for a, b, c in (v.split(":") for v in args.pop(0)[11:].split(","))]
    for fx, seed, spec in [(fx_, sd_, sp) for fx_ in fixtures for sd_ in seeds for sp in (["fp32"] if sd_ else []) + args]:
        t0 = time.time()
        r = run(spec, seed=seed, fixture=fx)
        if fx:
            r["fixture"] = "%s:%d:%d" % fx
        r.pop("_y", None), r.pop("_lat", None)
        r["seconds"] = round(time.time() - t0, 1)
        print(json.dumps(r), flush=True)
