"""Host-side wrappers of the training kernels (include/hsidm.h, "training step"): NHWC tensors of the mode's storage type in,
kernels enqueued on the current stream, nothing synchronised.  Forward convolutions and their input gradients go through
ops.conv2d (the input gradient of a convolution is a convolution with the transposed, flipped weights)."""
import ctypes as C

import torch

from . import _lib, ops


def _check(code, what):
    _lib.check(code, what)


def _seed_args(seed):
    """seed: a Python int (kernel argument) or a device int64 tensor holding the key (captured steps)."""
    if torch.is_tensor(seed):
        return 0, _lib.ptr(seed)
    return int(seed), None


def gn_act_apply(x0, x1, gn_ab, silu, precision, p_drop=0.0, seed=0, layer=0):
    """a = dropout(act(GroupNorm(cat(x0, x1)))) materialised (NHWC [B, H, W, C0+C1])."""
    B, H, W, C0 = x0.shape
    C1 = 0 if x1 is None else x1.shape[3]
    out = torch.empty((B, H, W, C0 + C1), dtype=x0.dtype, device=x0.device)
    sv, sp = _seed_args(seed)
    _check(_lib.lib().hsidm_gn_act_apply(_lib.prec_id(precision), _lib.ptr(x0), _lib.ptr(x1), C0, C1, _lib.ptr(gn_ab),
                                         ops.XF_AFFINE_SILU if silu else ops.XF_AFFINE, B, H * W, float(p_drop), sv, sp, int(layer),
                                         _lib.ptr(out), _lib.stream_ptr()), "gn_act_apply")
    return out


def gn_act_bwd(da, x0, x1, gn_ab, gamma, groups, silu, precision, dgamma, dbeta, p_drop=0.0, seed=0, layer=0, add=None):
    """Backward of gn_act_apply + GroupNorm: returns (dx0, dx1) and writes dgamma / dbeta (fp32 views of the gradient buffer)."""
    B, H, W, C0 = x0.shape
    C1 = 0 if x1 is None else x1.shape[3]
    C = C0 + C1
    HW = H * W
    nsplit = _nsplit(B, HW, C)
    L = _lib.lib()
    ws = torch.empty(L.hsidm_gn_act_bwd_workspace_floats(B, C, groups, nsplit), dtype=torch.float32, device=x0.device)
    dx0 = torch.empty_like(x0)
    dx1 = None if x1 is None else torch.empty_like(x1)
    assert da.shape == (B, H, W, C) and (add is None or add.shape == da.shape)
    sv, sp = _seed_args(seed)
    _check(L.hsidm_gn_act_bwd(_lib.prec_id(precision), _lib.ptr(da), _lib.ptr(x0), _lib.ptr(x1), C0, C1, _lib.ptr(gn_ab), _lib.ptr(gamma),
                              groups, ops.XF_AFFINE_SILU if silu else ops.XF_AFFINE, B, HW, float(p_drop), sv, sp, int(layer), nsplit,
                              _lib.ptr(ws), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(add), _lib.ptr(dx0), _lib.ptr(dx1),
                              _lib.stream_ptr()), "gn_act_bwd")
    return dx0, dx1


def _nsplit(B, HW, C=64):
    """Pixel ranges per image for the streaming reductions: about four workgroups per CU over the batch (training batches are
    small: 4 latents per GPU in the reference, sr_gae.py:182), at least 64 pixels each - 16 on the wide layers (C >= 256: a
    256-thread workgroup covers only 256 * 8 / C pixels at a time, and the deep levels have 64 ... 1024 pixels per image)."""
    return max(1, min(HW // (16 if C >= 256 else 64), (1024 + B - 1) // B))


_ws_cache = {}


def _workspace(nbytes, dev):
    """One growing scratch buffer per device for the weight-gradient partial sums (launches are stream-ordered)."""
    key = str(dev)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        _ws_cache[key] = buf
    return buf


def conv_wgrad(a0, a1, dy, dw, precision, stride=1, ups=False, deferred=None, db=None, db_images=None):
    """dw [Cout_w, Cin_w, k, k] (fp32, contiguous view of the gradient buffer) = weight gradient of the convolution that
    maps a = cat(a0, a1) to the tensor whose gradient is dy.  db [Cout_w] (optional): the bias gradient, the per-channel sum of dy,
    from the same launch.  db_images [B, Cout_pad] (instead of db): the same sums per image (FiLM's gradient), Cout_pad = Cout rounded
    up to 64 (bias_image_cols), padding columns zero.

    deferred: a DeferredReductions collector - the split-K partial tiles stay in a workspace of this layer's own and ONE launch sums
    all layers at the end (reduce_deferred)."""
    B, Hin, Win, C0 = a0.shape
    C1 = 0 if a1 is None else a1.shape[3]
    _, Ho, Wo, Ct = dy.shape
    cout_w, cin_w, k, _ = dw.shape
    L = _lib.lib()
    geo = (C0, C1, B, Hin, Win, Ho, Wo, Ct, k, stride, int(bool(ups)))
    nb = L.hsidm_conv_wgrad_workspace_bytes(*geo)
    if nb < 0:
        _check(int(nb), "conv_wgrad_workspace_bytes")
    assert dw.dtype == torch.float32
    layout = dw_layout(dw)
    mode = 0
    if db is not None:
        assert db_images is None and db.is_contiguous() and db.dtype == torch.float32 and db.numel() == cout_w
        mode = 1
    if db_images is not None:
        assert db_images.is_contiguous() and db_images.dtype == torch.float32 and tuple(db_images.shape) == (B, bias_image_cols(Ct))
        db, mode = db_images, 2
    if deferred is None:
        ws = _workspace(nb, a0.device)
        assert dw.is_cuda
        dwp, dbp = (dw.data_ptr() if layout else _lib.ptr(dw)), _lib.ptr(db)      # (channels-last view: dense, checked by dw_layout)
    else:
        ws = deferred.workspace(dw, nb, geo, db, mode)
        dwp, dbp = None, None
    _check(L.hsidm_conv_wgrad(_lib.prec_id(precision), _lib.ptr(a0), _lib.ptr(a1), C0, C1, _lib.ptr(dy), B, Hin, Win, Ho, Wo, Ct, k,
                              stride, int(bool(ups)), cout_w, cin_w, dwp, layout, mode, dbp, _lib.ptr(ws), int(nb),
                              _lib.stream_ptr()), "conv_wgrad")


def dw_layout(dw):
    """0: the weight gradient view is PyTorch-contiguous [Cout, Cin, k, k]; 1: channels-last memory order [Cout, k, k, Cin] (what
    training.Trainer keeps its 3x3 weights and their gradients in)."""
    if dw.is_contiguous():
        return 0
    co, ci, kh, kw = dw.shape
    assert dw.stride() == (ci * kh * kw, 1, kw * ci, ci), "weight gradient must be contiguous or channels-last"
    return 1


def bias_image_cols(cout):
    """Columns of conv_wgrad's per-image bias sums: the gradient tensor's channel count rounded up to the kernel's 64-cout tile."""
    return (cout + 63) // 64 * 64


class DeferredReductions:
    """Per-layer split-K workspaces (persistent: their addresses go into a device table once) and the one-launch reduction."""

    def __init__(self, device):
        self.dev = device
        self.ws, self.items, self.table, self.blocks = {}, [], None, 0

    def workspace(self, dw, nbytes, geo, db=None, mode=0):
        key = dw.data_ptr()
        hit = self.ws.get(key)
        dbp = None if db is None else (db.data_ptr(), mode)
        if hit is None or hit[0].numel() < nbytes or hit[1] != geo or hit[4] != dbp:
            plan = (C.c_int32 * 5)()
            _check(_lib.lib().hsidm_conv_wgrad_plan(*geo, plan), "conv_wgrad_plan")
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.dev)
            self.ws[key] = (buf, geo, tuple(plan), dw, dbp, db)
            self.table = None                           # addresses changed: rebuild the table
        return self.ws[key][0]

    def reduce(self):
        if not self.ws:
            return
        if self.table is None:
            n = len(self.ws) + sum(1 for v in self.ws.values() if v[4] is not None)
            arr = (_lib.WgradItem * n)()
            blk, i = 0, 0
            for buf, geo, plan, dw, dbp, db in self.ws.values():
                it = arr[i]
                it.ws, it.dw = buf.data_ptr(), dw.data_ptr()
                it.nsplit, it.NT, it.Cout_pad, it.Cin_pad, ppb = plan
                it.Cout_w, it.Cin_w, it.block0, it.layout = dw.shape[0], dw.shape[1], blk, dw_layout(dw)
                blk += (dw.shape[0] * dw.shape[1] + ppb - 1) // ppb
                i += 1
                if dbp is not None:                     # the bias partials: a [nsplit][1][rows][1] stack behind the weight partials
                    nsplit, NT, cout_pad, cin_pad, _ = plan
                    rows_pad = cout_pad * (geo[2] if dbp[1] == 2 else 1)      # per image: [nsplit][B][Cout_pad]
                    rows = rows_pad if dbp[1] == 2 else dw.shape[0]
                    it = arr[i]
                    it.ws, it.dw = buf.data_ptr() + 4 * nsplit * NT * cout_pad * cin_pad, dbp[0]
                    it.nsplit, it.NT, it.Cout_pad, it.Cin_pad = nsplit, 1, rows_pad, 1
                    it.Cout_w, it.Cin_w, it.block0 = rows, 1, blk
                    blk += (rows + ppb - 1) // ppb
                    i += 1
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self.table = host.to(self.dev)
            self.n, self.blocks = n, blk
        _check(_lib.lib().hsidm_wgrad_reduce_all(_lib.ptr(self.table), self.n, self.blocks, _lib.stream_ptr()), "wgrad_reduce_all")


def add(a, b, precision):
    out = torch.empty_like(a)
    _check(_lib.lib().hsidm_add(_lib.prec_id(precision), _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), a.numel(), _lib.stream_ptr()), "add")
    return out


def zero_insert2(x, Ho, Wo, precision):
    B, Hi, Wi, C = x.shape
    out = torch.empty((B, Ho, Wo, C), dtype=x.dtype, device=x.device)
    _check(_lib.lib().hsidm_zero_insert2(_lib.prec_id(precision), _lib.ptr(x), _lib.ptr(out), B, Hi, Wi, Ho, Wo, C, _lib.stream_ptr()),
           "zero_insert2")
    return out


def sum2x2(x, precision):
    B, H2, W2, C = x.shape
    out = torch.empty((B, H2 // 2, W2 // 2, C), dtype=x.dtype, device=x.device)
    _check(_lib.lib().hsidm_sum2x2(_lib.prec_id(precision), _lib.ptr(x), _lib.ptr(out), B, H2 // 2, W2 // 2, C, _lib.stream_ptr()), "sum2x2")
    return out


def channel_sums(x, precision, out_c=None, want_bc=False, cout=None):
    """Per-channel sums of an NHWC tensor: out_c[c] = sum over (b, h, w) (written into the given fp32 view), and optionally the
    per-image sums [B, cout]."""
    B, H, W, C = x.shape
    cout = C if cout is None else cout
    # (a quarter of the splits of the GroupNorm reductions: the column-sum kernel walks them per image)
    part, nsplit = ops.channel_partials(x, precision, nsplit=max(1, _nsplit(B, H * W) // 4))
    bc = torch.empty((B, cout), dtype=torch.float32, device=x.device) if want_bc else None
    _check(_lib.lib().hsidm_colsum(_lib.ptr(part), nsplit, B, C, cout, _lib.ptr(bc), _lib.ptr(out_c), _lib.stream_ptr()), "colsum")
    return bc


def loss_grad(noise, eps, kind, scale, precision):
    """d(scale * loss)/d(eps) as an NHWC tensor [B, H, W, 8-padded C]."""
    B, Ci, H, W = eps.shape
    cpad = (Ci + 7) // 8 * 8
    out = torch.empty((B, H, W, cpad), dtype=_lib.act_dtype(precision), device=eps.device)
    _check(_lib.lib().hsidm_loss_grad(_lib.prec_id(precision), _lib.ptr(noise), _lib.ptr(eps), B, Ci, H * W, cpad, {"l1": 0, "l2": 1}[kind],
                                      float(scale), _lib.ptr(out), _lib.stream_ptr()), "loss_grad")
    return out


def noise_film_bwd(gamma, t_emb, dfilm, mlp, wf, grads):
    """grads = (dw1, db1, dw2, db2, dwf, dbf): fp32 views of the gradient buffer."""
    B, dim = t_emb.shape
    F = wf.shape[0]
    L = _lib.lib()
    ws = torch.empty(L.hsidm_noise_film_bwd_workspace_floats(B, dim, F), dtype=torch.float32, device=t_emb.device)
    w1, b1, w2, _ = mlp
    dw1, db1, dw2, db2, dwf, dbf = grads
    _check(L.hsidm_noise_film_bwd(_lib.ptr(gamma), _lib.ptr(t_emb), _lib.ptr(dfilm), B, dim, _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2),
                                  _lib.ptr(wf), F, _lib.ptr(dw1), _lib.ptr(db1), _lib.ptr(dw2), _lib.ptr(db2), _lib.ptr(dwf), _lib.ptr(dbf),
                                  _lib.ptr(ws), _lib.stream_ptr()), "noise_film_bwd")


def _gemm(a, a_f32, sab, sam, sak, b, b_f32, sbb, sbk, sbn, c, c_f32, scb, scm, M, N, K, batch, alpha):
    _check(_lib.lib().hsidm_bgemm(a, int(a_f32), sab, sam, sak, b, int(b_f32), sbb, sbk, sbn, c, int(c_f32), scb, scm, M, N, K, batch,
                                  float(alpha), _lib.stream_ptr()), "bgemm")


def attention_bwd(qkv, do, precision):
    """Gradient of ops.attention: qkv [B, H, W, 3C] (q | k | v thirds), do [B, H, W, C] -> dqkv [B, H, W, 3C]."""
    B, H, W, C3 = qkv.shape
    C, N = C3 // 3, H * W
    f32 = qkv.dtype == torch.float32
    es = qkv.element_size()
    dev = qkv.device
    scale = 1.0 / (C ** 0.5)
    P = torch.empty((B, N, N), dtype=torch.float32, device=dev)
    dP = torch.empty((B, N, N), dtype=torch.float32, device=dev)
    dqkv = torch.empty_like(qkv)
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + C * es, qkv.data_ptr() + 2 * C * es
    dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + C * es, dqkv.data_ptr() + 2 * C * es
    assert qkv.is_contiguous() and do.is_contiguous() and qkv.is_cuda
    L = _lib.lib()
    _gemm(q, f32, N * C3, C3, 1, k, f32, N * C3, 1, C3, P.data_ptr(), True, N * N, N, N, N, C, B, scale)          # S = scale Q K^T
    _check(L.hsidm_softmax_rows(_lib.ptr(P), B * N, N, _lib.stream_ptr()), "softmax_rows")
    _gemm(do.data_ptr(), f32, N * C, C, 1, v, f32, N * C3, 1, C3, dP.data_ptr(), True, N * N, N, N, N, C, B, 1.0)   # dP = dO V^T
    _check(L.hsidm_softmax_bwd_rows(_lib.ptr(P), _lib.ptr(dP), B * N, N, scale, _lib.stream_ptr()), "softmax_bwd_rows")
    _gemm(dP.data_ptr(), True, N * N, N, 1, k, f32, N * C3, C3, 1, dq, f32, N * C3, C3, N, C, N, B, 1.0)            # dQ = dS K
    _gemm(dP.data_ptr(), True, N * N, 1, N, q, f32, N * C3, C3, 1, dk, f32, N * C3, C3, N, C, N, B, 1.0)            # dK = dS^T Q
    _gemm(P.data_ptr(), True, N * N, 1, N, do.data_ptr(), f32, N * C, C, 1, dv, f32, N * C3, C3, N, C, N, B, 1.0)   # dV = P^T dO
    return dqkv


def gather_pack(src, idx, out_hi, out_lo=None):
    _check(_lib.lib().hsidm_gather_pack(_lib.ptr(src), _lib.ptr(idx), idx.numel(), _lib.ptr(out_hi), _lib.ptr(out_lo), _lib.stream_ptr()),
           "gather_pack")


def adam_coefs(lr, beta1, beta2, step):
    """(lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)): what hsidm_adam_step derives from `step`, for the device-resident form."""
    return float(lr / (1.0 - beta1 ** step)), float(1.0 / (1.0 - beta2 ** step) ** 0.5)


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, coef=None):
    """coef: optional device tensor of two floats (adam_coefs) read by the kernel instead of deriving them from `step`."""
    _check(_lib.lib().hsidm_adam_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), p.numel(), float(lr), float(beta1), float(beta2),
                                      float(eps), int(step), float(grad_scale), _lib.ptr(coef), _lib.stream_ptr()), "adam_step")
