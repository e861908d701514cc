// Training step, streaming part (reference model/model.py:49-59, diffusion.py:222-250, unet.py:80-111):
//   gn_act_apply      a = dropout(act(GroupNorm(x)))            the conv operand of a Block, materialised in training mode
//   gn_act_bwd_*      its backward: per-channel sums -> (dgamma, dbeta, group means) -> dx
//   zero_insert2 / sum2x2   adjoint of the stride-2 sampling / of the nearest x2 upsample
//   colsum            bias and FiLM gradients from per-(image, split, channel) sums
//   loss_grad         d(sum-reduced L1 | L2)/d(eps)
//   noise_film_bwd    FiLM projections + noise-level MLP backward
//   adam_step         fused Adam over the flat parameter buffer (torch.optim.Adam semantics, model/model.py:37-41)
// All HBM-bound (16-byte channel vectors per lane) or tiny; fp32 arithmetic; deterministic (no atomics).
#include "common.h"
#include "philox.h"
#include "../../include/hsidm.h"

namespace hsidm {

// 8 channels c .. c+7 of pixel (b, p) of the channel concat (s0 | s1)
template <typename T>
__device__ __forceinline__ void load_cat8(const T* __restrict__ s0, const T* __restrict__ s1, int C0, int C1, size_t bp, int c, float (&v)[8]) {
    if (c < C0) Vec8<T>::load(s0 + bp * C0 + c, v);
    else Vec8<T>::load(s1 + bp * C1 + (c - C0), v);
}

__device__ __forceinline__ float sigmoid_precise(float u) { return 1.0f / (1.0f + expf(-u)); }

// (scale, shift) of 8 consecutive channels of the GroupNorm table (idx % 8 == 0: four 16-byte loads, requested together - a load
// per channel inside the arithmetic loop was a dependent L2 round trip each, which is what a launch of a few microseconds is made of)
__device__ __forceinline__ void load_ab8(const float2* __restrict__ ab, size_t idx, float (&sc)[8], float (&sh)[8]) {
    const f32x4* q = reinterpret_cast<const f32x4*>(ab + idx);
    f32x4 r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = q[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) { sc[2 * k] = r[k][0]; sh[2 * k] = r[k][1]; sc[2 * k + 1] = r[k][2]; sh[2 * k + 1] = r[k][3]; }
}

// keep/scale factors of the 8 elements e0 .. e0+7 (e0 % 8 == 0) of a dropout mask: element e keeps its value iff word (e & 3) of
// Philox4x32-10(key = seed, counter = (e >> 2, stream = layer)) >= thresh; restated in oracle/train.py
__device__ __forceinline__ void dropout8(uint64_t e0, uint32_t layer, uint64_t seed, uint32_t thresh, float inv_keep, float (&f)[8]) {
    const uint64_t q = e0 >> 2;
    const Philox4 r0 = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), layer, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    const Philox4 r1 = philox4x32_10((uint32_t)(q + 1), (uint32_t)((q + 1) >> 32), layer, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    f[0] = r0.x >= thresh ? inv_keep : 0.f; f[1] = r0.y >= thresh ? inv_keep : 0.f;
    f[2] = r0.z >= thresh ? inv_keep : 0.f; f[3] = r0.w >= thresh ? inv_keep : 0.f;
    f[4] = r1.x >= thresh ? inv_keep : 0.f; f[5] = r1.y >= thresh ? inv_keep : 0.f;
    f[6] = r1.z >= thresh ? inv_keep : 0.f; f[7] = r1.w >= thresh ? inv_keep : 0.f;
}

// seed_ptr != NULL: the key is read from device memory (a captured training step is replayed with a new key per iteration)
struct DropArgs { uint64_t seed; const uint64_t* seed_ptr; uint32_t layer; uint32_t thresh; float inv_keep; };
__device__ __forceinline__ uint64_t drop_key(const DropArgs& d) { return d.seed_ptr ? *d.seed_ptr : d.seed; }

// ---- forward: a = dropout(act(scale*x + shift)) ---------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_act_apply_kernel(const T* __restrict__ s0, const T* __restrict__ s1, int C0, int C1,
                                                           const float2* __restrict__ ab, int silu_on, int HW, int64_t nvec,
                                                           DropArgs dr, T* __restrict__ out) {
    const int C = C0 + C1, nv = C >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int64_t bp = i / nv;
        const int c = (int)(i - bp * nv) * 8;
        const int b = (int)(bp / HW);
        float x[8], a[8], sc[8], sh[8];
        load_cat8<T>(s0, s1, C0, C1, (size_t)bp, c, x);
        load_ab8(ab, (size_t)b * C + c, sc, sh);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float u = fmaf(x[k], sc[k], sh[k]);
            a[k] = silu_on ? u * sigmoid_precise(u) : u;
        }
        if (dr.thresh) {
            float f[8];
            dropout8((uint64_t)bp * C + c, dr.layer, drop_key(dr), dr.thresh, dr.inv_keep, f);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] *= f[k];
        }
        Vec8<T>::store(out + (size_t)bp * C + c, a);
    }
}

// gradient wrt u = gamma*xhat + beta of one vector: dy = da * dropout' * act'(u)
__device__ __forceinline__ void dy8(const float (&da)[8], const float (&u)[8], int silu_on, bool drop, const float (&f)[8], float (&dy)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float g = da[k];
        if (drop) g *= f[k];
        if (silu_on) {
            const float s = sigmoid_precise(u[k]);
            g *= s * (1.0f + u[k] * (1.0f - s));
        }
        dy[k] = g;
    }
}

// ---- backward pass 1: per-(image, split, channel) sums (S1, S2) = (sum dy, sum dy*xhat); grid (nsplit, B) --------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_act_bwd_reduce_kernel(const T* __restrict__ da, const T* __restrict__ s0, const T* __restrict__ s1,
                                                                int C0, int C1, const float2* __restrict__ ab, const float2* __restrict__ mr,
                                                                int groups, int silu_on, int HW, int nsplit, DropArgs dr,
                                                                float2* __restrict__ part) {
    __shared__ float red[256 * 8 * 2];
    const int C = C0 + C1, nvec = C >> 3, rows = 256 / nvec, cpg = C / groups;
    const int split = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const int per = (HW + nsplit - 1) / nsplit;
    const int p_begin = split * per, p_end = min(HW, p_begin + per);
    const int cvi = t % nvec, prow = t / nvec;
    float s1a[8], s2a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s1a[k] = s2a[k] = 0.f;
    if (prow < rows) {
        const int c = cvi * 8;
        float sc[8], sh[8], mu[8], rs[8];
        load_ab8(ab, (size_t)b * C + c, sc, sh);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float2 m = mr[(size_t)b * groups + (c + k) / cpg];
            mu[k] = m.x; rs[k] = m.y;
        }
        // the operands of the next pixel are requested before this one's arithmetic (a Philox mask and two expf per element):
        // with 4 ... 16 pixels per thread on the deep levels the loop was one exposed round trip per pixel
        float x[8], g[8], xn[8], gn[8];
        int p = p_begin + prow;
        if (p < p_end) {
            load_cat8<T>(s0, s1, C0, C1, (size_t)b * HW + p, c, x);
            Vec8<T>::load(da + ((size_t)b * HW + p) * C + c, g);
        }
        for (; p < p_end; p += rows) {
            const size_t bp = (size_t)b * HW + p;
            const bool more = p + rows < p_end;
            const size_t bpn = more ? bp + rows : bp;
            load_cat8<T>(s0, s1, C0, C1, bpn, c, xn);
            Vec8<T>::load(da + bpn * C + c, gn);
            float u[8], f[8], dy[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) u[k] = fmaf(x[k], sc[k], sh[k]);
            if (dr.thresh) dropout8((uint64_t)bp * C + c, dr.layer, drop_key(dr), dr.thresh, dr.inv_keep, f);
            dy8(g, u, silu_on, dr.thresh != 0, f, dy);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s1a[k] += dy[k];
                s2a[k] = fmaf(dy[k], (x[k] - mu[k]) * rs[k], s2a[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { x[k] = xn[k]; g[k] = gn[k]; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[(prow * C + c + k) * 2] = s1a[k];
            red[(prow * C + c + k) * 2 + 1] = s2a[k];
        }
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        float a = 0.f, d = 0.f;
        for (int r = 0; r < rows; ++r) { a += red[(r * C + c) * 2]; d += red[(r * C + c) * 2 + 1]; }
        part[((size_t)b * nsplit + split) * C + c] = make_float2(a, d);
    }
}

// ---- backward pass 2a: grid (B): S[b][c] = sum over splits; group means gm[b][g] = (sum_c gamma_c S1, sum_c gamma_c S2) / n ----
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float2* __restrict__ part, int nsplit, int C, int HW, int groups,
                                                              const float* __restrict__ gamma, float2* __restrict__ S, float2* __restrict__ gm) {
    extern __shared__ float sm[];          // [C][2] gamma-weighted sums, then [256][2] scratch
    float* scr = sm + 2 * C;
    const int b = blockIdx.x, t = threadIdx.x, cpg = C / groups;
    // channel windows of up to 256 channels; within a window thread = (channel, split group): the splits of a channel are
    // summed by 256 / window threads and combined through LDS in a fixed order
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int cw = min(256, C - c0), ng = 256 / cw;
        const int cl = t % cw, sg = t / cw;
        float a = 0.f, d = 0.f;
        if (sg < ng) {
            int s = sg;
            for (; s + 7 * ng < nsplit; s += 8 * ng) {          // eight entries in flight, added in order
                float2 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = part[((size_t)b * nsplit + s + j * ng) * C + c0 + cl];
#pragma unroll
                for (int j = 0; j < 8; ++j) { a += v[j].x; d += v[j].y; }
            }
            for (; s < nsplit; s += ng) {
                const float2 v = part[((size_t)b * nsplit + s) * C + c0 + cl];
                a += v.x; d += v.y;
            }
        }
        __syncthreads();
        scr[2 * t] = a; scr[2 * t + 1] = d;
        __syncthreads();
        if (t < cw) {
            float sa = 0.f, sd = 0.f;
            for (int g = 0; g < ng; ++g) { sa += scr[2 * (g * cw + t)]; sd += scr[2 * (g * cw + t) + 1]; }
            const int c = c0 + t;
            S[(size_t)b * C + c] = make_float2(sa, sd);
            sm[2 * c] = sa * gamma[c];
            sm[2 * c + 1] = sd * gamma[c];
        }
    }
    __syncthreads();
    for (int g = t; g < groups; g += 256) {
        double a = 0.0, d = 0.0;
        for (int k = 0; k < cpg; ++k) { a += sm[2 * (g * cpg + k)]; d += sm[2 * (g * cpg + k) + 1]; }
        const double n = (double)cpg * HW;
        gm[(size_t)b * groups + g] = make_float2((float)(a / n), (float)(d / n));
    }
}
// ---- backward pass 2b: dgamma[c] = sum_b S2[b][c], dbeta[c] = sum_b S1[b][c] --------------------------------------------------
__global__ __launch_bounds__(256) void gn_bwd_params_kernel(const float2* __restrict__ S, int B, int C, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, d = 0.0;
    for (int b = 0; b < B; ++b) { const float2 v = S[(size_t)b * C + c]; a += v.x; d += v.y; }
    dbeta[c] = (float)a;
    dgamma[c] = (float)d;
}

// ---- backward pass 3: dx = rstd * (gamma*dy - M1 - xhat*M2) (+ add), split over the two halves of a concat -------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_act_bwd_apply_kernel(const T* __restrict__ da, const T* __restrict__ s0, const T* __restrict__ s1,
                                                               int C0, int C1, const float2* __restrict__ ab, const float2* __restrict__ mr,
                                                               const float2* __restrict__ gm, const float* __restrict__ gamma, int groups,
                                                               int silu_on, int HW, int64_t nvec, DropArgs dr, const T* __restrict__ add,
                                                               T* __restrict__ dx0, T* __restrict__ dx1, const float2* __restrict__ S, int B,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int C = C0 + C1, nv = C >> 3, cpg = C / groups;
    if (blockIdx.x == 0) {              // parameter gradients: dgamma[c] = sum_b S2[b][c], dbeta[c] = sum_b S1[b][c] (one launch fewer)
        for (int c = threadIdx.x; c < C; c += 256) {
            double a = 0.0, d = 0.0;
            for (int b = 0; b < B; ++b) { const float2 v = S[(size_t)b * C + c]; a += v.x; d += v.y; }
            dbeta[c] = (float)a;
            dgamma[c] = (float)d;
        }
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int64_t bp = i / nv;
        const int c = (int)(i - bp * nv) * 8;
        const int b = (int)(bp / HW);
        float x[8], g[8], u[8], f[8], dy[8], o[8];
        load_cat8<T>(s0, s1, C0, C1, (size_t)bp, c, x);
        Vec8<T>::load(da + (size_t)bp * C + c, g);
        float sc[8], sh[8];
        load_ab8(ab, (size_t)b * C + c, sc, sh);
        float2 mk[8], Mk[8];                    // every table read is requested before the arithmetic (see load_ab8)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int gi = (c + k) / cpg;
            mk[k] = mr[(size_t)b * groups + gi];
            Mk[k] = gm[(size_t)b * groups + gi];
        }
        float rres[8];
        if (add) Vec8<T>::load(add + (size_t)bp * C + c, rres);
#pragma unroll
        for (int k = 0; k < 8; ++k) u[k] = fmaf(x[k], sc[k], sh[k]);
        if (dr.thresh) dropout8((uint64_t)bp * C + c, dr.layer, drop_key(dr), dr.thresh, dr.inv_keep, f);
        dy8(g, u, silu_on, dr.thresh != 0, f, dy);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float2 m = mk[k], M = Mk[k];
            const float xh = (x[k] - m.x) * m.y;
            // rstd*gamma*dy = scale*dy (scale is the table's rstd*gamma)
            o[k] = sc[k] * dy[k] - m.y * (M.x + xh * M.y);
        }
        const bool first = c < C0;
        T* dst = first ? dx0 : dx1;
        const size_t off = first ? (size_t)bp * C0 + c : (size_t)bp * C1 + (c - C0);
        if (add) {
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] += rres[k];
        }
        Vec8<T>::store(dst + off, o);
    }
}

// ---- adjoints of the resampling steps -------------------------------------------------------------------------------
// out [B][Ho][Wo][C]: out[2y][2x] = in[y][x] (in [B][Hi][Wi][C], Hi = (Ho+1)/2), zero elsewhere
template <typename T>
__global__ __launch_bounds__(256) void zero_insert2_kernel(const T* __restrict__ in, T* __restrict__ out, int Hi, int Wi, int Ho, int Wo,
                                                           int C, int64_t nvec) {
    const int nv = C >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        int64_t r = i / nv;
        const int c = (int)(i - r * nv) * 8;
        const int x = (int)(r % Wo); r /= Wo;
        const int y = (int)(r % Ho);
        const int64_t b = r / Ho;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
        if (!(x & 1) && !(y & 1)) Vec8<T>::load(in + ((size_t)(b * Hi + (y >> 1)) * Wi + (x >> 1)) * C + c, v);
        Vec8<T>::store(out + (size_t)i * 8, v);
    }
}
// out [B][H][W][C] = sum of the 2x2 block of in [B][2H][2W][C]
template <typename T>
__global__ __launch_bounds__(256) void sum2x2_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W, int C, int64_t nvec) {
    const int nv = C >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        int64_t r = i / nv;
        const int c = (int)(i - r * nv) * 8;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int64_t b = r / H;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float v[8];
                Vec8<T>::load(in + ((size_t)(b * 2 * H + 2 * y + dy) * (2 * W) + 2 * x + dx) * C + c, v);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += v[k];
            }
        Vec8<T>::store(out + (size_t)i * 8, acc);
    }
}

// out = a + b (gradients that meet at a fan-out of the network: a skip connection and the main path)
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float x[8], y[8];
        Vec8<T>::load(a + (size_t)i * 8, x);
        Vec8<T>::load(b + (size_t)i * 8, y);
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] += y[k];
        Vec8<T>::store(out + (size_t)i * 8, x);
    }
}

// ---- column sums of a statistics slab: out_bc[b][c] = sum_s part[b][s][c].x, out_c[c] = sum_b out_bc -------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float2* __restrict__ part, int nsplit, int B, int C, int Cout,
                                                     float* __restrict__ out_bc, float* __restrict__ out_c) {
    // workgroup = 64 channels x 4 split groups; the groups meet in LDS in a fixed order
    __shared__ float scr[4][64];
    const int cl = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float tot = 0.f;
    for (int b = 0; b < B; ++b) {
        float a = 0.f;
        if (c < Cout)
            for (int s = sg; s < nsplit; s += 4) a += part[((size_t)b * nsplit + s) * C + c].x;
        __syncthreads();
        scr[sg][cl] = a;
        __syncthreads();
        if (sg == 0 && c < Cout) {
            const float v = ((scr[0][cl] + scr[1][cl]) + scr[2][cl]) + scr[3][cl];
            if (out_bc) out_bc[(size_t)b * Cout + c] = v;
            tot += v;
        }
    }
    if (sg == 0 && c < Cout && out_c) out_c[c] = tot;
}

// ---- d(loss)/d(eps): loss = scale * sum |noise - eps| (L1) or scale * sum (noise - eps)^2 (L2); NCHW fp32 in, NHWC (Cpad) out ----
template <typename T>
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ noise, const float* __restrict__ eps, int Cimg, int HW,
                                                        int Cpad, int kind, float scale, int64_t npix, T* __restrict__ out) {
    for (int64_t bp = (int64_t)blockIdx.x * 256 + threadIdx.x; bp < npix; bp += (int64_t)gridDim.x * 256) {
        const int64_t b = bp / HW;
        const int p = (int)(bp - b * HW);
        for (int c0 = 0; c0 < Cpad; c0 += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = c0 + k;
                float g = 0.f;
                if (c < Cimg) {
                    const size_t idx = ((size_t)b * Cimg + c) * HW + p;
                    const float d = noise[idx] - eps[idx];
                    g = kind == HSIDM_LOSS_L1 ? (d > 0.f ? -scale : (d < 0.f ? scale : 0.f)) : -2.0f * scale * d;
                }
                v[k] = g;
            }
            Vec8<T>::store(out + (size_t)bp * Cpad + c0, v);
        }
    }
}

// ---- FiLM projections + noise-level MLP, backward (unet.py:23-50,182-187) -------------------------------------------------
// kernel 1, grid (ceil(F/256)): thread f: dwf[f][:] = sum_b dfilm[b][f] * t[b][:], dbf[f] = sum_b dfilm[b][f];
//           block partial of dt[b][k] = sum_f dfilm[b][f] * wf[f][k] -> dt_part[block][b][k]
__global__ __launch_bounds__(256) void film_bwd_kernel(const float* __restrict__ dfilm, const float* __restrict__ t_emb,
                                                       const float* __restrict__ wf, int B, int dim, int F, float* __restrict__ dwf,
                                                       float* __restrict__ dbf, float* __restrict__ dt_part) {
    extern __shared__ float sm[];          // [256][dim + 1] contribution rows, reused per image
    const int t = threadIdx.x, f = blockIdx.x * 256 + t;
    const int pitch = dim + 1;
    if (f < F) {
        float db = 0.f;
        for (int b = 0; b < B; ++b) db += dfilm[(size_t)b * F + f];
        dbf[f] = db;
        for (int k = 0; k < dim; ++k) {
            float a = 0.f;
            for (int b = 0; b < B; ++b) a = fmaf(dfilm[(size_t)b * F + f], t_emb[(size_t)b * dim + k], a);
            dwf[(size_t)f * dim + k] = a;
        }
    }
    for (int b = 0; b < B; ++b) {
        __syncthreads();
        const float d = f < F ? dfilm[(size_t)b * F + f] : 0.f;
        for (int k = 0; k < dim; ++k) sm[t * pitch + k] = f < F ? d * wf[(size_t)f * dim + k] : 0.f;
        __syncthreads();
        for (int k = t; k < dim; k += 256) {
            float a = 0.f;
            for (int r = 0; r < 256; ++r) a += sm[r * pitch + k];
            dt_part[((size_t)blockIdx.x * B + b) * dim + k] = a;
        }
    }
}
// kernel 2, one workgroup: dt = sum of block partials (+ dt_extra), then the MLP backward with recomputed activations
__global__ __launch_bounds__(256) void noise_mlp_bwd_kernel(const float* __restrict__ gamma_lvl, const float* __restrict__ dt_part, int nblk,
                                                            int B, int dim, const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ w2, float* __restrict__ dw1, float* __restrict__ db1,
                                                            float* __restrict__ dw2, float* __restrict__ db2, float* __restrict__ work) {
    // work: [B][dim] pe, [B][4dim] pre, [B][4dim] act, [B][dim] dt, [B][4dim] dpre   (global scratch, one workgroup)
    const int t = threadIdx.x, H = 4 * dim, half = dim >> 1;
    float* pe = work;
    float* pre = pe + (size_t)B * dim;
    float* act = pre + (size_t)B * H;
    float* dt = act + (size_t)B * H;
    float* dpre = dt + (size_t)B * dim;
    for (int i = t; i < B * dim; i += 256) {
        const int b = i / dim, k = i - b * dim;
        const int kk = k < half ? k : k - half;
        const float e = gamma_lvl[b] * expf(-9.210340371976184f * ((float)kk / (float)half));
        pe[i] = k < half ? sinf(e) : cosf(e);
        float a = 0.f;
        for (int j = 0; j < nblk; ++j) a += dt_part[((size_t)j * B + b) * dim + k];
        dt[i] = a;
    }
    __syncthreads();
    for (int i = t; i < B * H; i += 256) {
        const int b = i / H, j = i - b * H;
        float a = b1[j];
        for (int k = 0; k < dim; ++k) a = fmaf(w1[(size_t)j * dim + k], pe[b * dim + k], a);
        pre[i] = a;
        act[i] = a * sigmoid_precise(a);
    }
    __syncthreads();
    for (int i = t; i < dim * H; i += 256) {            // dw2[k][j] = sum_b dt[b][k] * act[b][j]
        const int k = i / H, j = i - k * H;
        float a = 0.f;
        for (int b = 0; b < B; ++b) a = fmaf(dt[b * dim + k], act[b * H + j], a);
        dw2[i] = a;
    }
    for (int k = t; k < dim; k += 256) {
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += dt[b * dim + k];
        db2[k] = a;
    }
    for (int i = t; i < B * H; i += 256) {              // dpre = (dt . w2[:, j]) * swish'(pre)
        const int b = i / H, j = i - b * H;
        float a = 0.f;
        for (int k = 0; k < dim; ++k) a = fmaf(dt[b * dim + k], w2[(size_t)k * H + j], a);
        const float s = sigmoid_precise(pre[i]);
        dpre[i] = a * s * (1.0f + pre[i] * (1.0f - s));
    }
    __syncthreads();
    for (int i = t; i < H * dim; i += 256) {            // dw1[j][k] = sum_b dpre[b][j] * pe[b][k]
        const int j = i / dim, k = i - j * dim;
        float a = 0.f;
        for (int b = 0; b < B; ++b) a = fmaf(dpre[b * H + j], pe[b * dim + k], a);
        dw1[i] = a;
    }
    for (int j = t; j < H; j += 256) {
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += dpre[b * H + j];
        db1[j] = a;
    }
}

// ---- Adam (torch.optim.Adam, no weight decay / amsgrad; model/model.py:37-41) -------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float beta1, float beta2, float step_size,
                                                   float inv_sqrt_bc2, float eps, float grad_scale, const float* __restrict__ coef) {
    if (coef) { step_size = coef[0]; inv_sqrt_bc2 = coef[1]; }      // a captured step: the bias corrections come from device memory
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i], mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gg[k] * grad_scale;
            mm[k] = beta1 * mm[k] + (1.0f - beta1) * gk;           // exp_avg.lerp_(grad, 1 - beta1)
            vv[k] = beta2 * vv[k] + (1.0f - beta2) * gk * gk;      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
            const float denom = sqrtf(vv[k]) * inv_sqrt_bc2 + eps;
            pp[k] -= step_size * (mm[k] / denom);
        }
        reinterpret_cast<f32x4*>(p)[i] = pp; reinterpret_cast<f32x4*>(m)[i] = mm; reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        const float gk = g[i] * grad_scale;
        const float mk = beta1 * m[i] + (1.0f - beta1) * gk;
        const float vk = beta2 * v[i] + (1.0f - beta2) * gk * gk;
        m[i] = mk; v[i] = vk;
        p[i] -= step_size * (mk / (sqrtf(vk) * inv_sqrt_bc2 + eps));
    }
}

// ---- packed weights from the flat fp32 master copy: out_hi[i] = bf16(src[idx[i]]), out_lo[i] = bf16(src[idx[i]] - hi) ----------------
// n % 8 == 0: a thread packs 8 consecutive outputs (32 bytes of indices in, 16 bytes of bf16 out)
__global__ __launch_bounds__(256) void gather_pack_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int64_t n8,
                                                          bf16* __restrict__ hi, bf16* __restrict__ lo) {
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const i32x4 k0 = reinterpret_cast<const i32x4*>(idx)[2 * i], k1 = reinterpret_cast<const i32x4*>(idx)[2 * i + 1];
        float w[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { w[j] = k0[j] >= 0 ? src[k0[j]] : 0.f; w[4 + j] = k1[j] >= 0 ? src[k1[j]] : 0.f; }
        bf16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) { h[j] = (bf16)w[j]; l[j] = (bf16)(w[j] - (float)h[j]); }
        reinterpret_cast<bf16x8*>(hi)[i] = h;
        if (lo) reinterpret_cast<bf16x8*>(lo)[i] = l;
    }
}

static inline int grid_for(int64_t n_items) {
    const int64_t g = (n_items + 255) / 256;
    const int64_t cap = (int64_t)device_cus() * 16;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
static inline DropArgs drop_args(float p_drop, uint64_t seed, const void* seed_dev, uint32_t layer) {
    DropArgs d;
    d.seed = seed; d.seed_ptr = reinterpret_cast<const uint64_t*>(seed_dev); d.layer = layer;
    // keep iff word >= thresh: P(keep) = 1 - thresh / 2^32
    const double th = (double)p_drop * 4294967296.0;
    d.thresh = p_drop > 0.f ? (uint32_t)(th > 4294967295.0 ? 4294967295.0 : th) : 0u;
    d.inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    return d;
}

}  // namespace hsidm

using namespace hsidm;

#define HSIDM_BY_PREC(prec, KERNEL, ...)                                                                      \
    if ((prec) == HSIDM_BF16) hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<bf16>), __VA_ARGS__);                 \
    else if ((prec) == HSIDM_F32X3) hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<float>), __VA_ARGS__);          \
    else return HSIDM_E_BADARG;

static bool cat_ok(const void* s0, const void* s1, int C0, int C1) {
    return s0 && C0 > 0 && !(C0 & 7) && C1 >= 0 && !(C1 & 7) && (C1 == 0 || s1) && C0 + C1 <= 2048;
}

extern "C" int hsidm_gn_act_apply(int prec, const void* src0, const void* src1, int C0, int C1, const float* gn_ab, int transform,
                                  int B, int HW, float p_drop, uint64_t seed, const void* seed_dev, uint32_t layer, void* out, void* stream) {
    if (!cat_ok(src0, src1, C0, C1) || !gn_ab || !out || B <= 0 || HW <= 0 || p_drop < 0.f || p_drop >= 1.f) return HSIDM_E_BADARG;
    if (transform != HSIDM_XF_AFFINE && transform != HSIDM_XF_AFFINE_SILU) return HSIDM_E_BADARG;
    const int C = C0 + C1;
    const int64_t nvec = (int64_t)B * HW * (C >> 3);
    const DropArgs dr = drop_args(p_drop, seed, seed_dev, layer);
    hipStream_t s = (hipStream_t)stream;
#define ARGS(T) dim3(grid_for(nvec)), dim3(256), 0, s, (const T*)src0, (const T*)src1, C0, C1, (const float2*)gn_ab, \
                (int)(transform == HSIDM_XF_AFFINE_SILU), HW, nvec, dr, (T*)out
    if (prec == HSIDM_BF16) hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_act_apply_kernel<bf16>), ARGS(bf16));
    else if (prec == HSIDM_F32X3) hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_act_apply_kernel<float>), ARGS(float));
    else return HSIDM_E_BADARG;
#undef ARGS
    return (int)hipGetLastError();
}

extern "C" int hsidm_gn_act_bwd_workspace_floats(int B, int C, int groups, int nsplit) {
    if (B <= 0 || C <= 0 || groups <= 0 || nsplit <= 0) return HSIDM_E_BADARG;
    return 2 * (B * nsplit * C + B * C + B * groups);
}

extern "C" int hsidm_gn_act_bwd(int prec, const void* da, const void* src0, const void* src1, int C0, int C1, const float* gn_ab,
                                const float* gamma, int groups, int transform, int B, int HW, float p_drop, uint64_t seed,
                                const void* seed_dev, uint32_t layer, int nsplit, float* workspace, float* dgamma, float* dbeta, const void* add,
                                void* dx0, void* dx1, void* stream) {
    if (!cat_ok(src0, src1, C0, C1) || !da || !gn_ab || !gamma || !workspace || !dgamma || !dbeta || !dx0 || B <= 0 || HW <= 0 ||
        groups <= 0 || nsplit <= 0 || p_drop < 0.f || p_drop >= 1.f) return HSIDM_E_BADARG;
    if (transform != HSIDM_XF_AFFINE && transform != HSIDM_XF_AFFINE_SILU) return HSIDM_E_BADARG;
    const int C = C0 + C1;
    if (C % groups || (C1 > 0 && !dx1)) return HSIDM_E_BADARG;
    const int silu_on = transform == HSIDM_XF_AFFINE_SILU;
    const DropArgs dr = drop_args(p_drop, seed, seed_dev, layer);
    const float2* ab = (const float2*)gn_ab;
    const float2* mr = (const float2*)(gn_ab + (size_t)4 * B * C);          // (mean, rstd) part of the table (hsidm_gn_finalize)
    float2* part = (float2*)workspace;
    float2* S = part + (size_t)B * nsplit * C;
    float2* gm = S + (size_t)B * C;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nvec = (int64_t)B * HW * (C >> 3);
#define RARGS(T) dim3(nsplit, B), dim3(256), 0, s, (const T*)da, (const T*)src0, (const T*)src1, C0, C1, ab, mr, groups, silu_on, HW, nsplit, dr, part
#define AARGS(T) dim3(grid_for(nvec)), dim3(256), 0, s, (const T*)da, (const T*)src0, (const T*)src1, C0, C1, ab, mr, (const float2*)gm, gamma, \
                 groups, silu_on, HW, nvec, dr, (const T*)add, (T*)dx0, (T*)dx1, (const float2*)S, B, dgamma, dbeta
    if (prec == HSIDM_BF16) hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_act_bwd_reduce_kernel<bf16>), RARGS(bf16));
    else if (prec == HSIDM_F32X3) hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_act_bwd_reduce_kernel<float>), RARGS(float));
    else return HSIDM_E_BADARG;
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B), dim3(256), (size_t)(2 * C + 512) * sizeof(float), s, (const float2*)part, nsplit, C, HW,
                       groups, gamma, S, gm);
    if (prec == HSIDM_BF16) hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_act_bwd_apply_kernel<bf16>), AARGS(bf16));
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_act_bwd_apply_kernel<float>), AARGS(float));
#undef RARGS
#undef AARGS
    return (int)hipGetLastError();
}

extern "C" int hsidm_zero_insert2(int prec, const void* in, void* out, int B, int Hi, int Wi, int Ho, int Wo, int C, void* stream) {
    if (!in || !out || B <= 0 || C <= 0 || (C & 7) || Hi != (Ho + 1) / 2 || Wi != (Wo + 1) / 2) return HSIDM_E_BADARG;
    const int64_t nvec = (int64_t)B * Ho * Wo * (C >> 3);
    hipStream_t s = (hipStream_t)stream;
    if (prec == HSIDM_BF16) hipLaunchKernelGGL(HIP_KERNEL_NAME(zero_insert2_kernel<bf16>), dim3(grid_for(nvec)), dim3(256), 0, s, (const bf16*)in, (bf16*)out, Hi, Wi, Ho, Wo, C, nvec);
    else if (prec == HSIDM_F32X3) hipLaunchKernelGGL(HIP_KERNEL_NAME(zero_insert2_kernel<float>), dim3(grid_for(nvec)), dim3(256), 0, s, (const float*)in, (float*)out, Hi, Wi, Ho, Wo, C, nvec);
    else return HSIDM_E_BADARG;
    return (int)hipGetLastError();
}

extern "C" int hsidm_sum2x2(int prec, const void* in, void* out, int B, int H, int W, int C, void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return HSIDM_E_BADARG;
    const int64_t nvec = (int64_t)B * H * W * (C >> 3);
    hipStream_t s = (hipStream_t)stream;
    if (prec == HSIDM_BF16) hipLaunchKernelGGL(HIP_KERNEL_NAME(sum2x2_kernel<bf16>), dim3(grid_for(nvec)), dim3(256), 0, s, (const bf16*)in, (bf16*)out, H, W, C, nvec);
    else if (prec == HSIDM_F32X3) hipLaunchKernelGGL(HIP_KERNEL_NAME(sum2x2_kernel<float>), dim3(grid_for(nvec)), dim3(256), 0, s, (const float*)in, (float*)out, H, W, C, nvec);
    else return HSIDM_E_BADARG;
    return (int)hipGetLastError();
}

extern "C" int hsidm_add(int prec, const void* a, const void* b, void* out, int64_t n, void* stream) {
    if (!a || !b || !out || n <= 0 || (n & 7)) return HSIDM_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (prec == HSIDM_BF16) hipLaunchKernelGGL(HIP_KERNEL_NAME(add_kernel<bf16>), dim3(grid_for(n >> 3)), dim3(256), 0, s, (const bf16*)a, (const bf16*)b, (bf16*)out, n >> 3);
    else if (prec == HSIDM_F32X3) hipLaunchKernelGGL(HIP_KERNEL_NAME(add_kernel<float>), dim3(grid_for(n >> 3)), dim3(256), 0, s, (const float*)a, (const float*)b, (float*)out, n >> 3);
    else return HSIDM_E_BADARG;
    return (int)hipGetLastError();
}

extern "C" int hsidm_colsum(const float* part, int nsplit, int B, int C, int Cout, float* out_bc, float* out_c, void* stream) {
    if (!part || nsplit <= 0 || B <= 0 || C <= 0 || Cout <= 0 || Cout > C || (!out_bc && !out_c)) return HSIDM_E_BADARG;
    hipLaunchKernelGGL(colsum_kernel, dim3((Cout + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const float2*)part, nsplit, B, C, Cout, out_bc, out_c);
    return (int)hipGetLastError();
}

extern "C" int hsidm_loss_grad(int prec, const float* noise, const float* eps, int B, int Cimg, int HW, int Cpad, int kind, float scale,
                               void* out, void* stream) {
    if (!noise || !eps || !out || B <= 0 || Cimg <= 0 || HW <= 0 || Cpad < Cimg || (Cpad & 7) || (kind != HSIDM_LOSS_L1 && kind != HSIDM_LOSS_L2))
        return HSIDM_E_BADARG;
    const int64_t npix = (int64_t)B * HW;
    hipStream_t s = (hipStream_t)stream;
    if (prec == HSIDM_BF16) hipLaunchKernelGGL(HIP_KERNEL_NAME(loss_grad_kernel<bf16>), dim3(grid_for(npix)), dim3(256), 0, s, noise, eps, Cimg, HW, Cpad, kind, scale, npix, (bf16*)out);
    else if (prec == HSIDM_F32X3) hipLaunchKernelGGL(HIP_KERNEL_NAME(loss_grad_kernel<float>), dim3(grid_for(npix)), dim3(256), 0, s, noise, eps, Cimg, HW, Cpad, kind, scale, npix, (float*)out);
    else return HSIDM_E_BADARG;
    return (int)hipGetLastError();
}

extern "C" int hsidm_noise_film_bwd_workspace_floats(int B, int dim, int F) {
    if (B <= 0 || dim <= 0 || F <= 0) return HSIDM_E_BADARG;
    return ((F + 255) / 256) * B * dim + B * (2 * dim + 12 * dim);
}

extern "C" int hsidm_noise_film_bwd(const float* gamma, const float* t_emb, const float* dfilm, int B, int dim, const float* w1,
                                    const float* b1, const float* w2, const float* wf, int F, float* dw1, float* db1, float* dw2,
                                    float* db2, float* dwf, float* dbf, float* workspace, void* stream) {
    if (!gamma || !t_emb || !dfilm || !w1 || !b1 || !w2 || !wf || !dw1 || !db1 || !dw2 || !db2 || !dwf || !dbf || !workspace) return HSIDM_E_BADARG;
    if (B <= 0 || dim <= 0 || (dim & 1) || dim > 1024 || F <= 0) return HSIDM_E_BADARG;
    const int nblk = (F + 255) / 256;
    const size_t lds = (size_t)256 * (dim + 1) * sizeof(float);
    if (lds > 160 * 1024) return HSIDM_E_UNSUPPORTED;
    static PerDeviceOnce once;
    if (int rc = raise_lds_cap(once, &film_bwd_kernel, lds)) return rc;
    hipStream_t s = (hipStream_t)stream;
    float* dt_part = workspace;
    float* work = workspace + (size_t)nblk * B * dim;
    hipLaunchKernelGGL(film_bwd_kernel, dim3(nblk), dim3(256), lds, s, dfilm, t_emb, wf, B, dim, F, dwf, dbf, dt_part);
    hipLaunchKernelGGL(noise_mlp_bwd_kernel, dim3(1), dim3(256), 0, s, gamma, (const float*)dt_part, nblk, B, dim, w1, b1, w2, dw1, db1, dw2, db2, work);
    return (int)hipGetLastError();
}

extern "C" int hsidm_gather_pack(const float* src, const int32_t* idx, int64_t n, void* out_hi, void* out_lo, void* stream) {
    if (!src || !idx || !out_hi || n <= 0 || (n & 7)) return HSIDM_E_BADARG;
    hipLaunchKernelGGL(gather_pack_kernel, dim3(grid_for(n >> 3)), dim3(256), 0, (hipStream_t)stream, src, idx, n >> 3, (bf16*)out_hi, (bf16*)out_lo);
    return (int)hipGetLastError();
}

extern "C" int hsidm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                               int step, float grad_scale, const float* coef_dev, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || (step <= 0 && !coef_dev)) return HSIDM_E_BADARG;
    const int st = step > 0 ? step : 1;
    const double bc1 = 1.0 - pow((double)beta1, st), bc2 = 1.0 - pow((double)beta2, st);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, beta1, beta2,
                       (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), eps, grad_scale, coef_dev);
    return (int)hipGetLastError();
}
