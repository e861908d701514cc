/* hsidm.h - C ABI of libhsidm.so, the MI355X (gfx950) kernels behind the HSI-DMGASR denoising path.
 *
 * The reference (handsomewzy/HSI-DMGASR) is pure PyTorch: its "FFI" for this path is the set of
 * torch.nn ops its nn.Modules call.  Each entry point below replaces one such op sequence; the
 * reference file:line it stands in for is cited per function.  The Python modules in
 * hsi-dmgasr_amd/ (same class names, constructor arguments and state_dict keys as the reference)
 * bind these symbols with ctypes - see INTEGRATION.md for the binding a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless named host_*;
 *   - the caller owns all memory (activations, packed weights, workspaces); nothing is allocated,
 *     freed or synchronised inside; every call only enqueues kernels on `stream` (a hipStream_t
 *     passed as void*), so calls can be captured into a hipGraph;
 *   - return value: 0 on success, otherwise a negative HSIDM_E_* code or a positive hipError_t;
 *   - internal activation tensors are NHWC ("pixel-major, channels contiguous") in the storage
 *     type of the precision mode; module boundaries of the reference (NCHW fp32) are converted by
 *     hsidm_nchw_to_nhwc / hsidm_nhwc_to_nchw.
 */
#ifndef HSIDM_H
#define HSIDM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* precision modes */
#define HSIDM_BF16  0   /* bf16 storage + bf16 MFMA operands, fp32 accumulate / statistics / softmax   */
#define HSIDM_F32X3 1   /* fp32 storage; operands split bf16 hi+lo, 3 MFMA passes (fp32-grade parity) */
#define HSIDM_F16   2   /* fp16 storage + fp16 MFMA operands (11-bit significands, |x| <= 65504: stores saturate), fp32
                           accumulate / statistics / softmax; same kernels, layouts and rate as HSIDM_BF16.  A convolution whose
                           descriptor carries w_v2_lo / w_lo runs a second MFMA pass on the low halves of its weights */
#define HSIDM_F32H  3   /* hsidm_conv2d only: fp32 storage like HSIDM_F32X3 (same tensors in and out, fp32 GroupNorm pairs), but ONE
                           fp16 activation operand - the staged value, rounded once behind the fp32 transform - against fp16 hi + lo
                           weights (w_v2 AND w_v2_lo required, fp16; w_hi / w_lo are not read): two MFMA passes instead of three.  Forms:
                           the persistent 3x3 kernel on 8x16 tiles (output maps >= 16 wide; stride 1 | 2, folded upsample) and the
                           plain 1x1 GEMM; every other shape returns HSIDM_E_UNSUPPORTED and runs with HSIDM_F32X3 weights instead.
                           The kernel set of a reverse chain's steps 2 .. 8 in the host's "fp16" policy (precision.py) */

/* error codes */
#define HSIDM_OK             0
#define HSIDM_E_BADARG      (-1)
#define HSIDM_E_UNSUPPORTED (-2)

/* input transforms fused into the convolution's operand staging */
#define HSIDM_XF_NONE        0
#define HSIDM_XF_AFFINE      1   /* y = scale[b,c]*x + shift[b,c]            (GroupNorm)            */
#define HSIDM_XF_AFFINE_SILU 2   /* y = silu(scale*x + shift)                (GroupNorm + Swish)    */

#define HSIDM_ACT_NONE  0
#define HSIDM_ACT_LEAKY 1        /* LeakyReLU(0.01)                                                  */

/* ABI version of this header: bumped whenever a struct below grows, a prototype changes or an entry point is added (2: hsidm_conv_desc
 * gained w_v2_ls / w_v2_li; 3: hsidm_conv1x1_pair, and the bn == 64 projection steps of w_v2 are padded to three).  hsidm_version() returns the version the LIBRARY was built against; a caller built against another header must refuse
 * to run (a shorter hsidm_conv_desc would be read 16 bytes past its end) - the ctypes binding does (hsi-dmgasr_amd/_lib.py). */
#define HSIDM_ABI_VERSION 3
int hsidm_version(void);
const char* hsidm_error_string(int code);
/* Diagnostic dispatch switches for A/B measurements and tests: "NO_V3", "V2_BN256", "ATTENTION_V1", "NO_XCD_MAP", "1X1_V1",
 * "V2_ABL", "SK_MULT", "NO_SPLIT_K", "NO_SPARSE_LO", "NO_FUSED_PROJ".  Initialised once from the environment (HSIDM_<name>) when the library is loaded; the launch path never
 * reads the environment.  Returns the previous value (>= 0) or HSIDM_E_BADARG for an unknown name. */
int hsidm_debug_switch(const char* name, int value);
/* Current value of a switch (>= 0) without changing it, or HSIDM_E_BADARG. */
int hsidm_debug_query(const char* name);

/* ---- convolution ------------------------------------------------------------------------------
 * One K-phase of hsidm_conv2d: an input tensor (optionally the channel concat of two tensors,
 * replacing torch.cat((x, feats.pop()), 1), unet.py:259) with an optional fused GroupNorm(+Swish).
 */
typedef struct hsidm_conv_phase {
    const void*  src0;        /* NHWC [B][Hin][Win][C0]                                             */
    const void*  src1;        /* NHWC [B][Hin][Win][C1] or NULL                                     */
    const float* gn_ab;       /* GroupNorm table of hsidm_gn_finalize: [B][C0+C1][2] fp32 (scale, shift)
                                 followed by its two fp16x2 parts (see hsidm_gn_finalize); or NULL      */
    int32_t C0, C1;           /* multiples of 8                                                     */
    int32_t transform;        /* HSIDM_XF_*                                                         */
    int32_t ntaps;            /* 9 (3x3) or 1 (1x1)                                                 */
} hsidm_conv_phase;

/* Replaces, in one launch:
 *   Block.forward            unet.py:83-91   GroupNorm -> Swish -> Conv2d 3x3          (phase 0)
 *   FeatureWiseAffine        unet.py:42-50   h + Linear(t)[b,c]                         (film)
 *   ResnetBlock tail         unet.py:105-111 block2(h) + res_conv(x) | + x             (phase 1 / res)
 *   Upsample / Downsample    unet.py:58-74   nearest x2 folded into addressing / stride 2
 *   SelfAttention qkv / out  unet.py:128,141 1x1 convs (+ residual)
 *   common.ResBlock / ResAttentionBlock convs common.py:163-182,250-271 (LeakyReLU, 0.1*res + x)
 * out = res_scale * act(conv + bias[c] + film[b,c]) + res.
 */
/* ups: 0 none; HSIDM_UPS_ADDRESS: the conv reads in[y>>1][x>>1] (any kernel path);
 * HSIDM_UPS_FOLDED (HSIDM_BF16, bn = 128, no input transform): w_v2 holds the four parity-folded 2x2 kernels
 *   [parity 2*py+px][chunk][tap 2*ty+tx][Cout_pad/32][4][64][8],
 *   W_par[ty][tx] = sum of the 3x3 taps (dy, dx) with (py+dy-1)>>1 == py+ty-1 and (px+dx-1)>>1 == px+tx-1,
 * because output pixel (2y+py, 2x+px) of conv3x3(nearest_x2(in)) only sees in[y-1+py+ty][x-1+px+tx]: 4/9 of the
 * multiplications of HSIDM_UPS_ADDRESS.  Statistics entries: one per (input tile, parity, wave row).
 *
 * stride = 2 with w_v2 != NULL (HSIDM_BF16, bn 64|128, even Hin/Win, no input transform): w_v2 holds
 *   [plane 2*ry+rx][chunk][tap 2*ty+tx][Cout_pad/32][4][64][8], plane (ry, rx) = in[2i+ry][2j+rx],
 *   W_plane[ty][tx] = W[dy][dx] with dy = 2*ty if ry else (ty == 1 ? 1 : none), dx likewise (absent taps are zero);
 * without w_v2 the stride is folded into the addressing of the generic kernel. */
#define HSIDM_UPS_ADDRESS 1
#define HSIDM_UPS_FOLDED  2
typedef struct hsidm_conv_desc {
    hsidm_conv_phase ph[2];
    int32_t nphase;           /* 1, or 2 = phase 1 is a fused 1x1 projection of a second input      */
    const void*  w_hi;        /* packed bf16 [step][Cout_pad][BK], step = (phase, chunk, tap)       */
    const void*  w_lo;        /* low halves (HSIDM_F32X3: required; HSIDM_F16: optional second weight pass) */
    const void*  w_v2;        /* optional (HSIDM_BF16): the same weights in the register-streaming order
                                 [step][Cout_pad/32][4][64 lanes][8]; enables the persistent kernels
                                 (csrc/conv_v2.h, conv1x1_g.hip).  1x1: K padded to a multiple of 128.
                                 3x3 with C0 + C1 == 8: tap-major GEMM layout, k = 8*tap + c padded to
                                 128 (2 steps); such a descriptor must satisfy: stride 1, no ups, no
                                 transform / act / film, Cout % bn == 0, Hout*Wout % 64 == 0, >= 128,
                                 Wout a power of two                                                   */
    const float* bias;        /* [Cout] or NULL                                                     */
    const float* film;        /* [B][film_stride], pre-offset to this layer's columns, or NULL      */
    int32_t film_stride;
    const void*  res;         /* residual in the layout/type of out, or NULL                        */
    float        res_scale;
    void*        out;         /* NHWC [B][Hout][Wout][Cout]; NCHW fp32 when out_nchw                */
    float*       stats;       /* optional [B][hsidm_conv_stats_nsplit(d)][Cout][2]: per-(image, tile part,
                                 channel) sum and sum of squares of `out` as stored, every entry
                                 written exactly once (no atomics, no zeroing); feeds hsidm_gn_finalize */
    int32_t B, Hin, Win, Hout, Wout, Cout;
    int32_t ksize;            /* 3 or 1 (phase 0)                                                   */
    int32_t stride;           /* 1 or 2                                                             */
    int32_t ups;              /* HSIDM_UPS_*: nearest x2 upsample folded in (Hout = 2*Hin)          */
    int32_t act;              /* HSIDM_ACT_*                                                        */
    int32_t out_nchw;         /* 1: write NCHW fp32 (network outputs)                               */
    int32_t prec;             /* HSIDM_BF16 | HSIDM_F32X3 | HSIDM_F16 | HSIDM_F32H                  */
    int32_t bn;               /* cout slice the weights were packed for: 32, 64 or 128             */
    void*   workspace;        /* optional scratch of hsidm_conv_workspace_bytes(d) bytes: enables the split-K form
                                 (csrc/conv_sk.hip) for launches with few pixel tiles and a long contraction -
                                 the 8x8 / 16x16 levels at small batches; NULL: the persistent kernels only.
                                 With nphase == 2 and w_v2 (its steps: 9 per chunk of phase 0, then 1 per chunk of
                                 phase 1) two w_v2 kernels apply: the split form (bn == 128) - ask hsidm_conv_workspace_bytes
                                 first - and, for 16-bit descriptors with Cout == 64 (bn == 64), GN + SiLU on phase 0, whole
                                 16x16 tiles and no residual, the persistent kernel's projection forms (csrc/conv_v3.hip, PROJ:
                                 hsidm_conv_kernel_id(d) & 15 == 4; the ResnetBlock's res_conv of reference unet.py:102-103,110
                                 inside block2's launch): one-pass weights (no w_v2_lo; HSIDM_BF16 | HSIDM_F16) with
                                 at most 192 projection channels.  For bn == 64 (one-pass layers)
                                 the phase-1 steps of w_v2 carry the projection's weights times log2(e)
                                 (the kernel removes the factor its SiLU staging leaves on every product) and are padded with
                                 zero steps to THREE per item: w_v2 holds 9 * chunks(phase 0) + 3 steps (the one-pass form pulls
                                 three steps per item through its weight ring, whatever the projection's width; w_hi / w_lo, the
                                 LDS-tiled kernel's order, stay unscaled and unpadded).  Neither: launch the projection separately */
    int64_t workspace_bytes;
    const void*  w_v2_lo;     /* required for HSIDM_F32H (fp16) and for the persistent forms of HSIDM_F32X3 (bf16); optional (HSIDM_F16 with w_v2): fp16(W - fp16(W)) in the layout of w_v2 - the weights then carry
                                 ~18-19 significant bits (the low halves are fp16 subnormals for |w| < 0.125: absolute granularity 6e-8) for twice the matrix instructions (the weight rounding
                                 is the one systematic error of a 16-bit mode; the pass is nearly free on layers bound by the
                                 staging transform).  Not taken by the split-K form and the 256-cout items                        */
    const void*  w_v2_ls;     /* optional (with w_v2_lo, plain 3x3): the low halves again, 2:4 structured-sparse - of every four consecutive
                                 input channels (per cout and tap) the two of larger magnitude - as the A operand of
                                 v_smfmac_f32_16x16x64_f16: fp16 [step][Cout_pad/32][half 2][lane 64][8]; lane = 16 * kgroup + cout % 16 of the
                                 16-cout half holds channels 16 * kgroup .. + 15 of the step's 64: stored slots 2m, 2m + 1 = the kept values of
                                 channels 16 * kgroup + 4m .. + 3 (measured semantics: tools/ubench/smfmac_probe.hip, its output:
                                 profiles/r05_ubench/smfmac_probe.txt: stored slot sa of A lane group ga with index field v pairs with K = 16 ga + 4 (sa / 2) + v).
                                 The instruction's B side (the activations; built inside the kernels, not part of this ABI) holds the
                                 step's 64 channels as TWO DENSE 16x16x32 FRAGMENTS SIDE BY SIDE: lane = 16 * g + pixel % 16, slots 0-7 =
                                 channels 8 g .. + 7, slots 8-15 = channels 32 + 8 g .. + 7 (probe: A (ga 1, sa 0, v 0) = K 16 meets B
                                 (g 2, slot 0); A (ga 2, sa 5, v 2) = K 42 meets B (g 1, slot 10)) - which is why the kernels form it by
                                 concatenating the tap's q = 0 and q = 1 activation fragments (conv_v3.hip / conv_v2.h: `bb`).
                                 tests/test_gpu_anchor.py::test_sparse_second_weight_pass_removes_the_weight_bias gives every input channel
                                 a distinct mean, so a wrong slot order, index word or k-group map shows as a per-cout mean shift.
                                 A kernel that takes it runs the second weight pass at ~0.6 of its dense
                                 cost on 80 % of the low halves' energy                                                                     */
    const void*  w_v2_li;     /* with w_v2_ls: int32 [step][Cout_pad/32][lane 64]: bits [2s+1 : 2s] of the low (high) 16 bits = position, within
                                 its group of four channels, of stored slot s of the first (second) 16-cout half                         */
} hsidm_conv_desc;

int hsidm_conv2d(const hsidm_conv_desc* d, void* stream);
/* Scratch bytes with which hsidm_conv2d(d) would take its split-K form (0: it would not, whatever the workspace). */
int64_t hsidm_conv_workspace_bytes(const hsidm_conv_desc* d);
/* Number of partial entries per image that hsidm_conv2d(d) writes into d->stats (>0), or an error code. */
int hsidm_conv_stats_nsplit(const hsidm_conv_desc* d);
/* Which kernel hsidm_conv2d(d) dispatches to, for measurement tools: bits 0-3 = 0 LDS-tiled (conv_igemm), 1 persistent 3x3
 * (conv_v2), 3 LDS-staged 1x1 GEMM (conv1x1_g), 4 256-pixel 3x3 (conv_v3), 5 split-K 3x3 (conv_sk, only with d->workspace); bits 4-5 = tile (0: 8x16, 1: 8x8 of two images, 2: 8x8 of one image); bits 8.. = couts per work item (256: the 8-wave form of a 128-packed GroupNorm+SiLU conv).
 * <0 on error. */
int hsidm_conv_kernel_id(const hsidm_conv_desc* d);
/* K-chunk (input channels per packed step) of a precision mode: 64 for BF16 and F16, 32 for F32X3. */
int hsidm_conv_bk(int prec);

/* ---- GroupNorm statistics (nn.GroupNorm inside Block / SelfAttention, unet.py:84,121) ----------
 * partial: per-(image, split, channel) sum and sum of squares of a (concatenated) NHWC tensor;
 * finalize: -> per-(image, channel) (scale, shift) = (rstd*gamma, beta - mean*rstd*gamma).
 * `part` is [B][nsplit][C][2] floats; hsidm_conv2d's `stats` output has this layout, so a tensor produced by
 * a convolution needs no statistics pass.  finalize takes the two halves of a channel concat separately
 * (part1 may be NULL): GroupNorm groups may straddle the seam (e.g. 192 = 128 + 64 channels, 6 per group).
 * gn_ab (16 * B * C + 8 * B * groups bytes): [B][C][2] fp32 (scale, shift), then [B][C] fp16x2 copies of the same pairs, then
 * [B][C] fp16x2 copies of log2(e) * (scale, shift) (what the bf16 kernels read: the affine-only ones the first copy, the
 * GroupNorm+SiLU ones the pre-scaled copy, whose factor they remove from their fp32 accumulators), then [B][groups][2] fp32
 * (mean, rstd), which only the backward pass reads (hsidm_gn_act_bwd).
 */
int hsidm_gn_partial(int prec, const void* src0, const void* src1, int C0, int C1, int B, int HW,
                     int nsplit, float* part, void* stream);
int hsidm_gn_finalize(const float* part0, int nsplit0, int C0, const float* part1, int nsplit1, int C1,
                      int B, int HW, int groups, const float* gamma, const float* beta, float eps,
                      float* gn_ab, void* stream);

/* ---- noise-level embedding + all FiLM projections of one UNet call ---------------------------------
 * PositionalEncoding + noise_level_mlp (unet.py:23-31,182-187) and the Linear of every
 * FeatureWiseAffine (unet.py:38-49) for all ResnetBlocks at once: film[b][f] = Wf[f,:]·t_b + bf[f].
 * gamma: [B] noise levels, or NULL with (level_table, t_ptr): gamma = level_table[*t_ptr + 1]
 * (diffusion.py:154-155) read on the device so that a captured graph can be replayed per step.
 * t_emb (optional, [B][dim]): use this embedding instead of running the MLP (ResnetBlock.forward(x, time_emb)
 * called on its own); t_out (optional, [B][dim]): also return the embedding.
 */
int hsidm_noise_film(const float* gamma, const float* level_table, const int32_t* t_ptr, const float* t_emb,
                     int B, int dim, const float* w1, const float* b1, const float* w2, const float* b2,
                     const float* wf, const float* bf, int F, float* film, float* t_out, void* stream);

/* FeatureWiseAffine with use_affine_level=True (unet.py:44-47; not reachable from the reference's UNet, which never passes it):
 * out = (1 + gamma[b][c]) * x + beta[b][c] with gamma_beta [B][2C] = (gamma | beta), the projection hsidm_noise_film emits for a
 * Linear(dim, 2C); NHWC tensors of the mode's storage type. */
int hsidm_film_affine(int prec, const void* x, const float* gamma_beta, void* out, int B, int HW, int C, void* stream);

/* ---- self-attention core (unet.py:130-140): softmax(q k^T / sqrt(C)) v, single head ---------------
 * qkv: NHWC [B][N][3C] (q | k | v channel thirds, unet.py:129), out: [B][N][C].  N <= 1024.
 */
int hsidm_attention(int prec, const void* qkv, void* out, int B, int N, int C, void* stream);

/* ---- layout conversion at the reference's module boundaries ----------------------------------------
 * nchw_to_nhwc: out[e][p][c] for c < Cpad (multiple of 8): channel c comes from fp32 NCHW planes
 * src0 + off0[e] + c*HW (c < C0), src1 + off1[e] + (c-C0)*HW (C0 <= c < C0+C1), else 0.
 * off0/off1: per-output-image element offsets (device int64); this one call implements
 * torch.cat([cond, x], 1) (diffusion.py:158) and the spectral-group slicing x[:, s:e]
 * (AE.py:316-321) with all groups stacked on the batch axis.
 */
int hsidm_nchw_to_nhwc(int prec, const float* src0, const int64_t* off0, int C0,
                       const float* src1, const int64_t* off1, int C1,
                       void* out, int Bout, int HW, int Cpad, void* stream);
int hsidm_nhwc_to_nchw(int prec, const void* src, float* out, int B, int HW, int C, void* stream);

/* ---- reverse-diffusion update (diffusion.py:142-175) -------------------------------------------------
 * x <- c1*clamp(a*x - b*eps, -1, 1) + c2*x + [t>0] * z * exp(0.5*logvar), t = *t_ptr.
 * coef: [T][5] = (sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod, posterior_mean_coef1,
 * posterior_mean_coef2, posterior_log_variance_clipped).  z = noise[(T-1-t)*noise_stride + i] when
 * noise != NULL (stored noise; noise_stride = n for a [T-1][n] table, 0 for one buffer refilled by the
 * caller before every step), otherwise Philox4x32-10(seed, stream = t) + Box-Muller.
 * snap (optional): x is also copied to snap + slot*n when t % inter == 0 (slot counts snapshots,
 * diffusion.py:196-197).
 */
int hsidm_p_sample_update(float* x, const float* eps, const float* coef, const int32_t* t_ptr, int T,
                          const float* noise, int64_t noise_stride, uint64_t seed, int64_t n,
                          float* snap, int32_t inter, void* stream);
/* t <- t - 1 (one thread); separate launch so that every kernel of a step reads the same t.
 * wrap_T > 0: restart at wrap_T - 1 after t = 0 (benchmark loops longer than one chain). */
int hsidm_step_advance(int32_t* t_ptr, int32_t wrap_T, void* stream);
/* out[i] = N(0,1) from Philox4x32-10(seed, stream), i < n  (x_T: stream = T). */
int hsidm_philox_normal(float* out, int64_t n, uint64_t seed, uint32_t stream_id, void* stream);

/* ---- forward process / training objective (diffusion.py:213-250) -----------------------------------
 * q_sample (diffusion.py:213-220): out = gamma[b]*x0 + sqrt(1-gamma[b]^2)*noise, fp32 [B][per_sample]. */
int hsidm_q_sample(const float* x0, const float* noise, const float* gamma, float* out, int B,
                   int64_t per_sample, void* stream);
/* loss_func of set_loss (diffusion.py:85-91): out[0] = sum |a-b| (HSIDM_LOSS_L1) or sum (a-b)^2
 * (HSIDM_LOSS_L2) over n fp32 elements; deterministic two-stage reduction (fp64 between stages).
 * workspace: hsidm_loss_workspace_bytes() bytes of device memory owned by the caller. */
#define HSIDM_LOSS_L1 0
#define HSIDM_LOSS_L2 1
int hsidm_loss_workspace_bytes(void);
int hsidm_loss_sum(const float* a, const float* b, int64_t n, int kind, void* workspace, float* out, void* stream);

/* ---- group-autoencoder pieces (AE.py / common.py) -----------------------------------------------------
 * The body of the spectral ResAttentionBlock (common.py:250-271 with kernel_size 1, as SSB builds it, AE.py:102-109): two 1x1 convolutions
 * of 64 channels with an activation between them in ONE launch,
 *     out[m][:] = W2 act(W1 x[m][:] + bias1) + bias2,      x, out: NHWC [M = B * HW][64] in the storage type of `prec`,
 * the intermediate tensor never leaves the chip.  prec: HSIDM_F16 (fp16 hi + lo weights) | HSIDM_F32X3 (fp32 storage, bf16 hi + lo
 * operands: the intermediate is split exactly as a stored fp32 tensor would be when staged).  w_pair / w_pair_lo: the two weight
 * matrices as the two steps of the register-streaming order of hsidm_conv_desc.w_v2 - [2][64/32][kk 4][lane 64][8], step 0 = W1,
 * step 1 = W2 - high and low halves.  act: HSIDM_ACT_NONE | HSIDM_ACT_LEAKY.  stats (optional): [B][HW / 64][64] (sum, sum of
 * squares) of `out` per 64-pixel group, the CALayer's global-average partials (hsidm_ca_vector's `part`, nsplit = HW / 64).
 * HW % 64 == 0.  (ABI version 3.) */
int hsidm_conv1x1_pair(int prec, const void* x, const void* w_pair, const void* w_pair_lo, const float* bias1, int act,
                       const float* bias2, void* out, void* stats, int64_t M, int HW, void* stream);

/* CALayer (common.py:231-247): ca[b][c] = sigmoid(W2 relu(W1 mean_b + b1) + b2), mean from `part`.
 */
int hsidm_ca_vector(const float* part, int nsplit, int B, int C, int HW, int R,
                    const float* w1, const float* b1, const float* w2, const float* b2,
                    float* ca, void* stream);
/* out = res_scale * r * ca[b][c] + skip (+ skip2)   (ResAttentionBlock tail common.py:267-271,
 * plus the SSPN skip AE.py:137-139 when skip2 != NULL); NHWC tensors of the mode's storage type. */
int hsidm_ca_apply(int prec, const void* r, const float* ca, const void* skip, const void* skip2,
                   float res_scale, void* out, int B, int HW, int C, void* stream);
/* Overlap-average of the decoded groups (AE.py:286-295): dec NCHW fp32 [B*G][n_subs][HW] ->
 * y NCHW fp32 [B][C][HW]; start[g] device int32. */
int hsidm_overlap_average(const float* dec, const int32_t* start, int G, int n_subs, int B, int C,
                          int HW, float* y, void* stream);

/* ---- quality indices of decoded cubes (eval_hsi.py:27-121; caller sr_gae.py:468-474) -----------------------
 * truth, pred: NCHW fp32 [P][C][HW].  out[p] = {MPSNR (dB, data_range), SAM (degrees), ERGAS (ratio = upsampling
 * factor), CC, RMSE}.  workspace: hsidm_hsi_metrics_workspace_bytes(P, C, HW) bytes of device memory.
 * Deterministic; the SAM cosine is clamped to [-1, 1] (the reference returns NaN when fp32 rounding exceeds 1). */
int hsidm_hsi_metrics_workspace_bytes(int P, int C, int HW);
int hsidm_hsi_metrics(const float* truth, const float* pred, int P, int C, int HW, float ratio, float data_range,
                      void* workspace, float* out, void* stream);

/* MSSIM (eval_hsi.py:124-135, the remaining index of quality_assessment :217-238): out[p] = mean over bands of the structural
 * similarity with skimage's defaults (7x7 uniform window, sample covariance, K1 0.01, K2 0.03, 3-pixel border cropped).
 * truth, pred NCHW fp32 [P][C][H][W], H, W >= 7.  workspace: hsidm_hsi_mssim_workspace_bytes(P, C, H, W) bytes.  Deterministic. */
int hsidm_hsi_mssim_workspace_bytes(int P, int C, int H, int W);
int hsidm_hsi_mssim(const float* truth, const float* pred, int P, int C, int H, int W, float data_range, void* workspace,
                    float* out, void* stream);

/* ---- patch preparation (HStest.py:37-45, HStrain.py:49-63, imsize.py) ------------------------------------
 * One axis of the MATLAB-compatible resize: dst[o][j][i] = sum_p weights[j][p] * src[o][indices[j][p]][i] for
 * src [outer][in_len][inner], dst [outer][out_len][inner] (fp32).  weights/indices [out_len][taps] are the tap
 * tables of imsize.py:35-60 (host-built, device-resident).  clamp01: clamp the result to [0, 1] (HStest.py:59-60). */
int hsidm_resample_axis(const float* src, float* dst, int64_t outer, int in_len, int out_len, int inner,
                        const float* weights, const int32_t* indices, int taps, int clamp01, void* stream);
/* out = (x - min) / (max - min) per cube, x [P][n] fp32; workspace: hsidm_minmax_workspace_bytes(P) bytes. */
int hsidm_minmax_workspace_bytes(int P);
int hsidm_minmax_normalize(const float* x, float* out, int P, int64_t n, void* workspace, void* stream);
/* The 8 training-set augmentations of utils.py:3-28 (caller HStrain.py:65) on the two spatial axes of src [outer][H][W]:
 * mode 0 identity, 1 flipud, 2 rot90 (counter-clockwise), 3 flipud(rot90), 4 rot180, 5 flipud(rot180), 6 rot270,
 * 7 flipud(rot270).  dst is [outer][H][W] for modes 0, 1, 4, 5 and [outer][W][H] for 2, 3, 6, 7; dst != src. */
int hsidm_augment(const float* src, float* dst, int64_t outer, int H, int W, int mode, void* stream);
/* Per-band mean / standard-deviation matching (eval_hsi.py:259-274, caller sr_gae.py:340):
 * out[p][c] = clip((x[p][c] - mean x) / std x * std guide + mean guide, 0, 1) for c < num_channels, 0 for the other
 * bands (as the reference leaves them).  guide [P][C][guide_HW], x and out [P][C][HW], NCHW fp32; population standard
 * deviation.  workspace: hsidm_color_correction_workspace_bytes(P, C) bytes. */
int hsidm_color_correction_workspace_bytes(int P, int C);
int hsidm_color_correction(const float* guide, int guide_HW, const float* x, float* out, int P, int C, int HW,
                           int num_channels, void* workspace, void* stream);

/* ---- training step: forward with materialised operands, backward, optimiser --------------------------------
 * The reference trains through autograd (model/model.py:49-59: l_pix = netG(data); l_pix.backward(); optG.step()) over
 * GaussianDiffusion.p_losses (diffusion.py:222-250).  The entry points below are the hand-written adjoints of the forward
 * kernels above; the host side (hsi-dmgasr_amd/training.py) chains them in reverse order of the forward pass.
 *
 * Training-mode operand of a Block's convolution (unet.py:83-88 with Dropout active, :100-101), materialised once and used by
 * the forward convolution (transform NONE) and by the weight gradient:
 *   out = dropout_p(act(scale[b,c] * x + shift[b,c])), x = cat(src0, src1), act = SiLU (HSIDM_XF_AFFINE_SILU) or identity
 *   (HSIDM_XF_AFFINE, the attention's GroupNorm, unet.py:127); (scale, shift) = the fp32 pairs of hsidm_gn_finalize.
 * Dropout mask: element e (flat NHWC index of out) keeps its value, scaled by 1/(1-p), iff word (e & 3) of
 * Philox4x32-10(key = seed, counter = (e >> 2, 0, layer, 0)) >= p * 2^32; p_drop = 0 disables it.  seed_dev != NULL: the key is
 * the uint64 at that device address instead of `seed` (a training step captured into a hipGraph is replayed with a new key per
 * iteration). */
int hsidm_gn_act_apply(int prec, const void* src0, const void* src1, int C0, int C1, const float* gn_ab, int transform,
                       int B, int HW, float p_drop, uint64_t seed, const void* seed_dev, uint32_t layer, void* out, void* stream);
/* Backward of hsidm_gn_act_apply including the GroupNorm statistics (nn.GroupNorm backward):
 *   dy = da * dropout' * act'(u);  dgamma[c] = sum dy * xhat;  dbeta[c] = sum dy;
 *   dx = rstd * (gamma * dy - mean_g(gamma * dy) - xhat * mean_g(gamma * dy * xhat))  (+ add [B][HW][C0+C1]: gradients that reach x
 *   along other paths, e.g. the residual projection), written to the two halves of the concat (dx1 NULL when C1 == 0).
 * gn_ab: the full table of hsidm_gn_finalize (its (mean, rstd) part is read); gamma [C0+C1]; da NHWC [B][HW][C0+C1].
 * Three launches: per-(image, split, channel) sums; group means; dx (whose first workgroup also writes the parameter gradients).  Deterministic.
 * workspace: hsidm_gn_act_bwd_workspace_floats(B, C0+C1, groups, nsplit) floats. */
int hsidm_gn_act_bwd_workspace_floats(int B, int C, int groups, int nsplit);
int hsidm_gn_act_bwd(int prec, const void* da, const void* src0, const void* src1, int C0, int C1, const float* gn_ab,
                     const float* gamma, int groups, int transform, int B, int HW, float p_drop, uint64_t seed, const void* seed_dev,
                     uint32_t layer, int nsplit, float* workspace, float* dgamma, float* dbeta, const void* add, void* dx0, void* dx1,
                     void* stream);
/* Weight gradient of hsidm_conv2d's convolution: dw[co][ci][ky][kx] (fp32, PyTorch layout [Cout_w][Cin_w][k][k]) =
 *   sum_{b,y,x} dy[b][y][x][co] * a[b][s*y+ky-1][s*x+kx-1][ci], a = cat(a0, a1) NHWC [B][Hin][Win][C0+C1] (the materialised operand),
 *   dy NHWC [B][Hout][Wout][Cout]; ksize 3 (pad 1) or 1; stride 1 | 2; ups: a is read through the nearest x2 upsample
 *   (unet.py:64-65).  Cout_w <= Cout and Cin_w <= C0 + C1 drop the zero-padding channels of the NHWC tensors.
 * Implicit GEMM over the pixel axis on MFMA (csrc/wgrad.hip), split K with a fixed summation order (deterministic).
 * dw_layout: 0 = dw [Cout_w][Cin_w][k][k] (PyTorch's contiguous layout), 1 = [Cout_w][k][k][Cin_w] (the same tensor in channels-last
 *   memory order: what the training step keeps its weights in, so that the re-pack gathers contiguous runs).
 * with_bias = 1: also the bias gradient db[co] = sum_{b,y,x} dy[b][y][x][co] (Cout_w values), from one more accumulator tile against
 *   an all-ones operand in the same launch; its partial sums [nsplit][Cout_pad] sit behind the weight partials in the workspace.
 * with_bias = 2: the same sums per image, db [B][Cout_pad] (padding columns zero) - FeatureWiseAffine's gradient (unet.py:42-50) is
 *   the per-image sum of the block's conv gradient; partial sums [nsplit][B][Cout_pad].
 * workspace: hsidm_conv_wgrad_workspace_bytes(...) bytes (<= 64 MiB). */
int64_t hsidm_conv_wgrad_workspace_bytes(int C0, int C1, int B, int Hin, int Win, int Hout, int Wout, int Cout, int ksize,
                                         int stride, int ups);
int hsidm_conv_wgrad(int prec, const void* a0, const void* a1, int C0, int C1, const void* dy, int B, int Hin, int Win,
                     int Hout, int Wout, int Cout, int ksize, int stride, int ups, int Cout_w, int Cin_w, float* dw,
                     int dw_layout, int with_bias, float* db, void* workspace, int64_t workspace_bytes, void* stream);
/* Deferred form: dw == NULL leaves the partial tiles in `workspace` ([nsplit][taps][Cout_pad][Cin_pad] fp32, then with_bias's
 * [nsplit][Cout_pad] ([nsplit][B][Cout_pad] per image: Cout_pad' = Cout_w' = B*Cout_pad) - as an item: taps 1, Cin_pad 1, Cin_w 1,
 * ws = workspace + nsplit*taps*Cout_pad*Cin_pad floats; the sizes from
 * hsidm_conv_wgrad_plan) and ONE hsidm_wgrad_reduce_all launch sums every layer's splits later - a training step on one GPU has 94
 * of these reductions, each 10-20 us of latency on its own.  items: device array sorted by block0, block0 = prefix sum of
 * ceil(Cout_w * Cin_w / ppb) over the items, ppb = plan5[4] (pairs per workgroup: 64 with many splits, 256 with few; a bias item uses
 * its layer's ppb); total_blocks = that sum.  plan5 = {nsplit, taps, Cout_pad, Cin_pad, ppb}. */
typedef struct hsidm_wgrad_item {
    const float* ws;
    float*       dw;
    int32_t nsplit, NT, Cout_pad, Cin_pad, Cout_w, Cin_w, block0, layout;    /* layout: dw_layout of hsidm_conv_wgrad */
} hsidm_wgrad_item;
int hsidm_conv_wgrad_plan(int C0, int C1, int B, int Hin, int Win, int Hout, int Wout, int Cout, int ksize, int stride, int ups,
                          int32_t* plan5);
int hsidm_wgrad_reduce_all(const hsidm_wgrad_item* items_dev, int n_items, int total_blocks, void* stream);
/* Adjoints of the resampling steps (the input gradient of a convolution itself is hsidm_conv2d with the transposed, flipped
 * weights): zero_insert2: out[2y][2x] = in[y][x], zero elsewhere (stride-2 conv, unet.py:73-74; Hi = (Ho+1)/2);
 * sum2x2: out[y][x] = sum of in's 2x2 block (nearest x2, unet.py:64).  NHWC tensors of the mode's storage type. */
/* out = a + b over n elements (n % 8 == 0) of the mode's storage type: gradients meeting at a fan-out (skip connections). */
int hsidm_add(int prec, const void* a, const void* b, void* out, int64_t n, void* stream);
int hsidm_zero_insert2(int prec, const void* in, void* out, int B, int Hi, int Wi, int Ho, int Wo, int C, void* stream);
int hsidm_sum2x2(int prec, const void* in, void* out, int B, int H, int W, int C, void* stream);
/* Column sums of a statistics slab [B][nsplit][C][2] (hsidm_gn_partial / hsidm_conv2d): out_bc[b][c] = sum_s part[b][s][c].sum for
 * c < Cout (FiLM gradients, unet.py:49), out_c[c] = sum_b out_bc[b][c] (bias gradients); either may be NULL. */
int hsidm_colsum(const float* part, int nsplit, int B, int C, int Cout, float* out_bc, float* out_c, void* stream);
/* d/d(eps) of scale * loss_func(noise, eps) (diffusion.py:248 with set_loss's sum reduction; model/model.py:53-54 supplies
 * scale = 1/(b*c*h*w)): -scale * sign(noise - eps) (L1) or -2 * scale * (noise - eps) (L2).  noise, eps NCHW fp32 [B][Cimg][HW];
 * out NHWC [B][HW][Cpad] in the storage type, channels >= Cimg zero. */
int hsidm_loss_grad(int prec, const float* noise, const float* eps, int B, int Cimg, int HW, int Cpad, int kind, float scale,
                    void* out, void* stream);
/* Backward of hsidm_noise_film: dfilm [B][F] (per-image channel sums of the gradient at every block1 output) ->
 * dwf [F][dim], dbf [F], and through t = noise_level_mlp(gamma) (recomputed) dw1 [4dim][dim], db1, dw2 [dim][4dim], db2.
 * t_emb [B][dim]: the embedding the forward returned.  workspace: hsidm_noise_film_bwd_workspace_floats(B, dim, F) floats. */
int hsidm_noise_film_bwd_workspace_floats(int B, int dim, int F);
int hsidm_noise_film_bwd(const float* gamma, const float* t_emb, const float* dfilm, int B, int dim, const float* w1,
                         const float* b1, const float* w2, const float* wf, int F, float* dw1, float* db1, float* dw2,
                         float* db2, float* dwf, float* dbf, float* workspace, void* stream);
/* Strided batched GEMM on the exact fp32 matrix instruction (attention backward, csrc/bgemm.hip):
 * C[i](m, n) = alpha * sum_k A[i](m, k) * B[i](k, n); A(m,k) = a[i*sab + m*sam + k*sak], B(k,n) = b[i*sbb + k*sbk + n*sbn],
 * C(m,n) = c[i*scb + m*scm + n]; *_f32: the array's element type - 0 bf16, 1 fp32, 2 fp16. */
int hsidm_bgemm(const void* a, int a_f32, int64_t sab, int64_t sam, int64_t sak, const void* b, int b_f32, int64_t sbb,
                int64_t sbk, int64_t sbn, void* c, int c_f32, int64_t scb, int64_t scm, int M, int N, int K, int batch,
                float alpha, void* stream);
/* s <- softmax over each row of s [rows][N] (N <= 1024);  dp <- p o (dp - rowsum(dp o p)) * scale (its backward). */
int hsidm_softmax_rows(float* s, int64_t rows, int N, void* stream);
int hsidm_softmax_bwd_rows(const float* p, float* dp, int64_t rows, int N, float scale, void* stream);
/* Packed (kernel-order) weights straight from the flat fp32 master copy: out_hi[i] = bf16(src[idx[i]]) and, when out_lo != NULL
 * (HSIDM_F32X3), out_lo[i] = bf16(src[idx[i]] - out_hi[i]); idx[i] < 0 gives a zero (padding); n % 8 == 0.  idx is the packed layout's
 * gather map, built once on the host; one launch re-packs every convolution of the network after an optimiser step. */
int hsidm_gather_pack(const float* src, const int32_t* idx, int64_t n, void* out_hi, void* out_lo, void* stream);
/* One Adam step over a flat fp32 buffer (torch.optim.Adam as built in model/model.py:37-41: betas (0.9, 0.999), eps 1e-8, no
 * weight decay): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps), g scaled by
 * grad_scale first (1/world size after a sum all-reduce).  coef_dev != NULL: (lr/(1-b1^t), 1/sqrt(1-b2^t)) are read from that
 * device address (two floats) instead of being derived from `step` (captured steps). */
int hsidm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    int step, float grad_scale, const float* coef_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HSIDM_H */
