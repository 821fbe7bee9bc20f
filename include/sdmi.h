/* sdmi.h - C ABI of libsdmi_hip.so: the MI355X (gfx950) kernels behind the
 * latent-diffusion denoising hot path (UNet forward + VAE decoder + DDIM update).
 *
 * The reference (Charles-yueyue831/StableDiffusion) is pure PyTorch and has no native
 * plugin/FFI boundary of its own; each entry point below therefore cites
 * the reference *call sites* whose ATen ops it replaces.  Paths are relative to the reference
 * repository root.
 *
 * Conventions (every entry point):
 *   - extern "C", plain pointers and sizes; no torch types.
 *   - all pointers are DEVICE pointers unless the name says host.
 *   - asynchronous on `stream` (a hipStream_t passed as void*); no implicit sync.
 *   - the caller owns and allocates inputs, outputs and workspaces; nothing is allocated or
 *     freed inside (safe under HIP graph capture).
 *   - returns SDMI_OK (0) or a negative sdmi_status; never throws.
 *   - activations are 16-bit (bf16 or f16, `dtype` argument / field), channels-last: images [B, H, W, C] == token matrices [B*H*W, C].
 *   - statistics, logits, accumulators and softmax are fp32.
 */
#ifndef SDMI_H
#define SDMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sdmi_status {
    SDMI_OK = 0,
    SDMI_ERR_BAD_ARG = -1,      /* null pointer / non-positive size / unsupported combination */
    SDMI_ERR_ALIGNMENT = -2,    /* a leading dimension or channel count breaks the 16-byte vector rule */
    SDMI_ERR_WORKSPACE = -3,    /* workspace missing or too small */
    SDMI_ERR_LAUNCH = -4,       /* hipLaunchKernel (or another HIP runtime call) reported an error */
    SDMI_ERR_TOO_LARGE = -5     /* the problem exceeds the fixed scratch of a no-allocation entry point */
} sdmi_status;

/* 16-bit storage / MFMA operand type of activations and packed weights.  Mirrors the reference's dtype
 * policy (openai_model.py:602-603: use_fp16 -> float16, use_bf16 -> bfloat16); both run at the same MFMA rate. */
typedef enum sdmi_dtype {
    SDMI_DTYPE_BF16 = 0, SDMI_DTYPE_F16 = 1,
    SDMI_DTYPE_F32_PLANES = 2,   /* sdmi_pack_* only: the six bf16 piece planes of the fp32 mode (see sdmi_f32_split3) */
    SDMI_DTYPE_F16_TO_BF16 = 3   /* sdmi_groupnorm / sdmi_groupnorm_noalloc / sdmi_layernorm only: input f16 (the residual
                                    stream of the bf16 operand policy), output bf16 (an MFMA operand) */
} sdmi_dtype;

const char* sdmi_status_str(int status);
int sdmi_abi_version(void);   /* 13: + sdmi_gemm_args.tile_n == 64 (the 64-column tile, tile_m 64 | 128), ln_stats_slots up to 48; 12: + sdmi_gemm_args.ln_colsum / ln_stats / row_stats_out (LayerNorm folded into the projection, row statistics travelling with the residual stream), sdmi_gemm_row_stats_slots, sdmi_row_stats, sdmi_xattn_block (the cross-attention half of a transformer block as one launch); 11: + sdmi_gemm_args.out_f16 / res_f16 / a_f16 / dup_rows, SDMI_DTYPE_F16_TO_BF16, sdmi_probe_mfma, sdmi_probe_copy (measurement aids); 10: + sdmi_gemm_args.tile_n == 256, sdmi_gemm_wide_applies; 9: + sdmi_f32_attention_vae, sdmi_f32_nchw_to_nhwc takes a scale; 8: + sdmi_gemm_args.ln_gamma/ln_beta/ln_eps, sdmi_gemm_panel_applies; 7: + sdmi_gemm_args.algo; 6: + sdmi_attention_vae; 5: + sdmi_pack_*; 4: + sdmi_groupnorm_noalloc, SDMI_ERR_TOO_LARGE (3: + sdmi_f32_embed_tokens,
                                 sdmi_f32_attention_d64_causal, sdmi_gemm_args.act == 2; 2: dtype, fp32 mode) */

/* ------------------------------------------------------------------------------------------
 * GEMM / implicit-GEMM convolution on 16-bit (bf16 | f16) MFMA, fp32 accumulate:
 *     out[m, n] = epilogue( alpha * sum_k A[m, k] * W[n, k] )
 * Replaces: nn.Linear at ldm/modules/attention.py:81,117,213,224,291-296 and
 * ldm/modules/diffusion/openai_model.py:218,618-620; nn.Conv2d (3x3 s1/s2, 1x1) at
 * openai_model.py:85,139,195,234,244-248,663,927, ldm/modules/diffusion/model.py:122,329,369,
 * 433,440,444-448,491-494 and ldm/models/autoencoder.py:54; F.interpolate(x2, nearest)+conv at
 * openai_model.py:106-114 / model.py:137-141 (fused gather); torch.cat([h, skip]) feeding a 1x1
 * conv at openai_model.py:1010 (two-pointer A).
 *
 * W is PRE-PACKED (sdmi_pack_* below): 16-bit [Npad][K], K contiguous, Npad = N_w rounded up to tile_n
 * (128, or 160 for widths that are multiples of 160 -- no GEGLU), rows >= N_w zero.  K must be a
 * multiple of 64 (zero columns).
 *   mode 0 (linear / 1x1 conv): A is [M, lda] row-major; if A2 != NULL columns [0,k_split) come
 *       from A and [k_split,K) from A2 (row stride lda2); k_split % 64 == 0.  If the logical width of A
 *       is not a multiple of 64, pass it as k_valid: columns [k_valid, K) read as zero.
 *   mode 1 (3x3 conv, pad 1): A is an NHWC image [nb, H, W, Cin]; row m = (b, oy, ox) of the
 *       [nb, Ho, Wo] output; k = (ky*3+kx)*Cin + ci (weights packed in that order); stride 1|2;
 *       upsample=1 reads the input through a nearest x2 gather (logical input 2H x 2W).
 *       Cin % 8 == 0; taps beyond 9 (K padding) read as zero.
 * Epilogue, in this order: *alpha, +bias (per column n, or per row m if bias_per_row),
 *   GEGLU (geglu=1: W rows hold [16 value | 16 gate] column blocks alternately, bias packed the
 *   same way; out = value * gelu_erf(gate); N counts gated output columns), act (1 = SiLU, 2 = erf GELU),
 *   +emb[m / rows_per_batch, n] (fp32), +res[m, n] (16-bit), store as out_kind.
 *   out_kind 0: 16-bit [M, ld_out]; 1: fp32 [M, ld_out]; 2: fp32 NCHW [M/rows_per_batch, N, rows_per_batch].
 * batch > 1 runs `batch` independent problems (grid.z) with the element strides given.
 * split_k > 1 splits K over grid.z (batch must be 1): needs `workspace` of
 *   sdmi_gemm_workspace_bytes() bytes; partial sums are reduced in a second kernel.
 * ---------------------------------------------------------------------------------------- */
typedef struct sdmi_gemm_args {
    const void* A;
    const void* A2;
    const void* W;
    const float* bias;
    const float* emb;
    const void* res;
    void* out;
    void* workspace;
    int64_t workspace_bytes;
    int64_t M, N, K;
    int64_t lda, lda2, k_split;
    int64_t k_valid;                    /* mode 0: columns >= k_valid of A read as zero (0 = K); % 8 == 0 */
    int64_t ld_res, ld_out, ld_emb, rows_per_batch;
    int64_t stride_a, stride_w, stride_out, stride_res;   /* elements, for batch > 1 */
    int32_t mode;
    int32_t H, W_in, Cin, Ho, Wo, stride, upsample;
    int32_t out_kind, geglu, act, bias_per_row;
    int32_t batch, split_k;
    int32_t dtype;                      /* sdmi_dtype of A, A2, W, res and of out when out_kind == 0 */
    float alpha;
    int32_t tile_n;                     /* column tile of the kernel: 128 (0 = 128) or 160; W rows padded to it.  256 = the
                                           256 x 256 x 32 tile (more FLOP per byte moved into LDS: for long K with N_w a
                                           multiple of 256): the caller vouches that W has N_w rounded up to 256 rows, and
                                           sdmi_gemm_wide_applies() must hold - SDMI_ERR_BAD_ARG otherwise.
                                           64 = the small tile (64 columns x tile_m = 128 | 64 rows - mode 1: 128 only -, three stages): for
                                           launches of a few hundred rows, where the larger tiles leave most CUs without a block
                                           (the UNet at batch 1 - 2: 512 x 1280 x 1280 18.0 -> 9.7 us); W must hold N_w rounded up
                                           to 64 rows (any buffer packed for 128 | 160 with N_w % 64 == 0 does); no GEGLU, no
                                           ln_gamma, algo != 2, stages != 2 - SDMI_ERR_BAD_ARG otherwise.  Same k order inside
                                           the fp32 sums as the larger tiles: unsplit results are bit-identical */
    int32_t tile_m;                     /* row tile: 0 = sdmi_gemm_tile_m() picks, or 128 | 256 (tile_n 128 | 160), 256 (tile_n
                                           256), 64 | 128 (tile_n 64) */
    int32_t asym_pad;                   /* mode 1: 0 = zero pad 1 on every side; 1 = pad bottom/right only - the VAE
                                           encoder's DownSample, F.pad(x, (0,1,0,1)) + conv stride 2 (model.py:87,99-102) */
    int32_t res_f32;                    /* 1: `res` is fp32 (only with an fp32 out_kind) - the fp32 mode of the path */
    int32_t stages;                     /* LDS ring depth: 0 | 3 = three stages; 2 (128-row tile only) = two stages,
                                           two blocks per CU - for short K */
    int32_t algo;                       /* 0 = the library picks (incl. the narrow conv3x3 kernel for N <= 16, stride 1, and - for stride-1 conv3x3 over whole
                                           256-pixel row strips of a 16 | 32 | 64-pixel-wide image on the 256 x 160 tile - the halo-tile kernel of
                                           sdmi_conv_halo.hip, which sums K channel-chunk-major: its results equal the tiled kernel's to fp32
                                           rounding of the sums, not bit for bit); 1 = the tiled kernel; 2 = the A-panel-stationary kernel
                                           (mode 0, K == 320, 16-bit output, N_w % 160 == 0 or GEGLU: every workgroup keeps
                                           its rows of A in LDS and streams all of W) - SDMI_ERR_BAD_ARG if it does not apply */
    /* ln_gamma non-NULL: A is first layer-normalised over its K columns, y = (a - mean) * rstd * ln_gamma[k] + ln_beta[k]
       (biased variance, fp32 statistics), rounded to the 16-bit operand type - the nn.LayerNorm in front of to_q/k/v, to_q
       and the GEGLU projection (attention.py:512-514, 528-541) fused into the projection.  Only where the
       A-panel-stationary kernel applies (sdmi_gemm_panel_applies() == 1: the whole row is resident in LDS);
       SDMI_ERR_BAD_ARG otherwise - run sdmi_layernorm first in that case. */
    const float* ln_gamma;
    const float* ln_beta;
    float ln_eps;
    int32_t out_f16, res_f16, a_f16;    /* dtype == SDMI_DTYPE_BF16 only (SDMI_ERR_BAD_ARG with f16 operands, where they would change
                                           nothing): the bf16 operand policy keeps its RESIDUAL STREAM in f16 - same bytes, three more
                                           significant bits - so that only the MFMA operands carry bf16 rounding.
                                           out_f16: `out` (out_kind 0) is stored as f16; res_f16: `res` holds f16; a_f16: A holds f16 -
                                           only together with ln_gamma, whose LayerNorm then writes the bf16 operand */
    int32_t dup_rows;                   /* 1: every output row m is stored twice, at rows m and m + M of `out` (which then has
                                           2 M rows): the torch.cat([t, t]) that turns the guidance-pair prefix of the UNet
                                           (one row per (uncond, cond) pair) into the full classifier-free-guidance batch where
                                           the context enters, written by the producing launch instead of a copy kernel.  Only
                                           with the 16-bit row-major fast epilogue (out_kind 0, no GEGLU, N % 8 == 0, column
                                           bias, ld_out % 8 == 0, ld_res % 8 == 0, 16-byte aligned out / res), batch 1,
                                           split_k 1 and N > 16 - SDMI_ERR_BAD_ARG otherwise */
    /* ---- the token stream of the transformer blocks (dtype SDMI_DTYPE_F16, mode 0, no A2, no emb, column bias, batch 1,
       N_w % 4 == 0, stages != 2, algo != 2, ln_gamma NULL - SDMI_ERR_BAD_ARG otherwise) ----
       ln_colsum non-NULL: the nn.LayerNorm in front of the projection (attention.py:512-514, 528-541) FOLDED into it - no
       normalised copy of A exists at any point.  With W' = W * gamma[k] (the caller packs W', rounded to f16) and
           colsum[n] = sum_k W'[n, k]   (of the rounded values, one float per PACKED weight row: the GEGLU interleave and the
                                         zero padding rows included - what summing the packed buffer's rows gives),
           bias'[n]  = bias[n] + sum_k beta[k] * W[n, k]   (passed as `bias`),
       LayerNorm(a) @ W^T + bias = rstd[m] * (a @ W'^T - mean[m] * colsum[n]) + bias'[n]: the kernel multiplies the RAW rows
       of A and applies two numbers per row in front of the usual epilogue (with split_k > 1: in the reduce pass).  They come
       from ln_stats [M][ln_stats_slots][2] = per row and slot the partial (sum a, sum a^2) over the row's K columns (biased
       variance, fp32, eps = ln_eps; ln_stats_slots even and <= 48, or 1 | 3; 16-byte aligned) - left there by the launch that PRODUCED A
       through row_stats_out, or by sdmi_row_stats. */
    const float* ln_colsum;
    const float* ln_stats;
    int32_t ln_stats_slots;
    /* row_stats_out non-NULL: besides storing `out`, the launch leaves the row sums of the values it stores (fp32, before
       the rounding to f16) in row_stats_out [M (2 M with dup_rows)][row_stats_slots][2]; row_stats_slots must equal
       sdmi_gemm_row_stats_slots(args) (one slot per (column tile, wave column) of the kernel the plan in args selects - two
       per 128 | 160 | 64 columns, four per 256 -; one per 1024 columns with split_k > 1; at most 48: what a consumer reads).  16-bit row-major output with the conditions of dup_rows (split_k > 1 allowed without dup_rows), no GEGLU.
       This is how to_out + residual, the feed-forward's second linear and proj_in hand the statistics of the residual
       stream to the LayerNorm-folded projection that reads it next. */
    int32_t row_stats_slots;
    float* row_stats_out;
} sdmi_gemm_args;

int sdmi_gemm(const sdmi_gemm_args* args, void* stream);
/* 1 if this problem can run on the A-panel-stationary kernel (algo 2, or algo 0 with ln_gamma set), else 0. */
int sdmi_gemm_panel_applies(const sdmi_gemm_args* args);
/* 1 if this problem can run on the 256 x 256 x 32 tile (tile_n = 256): whole 64-deep k-tiles (mode 0: no K padding in
   A; mode 1: Cin % 64 == 0), and either split_k > 1, a plain 16-bit row-major epilogue (N % 8 == 0, column bias, 16-byte
   aligned out / res with ld % 8 == 0) or the 16-bit GEGLU epilogue (N % 32 == 0, no res / emb / act, batch 1).  Has no
   say on whether W is padded to 256 rows. */
int sdmi_gemm_wide_applies(const sdmi_gemm_args* args);
/* Slots per row a launch with these args (tile_n, split_k, N as they will be passed) writes to row_stats_out. */
int32_t sdmi_gemm_row_stats_slots(const sdmi_gemm_args* args);
/* Row statistics of a 16-bit [rows, C] matrix in the layout of sdmi_gemm_args.ln_stats with one slot:
   stats[r] = (sum_c x[r, c], sum_c x[r, c]^2), fp32.  For a LayerNorm-folded projection whose input was not produced by a
   launch with row_stats_out (the reference has no counterpart: its nn.LayerNorm computes them on the fly).  C % 8 == 0. */
int sdmi_row_stats(const void* x, float* stats, int64_t rows, int32_t C, int32_t dtype, void* stream);
int64_t sdmi_gemm_workspace_bytes(int64_t M, int64_t N_w, int32_t split_k);
/* Row tile (128 | 256; 64 | 128 for tile_n = 64) sdmi_gemm uses for this problem when args.tile_m == 0 (for split-K planning). */
int32_t sdmi_gemm_tile_m(int64_t M, int64_t N_w, int32_t tile_n);

/* ------------------------------------------------------------------------------------------
 * The cross-attention half of a BasicTransformerBlock as one launch (C = 320, 5 heads of 64: level 0 of the SD2 UNet):
 *     out = to_out( softmax( scale * to_q(LayerNorm(x)) K^T ) V ) + bias_o + x
 * Replaces the chain norm2 -> attn2.to_q -> CrossAttention.forward -> attn2.to_out[0] + residual at attention.py:518-528
 * (CrossAttention: attention.py:300-380) - three launches and two round trips of a [tokens, C] tensor - for a context of
 * up to 128 keys.  Every step is local to a token row except the read of the prompt's keys / values, so a workgroup runs
 * the chain on 128 rows resident in LDS.  x, out: f16 [M, ld]; wq, wo: f16 [320][320] as sdmi_pack_linear packs them with
 * tile_n = 160 (PLAIN weights: the LayerNorm is computed on the resident rows, exact two-pass statistics); k, v: f16
 * [images * nk, ld] (image b's keys at rows b * nk ...), head h at columns h * 64; rows_per_image = tokens of one image.
 * bias_o may be NULL (no output bias); out may be x itself (in place: a workgroup reads its 128 rows before it writes them).
 * row_stats_out (optional, row_stats_slots = 4): as sdmi_gemm_args.row_stats_out, for a LayerNorm-folded consumer of `out`.
 * sdmi_xattn_block_applies() == 1 iff the problem fits: dtype SDMI_DTYPE_F16, C == 320, heads == 5, nk <= 128, M and
 * rows_per_image multiples of 128, 16-byte aligned pointers, leading dimensions % 8 == 0. */
typedef struct sdmi_xattn_args {
    const void* x; int64_t ldx;
    int64_t M; int32_t rows_per_image;
    int32_t C, heads;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    const void* wq; const void* wo; const float* bias_o;
    const void* k; const void* v; int64_t ldk, ldv; int32_t nk;
    float scale;
    void* out; int64_t ld_out;
    float* row_stats_out; int32_t row_stats_slots;
    int32_t dtype;
} sdmi_xattn_args;
int sdmi_xattn_block(const sdmi_xattn_args* args, void* stream);
int sdmi_xattn_block_applies(const sdmi_xattn_args* args);

/* ------------------------------------------------------------------------------------------
 * Weight packing: fp32 weights in the reference's own (PyTorch state-dict) layout -> the pre-packed W operand that
 * sdmi_gemm reads.  Replaces nothing at run time: it is the one-time preparation of nn.Linear.weight [N, K]
 * (attention.py:81,117,213,224,291-296; openai_model.py:218,618-620), of GEGLU.proj.weight [2*inner, K]
 * (attention.py:81-91) and of nn.Conv2d.weight [Cout, Cin, 3, 3] (openai_model.py:85,139,195,234,663,927;
 * model.py:122,329,369,433,440) that the reference leaves to cuDNN / cuBLAS heuristics.
 * Packed buffer: 16-bit [Npad][Kp], Npad = rows_total rounded up to tile_n (128 | 160; sdmi_pack_tile_n gives the
 * tile the Python host uses: 160 where it divides N and the layer is not GEGLU), Kp = sdmi_pack_linear_k /
 * sdmi_pack_conv3x3_k, padding zero; sdmi_pack_bytes(rows_total, Kp, tile_n) bytes, allocated by the caller.
 *   linear : row n, column k as they stand.  Several weights that share an input (q|k|v, the 22 embedding
 *            projections) are concatenated along N by one call each: rows [row_offset, row_offset + N) of a buffer
 *            of rows_total rows; the call that holds the last row also writes the zero rows up to Npad.
 *   geglu  : rows interleaved [16 value | 16 gate] per 32 (see sdmi_gemm above); bias (may be NULL) permuted alike.
 *   conv3x3: column (ky*3 + kx) * Cin_p + ci with Cin_p = sdmi_pack_conv3x3_cin (pass it as sdmi_gemm_args.Cin; the
 *            activation must carry that many channels, zeros beyond Cin).
 * dtype: SDMI_DTYPE_BF16 | SDMI_DTYPE_F16 (the MFMA operand type), or SDMI_DTYPE_F32_PLANES - the fp32 mode: six bf16
 * planes (w0 w1 w0 w2 w1 w0), w = w0 + w1 + w2 exactly, of width K (Cin) rounded up to 32, inside every conv tap.
 * Device entry points take device pointers and run asynchronously on `stream`; the *_host twins take host pointers
 * and run the same arithmetic on the CPU (weights prepared off the GPU; the CPU test suite) - bit-identical results.
 * ---------------------------------------------------------------------------------------- */
int32_t sdmi_pack_tile_n(int64_t N, int32_t geglu);
int64_t sdmi_pack_linear_k(int64_t K, int32_t dtype);
int64_t sdmi_pack_conv3x3_cin(int64_t Cin, int32_t dtype);
int64_t sdmi_pack_conv3x3_k(int64_t Cin, int32_t dtype);
int64_t sdmi_pack_bytes(int64_t rows_total, int64_t k_packed, int32_t tile_n);
int sdmi_pack_linear(const float* w, int64_t N, int64_t K, int64_t ld_w, int64_t row_offset, int64_t rows_total,
                     void* out, int32_t dtype, int32_t tile_n, void* stream);
int sdmi_pack_geglu(const float* w, const float* bias, int64_t inner, int64_t K, int64_t ld_w, void* out,
                    float* bias_out, int32_t dtype, void* stream);
int sdmi_pack_conv3x3(const float* w, int64_t Cout, int64_t Cin, void* out, int32_t dtype, int32_t tile_n, void* stream);
int sdmi_pack_linear_host(const float* w, int64_t N, int64_t K, int64_t ld_w, int64_t row_offset, int64_t rows_total,
                          void* out, int32_t dtype, int32_t tile_n);
int sdmi_pack_geglu_host(const float* w, const float* bias, int64_t inner, int64_t K, int64_t ld_w, void* out,
                         float* bias_out, int32_t dtype);
int sdmi_pack_conv3x3_host(const float* w, int64_t Cout, int64_t Cin, void* out, int32_t dtype, int32_t tile_n);

/* ------------------------------------------------------------------------------------------
 * GroupNorm(32 groups) [+ SiLU] over an NHWC 16-bit tensor, fp32 statistics.
 * Replaces: GroupNorm32 + nn.SiLU at ldm/utils.py:267-278 with openai_model.py:186-190,
 * 225-235,922-927; nn.GroupNorm(eps 1e-6) at attention.py:45,241; group_norm + swish at
 * model.py:24-40,404-405,458-466,508.
 * Input is the channel concatenation [x1 (C1 ch) | x2 (C2 ch)] (x2 may be NULL, C2 = 0): this is
 * the torch.cat([h, hs.pop()], 1) of openai_model.py:1010 read through two pointers.  Output is
 * one [nb, hw, C1+C2] tensor.  C1, C2 multiples of 8; (C1+C2) % 32 == 0.
 * workspace: sdmi_groupnorm_workspace_bytes(nb, hw) bytes of fp32 scratch.
 * ---------------------------------------------------------------------------------------- */
int sdmi_groupnorm(const void* x1, const void* x2, int32_t C1, int32_t C2,
                   const float* gamma, const float* beta, void* out,
                   int32_t nb, int64_t hw, float eps, int32_t silu, int32_t dtype,
                   void* workspace, int64_t workspace_bytes, void* stream);
int64_t sdmi_groupnorm_workspace_bytes(int32_t nb, int64_t hw);

/* sdmi_groupnorm without a workspace argument, for callers that cannot allocate (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  It has NO counterpart in the reference (which allocates freely
 * through PyTorch): same arguments, kernels and results as sdmi_groupnorm, but the per-slab statistics live in a
 * static device array inside the library (SDMI_GROUPNORM_NOALLOC_MAX_BYTES per GPU, held until the process ends)
 * instead of caller memory; no hipMalloc / hipFree / allocating routine is called.
 *   - limit: nb * slabs(hw) <= 4096 slab entries (SDMI_GROUPNORM_NOALLOC_MAX_BYTES / 384), where slabs(hw) is what the
 *     launch really uses: 1 up to 1024 pixels (the fused single launch), else ceil(hw / max(64, ceil(hw / 256))) <= 256
 *     - e.g. 16 images of any size, 4096 images of 32 x 32; beyond it the call returns SDMI_ERR_TOO_LARGE and
 *     launches nothing.  (sdmi_groupnorm_workspace_bytes() is an upper bound of the same count, not the limit.)
 *   - the array's device address is looked up on the first call made with a device current (hipGetSymbolAddress, once
 *     per device, per-device pointer kept): make one call outside any stream capture first.
 *   - every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams) race - serialise them.
 *     Nothing has to be cleared between calls: each call overwrites every entry it later reads. */
#define SDMI_GROUPNORM_NOALLOC_MAX_BYTES (4096 * 32 * 3 * 4)
int sdmi_groupnorm_noalloc(const void* x1, const void* x2, int32_t C1, int32_t C2,
                           const float* gamma, const float* beta, void* out,
                           int32_t nb, int64_t hw, float eps, int32_t silu, int32_t dtype, void* stream);

/* LayerNorm over the last dim C of a 16-bit [rows, C] matrix, fp32 statistics, affine.
 * Replaces nn.LayerNorm at attention.py:512-514 (called :525-527).  C % 8 == 0. */
int sdmi_layernorm(const void* x, const float* gamma, const float* beta, void* out,
                   int64_t rows, int32_t C, float eps, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Flash-style multi-head attention, head dim 64, 16-bit in/out, fp32 logits + online softmax.
 *   out[b, i, h*64 + :] = softmax_j( scale * q[b,i,h,:] . k[b,j,h,:] ) @ v[b,j,h,:]
 * Replaces CrossAttention.forward at attention.py:315-378 (the einsum/softmax/einsum chain and
 * the head split/merge rearranges); never materialises the [B*H, Nq, Nk] score tensor.
 * q rows are [nb*Nq, ldq] with head h at columns [h*64, h*64+64); likewise k, v ([nb*Nk, ld*]) and
 * out.  All leading dimensions multiples of 8.  Nq, Nk arbitrary (>0).
 * Limit: nb * heads <= 65535 (one grid.y slot per (image, head)); SDMI_ERR_BAD_ARG beyond it - split the batch.
 * ---------------------------------------------------------------------------------------- */
int sdmi_attention_d64(const void* q, const void* k, const void* v, void* out,
                       int32_t nb, int32_t heads, int32_t Nq, int32_t Nk,
                       int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                       float scale, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Single-head attention of the VAE's AttnBlock (ldm/modules/diffusion/model.py:514-525: the bmm / softmax / bmm chain
 * on [B, N, C] with C = 512), head dim D in {64, 128, 256, 512}, 16-bit in/out, fp32 logits and softmax, WITHOUT the
 * [B, N, N] score tensor the reference materialises (340 MB per image at 768^2).
 *   W[i, j] = softmax_j( scale * q_i . k_j )
 *   mode 0 ("standard", upstream):  out[b, i, :] = sum_j W[i, j] v[b, j, :]
 *   mode 1 ("reference", literally model.py:519-523 - the contraction runs over the QUERY index):
 *                                    out[b, j, :] = sum_i W[i, j] v[b, i, :]
 * Two launches (row statistics, then the weighted sum with the logits recomputed): 6 N^2 D FLOP.
 * q, k, v: [nb * N, ld*] rows, out likewise; leading dimensions multiples of 8 (ldo: 4); N even.
 * workspace: sdmi_attention_vae_workspace_bytes(nb, N) bytes (two floats per query).
 * ---------------------------------------------------------------------------------------- */
int sdmi_attention_vae(const void* q, const void* k, const void* v, void* out, int32_t nb, int32_t N, int32_t D,
                       int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, int32_t mode, int32_t dtype,
                       void* workspace, int64_t workspace_bytes, void* stream);
int64_t sdmi_attention_vae_workspace_bytes(int32_t nb, int32_t N);

/* Row / column softmax helpers (the three-GEMM form of the VAE attention that sdmi_attention_vae replaced; kept for
 * callers that want the probabilities themselves) (model.py:514-523), whose
 * score matrix S is produced by sdmi_gemm (out_kind 1).
 *   rows: P[i, j] = softmax_j S[i, j]            (16-bit out)     - "standard" mode
 *   cols: PT[j, i] = softmax_j ST[j, i] for each column i of ST    - "reference" mode, where
 *         ST[j, i] = scale * k_j . q_i, so PT = W^T with W row-normalised over j (model.py:521-523).
 * colstats: fp32 scratch of sdmi_softmax_cols_scratch_bytes(batch, cols) bytes. */
int sdmi_softmax_rows(const float* S, void* P, int64_t rows, int64_t cols, int64_t ld_s, int64_t ld_p,
                      int32_t dtype, void* stream);
int sdmi_softmax_cols(const float* ST, void* PT, float* colstats, int32_t batch, int64_t rows, int64_t cols,
                      int32_t dtype, void* stream);
int64_t sdmi_softmax_cols_scratch_bytes(int32_t batch, int64_t cols);

/* ------------------------------------------------------------------------------------------
 * Layout / elementwise kernels at the edges of the path.
 * ---------------------------------------------------------------------------------------- */
/* fp32 NCHW [nb, C, hw] -> 16-bit (dtype) NHWC [nb, hw, Cpad] (channels >= C zero), value * scale.
 * Entry of UNetModel.forward (openai_model.py:980) and of decode_first_stage (z / scale_factor,
 * configs/stable-diffusion/v2-inference.yaml:17). */
int sdmi_nchw_f32_to_nhwc(const float* x, void* out, int32_t nb, int32_t C, int32_t Cpad, int64_t hw,
                          float scale, int32_t dtype, void* stream);
/* fp32 -> 16-bit (dtype) cast of a contiguous buffer (context embeddings, openai_model.py:940 `context`). */
int sdmi_cast_f32(const float* x, void* out, int64_t n, int32_t dtype, void* stream);
/* Sinusoidal timestep embedding, cos first then sin (ldm/utils.py:301-318, repaired B1/B2):
 * out 16-bit (dtype) [nb, dim]. t is int64 [nb]. dim even. */
int sdmi_timestep_embedding(const int64_t* t, void* out, int32_t nb, int32_t dim, float max_period, int32_t dtype,
                            void* stream);
/* Classifier-free-guidance combine + DDIM update on fp32 NCHW latents (the loop body absent from
 * ldm/models/diffusion/ddim.py:184-249; semantics SURVEY.md §3.2).  model_out holds the UNet output
 * for [uncond batch | cond batch] (2*n elements) when e_uc_first != 0, else n elements.
 * v_param selects v-prediction.  noise may be NULL (sigma == 0). Writes x_prev and pred_x0 (may be NULL). */
int sdmi_ddim_step(const float* x, const float* model_out, const float* noise, float* x_prev, float* pred_x0,
                   int64_t n, int32_t cfg, float guidance_scale, float a_t, float a_prev, float sigma_t,
                   float sqrt_one_minus_a_t, int32_t v_param, void* stream);
/* clamp((x + 1) / 2, 0, 1) * 255 -> uint8, NCHW fp32 in, NHWC uint8 out (the image post-processing that
 * follows decode_first_stage in upstream txt2img; absent after scripts/txt2img.py:402). */
/* out = w0*x0 + w1*x1 + w2*x2 + w3*x3 over n fp32 elements; x1..x3 may be NULL.  The eps combinations of the PLMS
 * multistep sampler (absent from the reference, named by txt2img.py:96-100 `--plms`; upstream plms.py p_sample_plms)
 * and the classifier-free-guidance combine (1-s)*e_uc + s*e_c when it is not fused into sdmi_ddim_step. */
int sdmi_lincomb4(const float* x0, const float* x1, const float* x2, const float* x3, float w0, float w1, float w2,
                  float w3, float* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * fp32 mode (the reference's float32 dtype policy, openai_model.py:602-603 with use_fp16 = use_bf16 = False): fp32
 * channels-last activations; contractions run on sdmi_gemm through the 3 x bf16 split below; the rest is plain fp32.
 * Built for the "<= 1e-5 in fp32 mode" parity gate, not for speed.
 *   sdmi_f32_split3: x = [x1 | x2] fp32 [rows, C1 + C2] -> bf16 [rows][6][Cp] planes (x0 x0 x1 x0 x1 x2) with
 *       x = x0 + x1 + x2 exactly; the matching weight planes are (w0 w1 w0 w2 w1 w0), so a GEMM / conv over 6*Cp
 *       input channels accumulates the six piece products of weight <= 2 = the fp32 product.  Cp % 8 == 0.
 *   sdmi_f32_groupnorm / _layernorm / _attention_d64 / _nchw_to_nhwc / _timestep_embedding: fp32 counterparts of
 *       the 16-bit entry points above (same reference call sites), two-pass statistics, full-precision exp / erf.
 * ---------------------------------------------------------------------------------------- */
int sdmi_f32_split3(const float* x1, const float* x2, int32_t C1, int32_t C2, int64_t rows, void* out, int32_t Cp,
                    void* stream);
int sdmi_f32_groupnorm(const float* x1, const float* x2, int32_t C1, int32_t C2, const float* gamma, const float* beta,
                       float* out, int32_t nb, int64_t hw, float eps, int32_t silu, void* stream);
int sdmi_f32_layernorm(const float* x, const float* gamma, const float* beta, float* out, int64_t rows, int32_t C,
                       float eps, void* stream);
int sdmi_f32_attention_d64(const float* q, const float* k, const float* v, float* out, int32_t nb, int32_t heads,
                           int32_t Nq, int32_t Nk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale,
                           void* stream);
int sdmi_f32_nchw_to_nhwc(const float* x, float* out, int32_t nb, int32_t C, int64_t hw, float scale, void* stream);   /* out = scale * x */
/* fp32 counterpart of sdmi_attention_vae (AttnBlock, ldm/modules/diffusion/model.py:496-528): same arguments and modes
 * (0 standard, 1 the reference's contraction over the query index), same workspace (sdmi_attention_vae_workspace_bytes),
 * D in {64, 128, 256, 512}, leading dimensions multiples of 4, any N.  Two passes, no [B, N, N] tensor. */
int sdmi_f32_attention_vae(const float* q, const float* k, const float* v, float* out, int32_t nb, int32_t N, int32_t D,
                           int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, int32_t mode, void* workspace,
                           int64_t workspace_bytes, void* stream);
int sdmi_f32_timestep_embedding(const int64_t* t, float* out, int32_t nb, int32_t dim, float max_period, void* stream);

/* Text conditioning stage (cond_stage_config, configs/stable-diffusion/v2-inference.yaml:63-67, called from
 * LatentDiffusion.get_learned_conditioning, ldm/models/diffusion/ddpm.py:658-669).  The reference names
 * FrozenOpenCLIPEmbedder but does not contain it; these are the two pieces of a pre-LN text transformer that the
 * entry points above do not already cover.  fp32 like the rest of this section (the stage runs once per prompt).
 *   embed_tokens : out[r, :] = table[ids[r], :] + pos[r % T, :]   (token_embedding + positional_embedding)
 *   attention_d64_causal : sdmi_f32_attention_d64 with Nq == Nk == N and query i restricted to keys 0..i
 *                          (the additive -inf upper-triangular attn_mask of the text tower)                      */
int sdmi_f32_embed_tokens(const int64_t* ids, const float* table, const float* pos, float* out, int64_t rows, int32_t T,
                          int32_t C, int32_t vocab, void* stream);
int sdmi_f32_attention_d64_causal(const float* q, const float* k, const float* v, float* out, int32_t nb, int32_t heads,
                                  int32_t N, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale,
                                  void* stream);

/* DiagonalGaussianDistribution (ldm/modules/distribution/distribution.py:10-43) on fp32 NCHW moments
 * [nb, 2C, hw]: mean = first half, log_var = second half clamped to [-30, 20], std = exp(log_var / 2),
 * var = exp(log_var) (std = var = 0 when deterministic); sdmi_mul_add gives the sample mean + std * noise. */
int sdmi_diag_gaussian(const float* parameters, float* mean, float* log_var, float* stdv, float* var, int32_t nb, int32_t C,
                       int64_t hw, int32_t deterministic, void* stream);
int sdmi_mul_add(const float* a, const float* b, const float* c, float* out, int64_t n, void* stream);

int sdmi_image_to_u8(const float* x, uint8_t* out, int32_t nb, int32_t C, int64_t hw, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement aids (no counterpart in the reference; they compute nothing the path uses).  bench.py times them with
 * HIP events to state roofline fractions against the MEASURED peaks of the box it runs on as well as the nominal ones
 * (SURVEY.md section 8(d)).
 *   sdmi_probe_mfma: a register-only v_mfma_f32_16x16x32_f16 loop on pseudo-random operands, 256 workgroups x 8 waves,
 *       `iters` x 16 MFMAs per wave; *flop (host pointer, may be NULL) receives the FLOP count of the launch.  `out`:
 *       2 KiB of device memory (never written in practice).
 *   sdmi_probe_copy: dst[0, bytes) = src[0, bytes) by 16-byte lanes (bytes % 16 == 0, 16-byte aligned pointers): the
 *       streaming HBM rate is 2 * bytes / time.
 * ---------------------------------------------------------------------------------------- */
int sdmi_probe_mfma(float* out, int32_t iters, double* flop, void* stream);
int sdmi_probe_copy(const void* src, void* dst, int64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDMI_H */
