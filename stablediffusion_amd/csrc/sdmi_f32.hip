// sdmi_f32: the fp32 mode of the path (use_fp16 = use_bf16 = False, the reference's float32 dtype policy,
// openai_model.py:602-603) - built for the parity gate "per-step eps/v <= 1e-5 in fp32 mode", not for speed.
//
// Contractions still run on the bf16 MFMA kernels of sdmi_gemm.hip: every fp32 operand is split into three bf16
// pieces  x = x0 + x1 + x2  (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1): 3 x 8 = 24 mantissa bits, an
// exact decomposition of an fp32 value up to its last bit) and the product keeps the six piece pairs of weight
// i + j <= 2:   x*w ~= x0w0 + x0w1 + x1w0 + x0w2 + x1w1 + x2w0   (dropped terms are below 2^-24 relative).
// The six pairs are laid out along K - activation planes [x0 x0 x1 x0 x1 x2], weight planes [w0 w1 w0 w2 w1 w0] - so
// ONE fp32-accumulating bf16 GEMM / implicit-GEMM conv of 6x the K extent computes the fp32 product.  Everything
// else (norms, softmax, attention with head dim 64, activations) is plain fp32 arithmetic in the kernels below;
// activations are stored as fp32, channels-last.
#include "sdmi_common.h"

namespace {

__device__ __forceinline__ void split3(float x, unsigned short& a0, unsigned short& a1, unsigned short& a2) {
    a0 = f32_to_bf16_bits(x);
    const float r1 = x - bf16_bits_to_f32(a0);
    a1 = f32_to_bf16_bits(r1);
    const float r2 = r1 - bf16_bits_to_f32(a1);
    a2 = f32_to_bf16_bits(r2);
}

// x = [x1 | x2] (fp32 rows of C1 + C2 channels) -> out bf16 [rows][6][Cp], planes x0 x0 x1 x0 x1 x2, channels >= C zero
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int C1,
                                                     int C2, int64_t rows, unsigned short* __restrict__ out, int Cp) {
    const int64_t total = rows * Cp;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / Cp;
        const int c = (int)(idx - r * Cp);
        float v = 0.f;
        if (c < C1) v = x1[r * C1 + c];
        else if (c < C1 + C2) v = x2[r * C2 + (c - C1)];
        unsigned short a0, a1, a2;
        split3(v, a0, a1, a2);
        unsigned short* o = out + r * 6 * Cp + c;
        o[0] = a0; o[Cp] = a0; o[2 * Cp] = a1; o[3 * Cp] = a0; o[4 * Cp] = a1; o[5 * Cp] = a2;
    }
}

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

// GroupNorm(32)(+SiLU) on fp32 NHWC, input possibly the concat of two tensors: one block per (image, group),
// exact two-pass statistics.
__global__ __launch_bounds__(256) void gn_f32_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int C1,
                                                     int C2, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ out, int64_t hw,
                                                     float eps, int silu) {
    __shared__ float red[4];
    const int C = C1 + C2, cpg = C / 32;
    const int b = blockIdx.y, g = blockIdx.x;
    const int64_t n = hw * cpg;
    auto load = [&](int64_t e) -> float {
        const int64_t pix = e / cpg;
        const int c = g * cpg + (int)(e - pix * cpg);
        return c < C1 ? x1[((int64_t)b * hw + pix) * C1 + c] : x2[((int64_t)b * hw + pix) * C2 + (c - C1)];
    };
    float s = 0.f;
    for (int64_t e = threadIdx.x; e < n; e += blockDim.x) s += load(e);
    const float mean = block_sum(s, red) / (float)n;
    float q = 0.f;
    for (int64_t e = threadIdx.x; e < n; e += blockDim.x) { const float d = load(e) - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(block_sum(q, red) / (float)n + eps);
    for (int64_t e = threadIdx.x; e < n; e += blockDim.x) {
        const int64_t pix = e / cpg;
        const int c = g * cpg + (int)(e - pix * cpg);
        float y = (load(e) - mean) * rstd * gamma[c] + beta[c];
        if (silu) y = y / (1.0f + expf(-y));
        out[((int64_t)b * hw + pix) * C + c] = y;
    }
}

// LayerNorm over C, fp32: one wave per row, exact two-pass statistics.
__global__ __launch_bounds__(256) void ln_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ out,
                                                     int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    for (int c = lane; c < C; c += 64) out[row * C + c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
}

// Attention, head dim 64, fp32 throughout: block = 64 queries of one (batch, head), one query per thread; keys and
// values are staged through LDS 64 at a time and read by broadcast; online softmax per thread.
__global__ __launch_bounds__(64) void attn_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, float* __restrict__ out, int heads,
                                                      int Nq, int Nk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                                                      float scale, int causal) {
    __shared__ float Ks[64][64], Vs[64][64];
    const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
    const int qi = blockIdx.x * 64 + threadIdx.x;
    const bool qok = qi < Nq;
    const float* qp = q + ((int64_t)b * Nq + (qok ? qi : 0)) * ldq + h * 64;
    float qr[64], o[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) { qr[d] = qp[d] * scale; o[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < Nk; k0 += 64) {
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * 64; e += 64) {
            const int r = e >> 6, d = e & 63;
            const bool ok = k0 + r < Nk;
            Ks[r][d] = ok ? k[((int64_t)b * Nk + k0 + r) * ldk + h * 64 + d] : 0.f;
            Vs[r][d] = ok ? v[((int64_t)b * Nk + k0 + r) * ldv + h * 64 + d] : 0.f;
        }
        __syncthreads();
        int kn = Nk - k0 < 64 ? Nk - k0 : 64;
        if (causal && qi - k0 + 1 < kn) kn = qi - k0 + 1;       // query i attends to keys 0..i (per-thread bound)
        for (int j = 0; j < kn; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 64; ++d) s = fmaf(qr[d], Ks[j][d], s);
            const float mn = fmaxf(m, s);
            const float a = expf(m - mn), p = expf(s - mn);
            l = l * a + p;
#pragma unroll
            for (int d = 0; d < 64; ++d) o[d] = fmaf(p, Vs[j][d], o[d] * a);
            m = mn;
        }
    }
    if (qok) {
        float* op = out + ((int64_t)b * Nq + qi) * ldo + h * 64;
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < 64; ++d) op[d] = o[d] * inv;
    }
}

// Single-head attention of the VAE's AttnBlock (model.py:496-528) in fp32 - the counterpart of sdmi_attention_vae,
// same two passes (row statistics over all keys, then the weighted sum with the logits recomputed; no [B, N, N]
// tensor), plain FMA on the VALU.  16 output rows per 256-thread block, 16 lanes per row, each lane owning D / 16
// channels as interleaved float4s (the 16 lanes of a row read 256 contiguous bytes); Y (and V) rows are staged through
// LDS TK at a time and read by broadcast across the four rows of a wave.
//   MODE 0: statistics (m_i, 1 / l_i) of the rows of X against all Y.
//   MODE 1: O[r, :] = sum_c w(r, c) V[c, :], w = exp(s - m_r) / l_r           (standard: X = Q, Y = K)
//   MODE 2: ... w = exp(s - m_c) / l_c, the statistics of the contraction index (reference: X = K, Y = Q, defect B7)
template <int DC, int MODE>
__global__ __launch_bounds__(256) void attn_vae_f32_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ v, float* __restrict__ out,
                                                           float* __restrict__ stats, int N, int64_t ldx, int64_t ldy,
                                                           int64_t ldv, int64_t ldo, float scale) {
    constexpr int D = DC * 64, TK = MODE == 0 ? 16 : 8;
    __shared__ __attribute__((aligned(16))) float Ys[TK][D];
    __shared__ __attribute__((aligned(16))) float Vs[MODE == 0 ? 1 : TK][MODE == 0 ? 4 : D];
    __shared__ float Cs[TK][2];
    const int b = blockIdx.y, t = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int row = blockIdx.x * 16 + rl;
    const int rowc = row < N ? row : N - 1;
    const float* xp = x + ((int64_t)b * N + rowc) * ldx;
    const float* yb = y + (int64_t)b * N * ldy;
    const float* vb = v + (int64_t)b * N * ldv;
    float* sb = stats + (int64_t)b * N * 2;
    f32x4 xr[DC], o[DC];
#pragma unroll
    for (int u = 0; u < DC; ++u) {
        xr[u] = *reinterpret_cast<const f32x4*>(xp + u * 64 + t * 4);
        o[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float m = -INFINITY, l = 0.f, il = 0.f;
    if (MODE == 1) { m = sb[2 * rowc]; il = sb[2 * rowc + 1]; }
    for (int c0 = 0; c0 < N; c0 += TK) {
        __syncthreads();
        for (int e = threadIdx.x; e < TK * (D / 4); e += 256) {
            const int r = e / (D / 4), d4 = e - r * (D / 4);
            const int c = c0 + r < N ? c0 + r : N - 1;
            *reinterpret_cast<f32x4*>(&Ys[r][d4 * 4]) = *reinterpret_cast<const f32x4*>(yb + (int64_t)c * ldy + d4 * 4);
            if (MODE != 0)
                *reinterpret_cast<f32x4*>(&Vs[r][d4 * 4]) = *reinterpret_cast<const f32x4*>(vb + (int64_t)c * ldv + d4 * 4);
        }
        if (MODE == 2 && threadIdx.x < TK) {
            const int c = c0 + threadIdx.x < N ? c0 + threadIdx.x : N - 1;
            Cs[threadIdx.x][0] = sb[2 * c];
            Cs[threadIdx.x][1] = sb[2 * c + 1];
        }
        __syncthreads();
        const int kn = N - c0 < TK ? N - c0 : TK;
        for (int j = 0; j < kn; ++j) {
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < DC; ++u) {
                const f32x4 yv = *reinterpret_cast<const f32x4*>(&Ys[j][u * 64 + t * 4]);
                s = fmaf(xr[u][0], yv[0], s); s = fmaf(xr[u][1], yv[1], s);
                s = fmaf(xr[u][2], yv[2], s); s = fmaf(xr[u][3], yv[3], s);
            }
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
            s *= scale;
            if (MODE == 0) {
                const float mn = fmaxf(m, s);
                l = l * expf(m - mn) + expf(s - mn);
                m = mn;
            } else {
                const float w = MODE == 1 ? expf(s - m) * il : expf(s - Cs[j][0]) * Cs[j][1];
#pragma unroll
                for (int u = 0; u < DC; ++u) {
                    const f32x4 vv = *reinterpret_cast<const f32x4*>(&Vs[j][u * 64 + t * 4]);
                    o[u][0] = fmaf(w, vv[0], o[u][0]); o[u][1] = fmaf(w, vv[1], o[u][1]);
                    o[u][2] = fmaf(w, vv[2], o[u][2]); o[u][3] = fmaf(w, vv[3], o[u][3]);
                }
            }
        }
    }
    if (row >= N) return;
    if (MODE == 0) {
        if (t == 0) { sb[2 * row] = m; sb[2 * row + 1] = 1.0f / l; }
    } else {
        float* op = out + ((int64_t)b * N + row) * ldo;
#pragma unroll
        for (int u = 0; u < DC; ++u) *reinterpret_cast<f32x4*>(op + u * 64 + t * 4) = o[u];
    }
}

template <int DC>
void launch_attn_vae_f32(const float* q, const float* k, const float* v, float* out, float* stats, int nb, int N, int64_t ldq,
                         int64_t ldk, int64_t ldv, int64_t ldo, float scale, int mode, hipStream_t s) {
    const dim3 grid((unsigned)((N + 15) / 16), (unsigned)nb);
    hipLaunchKernelGGL((attn_vae_f32_kernel<DC, 0>), grid, dim3(256), 0, s, q, k, v, out, stats, N, ldq, ldk, ldv, ldo, scale);
    if (mode == 0)
        hipLaunchKernelGGL((attn_vae_f32_kernel<DC, 1>), grid, dim3(256), 0, s, q, k, v, out, stats, N, ldq, ldk, ldv, ldo, scale);
    else
        hipLaunchKernelGGL((attn_vae_f32_kernel<DC, 2>), grid, dim3(256), 0, s, k, q, v, out, stats, N, ldk, ldq, ldv, ldo, scale);
}

// out[r, :] = table[ids[r], :] + pos[r % T, :]  (token + positional embedding of a text transformer).  An id outside
// [0, vocab) is clamped: the host wrapper has already rejected it, this only keeps the access in bounds.
__global__ __launch_bounds__(256) void embed_tokens_f32_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                               const float* __restrict__ pos, float* __restrict__ out,
                                                               int64_t rows, int T, int C, int vocab) {
    const int cq = C / 4;
    const int64_t total = rows * cq;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / cq;
        const int c = (int)(idx - r * cq) * 4;
        int64_t id = ids[r];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const f32x4 a = *reinterpret_cast<const f32x4*>(table + id * C + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(pos + (r % T) * C + c);
        *reinterpret_cast<f32x4*>(out + r * C + c) = a + b;
    }
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int nb,
                                                               int C, int64_t hw, float scale) {
    const int64_t total = (int64_t)nb * C * hw;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = idx % hw;
        const int64_t r = idx / hw;
        const int c = (int)(r % C);
        const int64_t b = r / C;
        out[(b * hw + pix) * C + c] = x[idx] * scale;
    }
}

__global__ void timestep_embedding_f32_kernel(const int64_t* __restrict__ t, float* __restrict__ out, int nb, int dim,
                                              float max_period) {
    const int half = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= half || b >= nb) return;
    // ldm/utils.py:304-314: freq = exp(-ln(max_period) * i / half); [cos | sin]
    const float freq = expf(-logf(max_period) * (float)i / (float)half);
    const float arg = (float)t[b] * freq;
    out[(int64_t)b * dim + i] = cosf(arg);
    out[(int64_t)b * dim + half + i] = sinf(arg);
}

inline unsigned blocks_for(int64_t n, int64_t cap = 16384) {
    int64_t b = (n + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int sdmi_f32_split3(const float* x1, const float* x2, int32_t C1, int32_t C2, int64_t rows, void* out,
                               int32_t Cp, void* stream) {
    if (x1 == nullptr || out == nullptr || rows <= 0 || C1 <= 0 || C2 < 0) return SDMI_ERR_BAD_ARG;
    if ((x2 == nullptr) != (C2 == 0) || Cp < C1 + C2) return SDMI_ERR_BAD_ARG;
    if (Cp % 8 != 0) return SDMI_ERR_ALIGNMENT;
    hipLaunchKernelGGL(split3_kernel, dim3(blocks_for(rows * Cp)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x1, x2,
                       C1, C2, rows, reinterpret_cast<unsigned short*>(out), Cp);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

extern "C" int sdmi_f32_groupnorm(const float* x1, const float* x2, int32_t C1, int32_t C2, const float* gamma,
                                  const float* beta, float* out, int32_t nb, int64_t hw, float eps, int32_t silu,
                                  void* stream) {
    if (x1 == nullptr || gamma == nullptr || beta == nullptr || out == nullptr) return SDMI_ERR_BAD_ARG;
    if (nb <= 0 || hw <= 0 || C1 <= 0 || C2 < 0 || (x2 == nullptr) != (C2 == 0)) return SDMI_ERR_BAD_ARG;
    if ((C1 + C2) % 32 != 0) return SDMI_ERR_ALIGNMENT;
    hipLaunchKernelGGL(gn_f32_kernel, dim3(32, (unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x1, x2, C1,
                       C2, gamma, beta, out, hw, eps, silu);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

extern "C" int sdmi_f32_layernorm(const float* x, const float* gamma, const float* beta, float* out, int64_t rows,
                                  int32_t C, float eps, void* stream) {
    if (x == nullptr || gamma == nullptr || beta == nullptr || out == nullptr || rows <= 0 || C <= 0) return SDMI_ERR_BAD_ARG;
    if ((rows + 3) / 4 > 0x7fffffffLL) return SDMI_ERR_BAD_ARG;
    hipLaunchKernelGGL(ln_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                       gamma, beta, out, rows, C, eps);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

static int launch_attn_f32(const float* q, const float* k, const float* v, float* out, int32_t nb, int32_t heads,
                                      int32_t Nq, int32_t Nk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                                      float scale, int causal, void* stream) {
    if (q == nullptr || k == nullptr || v == nullptr || out == nullptr) return SDMI_ERR_BAD_ARG;
    if (nb <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0) return SDMI_ERR_BAD_ARG;
    if (ldq < (int64_t)heads * 64 || ldk < (int64_t)heads * 64 || ldv < (int64_t)heads * 64 || ldo < (int64_t)heads * 64)
        return SDMI_ERR_BAD_ARG;
    if ((int64_t)nb * heads > 65535) return SDMI_ERR_BAD_ARG;
    hipLaunchKernelGGL(attn_f32_kernel, dim3((unsigned)((Nq + 63) / 64), (unsigned)(nb * heads)), dim3(64), 0,
                       reinterpret_cast<hipStream_t>(stream), q, k, v, out, heads, Nq, Nk, ldq, ldk, ldv, ldo, scale,
                       causal);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

extern "C" int sdmi_f32_attention_d64(const float* q, const float* k, const float* v, float* out, int32_t nb, int32_t heads,
                                      int32_t Nq, int32_t Nk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                                      float scale, void* stream) {
    return launch_attn_f32(q, k, v, out, nb, heads, Nq, Nk, ldq, ldk, ldv, ldo, scale, 0, stream);
}

extern "C" int sdmi_f32_attention_d64_causal(const float* q, const float* k, const float* v, float* out, int32_t nb,
                                             int32_t heads, int32_t N, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                                             float scale, void* stream) {
    return launch_attn_f32(q, k, v, out, nb, heads, N, N, ldq, ldk, ldv, ldo, scale, 1, stream);
}

extern "C" int sdmi_f32_attention_vae(const float* q, const float* k, const float* v, float* out, int32_t nb, int32_t N,
                                      int32_t D, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale,
                                      int32_t mode, void* workspace, int64_t workspace_bytes, void* stream) {
    if (q == nullptr || k == nullptr || v == nullptr || out == nullptr) return SDMI_ERR_BAD_ARG;
    if (nb <= 0 || nb > 65535 || N <= 0 || (mode != 0 && mode != 1)) return SDMI_ERR_BAD_ARG;
    if (D != 64 && D != 128 && D != 256 && D != 512) return SDMI_ERR_BAD_ARG;
    if (ldq < D || ldk < D || ldv < D || ldo < D) return SDMI_ERR_BAD_ARG;
    if (((ldq | ldk | ldv | ldo) & 3) != 0 || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) != 0)
        return SDMI_ERR_ALIGNMENT;
    if (workspace == nullptr || workspace_bytes < (int64_t)nb * N * 2 * (int64_t)sizeof(float)) return SDMI_ERR_WORKSPACE;
    float* st = reinterpret_cast<float*>(workspace);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (D) {
        case 64:  launch_attn_vae_f32<1>(q, k, v, out, st, nb, N, ldq, ldk, ldv, ldo, scale, mode, s); break;
        case 128: launch_attn_vae_f32<2>(q, k, v, out, st, nb, N, ldq, ldk, ldv, ldo, scale, mode, s); break;
        case 256: launch_attn_vae_f32<4>(q, k, v, out, st, nb, N, ldq, ldk, ldv, ldo, scale, mode, s); break;
        default:  launch_attn_vae_f32<8>(q, k, v, out, st, nb, N, ldq, ldk, ldv, ldo, scale, mode, s); break;
    }
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

extern "C" int sdmi_f32_embed_tokens(const int64_t* ids, const float* table, const float* pos, float* out, int64_t rows,
                                     int32_t T, int32_t C, int32_t vocab, void* stream) {
    if (ids == nullptr || table == nullptr || pos == nullptr || out == nullptr) return SDMI_ERR_BAD_ARG;
    if (rows <= 0 || T <= 0 || C <= 0 || vocab <= 0 || C % 4 != 0 || rows % T != 0) return SDMI_ERR_BAD_ARG;
    hipLaunchKernelGGL(embed_tokens_f32_kernel, dim3(blocks_for(rows * (C / 4))), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), ids, table, pos, out, rows, T, C, vocab);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

extern "C" int sdmi_f32_nchw_to_nhwc(const float* x, float* out, int32_t nb, int32_t C, int64_t hw, float scale,
                                     void* stream) {
    if (x == nullptr || out == nullptr || nb <= 0 || C <= 0 || hw <= 0) return SDMI_ERR_BAD_ARG;
    hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel, dim3(blocks_for((int64_t)nb * C * hw)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, out, nb, C, hw, scale);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

extern "C" int sdmi_f32_timestep_embedding(const int64_t* t, float* out, int32_t nb, int32_t dim, float max_period,
                                           void* stream) {
    if (t == nullptr || out == nullptr || nb <= 0 || dim <= 0 || dim % 2 != 0) return SDMI_ERR_BAD_ARG;
    const int half = dim / 2;
    hipLaunchKernelGGL(timestep_embedding_f32_kernel, dim3((unsigned)((half + 127) / 128), (unsigned)nb), dim3(128), 0,
                       reinterpret_cast<hipStream_t>(stream), t, out, nb, dim, max_period);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}
