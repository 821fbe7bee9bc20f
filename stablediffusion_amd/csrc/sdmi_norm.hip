// sdmi_norm: GroupNorm(32)+SiLU and LayerNorm for NHWC bf16 activations on gfx950.
//
// Both are HBM-bound: every global access is a 16-byte (8 x bf16) vector, fully coalesced because an
// NHWC tensor [nb, hw, C] is one contiguous [nb*hw, C] matrix.  Statistics are fp32.
//
// GroupNorm runs as two launches on the caller's stream:
//   1. gn_stats: grid (slabs, nb) - each block reduces a slab of pixels to per-group (n, mean, M2)
//   2. gn_apply: grid (slabs, nb) - Chan/Welford merge of the image's slab partials -> (mean, rstd) per group
//                                   (redone by every block, in a fixed order), then
//                                   y = silu(x * scale[c] + shift[c]) with scale/shift tables in LDS
// The second read of x in pass 2 is served from L2 / Infinity Cache for the UNet's tensor sizes.
// The input may be the channel concatenation of two tensors (two base pointers).
#include <mutex>
#include "sdmi_common.h"

namespace {

constexpr int GN_GROUPS = 32;
constexpr int GN_THREADS = 256;

__host__ __device__ inline int64_t gn_slab_pixels(int64_t hw) {
    // at most 256 slabs per image, at least 64 pixels per slab
    int64_t p = (hw + 255) / 256;
    if (p < 64) p = 64;
    return p;
}

__device__ __forceinline__ uint4 gn_load(const unsigned short* x1, const unsigned short* x2, int C1, int C2,
                                         int64_t pixel, int v) {
    const int c = v * 8;
    if (c < C1) return *reinterpret_cast<const uint4*>(x1 + pixel * C1 + c);
    return *reinterpret_cast<const uint4*>(x2 + pixel * C2 + (c - C1));
}

// Chan/Welford merge of two (n, mean, M2) summaries; an empty right-hand side leaves the left unchanged.
__device__ __forceinline__ void gn_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    if (nb == 0.f) return;
    const float tot = n + nb;
    const float delta = mb - mean;
    mean += delta * (nb / tot);
    m2 += m2b + delta * delta * (n * nb / tot);
    n = tot;
}

__device__ __forceinline__ int64_t p_begin_of(int slab, int64_t P) { return (int64_t)slab * P; }
__device__ __forceinline__ int64_t p_end_of(int slab, int64_t P, int64_t hw) {
    const int64_t e = (int64_t)slab * P + P;
    return e < hw ? e : hw;
}

// y = silu?(x * scale[c] + shift[c]) over pixels [p_begin, p_end) x channels [c0, c0 + Cw) of image b; scale/shift
// tables are built in LDS (sh[2*Cw]) from grp_stat[g_local] = (mean, rstd).  Called by every thread of the block.
template <int DTI, int DTO>
__device__ __forceinline__ void gn_apply_stream(const unsigned short* __restrict__ x1, const unsigned short* __restrict__ x2,
                                                int C1, int C2, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, const float* grp_stat,
                                                unsigned short* __restrict__ out, int64_t hw, int64_t p_begin,
                                                int64_t p_end, int b, int c0, int Cw, int cpg, int silu, float* sh) {
    const int C = C1 + C2, tid = threadIdx.x, v0 = c0 >> 3, nvec = Cw >> 3;
    for (int c = tid; c < Cw; c += GN_THREADS) {
        const int g = c / cpg;
        const float mean = grp_stat[2 * g], rstd = grp_stat[2 * g + 1];
        const float sc = rstd * gamma[c0 + c];
        sh[c] = sc;
        sh[Cw + c] = beta[c0 + c] - mean * sc;
    }
    __syncthreads();
    const int cols = nvec < GN_THREADS ? nvec : GN_THREADS;
    const int rows = GN_THREADS / cols;
    const int tcol = tid % cols, trow = tid / cols;
    if (trow >= rows) return;
    for (int v = tcol; v < nvec; v += cols) {
        float sc[8], sf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = sh[v * 8 + j]; sf[j] = sh[Cw + v * 8 + j]; }
        int64_t pix = p_begin + trow;
        for (; pix + 3 * rows < p_end; pix += 4 * rows) {
            uint4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) raw[u] = gn_load(x1, x2, C1, C2, (int64_t)b * hw + pix + u * rows, v0 + v);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[8];
                elem_unpack8<DTI>(raw[u], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y = f[j] * sc[j] + sf[j];
                    f[j] = silu ? silu_f(y) : y;
                }
                *reinterpret_cast<uint4*>(out + ((int64_t)b * hw + pix + u * rows) * C + c0 + v * 8) = elem_pack8<DTO>(f);
            }
        }
        for (; pix < p_end; pix += rows) {
            const int64_t gp = (int64_t)b * hw + pix;
            const uint4 raw = gn_load(x1, x2, C1, C2, gp, v0 + v);
            float f[8];
            elem_unpack8<DTI>(raw, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float y = f[j] * sc[j] + sf[j];
                f[j] = silu ? silu_f(y) : y;
            }
            *reinterpret_cast<uint4*>(out + gp * C + c0 + v * 8) = elem_pack8<DTO>(f);
        }
    }
}

// grid (slabs, nb, gchunks): a block owns P pixels x Cw = C/gchunks channels (32/gchunks whole groups), so that
// low-resolution tensors (8x8, 16x16: one or four slabs per image) still fill the chip.
// FUSED (one slab per image: the block sees every pixel of its groups): statistics and apply in ONE launch - the
// second read of x comes from L2.  Otherwise the block writes its (n, mean, M2) partials for gn_apply_kernel.
template <int DTI, int DTO, bool FUSED>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const unsigned short* __restrict__ x1,
                                                              const unsigned short* __restrict__ x2, int C1, int C2,
                                                              int64_t hw, int64_t P, float* __restrict__ partial,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps,
                                                              unsigned short* __restrict__ out, int silu) {
    // LDS: per-(thread row) channel sums, [rows][2][Cw]; summed in a fixed order afterwards (no atomics, so
    // the result is bit-reproducible from call to call)
    extern __shared__ __attribute__((aligned(16))) float sh[];
    __shared__ float part_sum[GN_THREADS][2];
    __shared__ float grp_stat[GN_GROUPS][2];
    const int C = C1 + C2, cpg = C / GN_GROUPS;
    const int gch = gridDim.z, Gw = GN_GROUPS / gch, Cw = C / gch;
    const int v0 = (blockIdx.z * Cw) >> 3, nvec = Cw >> 3;
    const int b = blockIdx.y, slab = blockIdx.x, nslab = gridDim.x;
    const int tid = threadIdx.x;

    const int cols = nvec < GN_THREADS ? nvec : GN_THREADS;
    const int rows = GN_THREADS / cols;
    const int tcol = tid % cols, trow = tid / cols;
    const int64_t p_begin = (int64_t)slab * P;
    const int64_t p_end = (p_begin + P < hw) ? p_begin + P : hw;
    if (trow < rows) {
        float* mine = sh + (size_t)trow * 2 * Cw;
        for (int v = tcol; v < nvec; v += cols) {
            float s[8], ss[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] = 0.f; ss[j] = 0.f; }
            // four pixel rows per trip: the four 16-byte loads are in flight together
            int64_t pix = p_begin + trow;
            for (; pix + 3 * rows < p_end; pix += 4 * rows) {
                uint4 raw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) raw[u] = gn_load(x1, x2, C1, C2, (int64_t)b * hw + pix + u * rows, v0 + v);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float f[8];
                    elem_unpack8<DTI>(raw[u], f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { s[j] += f[j]; ss[j] += f[j] * f[j]; }
                }
            }
            for (; pix < p_end; pix += rows) {
                const uint4 raw = gn_load(x1, x2, C1, C2, (int64_t)b * hw + pix, v0 + v);
                float f[8];
                elem_unpack8<DTI>(raw, f);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s[j] += f[j]; ss[j] += f[j] * f[j]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                mine[v * 8 + j] = s[j];
                mine[Cw + v * 8 + j] = ss[j];
            }
        }
    }
    __syncthreads();
    // per-group totals: GN_THREADS/Gw threads per group each sum a fixed interleaved subset; the partials of one group
    // sit Gw lanes apart, so a butterfly over the lane offsets Gw, 2Gw, ... < 64 adds those of a wave (Gw is a power of
    // two <= 32), and one thread per group adds the four wave totals - a fixed order, bit-reproducible from call to call.
    // (A single thread walking all GN_THREADS/Gw partials cost 7 us of the 10 us an 8 x 8 x 1280 launch took.)
    const int nparts = GN_THREADS / Gw;
    {
        const int g = tid % Gw, part = tid / Gw;
        float S = 0.f, SS = 0.f;
        for (int e = part; e < rows * cpg; e += nparts) {
            const int r = e / cpg, j = e - r * cpg;
            const float* rowp = sh + (size_t)r * 2 * Cw;
            S += rowp[g * cpg + j];
            SS += rowp[Cw + g * cpg + j];
        }
        for (int off = Gw; off < 64; off <<= 1) {
            S += __shfl_xor(S, off);
            SS += __shfl_xor(SS, off);
        }
        if ((tid & 63) < Gw) {
            part_sum[(tid >> 6) * Gw + g][0] = S;
            part_sum[(tid >> 6) * Gw + g][1] = SS;
        }
    }
    __syncthreads();
    if (tid < Gw) {
        float S = 0.f, SS = 0.f;
        for (int w = 0; w < GN_THREADS / 64; ++w) { S += part_sum[w * Gw + tid][0]; SS += part_sum[w * Gw + tid][1]; }
        const float n = (float)((p_end - p_begin) * cpg);
        const float mean = S / n;
        float m2 = SS - S * mean;
        if (m2 < 0.f) m2 = 0.f;
        if (FUSED) {
            grp_stat[tid][0] = mean;
            grp_stat[tid][1] = rsqrtf(m2 / n + eps);
        } else {
            float* o = partial + (((int64_t)b * nslab + slab) * GN_GROUPS + blockIdx.z * Gw + tid) * 3;
            o[0] = n; o[1] = mean; o[2] = m2;
        }
    }
    if (FUSED) {
        __syncthreads();      // grp_stat visible; the per-row sums in `sh` are dead and become the scale/shift tables
        gn_apply_stream<DTI, DTO>(x1, x2, C1, C2, gamma, beta, &grp_stat[0][0], out, hw, p_begin, p_end, b, blockIdx.z * Cw, Cw,
                            cpg, silu, sh);
    }
}

template <int DTI, int DTO>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const unsigned short* __restrict__ x1,
                                                              const unsigned short* __restrict__ x2, int C1, int C2,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ partial, float eps,
                                                              unsigned short* __restrict__ out, int64_t hw, int64_t P,
                                                              int silu) {
    extern __shared__ __attribute__((aligned(16))) float sh[];   // scale[Cw], shift[Cw]
    __shared__ float sub_stat[GN_THREADS][3];
    __shared__ float grp_stat[GN_GROUPS][2];
    const int C = C1 + C2, cpg = C / GN_GROUPS;
    const int gch = gridDim.z, Gw = GN_GROUPS / gch, Cw = C / gch;
    const int c0 = blockIdx.z * Cw;
    const int b = blockIdx.y, slab = blockIdx.x, nslab = gridDim.x;
    const int tid = threadIdx.x;
    // Every block merges the slab partials of its groups itself (fixed order: up to 8 interleaved chains per
    // group, then the chain results in order), which replaces a separate single-block finalize launch.
    const int nsub = GN_THREADS / Gw < 8 ? GN_THREADS / Gw : 8;
    {
        const int g = tid % Gw, sub = tid / Gw;
        float n = 0.f, mean = 0.f, m2 = 0.f;
        if (sub < nsub) {
            // four slabs per trip: their 12 loads are in flight together, then merged in slab order
            for (int base = sub; base < nslab; base += 4 * nsub) {
                float pn[4], pm[4], p2[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int sl = base + u * nsub;
                    const bool ok = sl < nslab;
                    const float* pp = partial + (((int64_t)b * nslab + (ok ? sl : 0)) * GN_GROUPS + blockIdx.z * Gw + g) * 3;
                    pn[u] = ok ? pp[0] : 0.f; pm[u] = pp[1]; p2[u] = pp[2];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) gn_merge(n, mean, m2, pn[u], pm[u], p2[u]);
            }
            sub_stat[sub * Gw + g][0] = n; sub_stat[sub * Gw + g][1] = mean; sub_stat[sub * Gw + g][2] = m2;
        }
    }
    __syncthreads();
    if (tid < Gw) {
        float n = 0.f, mean = 0.f, m2 = 0.f;
        for (int sub = 0; sub < nsub; ++sub)
            gn_merge(n, mean, m2, sub_stat[sub * Gw + tid][0], sub_stat[sub * Gw + tid][1], sub_stat[sub * Gw + tid][2]);
        grp_stat[tid][0] = mean;
        grp_stat[tid][1] = rsqrtf(m2 / n + eps);
    }
    __syncthreads();
    gn_apply_stream<DTI, DTO>(x1, x2, C1, C2, gamma, beta, &grp_stat[0][0], out, hw, p_begin_of(slab, P), p_end_of(slab, P, hw),
                        b, c0, Cw, cpg, silu, sh);
}

template <int DTI, int DTO> constexpr auto gn_stats_fused = gn_stats_kernel<DTI, DTO, true>;
template <int DTI, int DTO> constexpr auto gn_stats_split = gn_stats_kernel<DTI, DTO, false>;

// LayerNorm: LPR lanes per row (a power of two <= 64 chosen by the host so that C/8 vectors split evenly, e.g.
// C = 320 -> 8 lanes x 5 vectors, and a wave covers 64/LPR rows), row held in registers, exact two-pass statistics
// with the reductions inside the LPR-lane group.  Consecutive lanes of a group read consecutive 16-byte vectors.
constexpr int LN_MAXV = 5;   // vectors per lane

template <int DTI, int DTO>
__global__ __launch_bounds__(256) void layernorm_kernel(const unsigned short* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        unsigned short* __restrict__ out, int64_t rows, int C,
                                                        float eps, int lpr) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nvec = C >> 3;
    const int rpw = 64 / lpr;                        // rows per wave
    const int sub = lane / lpr, li = lane - sub * lpr;
    const float inv_c = 1.0f / (float)C;
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wid) * rpw; row0 < rows; row0 += (int64_t)gridDim.x * 4 * rpw) {
        const int64_t row = row0 + sub;
        const bool rok = row < rows;
        const unsigned short* xr = x + (rok ? row : 0) * C;
        float f[LN_MAXV][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int v = li + lpr * i;
            if (v < nvec) {
                const uint4 raw = *reinterpret_cast<const uint4*>(xr + v * 8);
                elem_unpack8<DTI>(raw, f[i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) s += f[i][j];
            }
        }
        for (int o = lpr >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * inv_c;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int v = li + lpr * i;
            if (v < nvec) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; q += d * d; }
            }
        }
        for (int o = lpr >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = rsqrtf(q * inv_c + eps);
        if (!rok) continue;
        unsigned short* orow = out + row * C;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int v = li + lpr * i;
            if (v < nvec) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + v * 8);
                const f32x4 g1 = *reinterpret_cast<const f32x4*>(gamma + v * 8 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + v * 8);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(beta + v * 8 + 4);
                float y[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    y[j] = (f[i][j] - mean) * rstd * g0[j] + b0[j];
                    y[4 + j] = (f[i][4 + j] - mean) * rstd * g1[j] + b1[j];
                }
                *reinterpret_cast<uint4*>(orow + v * 8) = elem_pack8<DTO>(y);
            }
        }
    }
}

}  // namespace

extern "C" int64_t sdmi_groupnorm_workspace_bytes(int32_t nb, int64_t hw) {
    const int64_t P = gn_slab_pixels(hw);
    const int64_t nslab = (hw + P - 1) / P;
    return (int64_t)nb * nslab * GN_GROUPS * 3 * (int64_t)sizeof(float);
}

#ifndef SDMI_GN_FUSED_MAX_HW
#define SDMI_GN_FUSED_MAX_HW 1024
#endif
// Slabs per image groupnorm_launch really uses: one (the fused single launch) up to SDMI_GN_FUSED_MAX_HW pixels.
static int64_t gn_launch_slabs(int64_t hw) {
    if (hw <= SDMI_GN_FUSED_MAX_HW) return 1;
    const int64_t P = gn_slab_pixels(hw);
    return (hw + P - 1) / P;
}

// Argument checks shared by the two GroupNorm entry points (everything except the scratch buffer).
static int groupnorm_check(const void* x1, const void* x2, int32_t C1, int32_t C2, const float* gamma, const float* beta,
                           const void* out, int32_t nb, int64_t hw, int32_t dtype) {
    if (dtype != SDMI_DTYPE_BF16 && dtype != SDMI_DTYPE_F16 && dtype != SDMI_DTYPE_F16_TO_BF16) return SDMI_ERR_BAD_ARG;
    if (x1 == nullptr || gamma == nullptr || beta == nullptr || out == nullptr) return SDMI_ERR_BAD_ARG;
    if (nb <= 0 || hw <= 0 || C1 <= 0 || C2 < 0) return SDMI_ERR_BAD_ARG;
    if ((x2 == nullptr) != (C2 == 0)) return SDMI_ERR_BAD_ARG;
    const int C = C1 + C2;
    if (C1 % 8 != 0 || C2 % 8 != 0 || C % GN_GROUPS != 0) return SDMI_ERR_ALIGNMENT;
    if (2 * C * (int)sizeof(float) > 64 * 1024) return SDMI_ERR_BAD_ARG;
    return SDMI_OK;
}

// The launches of both entry points; `partial` = nb * nslab * 32 * 3 floats of scratch (per-slab group summaries:
// every entry is written by gn_stats_kernel before gn_apply_kernel reads it, so the buffer needs no clearing).
static int groupnorm_launch(const void* x1, const void* x2, int32_t C1, int32_t C2, const float* gamma, const float* beta,
                            void* out, int32_t nb, int64_t hw, float eps, int32_t silu, int32_t dtype, float* partial,
                            void* stream) {
    const int C = C1 + C2;
    // Small images (<= 1024 pixels): one slab per image and one fused launch; else P-pixel slabs and two launches.
    const bool fused = hw <= SDMI_GN_FUSED_MAX_HW;
    const int64_t P = fused ? hw : gn_slab_pixels(hw);
    const int nslab = (int)gn_launch_slabs(hw);
    // channel chunks (whole groups, 16-byte aligned) until the grid can fill the chip
    int gch = 1;
    while (gch < GN_GROUPS && (int64_t)nslab * nb * gch < 512 && (C / (gch * 2)) % 8 == 0) gch *= 2;
    const int Cw = C / gch;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)nslab, (unsigned)nb, (unsigned)gch);
    const size_t lds = 2 * (size_t)Cw * sizeof(float);
    const int nvec_h = Cw / 8;
    const int rows_h = GN_THREADS / (nvec_h < GN_THREADS ? nvec_h : GN_THREADS);
    const unsigned short* u1 = reinterpret_cast<const unsigned short*>(x1);
    const unsigned short* u2 = reinterpret_cast<const unsigned short*>(x2);
    unsigned short* uo = reinterpret_cast<unsigned short*>(out);
    // (input type, output type): bf16 -> bf16, f16 -> f16, or f16 -> bf16 (the f16 residual stream of the bf16 operand policy)
#define SDMI_GN_LAUNCH(KERNEL, LDS, ...)                                                                            \
    do {                                                                                                            \
        if (dtype == SDMI_DTYPE_F16)      hipLaunchKernelGGL((KERNEL<1, 1>), grid, dim3(GN_THREADS), LDS, s, __VA_ARGS__); \
        else if (dtype == SDMI_DTYPE_BF16) hipLaunchKernelGGL((KERNEL<0, 0>), grid, dim3(GN_THREADS), LDS, s, __VA_ARGS__); \
        else                              hipLaunchKernelGGL((KERNEL<1, 0>), grid, dim3(GN_THREADS), LDS, s, __VA_ARGS__); \
    } while (0)
    if (fused) {
        SDMI_GN_LAUNCH(gn_stats_fused, lds * rows_h, u1, u2, C1, C2, hw, P, partial, gamma, beta, eps, uo, silu);
        SDMI_CHECK_LAUNCH();
        return SDMI_OK;
    }
    SDMI_GN_LAUNCH(gn_stats_split, lds * rows_h, u1, u2, C1, C2, hw, P, partial, gamma, beta, eps, uo, silu);
    SDMI_CHECK_LAUNCH();
    SDMI_GN_LAUNCH(gn_apply_kernel, lds, u1, u2, C1, C2, gamma, beta, partial, eps, uo, hw, P, silu);
    SDMI_CHECK_LAUNCH();
#undef SDMI_GN_LAUNCH
    return SDMI_OK;
}

extern "C" int sdmi_groupnorm(const void* x1, const void* x2, int32_t C1, int32_t C2, const float* gamma,
                              const float* beta, void* out, int32_t nb, int64_t hw, float eps, int32_t silu,
                              int32_t dtype, void* workspace, int64_t workspace_bytes, void* stream) {
    const int st = groupnorm_check(x1, x2, C1, C2, gamma, beta, out, nb, hw, dtype);
    if (st != SDMI_OK) return st;
    if (workspace == nullptr || workspace_bytes < sdmi_groupnorm_workspace_bytes(nb, hw)) return SDMI_ERR_WORKSPACE;
    return groupnorm_launch(x1, x2, C1, C2, gamma, beta, out, nb, hw, eps, silu, dtype, reinterpret_cast<float*>(workspace),
                            stream);
}

// ---- sdmi_groupnorm_noalloc: the same GroupNorm for callers that cannot allocate ---------------------------------
// The slab partials live in a file-scope device array instead of a caller workspace.  Each GPU holds its own copy
// of the array at its own address: the address is looked up once per device with hipGetSymbolAddress (on the first
// call made with that device current) and only that pointer - never the array's name - is handed to the kernels.
constexpr int64_t GN_NOALLOC_MAX_SLABS = 4096;                                  // nb * slabs-per-image
constexpr int64_t GN_NOALLOC_FLOATS = GN_NOALLOC_MAX_SLABS * GN_GROUPS * 3;     // 1.5 MiB per GPU
static __device__ __align__(256) float scratch[GN_NOALLOC_FLOATS];

constexpr int GN_MAX_DEVICES = 64;
static std::mutex scratch_mutex;
static float* scratch_ptr[GN_MAX_DEVICES] = {};

static_assert(GN_NOALLOC_FLOATS * (int64_t)sizeof(float) == SDMI_GROUPNORM_NOALLOC_MAX_BYTES, "limit stated in sdmi.h");

extern "C" int sdmi_groupnorm_noalloc(const void* x1, const void* x2, int32_t C1, int32_t C2, const float* gamma,
                                      const float* beta, void* out, int32_t nb, int64_t hw, float eps, int32_t silu,
                                      int32_t dtype, void* stream) {
    const int st = groupnorm_check(x1, x2, C1, C2, gamma, beta, out, nb, hw, dtype);
    if (st != SDMI_OK) return st;
    // the limit counts the slab entries the launch really writes (one per image for the fused small-image form)
    if ((int64_t)nb * gn_launch_slabs(hw) > GN_NOALLOC_MAX_SLABS) return SDMI_ERR_TOO_LARGE;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GN_MAX_DEVICES) return SDMI_ERR_LAUNCH;
    float* ptr;
    {
        std::lock_guard<std::mutex> lock(scratch_mutex);
        if (scratch_ptr[dev] == nullptr) {
            void* sym = nullptr;
            if (hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)) != hipSuccess || sym == nullptr) return SDMI_ERR_LAUNCH;
            scratch_ptr[dev] = reinterpret_cast<float*>(sym);
        }
        ptr = scratch_ptr[dev];
    }
    return groupnorm_launch(x1, x2, C1, C2, gamma, beta, out, nb, hw, eps, silu, dtype, ptr, stream);
}

namespace {

// (sum, sum of squares) of every row: one wave per row, 16 bytes per lane and trip.
template <int DT>
__global__ __launch_bounds__(256) void row_stats_kernel(const unsigned short* __restrict__ x, float* __restrict__ stats,
                                                        int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const unsigned short* xr = x + row * C;
    float s1 = 0.f, s2 = 0.f;
    for (int v = lane; v < (C >> 3); v += 64) {
        float f[8];
        elem_unpack8<DT>(*reinterpret_cast<const uint4*>(xr + v * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s1 += f[j];
            s2 += f[j] * f[j];
        }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (lane == 0) *reinterpret_cast<f32x2*>(stats + 2 * row) = (f32x2){s1, s2};
}

}  // namespace

extern "C" int sdmi_row_stats(const void* x, float* stats, int64_t rows, int32_t C, int32_t dtype, void* stream) {
    if (dtype != SDMI_DTYPE_BF16 && dtype != SDMI_DTYPE_F16) return SDMI_ERR_BAD_ARG;
    if (x == nullptr || stats == nullptr || rows <= 0 || C <= 0 || (rows + 3) / 4 > 0x7fffffffLL) return SDMI_ERR_BAD_ARG;
    if (C % 8 != 0) return SDMI_ERR_ALIGNMENT;
    const unsigned short* xi = reinterpret_cast<const unsigned short*>(x);
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == SDMI_DTYPE_F16) hipLaunchKernelGGL(row_stats_kernel<1>, grid, dim3(256), 0, hs, xi, stats, rows, C);
    else                         hipLaunchKernelGGL(row_stats_kernel<0>, grid, dim3(256), 0, hs, xi, stats, rows, C);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

extern "C" int sdmi_layernorm(const void* x, const float* gamma, const float* beta, void* out, int64_t rows,
                              int32_t C, float eps, int32_t dtype, void* stream) {
    if (dtype != SDMI_DTYPE_BF16 && dtype != SDMI_DTYPE_F16 && dtype != SDMI_DTYPE_F16_TO_BF16) return SDMI_ERR_BAD_ARG;
    if (x == nullptr || gamma == nullptr || beta == nullptr || out == nullptr) return SDMI_ERR_BAD_ARG;
    if (rows <= 0 || C <= 0) return SDMI_ERR_BAD_ARG;
    if (C % 8 != 0 || C > 64 * LN_MAXV * 8) return SDMI_ERR_ALIGNMENT;
    // lanes per row: the smallest power of two that holds the row in LN_MAXV vectors per lane
    const int nvec = C / 8;
    int lpr = 1;
    while (lpr < 64 && nvec > lpr * LN_MAXV) lpr *= 2;
    const int rpw = 64 / lpr;
    int64_t blocks = (rows + 4 * rpw - 1) / (4 * rpw);
    if (blocks > 16384) blocks = 16384;
    const unsigned short* xi = reinterpret_cast<const unsigned short*>(x);
    unsigned short* xo = reinterpret_cast<unsigned short*>(out);
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SDMI_DTYPE_F16)       hipLaunchKernelGGL((layernorm_kernel<1, 1>), dim3((unsigned)blocks), dim3(256), 0, hs, xi, gamma, beta, xo, rows, C, eps, lpr);
    else if (dtype == SDMI_DTYPE_BF16) hipLaunchKernelGGL((layernorm_kernel<0, 0>), dim3((unsigned)blocks), dim3(256), 0, hs, xi, gamma, beta, xo, rows, C, eps, lpr);
    else                               hipLaunchKernelGGL((layernorm_kernel<1, 0>), dim3((unsigned)blocks), dim3(256), 0, hs, xi, gamma, beta, xo, rows, C, eps, lpr);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}
