// sdmi_gemm: 16-bit (bf16 | f16) MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
//   out[m, n] = epilogue(alpha * sum_k A[m, k] * W[n, k])          (see include/sdmi.h)
//
// One kernel template (sdmi_gemm_kernel.h) serves nn.Linear, conv1x1 and conv3x3 (stride 1/2, optional fused
// nearest-x2 gather, optional two-pointer channel concat): the modes differ only in how a tile of A is gathered.
// Both operands are K-contiguous, so both MFMA fragments are plain 16-byte LDS reads.
//
// Tiling: 8 waves (4 x 2) per workgroup, block tile (64*MI) x (32*TN) x 64 with MI in {4, 2} (256 | 128 rows) and
// TN in {5, 4} (160 | 128 columns); a wave owns MI x TN v_mfma_f32_16x16x32 accumulators.  The MFMA "A" operand is
// the W tile and the "B" operand the activation tile, so an accumulator quad holds 4 CONSECUTIVE output columns of
// one output row: epilogue loads / stores are 8-16 byte vectors.
// LDS: 3-stage ring (2 stages for the short-K variant that runs two workgroups per CU) of XOR-swizzled 16-byte
// chunks (conflict-free ds_read_b128), filled by global_load_lds_dwordx4 (LDS-DMA: no VGPR round trip, no
// ds_write) and retired with counted s_waitcnt vmcnt(N), one barrier per k-tile, half-tile software pipeline.
// The gather of the implicit GEMM is expressed in the per-lane SOURCE address (zero taps read a zero constant).
// Workgroup ids are remapped so that each XCD (private L2) walks a contiguous, group-M-swizzled range of tiles.
// This file: the split-K reduce kernel and the C entry point; the kernel template is instantiated per
// (mode, dtype) in sdmi_gemm_m?_*.hip so that the units compile in parallel.
#include <mutex>
#include <cstdlib>
#include "sdmi_gemm_kernel.h"

namespace {

// Second pass of split-K: sum the slabs and run the same epilogue.  One thread per output quad.
template <int DT>
__global__ __launch_bounds__(256) void sdmi_gemm_splitk_reduce(const sdmi_gemm_args p, int64_t ldw) {
    const int64_t n_w = p.geglu ? 2 * p.N : p.N;
    const int64_t quads_per_row = (n_w + 3) / 4;
    const int64_t total = p.M * quads_per_row;
    const float* ws = reinterpret_cast<const float*>(p.workspace);
    const int64_t slab = p.M * ldw;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = idx / quads_per_row;
        const int64_t wcol = (idx - m * quads_per_row) * 4;
        if (p.geglu && (wcol & 16)) continue;   // gate quads are consumed with their value quad
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f}, g = v;
        for (int s = 0; s < p.split_k; ++s) {
            v += *reinterpret_cast<const f32x4*>(ws + s * slab + m * ldw + wcol);
            if (p.geglu) g += *reinterpret_cast<const f32x4*>(ws + s * slab + m * ldw + wcol + 16);
        }
        v *= p.alpha;
        int64_t n = wcol;
        if (p.geglu) {
            g *= p.alpha;
            n = (wcol >> 5) * 16 + (wcol & 15);
            if (p.bias != nullptr) {
                v += *reinterpret_cast<const f32x4*>(p.bias + wcol);
                g += *reinterpret_cast<const f32x4*>(p.bias + wcol + 16);
            }
            if (p.out_kind != 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= gelu_erf_precise(g[j]);
            } else {
                v *= gelu_erf_f4(g);
            }
        } else if (p.bias != nullptr && !p.bias_per_row) {
            for (int j = 0; j < 4 && n + j < p.N; ++j) v[j] += p.bias[n + j];
        }
        if (n >= p.N) continue;
        store_quad<DT>(p, m, n, v, 0, 0);
    }
}

// The reduce pass of the token-stream launches (ln_colsum and / or row_stats_out): a workgroup owns 256 quads (1024 weight
// columns) of ONE output row - blockIdx.x = row, blockIdx.y = column chunk - so the row's statistics are one uniform read
// (folded LayerNorm in) and its sums one block reduction away (row statistics out: slot = chunk).
constexpr int REDUCE_ROWS_CHUNK = 1024;
template <int DT>
__global__ __launch_bounds__(256) void sdmi_gemm_splitk_reduce_rows(const sdmi_gemm_args p, int64_t ldw) {
    __shared__ float part[8];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t m = blockIdx.x;
    const int64_t n_w = p.geglu ? 2 * p.N : p.N;
    const int64_t wcol = (int64_t)blockIdx.y * REDUCE_ROWS_CHUNK + 4 * threadIdx.x;      // (host-checked: N_w % 4 == 0)
    const float* ws = reinterpret_cast<const float*>(p.workspace);
    const int64_t slab = p.M * ldw;
    float mean = 0.f, rstd = 1.f;
    if (p.ln_colsum != nullptr) {      // (uniform) lane l fetches slot l of the row, a butterfly adds them up: one round trip
        f32x2 t = (f32x2){0.f, 0.f};
        if (lane < p.ln_stats_slots) t = reinterpret_cast<const f32x2*>(p.ln_stats)[m * p.ln_stats_slots + lane];
        float s1 = t[0], s2 = t[1];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            s1 += __shfl_xor(s1, off);
            s2 += __shfl_xor(s2, off);
        }
        lnf_mean_rstd(s1, s2, 1.0f / (float)(p.k_valid > 0 ? p.k_valid : p.K), p.ln_eps, mean, rstd);
    }
    float st1 = 0.f, st2 = 0.f;
    if (wcol < n_w && !(p.geglu && (wcol & 16))) {
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f}, g = v;
        for (int s = 0; s < p.split_k; ++s) {
            v += *reinterpret_cast<const f32x4*>(ws + s * slab + m * ldw + wcol);
            if (p.geglu) g += *reinterpret_cast<const f32x4*>(ws + s * slab + m * ldw + wcol + 16);
        }
        if (p.ln_colsum != nullptr) {
            v = (v - mean * *reinterpret_cast<const f32x4*>(p.ln_colsum + wcol)) * rstd;
            if (p.geglu) g = (g - mean * *reinterpret_cast<const f32x4*>(p.ln_colsum + wcol + 16)) * rstd;
        }
        v *= p.alpha;
        int64_t n = wcol;
        if (p.geglu) {
            g *= p.alpha;
            n = (wcol >> 5) * 16 + (wcol & 15);
            if (p.bias != nullptr) {
                v += *reinterpret_cast<const f32x4*>(p.bias + wcol);
                g += *reinterpret_cast<const f32x4*>(p.bias + wcol + 16);
            }
            if (p.out_kind != 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= gelu_erf_precise(g[j]);
            } else {
                v *= gelu_erf_f4(g);
            }
        } else if (p.bias != nullptr) {
            for (int j = 0; j < 4 && n + j < p.N; ++j) v[j] += p.bias[n + j];
        }
        if (n < p.N) {
            v = store_quad<DT>(p, m, n, v, 0, 0);
            for (int j = 0; j < 4 && n + j < p.N; ++j) {
                st1 += v[j];
                st2 += v[j] * v[j];
            }
        }
    }
    if (p.row_stats_out != nullptr) {      // (uniform)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            st1 += __shfl_xor(st1, off);
            st2 += __shfl_xor(st2, off);
        }
        if (lane == 0) { part[wid] = st1; part[4 + wid] = st2; }
        __syncthreads();
        if (threadIdx.x == 0)
            reinterpret_cast<f32x2*>(p.row_stats_out)[m * p.row_stats_slots + blockIdx.y] =
                (f32x2){(part[0] + part[1]) + (part[2] + part[3]), (part[4] + part[5]) + (part[6] + part[7])};
    }
}

}  // namespace


#define SDMI_GEMM_UNIT_DECL(NAME)      \
    int sdmi_gemm_unit_attrs_##NAME(); \
    void sdmi_gemm_unit_launch_##NAME(int wide, int variant, dim3 grid, hipStream_t s, const sdmi_gemm_args& p);
SDMI_GEMM_UNIT_DECL(m0_bf16) SDMI_GEMM_UNIT_DECL(m0_f16) SDMI_GEMM_UNIT_DECL(m1_bf16) SDMI_GEMM_UNIT_DECL(m1_f16)
#undef SDMI_GEMM_UNIT_DECL

int sdmi_gemm_unit_attrs_m0_f16_lnf();                                   // sdmi_gemm_lnf.hip (LayerNorm folded, mode 0, f16)
void sdmi_gemm_unit_launch_m0_f16_lnf(int wide, int variant, dim3 grid, hipStream_t s, const sdmi_gemm_args& p);
int sdmi_gemm_wide_attrs_f16_lnf();
int sdmi_gemm_small_attrs_m0_bf16();                                     // the 64-column tile (mode 0): sdmi_gemm_m0_*.hip, sdmi_gemm_lnf.hip
int sdmi_gemm_small_attrs_m0_f16();
int sdmi_gemm_small_attrs_m0_f16_lnf();
void sdmi_gemm_small_launch_m0_bf16(int rows64, dim3 grid, hipStream_t s, const sdmi_gemm_args& p);
void sdmi_gemm_small_launch_m0_f16(int rows64, dim3 grid, hipStream_t s, const sdmi_gemm_args& p);
void sdmi_gemm_small_launch_m0_f16_lnf(int rows64, dim3 grid, hipStream_t s, const sdmi_gemm_args& p);
int sdmi_gemm_small_conv_attrs_m1_bf16();                                // conv3x3 on the 128 x 64 block: sdmi_gemm_m1_*.hip
int sdmi_gemm_small_conv_attrs_m1_f16();
void sdmi_gemm_small_conv_launch_m1_bf16(dim3 grid, hipStream_t s, const sdmi_gemm_args& p);
void sdmi_gemm_small_conv_launch_m1_f16(dim3 grid, hipStream_t s, const sdmi_gemm_args& p);
void sdmi_gemm_wide_launch_f16_lnf(int mode, dim3 grid, hipStream_t s, const sdmi_gemm_args& p);
int sdmi_gemm_wide_attrs_bf16();                                         // sdmi_gemm_wide_*.hip
int sdmi_gemm_wide_attrs_f16();
void sdmi_gemm_wide_launch_bf16(int mode, dim3 grid, hipStream_t s, const sdmi_gemm_args& p);
void sdmi_gemm_wide_launch_f16(int mode, dim3 grid, hipStream_t s, const sdmi_gemm_args& p);
int sdmi_gemm_panel_launch(const sdmi_gemm_args& p, hipStream_t s);      // sdmi_gemm_panel.hip
int sdmi_conv_halo_launch(const sdmi_gemm_args& p, hipStream_t s);       // sdmi_conv_halo.hip
int sdmi_conv_narrow_launch(const sdmi_gemm_args& p, hipStream_t s);     // sdmi_conv_narrow.hip

// Row tile the launcher picks: 256 rows once that still gives every CU a block, else 128 rows.
static int pick_tile_m(int64_t M, int64_t N_w, int tile_n, int64_t problems) {
    if (tile_n == 64)       // the small tile: 128 rows once that gives every CU a block, else 64
        return ((M + 127) / 128) * ((N_w + 63) / 64) * problems >= 256 ? 128 : 64;
    const int bn = tile_n == 160 ? 160 : 128;
    const int64_t tiles = ((M + 255) / 256) * ((N_w + bn - 1) / bn) * problems;
    return tiles >= 256 ? 256 : 128;
}

extern "C" int32_t sdmi_gemm_tile_m(int64_t M, int64_t N_w, int32_t tile_n) { return pick_tile_m(M, N_w, tile_n, 1); }

static int64_t slab_ld(int64_t N_w, int tile_n) { return (N_w + tile_n - 1) / tile_n * tile_n; }

extern "C" int64_t sdmi_gemm_workspace_bytes(int64_t M, int64_t N_w, int32_t split_k) {
    if (split_k <= 1) return 0;
    // enough for any column tile (128, 160 or 256)
    int64_t ldw = slab_ld(N_w, 128) > slab_ld(N_w, 160) ? slab_ld(N_w, 128) : slab_ld(N_w, 160);
    if (slab_ld(N_w, 256) > ldw) ldw = slab_ld(N_w, 256);
    return (int64_t)split_k * M * ldw * (int64_t)sizeof(float);
}

// The 256 x 256 x 32 tile (sdmi_gemm_wide.h) runs whole k-tiles from plain row / tap pointers and has two epilogues: the
// 16-bit fast path and the split-K slabs.
static bool wide_applies(const sdmi_gemm_args& p) {
    const int batch = p.batch > 0 ? p.batch : 1;
    const int split = p.split_k > 0 ? p.split_k : 1;
    if (p.ln_gamma != nullptr || p.algo == 2) return false;
    if (p.tile_m != 0 && p.tile_m != 256) return false;
    if (p.stages != 0 && p.stages != 3) return false;
    if (p.K % 64 != 0) return false;
    if (p.mode == 0) {
        if (p.k_valid != 0 && p.k_valid != p.K) return false;
    } else {
        if (p.Cin % 64 != 0 || p.Cin > ZERO_ELEMS || p.K != 9 * (int64_t)p.Cin) return false;
    }
    if (split > 1) return true;
    if (p.geglu) {      // the 16-bit GEGLU epilogue: nothing to add, whole 32-column output runs, no batching
        return p.out_kind == 0 && p.res == nullptr && p.emb == nullptr && !p.bias_per_row && p.act == 0 && (p.N & 31) == 0 &&
               (p.ld_out & 7) == 0 && (((uintptr_t)p.out) & 15) == 0 && batch == 1;
    }
    if (p.out_kind != 0 || (p.N & 7) != 0 || p.bias_per_row || (p.ld_out & 7) != 0) return false;
    if (p.res != nullptr && ((p.ld_res & 7) != 0 || p.res_f32)) return false;
    if ((((uintptr_t)p.out | (uintptr_t)p.res) & 15) != 0) return false;
    if (batch > 1 && (((p.stride_out | p.stride_res) & 7) != 0)) return false;
    return true;
}

extern "C" int sdmi_gemm_wide_applies(const sdmi_gemm_args* a) { return a != nullptr && wide_applies(*a) ? 1 : 0; }

// Slots per row a launch with row_stats_out writes: one per (column tile, wave column) of the tiled kernel, one per
// 64-column run of the 256 x 256 x 32 tile, one per 1024 columns of the split-K reduce pass.
static int row_stats_slots(const sdmi_gemm_args& p) {
    if (p.split_k > 1) return (int)((p.N + REDUCE_ROWS_CHUNK - 1) / REDUCE_ROWS_CHUNK);      // (the row-wise reduce pass)
    const int BN = p.tile_n == 0 ? 128 : p.tile_n;
    const int64_t tiles_n = (p.N + BN - 1) / BN;
    return (int)(tiles_n * (BN == 256 ? 4 : 2));
}

extern "C" int32_t sdmi_gemm_row_stats_slots(const sdmi_gemm_args* a) { return a == nullptr ? 0 : row_stats_slots(*a); }

extern "C" int sdmi_gemm(const sdmi_gemm_args* a, void* stream) {
    if (a == nullptr || a->A == nullptr || a->W == nullptr || a->out == nullptr) return SDMI_ERR_BAD_ARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return SDMI_ERR_BAD_ARG;
    if (a->K % BK != 0) return SDMI_ERR_ALIGNMENT;
    if (a->mode != 0 && a->mode != 1) return SDMI_ERR_BAD_ARG;
    if (a->out_kind < 0 || a->out_kind > 2) return SDMI_ERR_BAD_ARG;
    if (a->act < 0 || a->act > 2) return SDMI_ERR_BAD_ARG;
    if (a->dtype != SDMI_DTYPE_BF16 && a->dtype != SDMI_DTYPE_F16) return SDMI_ERR_BAD_ARG;
    const int batch = a->batch > 0 ? a->batch : 1;
    const int split = a->split_k > 0 ? a->split_k : 1;
    if (batch > 1 && split > 1) return SDMI_ERR_BAD_ARG;
    if (a->geglu && (a->N % 16 != 0)) return SDMI_ERR_ALIGNMENT;
    if ((a->emb != nullptr || a->out_kind == 2) && a->rows_per_batch <= 0) return SDMI_ERR_BAD_ARG;
    if (a->emb != nullptr && (a->ld_emb % 4 != 0)) return SDMI_ERR_ALIGNMENT;
    if (a->res != nullptr && (a->ld_res % 4 != 0)) return SDMI_ERR_ALIGNMENT;
    if (a->res_f32 && a->out_kind == 0) return SDMI_ERR_BAD_ARG;      // an fp32 residual goes with an fp32 output
    if (a->out_kind != 2 && (a->ld_out % 4 != 0)) return SDMI_ERR_ALIGNMENT;
    if (a->mode == 0) {
        if (a->lda % 8 != 0) return SDMI_ERR_ALIGNMENT;
        if (a->k_valid < 0 || a->k_valid > a->K || a->k_valid % 8 != 0) return SDMI_ERR_ALIGNMENT;
        if (a->A2 != nullptr) {
            if (a->lda2 % 8 != 0 || a->k_split % BK != 0 || a->k_split <= 0 || a->k_split >= a->K)
                return SDMI_ERR_ALIGNMENT;
        }
    } else {
        if (a->Cin <= 0 || a->Cin % 8 != 0) return SDMI_ERR_ALIGNMENT;
        if (a->H <= 0 || a->W_in <= 0 || a->Ho <= 0 || a->Wo <= 0) return SDMI_ERR_BAD_ARG;
        if (a->stride != 1 && a->stride != 2) return SDMI_ERR_BAD_ARG;
        if (a->upsample != 0 && a->upsample != 1) return SDMI_ERR_BAD_ARG;
        if (a->M % ((int64_t)a->Ho * a->Wo) != 0) return SDMI_ERR_BAD_ARG;
        if (a->K < 9 * (int64_t)a->Cin) return SDMI_ERR_BAD_ARG;
        // output geometry must match the (possibly upsampled) input
        const int Hl = a->H << a->upsample, Wl = a->W_in << a->upsample;
        const int padsum = a->asym_pad ? 1 : 2;
        if (a->asym_pad != 0 && a->asym_pad != 1) return SDMI_ERR_BAD_ARG;
        if ((Hl + padsum - 3) / a->stride + 1 != a->Ho || (Wl + padsum - 3) / a->stride + 1 != a->Wo) return SDMI_ERR_BAD_ARG;
        if (a->A2 != nullptr) return SDMI_ERR_BAD_ARG;
    }
    sdmi_gemm_args p = *a;
    p.batch = batch;
    p.split_k = split;
    const int64_t n_w = p.geglu ? 2 * p.N : p.N;
    const int BN = p.tile_n == 0 ? 128 : p.tile_n;
    if (BN != 64 && BN != 128 && BN != 160 && BN != 256) return SDMI_ERR_BAD_ARG;
    if (BN == 160 && p.geglu) return SDMI_ERR_BAD_ARG;      // GEGLU pairs 16-column blocks inside a wave
    if (BN == 64) {
        // the small tile: mode 0, 64 columns x 128 | 64 rows, three stages; no GEGLU (a wave holds two 16-column blocks),
        // no LayerNorm on resident rows (the panel kernel's)
        if (p.geglu || p.ln_gamma != nullptr || p.algo == 2 || p.stages == 2) return SDMI_ERR_BAD_ARG;
        if (p.tile_m != 0 && p.tile_m != 64 && p.tile_m != 128) return SDMI_ERR_BAD_ARG;
        if (p.mode == 1 && p.tile_m == 64) return SDMI_ERR_BAD_ARG;      // (the conv loader stages two 64-row pieces of A)
    } else if (p.tile_m != 0 && p.tile_m != 128 && p.tile_m != 256) return SDMI_ERR_BAD_ARG;
    if (p.stages != 0 && p.stages != 2 && p.stages != 3) return SDMI_ERR_BAD_ARG;
    if (BN == 256 && !wide_applies(p)) return SDMI_ERR_BAD_ARG;
    const int BM = BN == 256 ? 256 : p.tile_m != 0 ? p.tile_m : (BN == 64 && p.mode == 1) ? 128 : pick_tile_m(p.M, n_w, BN, batch);
    const int64_t tiles_m = (p.M + BM - 1) / BM, tiles_n = (n_w + BN - 1) / BN;
    if (tiles_m * tiles_n > 0x7fffffffLL) return SDMI_ERR_BAD_ARG;
    if (split > 1) {
        if (split > p.K / BK) return SDMI_ERR_BAD_ARG;
        if (p.workspace == nullptr || p.workspace_bytes < (int64_t)split * p.M * slab_ld(n_w, BN) * (int64_t)sizeof(float))
            return SDMI_ERR_WORKSPACE;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (p.algo < 0 || p.algo > 2) return SDMI_ERR_BAD_ARG;
    if ((p.out_f16 | p.res_f16 | p.a_f16) != 0) {
        // the f16 residual stream of the bf16 operand policy
        if (p.dtype != SDMI_DTYPE_BF16 || (p.out_f16 & ~1) || (p.res_f16 & ~1) || (p.a_f16 & ~1)) return SDMI_ERR_BAD_ARG;
        if (p.out_f16 && p.out_kind != 0) return SDMI_ERR_BAD_ARG;
        if (p.res_f16 && (p.res == nullptr || p.res_f32)) return SDMI_ERR_BAD_ARG;
        if (p.a_f16 && p.ln_gamma == nullptr) return SDMI_ERR_BAD_ARG;      // only a LayerNorm converts the operand
        if (p.out_f16 && p.geglu) return SDMI_ERR_BAD_ARG;                  // a GEGLU output is an operand, never the stream
    }
    if (p.dup_rows != 0) {
        // the second copy of every row is written by epilogue_rows16 only: the conditions under which every kernel
        // (tiled, 256 x 256 x 32, panel without GEGLU) takes that epilogue
        if (p.dup_rows != 1 || batch > 1 || split > 1 || p.geglu || p.out_kind != 0 || (p.N & 7) != 0 || p.N <= 16 ||
            p.bias_per_row || ((p.ld_out | p.ld_res) & 7) != 0 || p.res_f32 ||
            (((uintptr_t)p.out | (uintptr_t)p.res) & 15) != 0)
            return SDMI_ERR_BAD_ARG;
    }
    if (p.ln_colsum != nullptr || p.row_stats_out != nullptr) {
        // the token-stream launches (folded LayerNorm in, row statistics out): mode 0 / f16 instantiations of their own
        if (p.dtype != SDMI_DTYPE_F16 || p.mode != 0 || p.A2 != nullptr || batch > 1 || (n_w & 3) != 0 || p.stages == 2 ||
            p.algo == 2 || p.ln_gamma != nullptr || p.ln_beta != nullptr || p.emb != nullptr || p.bias_per_row)
            return SDMI_ERR_BAD_ARG;
        if (p.ln_colsum != nullptr && (p.ln_stats == nullptr || p.ln_stats_slots < 1 || p.ln_stats_slots > 48 ||
                                       ((p.ln_stats_slots & 1) && p.ln_stats_slots > 4) || (((uintptr_t)p.ln_stats) & 15) != 0 || !(p.ln_eps >= 0.f)))
            return SDMI_ERR_BAD_ARG;
        if (p.ln_colsum == nullptr && p.ln_stats != nullptr) return SDMI_ERR_BAD_ARG;
        if (p.geglu && p.ln_colsum == nullptr) return SDMI_ERR_BAD_ARG;       // (a GEGLU output is never the stream)
        if (p.row_stats_out != nullptr) {
            // written by the fast 16-bit epilogue (or the row-wise reduce pass): its conditions, as for dup_rows
            if (p.geglu || p.out_kind != 0 || (p.N & 7) != 0 || p.N <= 16 || ((p.ld_out | p.ld_res) & 7) != 0 || p.res_f32 ||
                (((uintptr_t)p.out | (uintptr_t)p.res) & 15) != 0 || p.row_stats_slots != row_stats_slots(p) ||
                p.row_stats_slots > 48)      // (what a LayerNorm-folded consumer can read)
                return SDMI_ERR_BAD_ARG;
        }
    }
    if (p.ln_gamma != nullptr && (p.ln_beta == nullptr || p.algo == 1 || !(p.ln_eps >= 0.f))) return SDMI_ERR_BAD_ARG;
    if (p.ln_gamma == nullptr && p.ln_beta != nullptr) return SDMI_ERR_BAD_ARG;
    if (p.algo == 0 && p.mode == 1 && p.ln_gamma == nullptr) {      // conv3x3 with <= 16 output channels: the byte-bound narrow kernel
        const int st = sdmi_conv_narrow_launch(p, s);
        if (st != 0) return st < 0 ? st : SDMI_OK;
    }
    // The A-panel-stationary kernel (sdmi_gemm_panel.h): picked when a LayerNorm rides along (it is the kernel that holds
    // whole rows), or on request (algo 2).  Without the LayerNorm the tiled kernel's 256-row tile is the faster one on the
    // same problems (65536 x 320 x 320 + residual: 34 us against 40; N = 960: 70 against 79), so algo 0 stays there.
#ifndef SDMI_PANEL_AUTO_ALWAYS
#define SDMI_PANEL_AUTO_ALWAYS 0      // A/B: 1 = algo 0 takes the panel kernel wherever it applies (the round-2 mid-state)
#endif
    if (SDMI_PANEL_AUTO_ALWAYS && p.algo == 0 && p.ln_gamma == nullptr) {
        const int st = sdmi_gemm_panel_launch(p, s);
        if (st != 0) return st < 0 ? st : SDMI_OK;
    }
    if (p.algo == 2 || (p.algo == 0 && p.ln_gamma != nullptr)) {
        const int st = sdmi_gemm_panel_launch(p, s);
        if (st != 0) return st < 0 ? st : SDMI_OK;
        return SDMI_ERR_BAD_ARG;      // (the tiled kernel never holds a whole row)
    }
    if (p.ln_gamma != nullptr) return SDMI_ERR_BAD_ARG;
    // dynamic-LDS attribute of every instantiation, once per DEVICE (SdmiPerDeviceOnce)
    static SdmiPerDeviceOnce attr_once;
    const int attr_status = attr_once.run([] {
        int (*const fns[4])() = {sdmi_gemm_unit_attrs_m0_bf16, sdmi_gemm_unit_attrs_m0_f16, sdmi_gemm_unit_attrs_m1_bf16,
                                 sdmi_gemm_unit_attrs_m1_f16};
        bool ok = true;
        for (int i = 0; i < 4; ++i) ok = (fns[i]() == SDMI_OK) && ok;
        ok = (sdmi_gemm_unit_attrs_m0_f16_lnf() == SDMI_OK) && (sdmi_gemm_wide_attrs_f16_lnf() == SDMI_OK) && ok;
        ok = (sdmi_gemm_small_attrs_m0_bf16() == SDMI_OK) && (sdmi_gemm_small_attrs_m0_f16() == SDMI_OK) &&
             (sdmi_gemm_small_attrs_m0_f16_lnf() == SDMI_OK) && (sdmi_gemm_small_conv_attrs_m1_bf16() == SDMI_OK) &&
             (sdmi_gemm_small_conv_attrs_m1_f16() == SDMI_OK) && ok;
        return (sdmi_gemm_wide_attrs_bf16() == SDMI_OK) && (sdmi_gemm_wide_attrs_f16() == SDMI_OK) && ok;
    });
    if (attr_status != SDMI_OK) return attr_status;
    const dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)(split > 1 ? split : batch));
    const bool f16 = p.dtype == SDMI_DTYPE_F16;
    const int wide = BN == 160, tall = BM == 256;
    const int variant = tall ? 1 : (p.stages == 2 ? 2 : 0);
    // conv3x3 on whole row strips of a 16 | 32 | 64-pixel-wide image: the halo-tile kernel (sdmi_conv_halo.hip) where it applies
    // (SDMI_CONV_HALO=0: development switch back to the tiled kernel)
    static const bool halo_on = [] { const char* e = getenv("SDMI_CONV_HALO"); return e == nullptr || e[0] != '0'; }();
    int halo_status = 0;
    if (halo_on && p.mode == 1 && BN == 160 && BM == 256 && p.algo == 0) {
        sdmi_gemm_args ph = p;
        ph.tile_n = BN; ph.tile_m = BM;
        halo_status = sdmi_conv_halo_launch(ph, s);
        if (halo_status < 0) return halo_status;
    }
    if (halo_status == 1) {
    } else if (BN == 64 && p.mode == 1) {
        (f16 ? sdmi_gemm_small_conv_launch_m1_f16 : sdmi_gemm_small_conv_launch_m1_bf16)(grid, s, p);
    } else if (BN == 64) {
        if (p.ln_colsum != nullptr || p.row_stats_out != nullptr) sdmi_gemm_small_launch_m0_f16_lnf(BM == 64, grid, s, p);
        else (f16 ? sdmi_gemm_small_launch_m0_f16 : sdmi_gemm_small_launch_m0_bf16)(BM == 64, grid, s, p);
    } else if (p.ln_colsum != nullptr || p.row_stats_out != nullptr) {
        if (BN == 256) sdmi_gemm_wide_launch_f16_lnf(0, grid, s, p);
        else           sdmi_gemm_unit_launch_m0_f16_lnf(wide, tall ? 1 : 0, grid, s, p);
    }
    else if (BN == 256)   (f16 ? sdmi_gemm_wide_launch_f16 : sdmi_gemm_wide_launch_bf16)(p.mode, grid, s, p);
    else if (p.mode == 0) (f16 ? sdmi_gemm_unit_launch_m0_f16 : sdmi_gemm_unit_launch_m0_bf16)(wide, variant, grid, s, p);
    else                  (f16 ? sdmi_gemm_unit_launch_m1_f16 : sdmi_gemm_unit_launch_m1_bf16)(wide, variant, grid, s, p);
    SDMI_CHECK_LAUNCH();
    if (split > 1 && (p.ln_colsum != nullptr || p.row_stats_out != nullptr)) {
        if (p.M > 0x7fffffffLL) return SDMI_ERR_BAD_ARG;      // (one workgroup row per output row)
        const dim3 rgrid((unsigned)p.M, (unsigned)((n_w + REDUCE_ROWS_CHUNK - 1) / REDUCE_ROWS_CHUNK));
        hipLaunchKernelGGL(sdmi_gemm_splitk_reduce_rows<1>, rgrid, dim3(256), 0, s, p, tiles_n * BN);
        SDMI_CHECK_LAUNCH();
    } else if (split > 1) {
        const int64_t quads = p.M * ((n_w + 3) / 4);
        const int64_t blocks = (quads + 255) / 256;
        const dim3 rgrid((unsigned)(blocks < 4096 ? blocks : 4096));
        if (f16) hipLaunchKernelGGL(sdmi_gemm_splitk_reduce<1>, rgrid, dim3(256), 0, s, p, tiles_n * BN);
        else     hipLaunchKernelGGL(sdmi_gemm_splitk_reduce<0>, rgrid, dim3(256), 0, s, p, tiles_n * BN);
        SDMI_CHECK_LAUNCH();
    }
    return SDMI_OK;
}
