// sdmi_gemm_wide.h: the 256 x 256 x 32 block tile of sdmi_gemm (round 2).
//
// Why: a per-k-step timing of the tiled kernel over its four tile shapes (tools/bench_gemm.py) shows the
// bytes a CU pulls into LDS per second to be the same for all of them (43-52 GB/s per CU = 11-13 TB/s over the chip)
// while the MFMA rate moves with the tile's FLOP per byte - the L2 -> LDS path, not the matrix pipe, bounds the long-K
// GEMMs.  The 256 x 160 x 64 tile moves 53 KB for 5.2 MFLOP (98 FLOP/B); this one moves 32 KB for 4.2 MFLOP
// (128 FLOP/B).  A 64-deep k-tile of it would need 64 KB per ring stage, so the k-tile is 32 deep (64-byte rows) and
// the ring has four stages (128 KB, three in flight).
//
//   * 8 waves (4 x 2), wave tile 64 x 128: MI = 4 x TN = 8 v_mfma_f32_16x16x32 accumulators (128 VGPRs);
//   * LDS rows are 64 bytes = 4 chunks of 16 B; physical chunk = logical ^ s(row), s = [0, 3, 2, 1][(row >> 2) & 3]:
//     conflict-free for ds_read_b128's lane groups ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md);
//   * software pipeline over the two column halves of the wave tile (LO = blocks 0..3, HI = 4..7): after the barrier
//     that publishes k-tile t + 1 the wave reads A(t + 1) and W_LO(t + 1), runs the held-back HI MFMAs of tile t, issues
//     its DMA share of tile t + 4, reads W_HI(t + 1) and runs LO(t + 1); A fragments are double-buffered by parity;
//   * epilogue: the 16-bit fast path of the tiled kernel (epilogue_rows16), the 16-bit GEGLU epilogue (epilogue_geglu16)
//     or fp32 split-K slabs - the host takes this tile only where one of the three applies.
#pragma once
#include "sdmi_gemm_kernel.h"

namespace {

constexpr int WBK = 32, WBM = 256, WBN = 256, WNST = 4;
constexpr int WIDE_STAGE_CHUNKS = (WBM + WBN) * (WBK / 8);      // 2048 x 16 B
constexpr int WIDE_LDS_BYTES = WNST * WIDE_STAGE_CHUNKS * 16;   // 131072

template <int MODE, int DT, bool LNF = false>
__global__ __launch_bounds__(512, 2) void sdmi_gemm_wide_kernel(const sdmi_gemm_args p) {
    constexpr int TN = 8, MI = 4, NA = 2;      // (NA = NW = 2 block-wide DMA instructions of 128 rows per operand)
    constexpr int A_CHUNKS = WBM * (WBK / 8);
    constexpr int PER = 4;                     // DMA instructions per wave and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint4* const lds = reinterpret_cast<const uint4*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    const int64_t n_w = p.geglu ? 2 * p.N : p.N;
    const int tiles_m = (int)((p.M + WBM - 1) / WBM);
    const int tiles_n = (int)((n_w + WBN - 1) / WBN);
    const TileCoord tc = tile_of_block(blockIdx.x, tiles_m, tiles_n);
    const int64_t m0 = (int64_t)tc.tm * WBM;
    const int64_t n0 = (int64_t)tc.tn * WBN;

    const int z = blockIdx.z;
    const int nk_total = (int)(p.K / WBK);
    int kt_begin = 0, kt_end = nk_total;
    int64_t zoff_a = 0, zoff_w = 0, zoff_out = 0, zoff_res = 0;
    if (p.split_k > 1) {
        const int per = (nk_total + p.split_k - 1) / p.split_k;
        kt_begin = z * per;
        kt_end = min(nk_total, kt_begin + per);
    } else {
        zoff_a = z * p.stride_a; zoff_w = z * p.stride_w; zoff_out = z * p.stride_out; zoff_res = z * p.stride_res;
    }
    const unsigned short* const A = reinterpret_cast<const unsigned short*>(p.A) + zoff_a;
    const unsigned short* const A2 = reinterpret_cast<const unsigned short*>(p.A2);
    const unsigned short* const W = reinterpret_cast<const unsigned short*>(p.W) + zoff_w;

    // loader: thread owns physical chunk tid & 3 of tile rows lr + 128 * i and fetches logical chunk lc into it
    const int lr = tid >> 2;
    const int lc = (tid & 3) ^ ((4 - ((lr >> 2) & 3)) & 3);
    const int wave_u = __builtin_amdgcn_readfirstlane(wid);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned d_wave = lds_base + wave_u * 1024;
    const unsigned short* const zsrc = reinterpret_cast<const unsigned short*>(sdmi_zeros);

    const unsigned short* const w_tile = W + n0 * p.K;
    unsigned w_voff[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) w_voff[i] = (unsigned)((((int64_t)lr + 128 * i) * p.K + lc * 8) * 2);

    unsigned a_voff[NA], a2_voff[NA];        // MODE 0
    int64_t a_pix[NA];                       // MODE 1: pixel base b * H * W
    int iy0[NA], ix0[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int64_t m = m0 + lr + 128 * i;
        const bool rvalid = m < p.M;
        if (MODE == 0) {
            const int64_t rrel = (rvalid ? m : p.M - 1) - m0;      // rows past M re-read row M - 1 (never stored)
            a_voff[i] = (unsigned)((rrel * p.lda + lc * 8) * 2);
            a2_voff[i] = (unsigned)((rrel * p.lda2 + lc * 8) * 2);
        } else {
            const int64_t hw = (int64_t)p.Ho * p.Wo;
            const int64_t b = m / hw;
            const int rem = (int)(m - b * hw);
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_pix[i] = b * (int64_t)p.H * p.W_in;
            const int pad_lo = p.asym_pad ? 0 : 1;
            iy0[i] = rvalid ? oy * p.stride - pad_lo : -(1 << 20);
            ix0[i] = ox * p.stride - pad_lo;
        }
    }
    const int Hl = p.H << p.upsample, Wl = p.W_in << p.upsample;
    // conv (host-checked: Cin % 64 == 0, so K = 9 Cin and a k-tile never straddles a tap): (tap, channel run) is one
    // scalar state per wave; each row keeps a pointer rebuilt when the tap changes and bumped by 32 channels otherwise
    int f_tap = 0, f_ci = 0;
    bool f_new_tap = true;
    const unsigned short* a_ptr[NA];
    if (MODE == 1) {
        f_tap = (kt_begin * WBK) / p.Cin;
        f_ci = kt_begin * WBK - f_tap * p.Cin;
#pragma unroll
        for (int i = 0; i < NA; ++i) a_ptr[i] = zsrc;
    }
#define SDMI_W_TAP_PTR(I, KY, KX)                                                                          \
    do {                                                                                                   \
        int iy_ = iy0[I] + (KY), ix_ = ix0[I] + (KX);                                                      \
        const bool ok_ = (unsigned)iy_ < (unsigned)Hl && (unsigned)ix_ < (unsigned)Wl;                     \
        iy_ >>= p.upsample; ix_ >>= p.upsample;                                                            \
        const int64_t off_ = (a_pix[I] + (int64_t)iy_ * p.W_in + ix_) * p.Cin;                             \
        a_ptr[I] = (ok_ ? A + off_ : zsrc) + (f_ci + lc * 8);                                              \
    } while (0)
#define SDMI_W_STAGE_TILE(KT, BUF)                                                                         \
    do {                                                                                                   \
        const unsigned d0_ = d_wave + (BUF) * (WIDE_STAGE_CHUNKS * 16);                                    \
        const int64_t k0_ = (int64_t)(KT) * WBK;                                                           \
        if (MODE == 0) {                                                                                   \
            const bool second_ = (A2 != nullptr) && (k0_ >= p.k_split);                                    \
            const unsigned short* sa_ = second_ ? A2 + m0 * p.lda2 + (k0_ - p.k_split) : A + m0 * p.lda + k0_; \
            glds_s2<0>(d0_, sa_, second_ ? a2_voff[0] : a_voff[0], second_ ? a2_voff[1] : a_voff[1]);      \
        } else {                                                                                           \
            if (f_new_tap) {                                                                               \
                const int ky_ = f_tap / 3, kx_ = f_tap - 3 * ky_;                                          \
                SDMI_W_TAP_PTR(0, ky_, kx_); SDMI_W_TAP_PTR(1, ky_, kx_);                                  \
            }                                                                                              \
            if (!(SDMI_DEBUG_A_FILL_NINTH && ((KT) % 9) != 0)) glds_v2<0>(d0_, a_ptr[0], a_ptr[1]);        \
            a_ptr[0] += WBK; a_ptr[1] += WBK;                                                              \
            f_ci += WBK;                                                                                   \
            f_new_tap = f_ci == p.Cin;                                                                     \
            if (f_new_tap) { f_ci = 0; ++f_tap; }                                                          \
        }                                                                                                  \
        glds_s2<A_CHUNKS * 16>(d0_, w_tile + k0_, w_voff[0], w_voff[1]);                                   \
    } while (0)

    f32x4 acc[TN][MI];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    const int pc = fq ^ ((4 - ((frow >> 2) & 3)) & 3);
    const int a_base = (wm * 64 + frow) * 4 + pc;
    const int w_base = A_CHUNKS + (wn * 128 + frow) * 4 + pc;
    const int nk = kt_end - kt_begin;

    uint4 afE[MI], afO[MI], wfL[4], wfH[4];
#define SDMI_W_READ_A(AF, STAGE)                                                                           \
    do {                                                                                                   \
        const uint4* St_ = lds + (STAGE) * WIDE_STAGE_CHUNKS;                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_) AF[i_] = St_[a_base + i_ * 64];                  \
    } while (0)
#define SDMI_W_READ_W(WF, STAGE, HALF)                                                                     \
    do {                                                                                                   \
        const uint4* St_ = lds + (STAGE) * WIDE_STAGE_CHUNKS;                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) WF[i_] = St_[w_base + ((HALF) * 4 + i_) * 64];    \
    } while (0)
#define SDMI_W_MFMA(AF, WF, HALF)                                                                          \
    do {                                                                                                   \
        _Pragma("unroll") for (int ni_ = 0; ni_ < 4; ++ni_)                                                \
            _Pragma("unroll") for (int mi_ = 0; mi_ < MI; ++mi_)                                           \
                acc[(HALF) * 4 + ni_][mi_] = mfma16<DT>(WF[ni_], AF[mi_], acc[(HALF) * 4 + ni_][mi_]);     \
    } while (0)
#ifndef SDMI_WIDE_SCHED
#define SDMI_WIDE_SCHED 1
#endif
#if SDMI_WIDE_SCHED == 1
#define SDMI_W_INTERLEAVE(READS, PER_READ)                                                                 \
    do {                                                                                                   \
        _Pragma("unroll") for (int g_ = 0; g_ < (READS); ++g_) {                                           \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                             \
            __builtin_amdgcn_sched_group_barrier(0x008, (PER_READ), 0);                                    \
        }                                                                                                  \
    } while (0)
#elif SDMI_WIDE_SCHED == 2
#define SDMI_W_INTERLEAVE(READS, PER_READ) __builtin_amdgcn_sched_barrier(0)
#else
#define SDMI_W_INTERLEAVE(READS, PER_READ) do {} while (0)
#endif
    // one k-tile step: CUR = the A fragments of tile kt (its LO half done), NXT receives tile kt + 1
#define SDMI_W_STEP(CUR, NXT)                                                                              \
    do {                                                                                                   \
        wait_dma<PER>(kt_issue - kt - 2);           /* tile kt + 1 landed; younger tiles may still fly */  \
        __syncthreads();                            /* ... for every wave, and tile kt is read out */      \
        SDMI_W_READ_A(NXT, rd);                                                                            \
        SDMI_W_READ_W(wfL, rd, 0);                                                                         \
        SDMI_W_MFMA(CUR, wfH, 1);                   /* HI half of tile kt */                               \
        SDMI_W_INTERLEAVE(8, 2);                                                                           \
        if (kt_issue < nk) {                                                                               \
            SDMI_W_STAGE_TILE(kt_begin + kt_issue, wr);                                                    \
            ++kt_issue;                                                                                    \
        }                                                                                                  \
        SDMI_W_READ_W(wfH, rd, 1);                                                                         \
        SDMI_W_MFMA(NXT, wfL, 0);                   /* LO half of tile kt + 1 */                           \
        SDMI_W_INTERLEAVE(4, 4);                                                                           \
        wr = rd;                                                                                           \
        rd = (rd + 1) & (WNST - 1);                                                                        \
        ++kt;                                                                                              \
    } while (0)

    int kt_issue = 0;
#pragma unroll
    for (int st = 0; st < WNST; ++st) {
        if (kt_issue < nk) {
            SDMI_W_STAGE_TILE(kt_begin + kt_issue, st);
            ++kt_issue;
        }
    }
    if (nk > 0) {
        wait_dma<PER>(kt_issue - 1);
        __syncthreads();
        SDMI_W_READ_A(afE, 0);
        SDMI_W_READ_W(wfL, 0, 0);
        SDMI_W_READ_W(wfH, 0, 1);
        SDMI_W_MFMA(afE, wfL, 0);
        int rd = 1, wr = 0, kt = 0;
        // nk - 1 steps, two per trip (the A fragment sets swap roles); an odd count runs one step ahead of the loop and
        // moves its fragments back, so the loop has one shape and one exit (with two exits hipcc stops accumulating
        // in place and spills)
        if (((nk - 1) & 1) != 0) {
            SDMI_W_STEP(afE, afO);
#pragma unroll
            for (int i = 0; i < MI; ++i) afE[i] = afO[i];
        }
        while (kt + 1 < nk) {
            SDMI_W_STEP(afE, afO);
            SDMI_W_STEP(afO, afE);
        }
        SDMI_W_MFMA(afE, wfH, 1);                   // HI half of the last tile
    }
#undef SDMI_W_STEP
#undef SDMI_W_INTERLEAVE
#undef SDMI_W_MFMA
#undef SDMI_W_READ_W
#undef SDMI_W_READ_A
#undef SDMI_W_STAGE_TILE
#undef SDMI_W_TAP_PTR

    // ---- epilogue: acc[ni][mi][j] = row m0 + wm*64 + mi*16 + (lane & 15), weight row n0 + wn*128 + ni*16 + 4*(lane >> 4) + j
    if constexpr (LNF) {
        if (p.split_k <= 1) {
            // LNF (sdmi_gemm_kernel.h): the token-stream form.  A folded LayerNorm (ln_colsum) is applied inside the 16-bit
            // epilogues - a separate pass over the 128 accumulator registers spills - from mean / rstd parked in LDS.
            // For the same reason a residual is requested row block by row block here.  Row statistics out: one slot per
            // 64-column run (4 per column tile).
            const bool ln_fold = p.ln_colsum != nullptr;
            float* const xs = reinterpret_cast<float*>(smem + WIDE_LDS_BYTES);      // [2][256] behind the ring: mean | rstd
            if (ln_fold) {
                float mean_r[MI], rstd_r[MI];
                lnf_rows_mean_rstd<MI>(p, m0 + wm * 64 + frow, fq, mean_r, rstd_r);
                if (wn == 0 && fq == 0) {      // (both wave columns hold the same rows)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        xs[wm * 64 + mi * 16 + frow] = mean_r[mi];
                        xs[WBM + wm * 64 + mi * 16 + frow] = rstd_r[mi];
                    }
                }
                __syncthreads();
            }
            const float* const mean = ln_fold ? xs + wm * 64 + frow : nullptr;
            const float* const rstd = ln_fold ? xs + WBM + wm * 64 + frow : nullptr;
            typedef f32x4 (&lhalf_t)[4][MI];
            const int64_t mrow = m0 + wm * 64 + frow, ncol = n0 + wn * 128;
            if (p.geglu) {      // (host-checked: GEGLU comes here only with a folded LayerNorm)
                epilogue_geglu16<DT, MI, true, 16>(p, reinterpret_cast<lhalf_t>(acc[0]), mrow, ncol, fq, mean, rstd);
                epilogue_geglu16<DT, MI, true, 16>(p, reinterpret_cast<lhalf_t>(acc[4]), mrow, ncol + 64, fq, mean, rstd);
            } else {
                const int slot = p.row_stats_out != nullptr ? tc.tn * 4 + wn * 2 : -1;
                epilogue_rows16<DT, 4, MI, 0, false, true, 16>(p, reinterpret_cast<lhalf_t>(acc[0]), mrow, ncol, fq,
                                                               zoff_out, zoff_res, mean, rstd, slot);
                epilogue_rows16<DT, 4, MI, 0, false, true, 16>(p, reinterpret_cast<lhalf_t>(acc[4]), mrow, ncol + 64, fq,
                                                               zoff_out, zoff_res, mean, rstd, slot < 0 ? -1 : slot + 1);
            }
            return;
        }
    }
    if (p.split_k > 1) {
        const int64_t ldw = (int64_t)tiles_n * WBN;
        float* ws = reinterpret_cast<float*>(p.workspace) + (int64_t)z * p.M * ldw;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int64_t m = m0 + wm * 64 + mi * 16 + frow;
            if (m >= p.M) continue;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int64_t wcol = n0 + wn * 128 + ni * 16 + 4 * fq;
                *reinterpret_cast<f32x4*>(ws + m * ldw + wcol) = acc[ni][mi];
            }
        }
        return;
    }
    // the wave's 128 columns as two runs of 64 (four 16-column blocks each): half the bias / residual / embedding
    // registers of one 8-block pass, which would spill beside the 128 accumulator registers
    typedef f32x4 (&half_t)[4][MI];
    if (p.geglu) {
        // (value, gate) pairs of 16-column weight blocks -> 64 gated output columns per wave, in two runs of 32
        // (host-checked: 16-bit row-major output, nothing to add, N % 32 == 0)
        epilogue_geglu16<DT, MI>(p, reinterpret_cast<half_t>(acc[0]), m0 + wm * 64 + frow, n0 + wn * 128, fq);
        epilogue_geglu16<DT, MI>(p, reinterpret_cast<half_t>(acc[4]), m0 + wm * 64 + frow, n0 + wn * 128 + 64, fq);
        return;
    }
    epilogue_rows16<DT, 4, MI, 0>(p, reinterpret_cast<half_t>(acc[0]), m0 + wm * 64 + frow, n0 + wn * 128, fq, zoff_out, zoff_res);
    epilogue_rows16<DT, 4, MI, 0>(p, reinterpret_cast<half_t>(acc[4]), m0 + wm * 64 + frow, n0 + wn * 128 + 64, fq, zoff_out, zoff_res);
}

}  // namespace

#define SDMI_GEMM_WIDE_UNIT(DT, NAME)                                                                               \
    int sdmi_gemm_wide_attrs_##NAME() {                                                                             \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(sdmi_gemm_wide_kernel<0, DT>),                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS_BYTES) != hipSuccess ||        \
            hipFuncSetAttribute(reinterpret_cast<const void*>(sdmi_gemm_wide_kernel<1, DT>),                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS_BYTES) != hipSuccess)          \
            return SDMI_ERR_LAUNCH;                                                                                 \
        return SDMI_OK;                                                                                             \
    }                                                                                                               \
    void sdmi_gemm_wide_launch_##NAME(int mode, dim3 grid, hipStream_t s, const sdmi_gemm_args& p) {                \
        if (mode == 0) hipLaunchKernelGGL((sdmi_gemm_wide_kernel<0, DT>), grid, dim3(512), WIDE_LDS_BYTES, s, p);   \
        else           hipLaunchKernelGGL((sdmi_gemm_wide_kernel<1, DT>), grid, dim3(512), WIDE_LDS_BYTES, s, p);   \
    }

// The token-stream instantiation (LNF: folded LayerNorm in, row statistics out; mode 0, f16).
#define SDMI_GEMM_WIDE_UNIT_LNF(NAME)                                                                               \
    int sdmi_gemm_wide_attrs_##NAME() {                                                                             \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(sdmi_gemm_wide_kernel<0, 1, true>),                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS_BYTES + 2048) != hipSuccess)   \
            return SDMI_ERR_LAUNCH;                                                                                 \
        return SDMI_OK;                                                                                             \
    }                                                                                                               \
    void sdmi_gemm_wide_launch_##NAME(int mode, dim3 grid, hipStream_t s, const sdmi_gemm_args& p) {                \
        (void)mode;                                                                                                 \
        hipLaunchKernelGGL((sdmi_gemm_wide_kernel<0, 1, true>), grid, dim3(512), WIDE_LDS_BYTES + 2048, s, p);      \
    }
