// sdmi_gemm_panel.h: the "A-panel-stationary" form of sdmi_gemm for linears with a SHORT K (K = 64 * KT <= 640) and many
// rows - the level-0 / level-1 transformer projections of the UNet (M = tokens, K = 320 | 640), which the tiled kernel
// runs at 250-650 TFLOP/s because every 128 x 160 block is dominated by its own prologue / epilogue latency and the
// A tile is re-fetched once per column tile.
//
// One workgroup owns BM rows and ALL columns:
//   * its A panel [BM x K] is fetched ONCE (a single burst of LDS-DMA - the whole panel is in flight together) and stays
//     in LDS;
//   * the weight tiles [BN x 64] stream through an NSTW-stage LDS ring without a break across column tiles: while the
//     epilogue of column tile j stores, the first k-tiles of tile j + 1 are already landing;
//   * two workgroups of 4 waves share a CU (BM = 64, K = 320: 80 KiB each), so one block's panel burst and epilogues
//     overlap the other's MFMA work.
// Measured (MI355X, M = 65536, K = 320, against the tiled kernel): N = 320 38 -> 31.5 us (45 -> 40 with a residual),
// N = 960 101 -> 75 us, GEGLU 2 x 1280 202 -> 186 us.  What still bounds it: loads, stores and LDS-DMA retire in issue
// order on ONE counter, so the first ring wait behind an epilogue also waits for that epilogue's stores to be
// acknowledged (~1 us under load), once per column tile; requesting the residual of both column tiles at block start
// (so that it lands with the panel) was measured SLOWER (49 us: 40 more VGPRs held across the loop) and is not kept.
// vmcnt bookkeeping: loads, stores and LDS-DMA retire in issue order on one counter.  Each epilogue starts with a full
// drain (the ring stages issued ahead of it are then known to have landed: the first NSTW - 1 k-steps of the next
// column tile need no wait), and every later wait counts only the ring stages issued after its target - all of them
// younger than the epilogue's stores, so no store is ever waited for explicitly.
// Fragment layout, swizzle, MFMA operand roles and the 16-byte epilogue are those of sdmi_gemm_kernel.h.
#pragma once
#include "sdmi_gemm_kernel.h"

namespace {

__device__ __forceinline__ void glds1_s(unsigned dst, const void* sbase, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v], %[sb]\n\t"
                 "s_mov_b32 m0, %[keep]"
                 : [keep] "=&s"(keep)
                 : [d] "s"(dst), [sb] "s"(sbase), [v] "v"(voff)
                 : "memory");
}

template <int PER>
__device__ __forceinline__ void wait_young(int young) {       // all but the `young` youngest ring stages have landed
    if (young >= 2)      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER) : "memory");
    else if (young == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");
    else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

constexpr int panel_lds_bytes(int tn, int wmw, int kt, int nstw) { return (kt * wmw * 32 + nstw * tn * 32) * 128; }

// WMW x 2 waves; block tile (32 * WMW) x (32 * TN); K = 64 * KT; NSTW-stage weight ring.
template <int DT, int TN, int WMW, int KT, int NSTW, bool GEGLU>
__global__ __launch_bounds__(WMW * 128, 2) void sdmi_gemm_panel_kernel(const sdmi_gemm_args p) {
    constexpr int MI = 2;
    constexpr int BM = WMW * 32, BN = TN * 32, NWAVES = WMW * 2;
    constexpr int A_KT_CHUNKS = BM * 8;                 // 16-byte chunks of one k-tile of the panel
    constexpr int W_CHUNKS = BN * 8;
    constexpr int RPI = NWAVES * 8;                     // tile rows one DMA instruction of the whole block covers
    constexpr int NAI = BM / RPI;                       // A instructions per wave and k-tile (2)
    constexpr int NWI = BN / RPI;                       // W instructions every wave issues per stage
    constexpr int W_TAIL_WAVES = (BN - NWI * RPI) / 8;  // leftover 8-row groups: waves 0 .. W_TAIL_WAVES-1 issue one more
    static_assert(BM % RPI == 0 && BN % 8 == 0, "tile rows must split over the waves");
    static_assert(!GEGLU || TN == 4, "GEGLU pairs the four 16-column blocks of a wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint4* const lds = reinterpret_cast<const uint4*>(smem);
    constexpr int W_RING0 = KT * A_KT_CHUNKS;           // first chunk of the weight ring

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int64_t n_w = GEGLU ? 2 * p.N : p.N;
    const int tiles_n = (int)(n_w / BN);                // exact (host-checked)
    const int64_t m0 = (int64_t)blockIdx.x * BM;

    const unsigned short* const A = reinterpret_cast<const unsigned short*>(p.A);
    const unsigned short* const W = reinterpret_cast<const unsigned short*>(p.W);

    // loader geometry (as sdmi_gemm_kernel.h): lane owns physical chunk tid & 7 of rows lr + RPI * i and fetches the
    // logical chunk (tid & 7) ^ (row & 7)
    const int lr = tid >> 3;
    const int lc = (tid & 7) ^ (lr & 7);
    const int wave_u = __builtin_amdgcn_readfirstlane(wid);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned d_wave = lds_base + wave_u * 1024;
    const bool w_tail_wave = W_TAIL_WAVES > 0 && wave_u < W_TAIL_WAVES;
    unsigned a_voff[NAI], w_voff[NWI + 1];
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
        const int64_t m = m0 + lr + RPI * i;
        const int64_t rrel = (m < p.M ? m : p.M - 1) - m0;      // rows past M re-read row M - 1 (never stored)
        a_voff[i] = (unsigned)((rrel * p.lda + lc * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < NWI; ++i) w_voff[i] = (unsigned)((((int64_t)lr + RPI * i) * p.K + lc * 8) * 2);
    // tail: the 8 rows NWI * RPI + 8 * wave + (lane >> 3) (only waves < W_TAIL_WAVES); LDS slot right behind the full rounds
    w_voff[NWI] = (unsigned)((((int64_t)NWI * RPI + wave_u * 8 + (lane >> 3)) * p.K + (((lane & 7) ^ ((lane >> 3) & 7))) * 8) * 2);

    // ---- A panel: every k-tile in one burst ----
    const unsigned short* const a_tile = A + m0 * p.lda;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < NAI; ++i)
            glds1_s(d_wave + (kt * A_KT_CHUNKS + i * RPI * 8) * 16, a_tile + kt * BK, a_voff[i]);

    const int steps = tiles_n * KT;
    auto issue_w = [&](int step) {      // weight k-tile of flat step `step` (column tile step / KT) into its ring slot
        const int nt = step / KT, kt = step - nt * KT;
        const unsigned short* src = W + ((int64_t)nt * BN) * p.K + kt * BK;
        const unsigned d0 = d_wave + (W_RING0 + (step % NSTW) * W_CHUNKS) * 16;
#pragma unroll
        for (int i = 0; i < NWI; ++i) glds1_s(d0 + i * RPI * 8 * 16, src, w_voff[i]);
        if (w_tail_wave) glds1_s(d0 + NWI * RPI * 8 * 16, src, w_voff[NWI]);
    };
    auto wait_w = [&](int young) {
        if (w_tail_wave) wait_young<NWI + 1>(young); else wait_young<NWI>(young);
    };
    int issued = 0;
#pragma unroll
    for (int s = 0; s < NSTW - 1; ++s)
        if (issued < steps) { issue_w(issued); ++issued; }

    // ---- optional LayerNorm of the resident panel (the nn.LayerNorm in front of the projection, fused) ----
    // Four threads per row, each owning the 16-byte chunks c with c % 4 == tid & 3 of every k-tile (2 * KT chunks,
    // 16 * KT channels): exact two-pass statistics in fp32 as in layernorm_kernel, the normalised row rounded to the
    // operand type and written back in place, so the MFMAs read what a separate LayerNorm launch would have stored.
    if (p.ln_gamma != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the panel (and the first ring stages) have landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        uint4* const ldsw = reinterpret_cast<uint4*>(smem);
        const int row = tid >> 2, part = tid & 3;
        float f[2 * KT][8];
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint4 raw = ldsw[kt * A_KT_CHUNKS + row * 8 + ((part + 4 * h) ^ (row & 7))];
                elem_unpack8_sel<DT>(raw, f[2 * kt + h], p.a_f16);      // (bf16 policy: the rows are the f16 residual stream)
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += f[2 * kt + h][j];
            }
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        const float inv_k = 1.0f / (float)(KT * BK);
        const float mean = sum * inv_k;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 2 * KT; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = f[c][j] - mean; q += d * d; }
        q += __shfl_xor(q, 1);
        q += __shfl_xor(q, 2);
        const float rstd = rsqrtf(q * inv_k + p.ln_eps);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ch = kt * BK + (part + 4 * h) * 8;
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.ln_gamma + ch);
                const f32x4 g1 = *reinterpret_cast<const f32x4*>(p.ln_gamma + ch + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.ln_beta + ch);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.ln_beta + ch + 4);
                float y[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    y[j] = (f[2 * kt + h][j] - mean) * rstd * g0[j] + b0[j];
                    y[4 + j] = (f[2 * kt + h][4 + j] - mean) * rstd * g1[j] + b1[j];
                }
                ldsw[kt * A_KT_CHUNKS + row * 8 + ((part + 4 * h) ^ (row & 7))] = elem_pack8<DT>(y);
            }
        __syncthreads();      // the rewritten panel is visible to every wave (lgkmcnt + barrier)
    }

    f32x4 acc[TN][MI];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    const int a_base = (wm * (MI * 16) + frow) * 8;
    const int w_base = (wn * (TN * 16) + frow) * 8;

    int step = 0;
    for (int nt = 0; nt < tiles_n; ++nt) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt, ++step) {
            // weight stage of this step landed?  (first NSTW - 1 steps of a later column tile: drained by the epilogue)
            if (nt == 0 || kt >= NSTW - 1) wait_w(issued - step - 1);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();             // everyone's share landed; everyone finished reading step - 1
            asm volatile("" ::: "memory");            // (the raw barrier is not a compiler fence for the LDS reads below)
            if (issued < steps) { issue_w(issued); ++issued; }      // into the slot step - 1 vacated
            const uint4* Ap = lds + kt * A_KT_CHUNKS;
            const uint4* Wp = lds + W_RING0 + (step % NSTW) * W_CHUNKS;
            uint4 af[2][MI], wf[2][TN];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int pc = (ks * 4 + fq) ^ (frow & 7);
#pragma unroll
                for (int i = 0; i < MI; ++i) af[ks][i] = Ap[a_base + i * 128 + pc];
#pragma unroll
                for (int i = 0; i < TN; ++i) wf[ks][i] = Wp[w_base + i * 128 + pc];
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = mfma16<DT>(wf[ks][ni], af[ks][mi], acc[ni][mi]);
        }
        // ---- epilogue of column tile nt ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the drain the wait bookkeeping above relies on
        const int64_t mrow = m0 + wm * (MI * 16) + frow;
        const int64_t ncol_w = (int64_t)nt * BN + wn * (TN * 16);
        if constexpr (GEGLU) {
            epilogue_geglu16<DT, MI>(p, acc, mrow, ncol_w, fq);
        } else {
            // pairs start on even 16-column blocks of the output: with 5 blocks per wave the odd wave column shifts by one
            if ((TN & 1) && (wn & 1)) epilogue_rows16<DT, TN, MI, (TN & 1)>(p, acc, mrow, ncol_w, fq, 0, 0);
            else                      epilogue_rows16<DT, TN, MI, 0>(p, acc, mrow, ncol_w, fq, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < MI; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

}  // namespace
