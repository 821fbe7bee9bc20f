// sdmi_gemm_kernel.h: the GEMM / implicit-GEMM kernel template of sdmi_gemm (see sdmi_gemm.hip for the design notes).
// Included by the per-(mode, dtype) instantiation units sdmi_gemm_m?_*.hip, which are compiled in parallel, and by
// sdmi_gemm.hip (split-K reduce + the C entry point).
#pragma once
#include "sdmi_common.h"

#ifndef SDMI_SCHED_INTERLEAVE
#define SDMI_SCHED_INTERLEAVE 1   // A/B: +1..3% on MI355X
#endif
#ifndef SDMI_SETPRIO
#define SDMI_SETPRIO 0   // A/B on MI355X (tools/ab_build.sh): raising the wave priority around the MFMA groups gains nothing here
#endif

#ifndef SDMI_DEBUG_A_FILL_NINTH
#ifndef SDMI_DEBUG_HALF_FRAG_READS
#define SDMI_DEBUG_HALF_FRAG_READS 0   // timing experiment only (results wrong): every second fragment read of the k-loop is dropped (the compiler merges the duplicate addresses) - what the LDS reads cost the k-step
#endif
#ifndef SDMI_DEBUG_ABLATE
#define SDMI_DEBUG_ABLATE 0   // timing experiments only (results wrong), the k-loop of the tiled kernel with one component removed: 1 = no MFMA (the fragments are still read and waited for), 2 = no LDS-DMA and no DMA wait (the stages of the prologue are re-read), 3 = no barrier, 4 = the DMA is issued but never waited for
#endif
#define SDMI_DEBUG_A_FILL_NINTH 0   // timing experiment only (results wrong, ring waits under-counted): the conv's A tile is moved into LDS on every ninth k-tile only - the LDS fill bytes of a halo-tile kernel (each input pixel staged once per channel chunk instead of once per tap) with everything else unchanged
#endif

namespace {

#ifndef SDMI_DEBUG_CLOCKS
#define SDMI_DEBUG_CLOCKS 0   // measurement build only: block 0 of every tiled-kernel launch leaves (shader clocks, 100 MHz ticks) of its k-loop in sdmi_debug_clk - the clock the part holds under this kernel (tools/kloop_clock.py)
#endif
#if SDMI_DEBUG_CLOCKS
__device__ long long sdmi_debug_clk[2];
#endif

// Zeros: the DMA source of out-of-image conv taps and padded rows / columns (the DMA cannot mask).  Long enough
// that a tap's pointer can walk a whole channel run (Cin <= ZERO_ELEMS) without leaving it.
constexpr int ZERO_ELEMS = 16384;
__device__ const uint4 sdmi_zeros[ZERO_ELEMS / 8 + 1] = {};

// Block tile BM x BN x BK with BN = 32*TN: TN = 4 (128 columns) or TN = 5 (160 columns; every UNet width is a
// multiple of 160, so those layers lose nothing to column padding).
// and BM = 64*MI: MI = 4 (256 rows) or MI = 2 (128 rows); 8 waves, 3-stage ring, one block per CU.
// The small tile (round 3; mode 0): TN = 2 (64 columns) with MI = 2 | 1 (128 | 64 rows) - for launches of a few hundred rows
// (the UNet at batch 1 - 2), where the larger tiles leave most of the chip without a block: 48 - 72 KiB of LDS, so two or
// three of its blocks also share a CU.
constexpr int BK = 64;
constexpr int lds_bytes(int tn, int mi, int nst) { return nst * (mi * 64 + tn * 32) * (BK / 8) * 16; }
constexpr int GROUP_M = 8;

struct TileCoord { int tm, tn; };

__device__ __forceinline__ TileCoord tile_of_block(int bid, int tiles_m, int tiles_n) {
    const int nwg = tiles_m * tiles_n;
    // XCD-aware bijective remap: blocks b and b+8 share an XCD; give each XCD a contiguous range.
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    // group-M swizzle inside the range
    const int per_group = GROUP_M * tiles_n;
    const int g = lid / per_group;
    const int first_m = g * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int in_g = lid - g * per_group;
    TileCoord t;
    t.tm = first_m + in_g % gsz;
    t.tn = in_g / gsz;
    return t;
}

// Epilogue for 4 consecutive output columns of one row.  `v` is the raw accumulator quad (already
// scaled and biased by the caller for the column-bias / GEGLU cases).  Returns the fp32 values it stored (rounded).
template <int DT>
__device__ __forceinline__ f32x4 store_quad(const sdmi_gemm_args& p, int64_t m, int64_t n, f32x4 v,
                                            int64_t zoff_out, int64_t zoff_res) {
    if (p.bias != nullptr && p.bias_per_row) {      // epilogue order of include/sdmi.h: bias, then act
        const float b = p.bias[m];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += b;
    }
    if (p.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = p.out_kind != 0 ? silu_precise(v[j]) : silu_f(v[j]);
    } else if (p.act == 2) {
        if (p.out_kind != 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = gelu_erf_precise(v[j]);
        } else {
            v = gelu_erf_f4(v);
        }
    }
    const bool full = (n + 3 < p.N);
    if (p.emb != nullptr) {
        const float* e = p.emb + (m / p.rows_per_batch) * p.ld_emb + n;
        if (full) {
            const f32x4 ev = *reinterpret_cast<const f32x4*>(e);
            v += ev;
        } else {
            for (int j = 0; j < 4 && n + j < p.N; ++j) v[j] += e[j];
        }
    }
    if (p.res != nullptr && p.res_f32) {
        const float* r = reinterpret_cast<const float*>(p.res) + zoff_res + m * p.ld_res + n;
        for (int j = 0; j < 4 && n + j < p.N; ++j) v[j] += r[j];
    } else if (p.res != nullptr) {
        const unsigned short* r = reinterpret_cast<const unsigned short*>(p.res) + zoff_res + m * p.ld_res + n;
        if (full) {
            const uint2 rv = *reinterpret_cast<const uint2*>(r);
            float rf[4];
            elem_unpack4_sel<DT>(rv, rf, p.res_f16);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += rf[j];
        } else {
            for (int j = 0; j < 4 && n + j < p.N; ++j) v[j] += elem_to_f32_sel<DT>(r[j], p.res_f16);
        }
    }
    if (p.out_kind == 0) {
        unsigned short* o = reinterpret_cast<unsigned short*>(p.out) + zoff_out + m * p.ld_out + n;
        if (full) {
            uint2 ov;
            ov.x = elem_pack2_sel<DT>(v[0], v[1], p.out_f16);
            ov.y = elem_pack2_sel<DT>(v[2], v[3], p.out_f16);
            *reinterpret_cast<uint2*>(o) = ov;
        } else {
            for (int j = 0; j < 4 && n + j < p.N; ++j) o[j] = elem_from_f32_sel<DT>(v[j], p.out_f16);
        }
    } else if (p.out_kind == 1) {
        float* o = reinterpret_cast<float*>(p.out) + zoff_out + m * p.ld_out + n;
        if (full) {
            *reinterpret_cast<f32x4*>(o) = v;
        } else {
            for (int j = 0; j < 4 && n + j < p.N; ++j) o[j] = v[j];
        }
    } else {  // fp32 NCHW
        const int64_t b = m / p.rows_per_batch, pix = m - b * p.rows_per_batch;
        float* o = reinterpret_cast<float*>(p.out) + zoff_out + (b * p.N + n) * p.rows_per_batch + pix;
        for (int j = 0; j < 4 && n + j < p.N; ++j) o[j * p.rows_per_batch] = v[j];
    }
    return v;
}

// Wait until at most `keep_stages` (0, 1, 2) of this wave's youngest ring stages (PER LDS-DMA instructions each)
// are still in flight.
template <int PER>
__device__ __forceinline__ void wait_dma(int keep_stages) {
    if (keep_stages >= 2)      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER) : "memory");
    else if (keep_stages == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");
    else                       asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- LayerNorm folded into the projection (sdmi_gemm_args.ln_colsum / ln_stats; f16 operands) ----
// The kernel multiplies the RAW rows of A by W' = W * gamma; LayerNorm(a) @ W^T = rstd[m] * (a @ W'^T - mean[m] * colsum[n]).
// The two numbers per row come from the launch that PRODUCED A (the residual stream of the transformer blocks:
// to_out + residual, ff2 + residual, proj_in), whose epilogue leaves per row and per (column tile, wave column) "slot" the
// partial sum(a), sum(a^2) of the fp32 values it is about to round and store (sdmi_gemm_args.row_stats_out).  A first
// form gathered the sums inside the consuming kernel (v_dot2_f32_f16 on the activation fragments beside the MFMAs): 5-26 %
// of the GEMM, more than the LayerNorm launch it replaced on the GEGLU projections
// (profiles/r03_lnfold_inkernel_experiment.txt).
// x summed over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (every lane gets the total): two swaps on the VALU
// (v_permlane16_swap: rows 1 / 3 of the first operand <-> rows 0 / 2 of the second; v_permlane32_swap: upper half of the
// first <-> lower half of the second) instead of two trips through the LDS crossbar (ds_bpermute).
__device__ __forceinline__ float sum_lane_groups(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
    const unsigned v = __builtin_bit_cast(unsigned, y);
    const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}

__device__ __forceinline__ void lnf_mean_rstd(float s1, float s2, float inv_k, float eps, float& mean, float& rstd) {
    mean = s1 * inv_k;
    const float var = fmaxf(s2 * inv_k - mean * mean, 0.f);
    rstd = rsqrtf(var + eps);
}

// mean / rstd of the MI rows a lane's accumulator quads belong to (rows mrow0 + 16 * mi) from the producer's slots,
// ln_stats[m][slot] = (sum, sum of squares).  The four lanes that share a row (lane >> 4 = 0..3) split the slots between
// them - pairs of slots as one 16-byte load where the slot count is even -, every load of a batch (MI rows x 2 steps = 16
// slots per row) is in flight at once, and a butterfly adds up: one round trip for up to 16 slots, two for up to 32, issued
// in the epilogue next to the bias / colsum loads (a serial walk over 16 slots per row measured +78 us on a 120 us launch;
// requesting them ahead of the main loop, beside the LDS-DMA, makes the compiler's vmcnt waits cover the DMA as well:
// +13 .. +44 us on the same launch).
template <int MI>
__device__ __forceinline__ void lnf_rows_mean_rstd(const sdmi_gemm_args& p, int64_t mrow0, int fq, float (&mean)[MI],
                                                   float (&rstd)[MI]) {
    const int slots = p.ln_stats_slots;
    const bool paired = (slots & 1) == 0;
    const int units = paired ? slots >> 1 : slots;      // (odd counts: host-checked <= 4, one 8-byte load per lane)
    float s1[MI], s2[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) s1[mi] = s2[mi] = 0.f;
#pragma unroll
    for (int batch = 0; batch < 3; ++batch) {
        if (batch * 8 >= units) break;                  // (uniform)
        f32x4 raw[2][MI];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int u = batch * 8 + fq + 4 * j;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int64_t m = mrow0 + mi * 16;
                raw[j][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (u < units && m < p.M) {
                    if (paired) {
                        raw[j][mi] = reinterpret_cast<const f32x4*>(p.ln_stats)[m * units + u];
                    } else {
                        const f32x2 t = reinterpret_cast<const f32x2*>(p.ln_stats)[m * units + u];
                        raw[j][mi] = (f32x4){t[0], t[1], 0.f, 0.f};
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                s1[mi] += raw[j][mi][0] + raw[j][mi][2];
                s2[mi] += raw[j][mi][1] + raw[j][mi][3];
            }
    }
    const float inv_k = 1.0f / (float)(p.k_valid > 0 ? p.k_valid : p.K);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        lnf_mean_rstd(sum_lane_groups(s1[mi]), sum_lane_groups(s2[mi]), inv_k, p.ln_eps, mean[mi], rstd[mi]);
    }
}

// acc <- rstd[m] * (acc - mean[m] * colsum[n]) for the wave's TN x MI accumulator quads; wcol0 = the wave's first weight
// row + 4 * (lane >> 4)
template <int TN, int MI>
__device__ __forceinline__ void lnf_apply(const sdmi_gemm_args& p, f32x4 (&acc)[TN][MI], const float (&mean)[MI],
                                          const float (&rstd)[MI], int64_t wcol0) {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int64_t wcol = wcol0 + ni * 16;
        const f32x4 cs = *reinterpret_cast<const f32x4*>(p.ln_colsum + wcol);      // (padded to the column tile: sdmi.h)
        // (Element by element on purpose.  Written as one f32x4 expression, hipcc (ROCm 7.2) put the row's mean on the SECOND
        // source of two of the packed fmas - v_pk_fma_f32 d, cs, mean, d op_sel:[0,1,0] - and on this part that form dropped
        // the product from the LOW result in lanes 48-63 now and then, whenever two of these workgroups shared a CU (the
        // 128 x 64 tile past 256 blocks: a handful of outputs per launch came out as if the mean were zero; 28 of 30
        // launches; never with the mean broadcast from the first source, which is what this form compiles to.  The instruction
        // alone computes correctly under the same co-residency (tools/micro/pk_fma_opsel_probe.hip), so the cause is an
        // interaction of that compiled sequence and is open; tests/test_host_cpu.py::
        // test_no_packed_fma_with_a_broadcast_second_source keeps the form out of the library and
        // test_small_tile_in_the_token_stream repeats the launches that showed it.)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[ni][mi][j] = __builtin_fmaf(-mean[mi], cs[j], acc[ni][mi][j]) * rstd[mi];
    }
}

// Fast 16-bit epilogue of one wave (row-major output, N % 8 == 0, column bias).  Two 16-column MFMA blocks that are
// neighbours in N are exchanged between the lane rows with v_permlane16_swap, so a lane owns 8 consecutive output
// columns of ONE block: stores (and residual loads) become 16 bytes per lane and 64 contiguous bytes per output row
// instead of 8 and 32.  P0 = first block of the first pair, chosen by the caller so every pair starts on a 64-byte
// boundary; blocks left over (TN odd) keep the 8-byte form.  All residual / embedding loads of a row block are
// requested before the first one is consumed.
#ifndef SDMI_LNF_DEBUG_NOSTAT
#define SDMI_LNF_DEBUG_NOSTAT 0   // timing experiment only: the token-stream kernels without the row-statistics stores
#endif
#ifndef SDMI_RES_HOIST
#define SDMI_RES_HOIST 1      // A/B (tools/ab_build.sh): 0 = residual requested row block by row block everywhere
#endif
// LNF (the instantiations of the transformer blocks' token stream): ln_mean != nullptr - the LayerNorm-folded form applied
// here instead of in place (ln_mean / ln_rstd: the lane's MI rows, LNS floats apart - in LDS for the 256 x 256 x 32 tile,
// which has no registers for a separate pass over its accumulators); stat_slot >= 0 - the row sums of the values stored
// go to p.row_stats_out[m][stat_slot] (the statistics a LayerNorm-folded consumer of this output reads).
// PRE (sdmi_xattn.hip, one resident workgroup per CU - nobody to hide the trip behind): bias and residual were requested
// ahead of the main loop by epilogue_rows16_prefetch (same template arguments, same lane -> element map) and arrive in
// pre_bias / pre_res.
template <int DT, int TN, int MI, int P0, bool HOIST = (SDMI_RES_HOIST != 0), bool LNF = false, int LNS = 1, bool PRE = false>
__device__ __forceinline__ void epilogue_rows16(const sdmi_gemm_args& p, f32x4 (&acc)[TN][MI], int64_t mrow, int64_t ncol_w,
                                                int fq, int64_t zoff_out, int64_t zoff_res, const float* ln_mean = nullptr,
                                                const float* ln_rstd = nullptr, int stat_slot = -1,
                                                const f32x4* pre_bias = nullptr, const uint2 (*pre_res)[TN] = nullptr) {
    const int odd = fq & 1;
    const int64_t ncol0 = ncol_w + 4 * fq;                  // own quad inside block 0
    const int64_t ncol8 = ncol_w + odd * 16 + 4 * (fq & 2); // own 8 columns inside pair 0 (after the swap)
    f32x4 bq[TN];
    bool nval[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int64_t n = ncol0 + ni * 16;
        nval[ni] = n < p.N;
        bq[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (PRE) bq[ni] = pre_bias[ni];
        else if (p.bias != nullptr && nval[ni]) bq[ni] = *reinterpret_cast<const f32x4*>(p.bias + n);
    }
    f32x4 csq[LNF ? TN : 1];
    if constexpr (LNF) {
        if (ln_mean != nullptr) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) csq[ni] = *reinterpret_cast<const f32x4*>(p.ln_colsum + ncol0 + ni * 16);
        }
    }
    const unsigned short* const resp = reinterpret_cast<const unsigned short*>(p.res) + zoff_res;
    unsigned short* const outp = reinterpret_cast<unsigned short*>(p.out) + zoff_out;
    // The residual of ALL row blocks is requested before the first one is consumed: one trip to memory per wave instead
    // of one per row block.  In the step the residual is cold (written several kernels earlier, long evicted from the
    // Infinity Cache): 65536 x 320 x 320 + residual on operands that are not cache-warm 45 -> 38 us, 16384 x 640 x 640
    // 30.3 -> 26.7, 65536 x 960 x 320 120 -> 102 (tools/bench_gemm.py ROTATE=8).  Touching the residual tile's lines ahead
    // of the main loop as well (three scattered LDS-DMA instructions per wave into a dump area) was measured slower
    // (38 -> 44 us): 64 lines per instruction cost more in the address path than the late trip they save.  (Each lane reads exactly the elements it later writes, so out == res stays legal.)
    // (HOIST = false - the two-stage variant, capped at 128 VGPRs - requests them row block by row block.)
    uint2 rq_all[MI][TN];
#define SDMI_RES_LOAD_ROWBLOCK(mi)                                                                         \
    do {                                                                                                   \
        const int64_t m = mrow + (mi) * 16;                                                                \
        const bool mval = m < p.M;                                                                         \
        _Pragma("unroll") for (int ni = 0; ni < TN; ++ni) rq_all[mi][ni] = make_uint2(0u, 0u);             \
        _Pragma("unroll") for (int ni = P0; ni + 1 < TN; ni += 2) {                                        \
            uint4 r4 = make_uint4(0u, 0u, 0u, 0u);                                                         \
            const int64_t n = ncol8 + ni * 16;                                                             \
            if (mval && n < p.N) r4 = *reinterpret_cast<const uint4*>(resp + m * p.ld_res + n);            \
            rq_all[mi][ni] = make_uint2(r4.x, r4.y);      /* (8-columns-per-lane order: swapped below) */  \
            rq_all[mi][ni + 1] = make_uint2(r4.z, r4.w);                                                   \
        }                                                                                                  \
        _Pragma("unroll") for (int ni = 0; ni < TN; ++ni) {                                                \
            const bool single = ni < P0 || (ni == TN - 1 && ((TN - P0) & 1));                              \
            if (single && mval && nval[ni])                                                                \
                rq_all[mi][ni] = *reinterpret_cast<const uint2*>(resp + m * p.ld_res + ncol0 + ni * 16);   \
        }                                                                                                  \
    } while (0)
    if constexpr (PRE) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) rq_all[mi][ni] = pre_res[mi][ni];
    } else if (HOIST && p.res != nullptr) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) SDMI_RES_LOAD_ROWBLOCK(mi);
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int64_t m = mrow + mi * 16;
        const bool mval = m < p.M;
        uint2 rq[TN];
        f32x4 eq[TN];
        if (p.res != nullptr) {
            if (!HOIST && !PRE) SDMI_RES_LOAD_ROWBLOCK(mi);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) rq[ni] = rq_all[mi][ni];
#pragma unroll
            for (int ni = P0; ni + 1 < TN; ni += 2) {
                const auto sx = __builtin_amdgcn_permlane16_swap(rq_all[mi][ni].x, rq_all[mi][ni + 1].x, false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(rq_all[mi][ni].y, rq_all[mi][ni + 1].y, false, false);
                rq[ni] = make_uint2(sx[0], sy[0]);
                rq[ni + 1] = make_uint2(sx[1], sy[1]);
            }
        }
        if (!LNF && p.emb != nullptr) {
            const float* e = p.emb + (mval ? m / p.rows_per_batch : 0) * p.ld_emb;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                eq[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (mval && nval[ni]) eq[ni] = *reinterpret_cast<const f32x4*>(e + ncol0 + ni * 16);
            }
        }
        uint2 ov[TN];
        float st1 = 0.f, st2 = 0.f;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            f32x4 v;
            if (LNF && ln_mean != nullptr) v = (acc[ni][mi] - ln_mean[mi * LNS] * csq[ni]) * (ln_rstd[mi * LNS] * p.alpha) + bq[ni];
            else                           v = acc[ni][mi] * p.alpha + bq[ni];
            if (p.act == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = silu_f(v[j]);
            } else if (p.act == 2) {
                v = gelu_erf_f4(v);
            }
            if (!LNF && p.emb != nullptr) v += eq[ni];
            if (p.res != nullptr) {
                float rf[4];
                elem_unpack4_sel<DT>(rq[ni], rf, p.res_f16);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += rf[j];
            }
            ov[ni].x = elem_pack2_sel<DT>(v[0], v[1], p.out_f16);
            ov[ni].y = elem_pack2_sel<DT>(v[2], v[3], p.out_f16);
            if constexpr (LNF) {      // (columns past N hold exact zeros: zero weight rows, no bias / residual read)
                st1 += (v[0] + v[1]) + (v[2] + v[3]);
                st2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
        }
        if constexpr (LNF) {
            if (stat_slot >= 0 && !SDMI_LNF_DEBUG_NOSTAT) {     // (wave-uniform) the four lane groups hold different columns of the same row
                st1 = sum_lane_groups(st1);
                st2 = sum_lane_groups(st2);
                if (fq == 0 && mval) {
                    f32x2* const so = reinterpret_cast<f32x2*>(p.row_stats_out);
                    so[m * p.row_stats_slots + stat_slot] = (f32x2){st1, st2};
                    if (p.dup_rows) so[(m + p.M) * p.row_stats_slots + stat_slot] = (f32x2){st1, st2};
                }
            }
        }
#pragma unroll
        for (int ni = P0; ni + 1 < TN; ni += 2) {
            const auto sx = __builtin_amdgcn_permlane16_swap(ov[ni].x, ov[ni + 1].x, false, false);
            const auto sy = __builtin_amdgcn_permlane16_swap(ov[ni].y, ov[ni + 1].y, false, false);
            const int64_t n = ncol8 + ni * 16;
            if (mval && n < p.N) {
                *reinterpret_cast<uint4*>(outp + m * p.ld_out + n) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                if (p.dup_rows) *reinterpret_cast<uint4*>(outp + (m + p.M) * p.ld_out + n) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            }
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const bool single = ni < P0 || (ni == TN - 1 && ((TN - P0) & 1));
            if (single && mval && nval[ni]) {
                *reinterpret_cast<uint2*>(outp + m * p.ld_out + ncol0 + ni * 16) = ov[ni];
                if (p.dup_rows) *reinterpret_cast<uint2*>(outp + (m + p.M) * p.ld_out + ncol0 + ni * 16) = ov[ni];
            }
        }
    }
}

// The loads of epilogue_rows16<..., PRE = true>: column bias and residual of the wave's tile, in the epilogue's own lane ->
// element map (p.res must be set; a null p.bias reads as zeros).
template <int DT, int TN, int MI, int P0>
__device__ __forceinline__ void epilogue_rows16_prefetch(const sdmi_gemm_args& p, int64_t mrow, int64_t ncol_w, int fq,
                                                         int64_t zoff_res, f32x4 (&bq)[TN], uint2 (&rq_all)[MI][TN]) {
    const int odd = fq & 1;
    const int64_t ncol0 = ncol_w + 4 * fq;
    const int64_t ncol8 = ncol_w + odd * 16 + 4 * (fq & 2);
    bool nval[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int64_t n = ncol0 + ni * 16;
        nval[ni] = n < p.N;
        bq[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr && nval[ni]) bq[ni] = *reinterpret_cast<const f32x4*>(p.bias + n);
    }
    const unsigned short* const resp = reinterpret_cast<const unsigned short*>(p.res) + zoff_res;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) SDMI_RES_LOAD_ROWBLOCK(mi);
}

#undef SDMI_RES_LOAD_ROWBLOCK

// GEGLU epilogue of one wave over FOUR 16-column weight blocks, 16-bit output (blocks (0,1) and (2,3) are (value, gate)
// pairs; used by the panel kernel - TN == 4 - and, twice per wave, by the 256 x 256 x 32 tile): the two
// gated 16-column blocks are paired through v_permlane16_swap - 16-byte stores, 64 contiguous bytes per row.
template <int DT, int MI, bool LNF = false, int LNS = 1>
__device__ __forceinline__ void epilogue_geglu16(const sdmi_gemm_args& p, f32x4 (&acc)[4][MI], int64_t mrow, int64_t wcol_w,
                                                 int fq, const float* ln_mean = nullptr, const float* ln_rstd = nullptr) {
    f32x4 bv[2], bg[2];
    f32x4 cv[LNF ? 2 : 1], cg[LNF ? 2 : 1];
    if constexpr (LNF) {      // (the LNF GEGLU epilogue is only entered with a folded LayerNorm)
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            cv[pi] = *reinterpret_cast<const f32x4*>(p.ln_colsum + wcol_w + pi * 32 + 4 * fq);
            cg[pi] = *reinterpret_cast<const f32x4*>(p.ln_colsum + wcol_w + pi * 32 + 4 * fq + 16);
        }
    }
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
        const int64_t wcol = wcol_w + pi * 32 + 4 * fq;
        bv[pi] = bg[pi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr && wcol + 19 < 2 * p.N) {
            bv[pi] = *reinterpret_cast<const f32x4*>(p.bias + wcol);
            bg[pi] = *reinterpret_cast<const f32x4*>(p.bias + wcol + 16);
        }
    }
    const int64_t n = (wcol_w >> 1) + (fq & 1) * 16 + 4 * (fq & 2);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int64_t m = mrow + mi * 16;
        uint2 ov[2];
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            f32x4 v, g;
            if constexpr (LNF) {
                const float sc = ln_rstd[mi * LNS] * p.alpha, mu = ln_mean[mi * LNS];
                v = (acc[2 * pi][mi] - mu * cv[pi]) * sc + bv[pi];
                g = (acc[2 * pi + 1][mi] - mu * cg[pi]) * sc + bg[pi];
            } else {
                v = acc[2 * pi][mi] * p.alpha + bv[pi];
                g = acc[2 * pi + 1][mi] * p.alpha + bg[pi];
            }
            v *= gelu_erf_f4(g);
            ov[pi].x = elem_pack2<DT>(v[0], v[1]);
            ov[pi].y = elem_pack2<DT>(v[2], v[3]);
        }
        const auto sx = __builtin_amdgcn_permlane16_swap(ov[0].x, ov[1].x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(ov[0].y, ov[1].y, false, false);
        if (m < p.M && n < p.N)
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p.out) + m * p.ld_out + n) =
                make_uint4(sx[0], sy[0], sx[1], sy[1]);
    }
}


// One workgroup = 8 waves (4 x 2), block tile (64*MI) x (32*TN) x 64, one workgroup per CU (two waves per SIMD),
// 3-stage LDS ring fed by LDS-DMA with counted vmcnt waits and one barrier per k-tile.
// (the 2-stage variant: two workgroups per CU).  The second __launch_bounds__ argument is waves per SIMD in HIP, not
// blocks per CU: 4 caps the 2-stage variant at 128 VGPRs so that two of its workgroups really fit.
// LNF: the instantiations that consume a folded LayerNorm (ln_colsum + ln_stats) and / or emit row statistics
// (row_stats_out); mode 0, f16 - their own, so no other launch carries the extra epilogue registers.
template <int MODE, int DT, int TN, int MI, int NST, bool LNF = false>
__global__ __launch_bounds__(512, NST == 2 ? 4 : 2) void sdmi_gemm_kernel(const sdmi_gemm_args p) {
    constexpr int BM = 64 * MI;                 // 4 waves across M, MI 16-row blocks each
    constexpr int BN = TN * 32;                 // 2 waves across N, TN 16-column blocks each
    constexpr int A_CHUNKS = BM * (BK / 8);     // 16-byte chunks per A tile
    constexpr int W_CHUNKS = BN * (BK / 8);     // 16-byte chunks per W tile
    constexpr int STAGE_CHUNKS = A_CHUNKS + W_CHUNKS;
    constexpr int RPI = 64;                     // tile rows one DMA instruction of the whole block covers
    constexpr int NA = BM / RPI;                // A instructions per thread and stage (4 | 2 | 1)
    constexpr int NW = BN / RPI;                // W instructions every thread issues (2 | 1)
    static_assert(NA == 4 || NA == 2 || NA == 1, "row tile");
    static_assert(NW == 2 || NW == 1, "column tile");
    static_assert(MODE == 0 || NA >= 2, "the conv loader stages two or four 64-row pieces of A");
    constexpr bool W_TAIL = (BN % RPI) != 0;    // leftover 32 W rows (TN = 5): waves 0..3 issue one more
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint4* const lds = reinterpret_cast<const uint4*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    const int64_t n_w = p.geglu ? 2 * p.N : p.N;          // weight rows in use
    const int tiles_m = (int)((p.M + BM - 1) / BM);
    const int tiles_n = (int)((n_w + BN - 1) / BN);
    const TileCoord tc = tile_of_block(blockIdx.x, tiles_m, tiles_n);
    const int64_t m0 = (int64_t)tc.tm * BM;
    const int64_t n0 = (int64_t)tc.tn * BN;

    // grid.z is either the batch index or the split-K slice
    const int z = blockIdx.z;
    const int nk_total = (int)(p.K / BK);
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

    // ---- loader geometry: thread owns 16-byte chunk `lc` of rows lr + 64*i ----
    // A wave-instruction writes LDS at (wave-uniform base) + lane*16, i.e. 8 consecutive tile rows x 8 physical
    // chunks.  The XOR swizzle therefore goes on the SOURCE address: lane l owns physical chunk l&7 of row
    // 64*i + 8*wave + (l>>3) and fetches logical chunk (l&7) ^ (row&7).
    const int lr = tid >> 3;                    // tile rows lr + 64*i
    const int lc = (tid & 7) ^ (lr & 7);        // LOGICAL 16-byte chunk fetched into physical chunk tid&7
    const int wave_u = __builtin_amdgcn_readfirstlane(wid);
    const bool w_tail_wave = W_TAIL && wave_u < (BN - NW * RPI) / 8;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned d_wave = lds_base + wave_u * 1024;
    const unsigned short* const zsrc = reinterpret_cast<const unsigned short*>(sdmi_zeros);
    const int64_t k_valid = p.k_valid > 0 ? p.k_valid : p.K;

    // W: scalar base (tile row n0, k-tile) + constant per-lane byte offsets
    const unsigned short* const w_tile = W + n0 * p.K;
    unsigned w_voff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) w_voff[i] = (unsigned)((((int64_t)lr + RPI * i) * p.K + lc * 8) * 2);

    // A, mode 0: scalar base (tile row m0, k-tile) + constant per-lane byte offsets; rows past M re-read row M-1
    // (their results are never stored).  A k-tile that crosses k_valid takes the general path below.
    unsigned a_voff[NA], a2_voff[NA];
    // A, general path and conv
    int64_t a_off[NA];      // MODE 0: row m;  MODE 1: pixel base b*H*W
    int iy0[NA], ix0[NA];   // MODE 1
    bool rvalid[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int64_t m = m0 + lr + RPI * i;
        rvalid[i] = m < p.M;
        if (MODE == 0) {
            a_off[i] = m;
            const int64_t rrel = (rvalid[i] ? m : p.M - 1) - m0;
            a_voff[i] = (unsigned)((rrel * p.lda + lc * 8) * 2);
            a2_voff[i] = (unsigned)((rrel * p.lda2 + lc * 8) * 2);
        } else {
            const int64_t hw = (int64_t)p.Ho * p.Wo;
            const int64_t b = m / hw;
            const int rem = (int)(m - b * hw);
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_off[i] = b * (int64_t)p.H * p.W_in;
            const int pad_lo = p.asym_pad ? 0 : 1;          // asym_pad: zero pad on the bottom / right only
            iy0[i] = rvalid[i] ? oy * p.stride - pad_lo : -(1 << 20);
            ix0[i] = ox * p.stride - pad_lo;
        }
    }
    const int Hl = p.H << p.upsample, Wl = p.W_in << p.upsample;
    // Conv with Cin % 64 == 0: a k-tile never straddles a tap, so (tap, channel run) is one scalar state per wave
    // and each row keeps a 64-bit pointer that is rebuilt when the tap changes and bumped by 64 channels otherwise.
    const bool conv_fast = MODE == 1 && (p.Cin % BK) == 0 && p.Cin <= ZERO_ELEMS;
    int f_tap = 0, f_ci = 0;
    bool f_new_tap = true;
    const unsigned short* a_ptr[NA];
    // general conv path: per-lane (tap, ci) of this thread's chunk, k = kt*64 + lc*8
    int c_tap = 0, c_ci = 0;
    if (MODE == 1) {
        f_tap = (kt_begin * BK) / p.Cin;
        f_ci = kt_begin * BK - f_tap * p.Cin;
        const int k = kt_begin * BK + lc * 8;
        c_tap = k / p.Cin;
        c_ci = k - c_tap * p.Cin;
#pragma unroll
        for (int i = 0; i < NA; ++i) a_ptr[i] = zsrc;
    }

    // general path: one load per statement, 64-bit pointer, zero source where masked
#define SDMI_GLDS(SRC, D0, OFF)                                                                            \
    do {                                                                                                   \
        unsigned keep_;                                                                                    \
        const void* src_ = (SRC);                                                                          \
        const unsigned dst_ = (D0) + (OFF);                                                                \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"                              \
                     "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"                                 \
                     : "=&s"(keep_) : "v"(src_), "s"(dst_) : "memory");                                    \
    } while (0)
#define SDMI_STAGE_A0(I, KK, BASE, LD, D0)                                                                 \
    SDMI_GLDS((rvalid[I] && kin_) ? (BASE) + a_off[I] * (LD) + (KK) : zsrc, D0, 8192 * (I))
#define SDMI_STAGE_A1(I, KY, KX, TV, D0)                                                                   \
    do {                                                                                                   \
        int iy_ = iy0[I] + (KY), ix_ = ix0[I] + (KX);                                                      \
        const bool ok_ = (TV) && (unsigned)iy_ < (unsigned)Hl && (unsigned)ix_ < (unsigned)Wl;             \
        iy_ >>= p.upsample; ix_ >>= p.upsample;                                                            \
        const int64_t off_ = (a_off[I] + (int64_t)iy_ * p.W_in + ix_) * p.Cin + c_ci;                      \
        SDMI_GLDS(ok_ ? A + off_ : zsrc, D0, 8192 * (I));                                                  \
    } while (0)
#ifndef SDMI_DEBUG_CENTER_TAP
#define SDMI_DEBUG_CENTER_TAP 0   // timing experiment only (results wrong): every tap reads the centre tap's pixels, i.e. perfect L2 reuse of A
#endif
#define SDMI_TAP_PTR(I, KY, KX)                                                                            \
    do {                                                                                                   \
        int iy_ = iy0[I] + (SDMI_DEBUG_CENTER_TAP ? 1 : (KY)), ix_ = ix0[I] + (SDMI_DEBUG_CENTER_TAP ? 1 : (KX)); \
        const bool ok_ = (unsigned)iy_ < (unsigned)Hl && (unsigned)ix_ < (unsigned)Wl;                     \
        iy_ >>= p.upsample; ix_ >>= p.upsample;                                                            \
        const int64_t off_ = (a_off[I] + (int64_t)iy_ * p.W_in + ix_) * p.Cin;                             \
        a_ptr[I] = (ok_ ? A + off_ : zsrc) + (f_ci + lc * 8);                                              \
    } while (0)

    // Issue this wave's share of the DMA of k-tile KT into ring stage BUF (A tile first, W tile behind it).
#define SDMI_STAGE_TILE(KT, BUF)                                                                           \
    do {                                                                                                   \
        const unsigned d0_ = d_wave + (BUF) * (STAGE_CHUNKS * 16);                                         \
        const int64_t k0_ = (int64_t)(KT) * BK;                                                            \
        if constexpr (MODE == 0) {                                                                         \
            if (k0_ + BK <= k_valid) {                                                                     \
                const bool second_ = (A2 != nullptr) && (k0_ >= p.k_split);                                \
                const unsigned short* sa_ = second_ ? A2 + m0 * p.lda2 + (k0_ - p.k_split) : A + m0 * p.lda + k0_; \
                if constexpr (NA == 4)                                                                     \
                    glds_s4<0>(d0_, sa_, second_ ? a2_voff[0] : a_voff[0], second_ ? a2_voff[1] : a_voff[1], \
                               second_ ? a2_voff[NA - 2] : a_voff[NA - 2], second_ ? a2_voff[NA - 1] : a_voff[NA - 1]); \
                else if constexpr (NA == 2)                                                                \
                    glds_s2<0>(d0_, sa_, second_ ? a2_voff[0] : a_voff[0], second_ ? a2_voff[NA - 1] : a_voff[NA - 1]); \
                else                                                                                       \
                    glds_s1<0>(d0_, sa_, second_ ? a2_voff[0] : a_voff[0]);                                \
            } else {                                                                                       \
                const int64_t k_ = k0_ + lc * 8;                                                           \
                const bool kin_ = k_ < k_valid;      /* columns past k_valid (K padding) read as zero */   \
                const bool second_ = (A2 != nullptr) && (k_ >= p.k_split);                                 \
                const unsigned short* base_ = second_ ? A2 : A;                                            \
                const int64_t ld_ = second_ ? p.lda2 : p.lda;                                              \
                const int64_t kk_ = second_ ? k_ - p.k_split : k_;                                         \
                SDMI_STAGE_A0(0, kk_, base_, ld_, d0_);                                                    \
                if constexpr (NA >= 2) SDMI_STAGE_A0(1, kk_, base_, ld_, d0_);                             \
                if constexpr (NA == 4) { SDMI_STAGE_A0(NA - 2, kk_, base_, ld_, d0_); SDMI_STAGE_A0(NA - 1, kk_, base_, ld_, d0_); } \
            }                                                                                              \
        } else if (conv_fast) {                                                                            \
            if (f_new_tap) {                                                                               \
                const int ky_ = f_tap / 3, kx_ = f_tap - 3 * ky_;                                          \
                SDMI_TAP_PTR(0, ky_, kx_); SDMI_TAP_PTR(1, ky_, kx_);                                      \
                if (NA == 4) { SDMI_TAP_PTR(NA - 2, ky_, kx_); SDMI_TAP_PTR(NA - 1, ky_, kx_); }           \
            }                                                                                              \
            if (SDMI_DEBUG_A_FILL_NINTH && ((KT) % 9) != 0) { /* timing experiment: see the macro */ }           \
            else if (NA == 4) glds_v4<0>(d0_, a_ptr[0], a_ptr[1], a_ptr[NA - 2], a_ptr[NA - 1]);           \
            else              glds_v2<0>(d0_, a_ptr[0], a_ptr[1]);                                         \
            _Pragma("unroll")                                                                              \
            for (int i_ = 0; i_ < NA; ++i_) a_ptr[i_] += BK;                                               \
            f_ci += BK;                                                                                    \
            f_new_tap = f_ci == p.Cin;                                                                     \
            if (f_new_tap) { f_ci = 0; ++f_tap; }                                                          \
        } else {                                                                                           \
            const int ky_ = c_tap / 3, kx_ = c_tap - 3 * ky_;                                              \
            const bool tapvalid_ = c_tap < 9;                                                              \
            SDMI_STAGE_A1(0, ky_, kx_, tapvalid_, d0_); SDMI_STAGE_A1(1, ky_, kx_, tapvalid_, d0_);        \
            if (NA == 4) { SDMI_STAGE_A1(NA - 2, ky_, kx_, tapvalid_, d0_); SDMI_STAGE_A1(NA - 1, ky_, kx_, tapvalid_, d0_); } \
            c_ci += BK;                                                                                    \
            while (c_ci >= p.Cin) { c_ci -= p.Cin; ++c_tap; }                                              \
        }                                                                                                  \
        if constexpr (NW == 1) glds_s1<A_CHUNKS * 16>(d0_, w_tile + k0_, w_voff[0]);                       \
        else if (w_tail_wave)  glds_s3<A_CHUNKS * 16>(d0_, w_tile + k0_, w_voff[0], w_voff[1], w_voff[2]); \
        else                   glds_s2<A_CHUNKS * 16>(d0_, w_tile + k0_, w_voff[0], w_voff[1]);            \
    } while (0)
#define SDMI_WAIT_DMA(KEEP)                                                                                \
    do {                                                                                                   \
        if (w_tail_wave) wait_dma<NA + NW + 1>(KEEP); else wait_dma<NA + NW>(KEEP);                        \
    } while (0)

    f32x4 acc[TN][MI];   // [ni][mi]
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fq = lane >> 4;
    const int a_base = (wm * (MI * 16) + frow) * 8;
    const int w_base = A_CHUNKS + (wn * (TN * 16) + frow) * 8;

    const int nk = kt_end - kt_begin;

    // ---- main loop ----
    // Software pipeline over half k-tiles so that every barrier is followed by MFMA work whose operands are
    // already in registers: after the barrier that publishes tile t+1 the wave issues the ds_reads of its first
    // half (f0), runs the held-back second-half MFMAs of tile t (f1, read before the barrier), issues its DMA
    // share of tile t+3 into the stage tile t just vacated, reads the second half of tile t+1 and runs its
    // first-half MFMAs.  Ring: tile t+1 being read, t+2 landed or landing, t+3 in flight.
    uint4 af0[MI], wf0[TN], af1[MI], wf1[TN];
#define SDMI_READ_FRAGS(AF, WF, STAGE, KS)                                                                 \
    do {                                                                                                   \
        const uint4* St_ = lds + (STAGE) * STAGE_CHUNKS;                                                   \
        const int pc_ = ((KS) * 4 + fq) ^ (frow & 7);                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_)                                                  \
            AF[i_] = St_[a_base + (SDMI_DEBUG_HALF_FRAG_READS ? i_ / 2 * 2 : i_) * 128 + pc_];             \
        _Pragma("unroll") for (int i_ = 0; i_ < TN; ++i_)                                                  \
            WF[i_] = St_[w_base + (SDMI_DEBUG_HALF_FRAG_READS ? i_ / 2 * 2 : i_) * 128 + pc_];             \
    } while (0)
#define SDMI_MFMA_HALF(AF, WF)                                                                             \
    do {                                                                                                   \
        if (SDMI_SETPRIO) __builtin_amdgcn_s_setprio(1);                                                   \
        if (SDMI_DEBUG_ABLATE == 1) {                                                                      \
            _Pragma("unroll") for (int mi_ = 0; mi_ < MI; ++mi_)                                           \
                asm volatile("" :: "v"(AF[mi_].x), "v"(AF[mi_].y), "v"(AF[mi_].z), "v"(AF[mi_].w));        \
            _Pragma("unroll") for (int ni_ = 0; ni_ < TN; ++ni_)                                           \
                asm volatile("" :: "v"(WF[ni_].x), "v"(WF[ni_].y), "v"(WF[ni_].z), "v"(WF[ni_].w));        \
        } else {                                                                                           \
        _Pragma("unroll") for (int ni_ = 0; ni_ < TN; ++ni_)                                               \
            _Pragma("unroll") for (int mi_ = 0; mi_ < MI; ++mi_)                                           \
                acc[ni_][mi_] = mfma16<DT>(WF[ni_], AF[mi_], acc[ni_][mi_]);                               \
        }                                                                                                  \
        if (SDMI_SETPRIO) __builtin_amdgcn_s_setprio(0);                                                   \
    } while (0)

    // scheduling hint for the region just written: one ds_read, then two MFMAs, MI+TN times (spreads the LDS reads
    // of the next half under the MFMAs of this one instead of bunching them)
#if SDMI_SCHED_INTERLEAVE
#define SDMI_INTERLEAVE()                                                                                  \
    do {                                                                                                   \
        _Pragma("unroll") for (int g_ = 0; g_ < MI + TN; ++g_) {                                           \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                             \
        }                                                                                                  \
    } while (0)
#else
#define SDMI_INTERLEAVE() do {} while (0)
#endif

#if SDMI_DEBUG_CLOCKS
    long long dbg_c0 = 0, dbg_w0 = 0;
    if (blockIdx.x == 0 && blockIdx.z == 0 && tid == 0) { dbg_c0 = clock64(); dbg_w0 = wall_clock64(); }
#endif
    int kt_issue = 0;
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        if (kt_issue < nk) {
            SDMI_STAGE_TILE(kt_begin + kt_issue, st);
            ++kt_issue;
        }
    }
    if (nk > 0) {
        SDMI_WAIT_DMA(kt_issue - 1);                 // tile 0 has landed (this wave's share)
        __syncthreads();                             // ... and everyone's
        SDMI_READ_FRAGS(af0, wf0, 0, 0);
        SDMI_READ_FRAGS(af1, wf1, 0, 1);
        SDMI_MFMA_HALF(af0, wf0);
        int rd = 1, wr = 0;                          // ring stage of tile kt+1 / of tile kt (free after the barrier)
        for (int kt = 0; kt + 1 < nk; ++kt) {
            if (SDMI_DEBUG_ABLATE != 2 && SDMI_DEBUG_ABLATE != 4) SDMI_WAIT_DMA(kt_issue - kt - 2);        // tile kt+1 landed; younger tiles may still fly
            if (SDMI_DEBUG_ABLATE != 3) __syncthreads();                         // every wave has finished READING tile kt as well
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SDMI_READ_FRAGS(af0, wf0, rd, 0);
            SDMI_MFMA_HALF(af1, wf1);                // second half of tile kt
            SDMI_INTERLEAVE();
            if (kt_issue < nk) {                     // (issued right behind the barrier instead: + 3-8 % on the conv shapes, profiles/r03f_kloop_ablation.txt)
                if (SDMI_DEBUG_ABLATE != 2) SDMI_STAGE_TILE(kt_begin + kt_issue, wr);
                ++kt_issue;
            }
            SDMI_READ_FRAGS(af1, wf1, rd, 1);
            SDMI_MFMA_HALF(af0, wf0);                // first half of tile kt+1
            SDMI_INTERLEAVE();
            wr = rd;
            rd = rd == NST - 1 ? 0 : rd + 1;
        }
        SDMI_MFMA_HALF(af1, wf1);                    // second half of the last tile
    }
#if SDMI_DEBUG_CLOCKS
    if (blockIdx.x == 0 && blockIdx.z == 0 && tid == 0) {
        sdmi_debug_clk[0] = clock64() - dbg_c0;
        sdmi_debug_clk[1] = wall_clock64() - dbg_w0;
    }
#endif

    // ---- epilogue ----
    // accumulator quad acc[ni][mi][j]: output row m = m0 + wm*16*MI + mi*16 + (lane&15),
    // weight row (column of out) wcol = n0 + wn*16*TN + ni*16 + 4*(lane>>4) + j.
    if constexpr (LNF) {
        // the folded LayerNorm, from the producer's row statistics (split-K: the reduce pass applies it)
        if (p.ln_colsum != nullptr && p.split_k <= 1) {
            float mean[MI], rstd[MI];
            lnf_rows_mean_rstd<MI>(p, m0 + wm * (MI * 16) + frow, fq, mean, rstd);
            lnf_apply<TN, MI>(p, acc, mean, rstd, n0 + wn * (TN * 16) + 4 * fq);
        }
    }
    if (p.split_k > 1) {
        // raw partial sums to the workspace slab of this slice: [split][M][tiles_n*BN] fp32
        float* ws = reinterpret_cast<float*>(p.workspace) + (int64_t)z * p.M * ((int64_t)tiles_n * BN);
        const int64_t ldw = (int64_t)tiles_n * BN;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int64_t m = m0 + wm * (MI * 16) + mi * 16 + frow;
            if (m >= p.M) continue;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int64_t wcol = n0 + wn * (TN * 16) + ni * 16 + 4 * fq;
                *reinterpret_cast<f32x4*>(ws + m * ldw + wcol) = acc[ni][mi];
            }
        }
        return;
    }

    // Common case (no GEGLU, row-major output, N % 4 == 0, column bias): all residual / embedding quads of a row
    // block are requested before the first one is consumed, so the loads overlap instead of forming one
    // load -> wait -> store round trip per quad.
    if (!p.geglu && p.out_kind == 0 && (p.N & 7) == 0 && !p.bias_per_row && ((p.ld_out | p.ld_res) & 7) == 0 &&
        (((uintptr_t)p.out | (uintptr_t)p.res | (uintptr_t)(2 * zoff_out) | (uintptr_t)(2 * zoff_res)) & 15) == 0) {
        const int64_t mrow = m0 + wm * (MI * 16) + frow, ncol_w = n0 + wn * (TN * 16);
        // pairs start on even 16-column blocks of the output: with 5 blocks per wave the odd wave column shifts by one
        // (LNF: the row sums of what is stored, one slot per column tile and wave column - sdmi_gemm_args.row_stats_out)
        const int slot = (LNF && p.row_stats_out != nullptr) ? tc.tn * 2 + wn : -1;
        if ((TN & 1) && (wn & 1)) epilogue_rows16<DT, TN, MI, 1, (SDMI_RES_HOIST != 0) && NST != 2, LNF>(p, acc, mrow, ncol_w, fq, zoff_out, zoff_res, nullptr, nullptr, slot);
        else                      epilogue_rows16<DT, TN, MI, 0, (SDMI_RES_HOIST != 0) && NST != 2, LNF>(p, acc, mrow, ncol_w, fq, zoff_out, zoff_res, nullptr, nullptr, slot);
        return;
    }
    if (!p.geglu && p.out_kind != 2 && (p.N & 3) == 0 && !p.bias_per_row && !p.res_f32) {
        const int64_t ncol0 = n0 + wn * (TN * 16) + 4 * fq;
        f32x4 bq[TN];
        bool nval[TN];
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int64_t n = ncol0 + ni * 16;
            nval[ni] = n < p.N;
            bq[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p.bias != nullptr && nval[ni]) bq[ni] = *reinterpret_cast<const f32x4*>(p.bias + n);
        }
        const unsigned short* const resp = reinterpret_cast<const unsigned short*>(p.res) + zoff_res;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int64_t m = m0 + wm * (MI * 16) + mi * 16 + frow;
            const bool mval = m < p.M;
            uint2 rq[TN];
            f32x4 eq[TN];
            if (p.res != nullptr) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    rq[ni] = make_uint2(0u, 0u);
                    if (mval && nval[ni]) rq[ni] = *reinterpret_cast<const uint2*>(resp + m * p.ld_res + ncol0 + ni * 16);
                }
            }
            if (p.emb != nullptr) {
                const float* e = p.emb + (mval ? m / p.rows_per_batch : 0) * p.ld_emb;
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    eq[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (mval && nval[ni]) eq[ni] = *reinterpret_cast<const f32x4*>(e + ncol0 + ni * 16);
                }
            }
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                if (!(mval && nval[ni])) continue;
                const int64_t n = ncol0 + ni * 16;
                f32x4 v = acc[ni][mi] * p.alpha + bq[ni];
                if (p.act == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = p.out_kind != 0 ? silu_precise(v[j]) : silu_f(v[j]);
                } else if (p.act == 2) {
                    if (p.out_kind != 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = gelu_erf_precise(v[j]);
                    } else {
                        v = gelu_erf_f4(v);
                    }
                }
                if (p.emb != nullptr) v += eq[ni];
                if (p.res != nullptr) {
                    float rf[4];
                    elem_unpack4_sel<DT>(rq[ni], rf, p.res_f16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += rf[j];
                }
                if (p.out_kind == 0) {
                    uint2 ov;
                    ov.x = elem_pack2_sel<DT>(v[0], v[1], p.out_f16);
                    ov.y = elem_pack2_sel<DT>(v[2], v[3], p.out_f16);
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.out) + zoff_out + m * p.ld_out + n) = ov;
                } else {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + zoff_out + m * p.ld_out + n) = v;
                }
            }
        }
        return;
    }

    if (p.geglu) {
        // GEGLU runs with TN == 4 only (host-checked): blocks (0,1) and (2,3) are (value, gate) pairs
        f32x4 bv[TN / 2], bg[TN / 2];
#pragma unroll
        for (int pi = 0; pi < TN / 2; ++pi) {
            const int64_t wcol = n0 + wn * (TN * 16) + pi * 32 + 4 * fq;
            bv[pi] = bg[pi] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p.bias != nullptr && wcol + 19 < 2 * p.N) {
                bv[pi] = *reinterpret_cast<const f32x4*>(p.bias + wcol);
                bg[pi] = *reinterpret_cast<const f32x4*>(p.bias + wcol + 16);
            }
        }
        // 16-bit output with nothing else to add: the wave's two output blocks (pi = 0, 1: 32 adjacent columns) are
        // paired through v_permlane16_swap as in epilogue_rows16 - 16-byte stores, 64 contiguous bytes per row
        const bool wide = TN == 4 && p.out_kind == 0 && p.res == nullptr && p.emb == nullptr && !p.bias_per_row &&
                          (p.N & 31) == 0 && (p.ld_out & 7) == 0 &&
                          (((uintptr_t)p.out | (uintptr_t)(2 * zoff_out)) & 15) == 0;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int64_t m = m0 + wm * (MI * 16) + mi * 16 + frow;
            if (wide) {
                uint2 ov[TN / 2];
#pragma unroll
                for (int pi = 0; pi < TN / 2; ++pi) {
                    f32x4 v = acc[2 * pi][mi] * p.alpha + bv[pi], g = acc[2 * pi + 1][mi] * p.alpha + bg[pi];
                    v *= gelu_erf_f4(g);
                    ov[pi].x = elem_pack2<DT>(v[0], v[1]);
                    ov[pi].y = elem_pack2<DT>(v[2], v[3]);
                }
                const auto sx = __builtin_amdgcn_permlane16_swap(ov[0].x, ov[TN / 2 - 1].x, false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(ov[0].y, ov[TN / 2 - 1].y, false, false);
                // N % 32 == 0: the wave's 32 output columns are inside N or outside it as a whole
                const int64_t n = ((n0 + wn * (TN * 16)) >> 1) + (fq & 1) * 16 + 4 * (fq & 2);
                if (m < p.M && n < p.N)
                    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p.out) + zoff_out + m * p.ld_out + n) =
                        make_uint4(sx[0], sy[0], sx[1], sy[1]);
                continue;
            }
            if (m >= p.M) continue;
#pragma unroll
            for (int pi = 0; pi < TN / 2; ++pi) {
                const int64_t wcol = n0 + wn * (TN * 16) + pi * 32 + 4 * fq;
                const int64_t n = (wcol >> 5) * 16 + (wcol & 15);
                if (n >= p.N) continue;
                f32x4 v = acc[2 * pi][mi] * p.alpha + bv[pi], g = acc[2 * pi + 1][mi] * p.alpha + bg[pi];
                if (p.out_kind != 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= gelu_erf_precise(g[j]);
                } else {
                    v *= gelu_erf_f4(g);
                }
                store_quad<DT>(p, m, n, v, zoff_out, zoff_res);
            }
        }
        return;
    }

    // general path: partial column quads (N % 4 != 0), per-row bias, fp32 NCHW output
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int64_t m = m0 + wm * (MI * 16) + mi * 16 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int64_t n = n0 + wn * (TN * 16) + ni * 16 + 4 * fq;
            if (n >= p.N) continue;
            f32x4 v = acc[ni][mi] * p.alpha;
            if (p.bias != nullptr && !p.bias_per_row) {
                if (n + 3 < p.N) {
                    v += *reinterpret_cast<const f32x4*>(p.bias + n);
                } else {
                    for (int j = 0; j < 4 && n + j < p.N; ++j) v[j] += p.bias[n + j];
                }
            }
            store_quad<DT>(p, m, n, v, zoff_out, zoff_res);
        }
    }
}

}  // namespace

// One launcher per (MODE, DT) unit: [tile_n == 160][variant], variant 0 = 128 rows / 3 stages, 1 = 256 rows / 3 stages,
// 2 = 128 rows / 2 stages (64-72 KiB of LDS: two blocks per CU, for short K where the epilogue of one block should
// overlap the main loop of another).
#define SDMI_GEMM_UNIT(MODE, DT, NAME)                                                                              \
    namespace {                                                                                                     \
    typedef void (*kern_t)(const sdmi_gemm_args);                                                                   \
    const kern_t unit_kernels[2][3] = {                                                                             \
        {sdmi_gemm_kernel<MODE, DT, 4, 2, 3>, sdmi_gemm_kernel<MODE, DT, 4, 4, 3>, sdmi_gemm_kernel<MODE, DT, 4, 2, 2>}, \
        {sdmi_gemm_kernel<MODE, DT, 5, 2, 3>, sdmi_gemm_kernel<MODE, DT, 5, 4, 3>, sdmi_gemm_kernel<MODE, DT, 5, 2, 2>}}; \
    }                                                                                                               \
    int sdmi_gemm_unit_attrs_##NAME() {                                                                             \
        for (int c = 0; c < 2; ++c)                                                                                 \
            for (int d = 0; d < 3; ++d)                                                                             \
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(unit_kernels[c][d]),                          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize,                                 \
                                        lds_bytes(c ? 5 : 4, d == 1 ? 4 : 2, d == 2 ? 2 : 3)) != hipSuccess)        \
                    return SDMI_ERR_LAUNCH;                                                                         \
        return SDMI_OK;                                                                                             \
    }                                                                                                               \
    void sdmi_gemm_unit_launch_##NAME(int wide, int variant, dim3 grid, hipStream_t s, const sdmi_gemm_args& p) {   \
        hipLaunchKernelGGL(unit_kernels[wide][variant], grid, dim3(512),                                            \
                           lds_bytes(wide ? 5 : 4, variant == 1 ? 4 : 2, variant == 2 ? 2 : 3), s, p);              \
    }

// The token-stream instantiations (LNF: folded LayerNorm in, row statistics out; mode 0, f16; 128- and 256-row tiles, three stages).
#define SDMI_GEMM_UNIT_LNF(NAME)                                                                                    \
    namespace {                                                                                                     \
    typedef void (*lnf_kern_t)(const sdmi_gemm_args);                                                               \
    const lnf_kern_t lnf_kernels[2][2] = {                                                                          \
        {sdmi_gemm_kernel<0, 1, 4, 2, 3, true>, sdmi_gemm_kernel<0, 1, 4, 4, 3, true>},                             \
        {sdmi_gemm_kernel<0, 1, 5, 2, 3, true>, sdmi_gemm_kernel<0, 1, 5, 4, 3, true>}};                            \
    }                                                                                                               \
    int sdmi_gemm_unit_attrs_##NAME() {                                                                             \
        for (int c = 0; c < 2; ++c)                                                                                 \
            for (int d = 0; d < 2; ++d)                                                                             \
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(lnf_kernels[c][d]),                           \
                                        hipFuncAttributeMaxDynamicSharedMemorySize,                                 \
                                        lds_bytes(c ? 5 : 4, d == 1 ? 4 : 2, 3)) != hipSuccess)                     \
                    return SDMI_ERR_LAUNCH;                                                                         \
        return SDMI_OK;                                                                                             \
    }                                                                                                               \
    void sdmi_gemm_unit_launch_##NAME(int wide, int variant, dim3 grid, hipStream_t s, const sdmi_gemm_args& p) {   \
        hipLaunchKernelGGL(lnf_kernels[wide][variant], grid, dim3(512),                                             \
                           lds_bytes(wide ? 5 : 4, variant == 1 ? 4 : 2, 3), s, p);                                 \
    }

// The small tile (64 columns; 128 | 64 rows; three stages; mode 0): [0] = 128 rows, [1] = 64 rows.
// (A 64-row x 128-column variant for the GEGLU projections - two workgroups per CU, one block's erf epilogue under the
// other's k-steps - was built and measured: 8192 x 2560 x 320 29.4 -> 33.1 us, 2048 x 5120 x 640 24.8 -> 24.3, 512 x 10240 x
// 1280 21.7 -> 25.3, and 25-40 % slower from 4096 rows on: these launches are bound by the bytes a CU pulls into LDS, which a
// smaller tile only adds to.  Not kept.)
#define SDMI_GEMM_UNIT_SMALL(DT, LNF, NAME)                                                                         \
    namespace {                                                                                                     \
    typedef void (*small_kern_t_##NAME)(const sdmi_gemm_args);                                                      \
    const small_kern_t_##NAME small_kernels_##NAME[2] = {sdmi_gemm_kernel<0, DT, 2, 2, 3, LNF>,                     \
                                                         sdmi_gemm_kernel<0, DT, 2, 1, 3, LNF>};                    \
    }                                                                                                               \
    int sdmi_gemm_small_attrs_##NAME() {                                                                            \
        for (int d = 0; d < 2; ++d)                                                                                 \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(small_kernels_##NAME[d]),                         \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(2, d ? 1 : 2, 3)) != hipSuccess) \
                return SDMI_ERR_LAUNCH;                                                                             \
        return SDMI_OK;                                                                                             \
    }                                                                                                               \
    void sdmi_gemm_small_launch_##NAME(int rows64, dim3 grid, hipStream_t s, const sdmi_gemm_args& p) {             \
        hipLaunchKernelGGL(small_kernels_##NAME[rows64 ? 1 : 0], grid, dim3(512), lds_bytes(2, rows64 ? 1 : 2, 3), s, p); \
    }

// conv3x3 on the 128 x 64 block (mode 1; the few-hundred-row convs of the UNet at batch 1 - 2)
#define SDMI_GEMM_UNIT_SMALL_CONV(DT, NAME)                                                                         \
    int sdmi_gemm_small_conv_attrs_##NAME() {                                                                       \
        return hipFuncSetAttribute(reinterpret_cast<const void*>(sdmi_gemm_kernel<1, DT, 2, 2, 3>),                 \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(2, 2, 3)) == hipSuccess    \
                   ? SDMI_OK : SDMI_ERR_LAUNCH;                                                                     \
    }                                                                                                               \
    void sdmi_gemm_small_conv_launch_##NAME(dim3 grid, hipStream_t s, const sdmi_gemm_args& p) {                    \
        hipLaunchKernelGGL((sdmi_gemm_kernel<1, DT, 2, 2, 3>), grid, dim3(512), lds_bytes(2, 2, 3), s, p);          \
    }
