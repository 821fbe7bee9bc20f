// sdmi_xattn_block: the cross-attention half of a BasicTransformerBlock as ONE kernel, for C = 320 (level 0 of the UNet):
//
//     out = to_out( softmax( scale * to_q(LayerNorm(x)) K^T ) V ) + bias + x          (attention.py:300-380, 518-528)
//
// Every step is local to a token row except the read of the prompt's 77 keys / values, so a workgroup that owns 128
// rows can run the whole chain without a trip to memory in between - today three launches (to_q, attention, to_out) and
// two round trips of a [tokens, C] tensor:
//   1. the 128 x 320 panel of x lands in LDS in one LDS-DMA burst and is layer-normalised in place (as sdmi_gemm_panel.h);
//   2. q = panel @ Wq^T: Wq streams through a 3-stage ring, all 128 x 320 accumulators stay in registers (80 VGPRs);
//   3. q (f16) overwrites the panel - head h is exactly k-tile h of it;
//   4. attention: waves 0-3 take heads 0, 1, 2 and waves 4-7 heads 3, 4 of their 32 rows; V of a head (2 x 64-key tiles,
//      keys past nk zero) is staged in the idle ring, double-buffered one head ahead; K fragments come straight from
//      global memory into registers one (head, tile) ahead; the inner loop is the one of sdmi_attn.hip (S^T = K Q^T on
//      v_mfma_32x32x16, lane-local online softmax, P^T from the accumulators, V^T by transposed LDS reads) with the Q
//      fragments read from the panel, in steps of 32 keys (nk = 77: 3 steps where 64-key tiles pay for 4); the normalised
//      O overwrites q in the panel;
//   5. out = panel @ Wo^T + bias + x through the same ring, stored by the 16-bit epilogue of sdmi_gemm (which also leaves
//      the row statistics for a LayerNorm-folded consumer).
// Fragment layout, swizzles and MFMA operand roles are those of sdmi_gemm_kernel.h / sdmi_attn.hip.
#include "sdmi_gemm_panel.h"

namespace {

constexpr int XKT = 5, XTN = 5, XMI = 2, XBM = 128, XBN = 160, XNSTW = 3;
constexpr int X_A_KT_CHUNKS = XBM * 8;                 // 16-byte chunks of one k-tile of the panel (1024)
constexpr int X_PANEL_CHUNKS = XKT * X_A_KT_CHUNKS;    // 5120 (80 KiB)
constexpr int X_W_CHUNKS = XBN * 8;                    // one weight ring stage (1280 chunks, 20 KiB)
constexpr int X_RING_CHUNKS = 4096;                    // ring / K-V staging region (64 KiB >= 3 stages = 60 KiB)
constexpr int X_LDS_BYTES = (X_PANEL_CHUNKS + X_RING_CHUNKS) * 16;      // 147456
constexpr int X_TILE_CH = 512;                         // chunks of one 64-key K or V tile
constexpr int X_VBUF_CH = 2 * X_TILE_CH;               // a V staging buffer: two 64-key tiles (16 KiB)
constexpr float X_RESCALE_THR = 8.0f;

struct XattnArgs {
    const unsigned short* x; int64_t ldx; int64_t M; int rows_per_image;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    const unsigned short* wq; const unsigned short* wo; const float* bias_o;
    const unsigned short* k; const unsigned short* v; int64_t ldk, ldv; int nk;
    float c;                                            // scale * log2(e)
    unsigned short* out; int64_t ld_out;
    float* row_stats_out;
};

__device__ __forceinline__ uint4 x_cvt8(const f32x16& s, int base) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = s[base + j];
    return elem_pack8<1>(f);
}

__global__ __launch_bounds__(512, 2) void xattn_c320_kernel(const XattnArgs p) {
    constexpr int DT = 1, KT = XKT, TN = XTN, MI = XMI, NSTW = XNSTW, BK64 = 64;
    constexpr int RPI = 64, NAI = 2, NWI = 2, W_TAIL_WAVES = 4;       // 160 weight rows = 2 x 64 + 4 waves x 8
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* const lds = reinterpret_cast<uint4*>(smem);
    constexpr int W_RING0 = X_PANEL_CHUNKS;

    const int tid = threadIdx.x;
#ifdef SDMI_XATTN_TIMING
    long long tstamp[8]; int nts = 0;
#define X_TS() do { tstamp[nts++] = __builtin_readcyclecounter(); } while (0)
#else
#define X_TS() do {} while (0)
#endif
    X_TS();
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int64_t m0 = (int64_t)blockIdx.x * XBM;
    const int img = (int)(m0 / p.rows_per_image);      // (host-checked: rows_per_image % 128 == 0)

    const int lr = tid >> 3;
    const int lc = (tid & 7) ^ (lr & 7);
    const int wave_u = __builtin_amdgcn_readfirstlane(wid);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned d_wave = lds_base + wave_u * 1024;
    const bool w_tail_wave = wave_u < W_TAIL_WAVES;
    unsigned a_voff[NAI], w_voff[NWI + 1];
#pragma unroll
    for (int i = 0; i < NAI; ++i) a_voff[i] = (unsigned)((((int64_t)lr + RPI * i) * p.ldx + lc * 8) * 2);
#pragma unroll
    for (int i = 0; i < NWI; ++i) w_voff[i] = (unsigned)((((int64_t)lr + RPI * i) * 320 + lc * 8) * 2);
    w_voff[NWI] = (unsigned)((((int64_t)NWI * RPI + wave_u * 8 + (lane >> 3)) * 320 + (((lane & 7) ^ ((lane >> 3) & 7))) * 8) * 2);

    // ---- 1. the panel of x, one burst ----
    const unsigned short* const a_tile = p.x + m0 * p.ldx;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < NAI; ++i)
            glds1_s(d_wave + (kt * X_A_KT_CHUNKS + i * RPI * 8) * 16, a_tile + kt * BK64, a_voff[i]);

    constexpr int steps = 2 * KT;      // two 160-column tiles of a 320 x 320 weight
    const unsigned short* Wcur = p.wq;
    auto issue_w = [&](int step) {
        const int nt = step / KT, kt = step - nt * KT;
        const unsigned short* src = Wcur + ((int64_t)nt * XBN) * 320 + kt * BK64;
        const unsigned d0 = d_wave + (W_RING0 + (step % NSTW) * X_W_CHUNKS) * 16;
#pragma unroll
        for (int i = 0; i < NWI; ++i) glds1_s(d0 + i * RPI * 8 * 16, src, w_voff[i]);
        if (w_tail_wave) glds1_s(d0 + NWI * RPI * 8 * 16, src, w_voff[NWI]);
    };
    auto wait_w = [&](int young) {
        if (w_tail_wave) wait_young<NWI + 1>(young); else wait_young<NWI>(young);
    };
    int issued = 0;
#pragma unroll
    for (int s = 0; s < NSTW - 1; ++s) { issue_w(issued); ++issued; }

    // ---- LayerNorm of the resident panel (sdmi_gemm_panel.h) ----
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int row = tid >> 2, part = tid & 3;
        float f[2 * KT][8];
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint4 raw = lds[kt * X_A_KT_CHUNKS + row * 8 + ((part + 4 * h) ^ (row & 7))];
                elem_unpack8<DT>(raw, f[2 * kt + h]);
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += f[2 * kt + h][j];
            }
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        const float inv_k = 1.0f / 320.0f;
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
                const int ch = kt * BK64 + (part + 4 * h) * 8;
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
                lds[kt * X_A_KT_CHUNKS + row * 8 + ((part + 4 * h) ^ (row & 7))] = elem_pack8<DT>(y);
            }
        __syncthreads();
    }
    X_TS();

    const int frow = lane & 15, fq = lane >> 4;
    const int a_base = (wm * (MI * 16) + frow) * 8;
    const int w_base = (wn * (TN * 16) + frow) * 8;

    // one k-step of a panel GEMM into `acc` (column tile nt, k-tile kt, flat step `step`)
#define X_GEMM_STEP(ACC, step, kt)                                                                              \
    do {                                                                                                        \
        const uint4* Ap = lds + (kt) * X_A_KT_CHUNKS;                                                           \
        const uint4* Wp = lds + W_RING0 + ((step) % NSTW) * X_W_CHUNKS;                                         \
        uint4 af[2][MI], wf[2][TN];                                                                             \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                      \
            const int pc = (ks * 4 + fq) ^ (frow & 7);                                                          \
            _Pragma("unroll") for (int i = 0; i < MI; ++i) af[ks][i] = Ap[a_base + i * 128 + pc];               \
            _Pragma("unroll") for (int i = 0; i < TN; ++i) wf[ks][i] = Wp[w_base + i * 128 + pc];               \
        }                                                                                                       \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                        \
            _Pragma("unroll") for (int ni = 0; ni < TN; ++ni)                                                   \
                _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                               \
                    ACC[ni][mi] = mfma16<DT>(wf[ks][ni], af[ks][mi], ACC[ni][mi]);                              \
    } while (0)

    // ---- 2. q = LayerNorm(x) @ Wq^T: both column tiles stay in registers ----
    f32x4 accq[2][TN][MI];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < MI; ++b) accq[t][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        int step = 0;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt, ++step) {
                wait_w(issued - step - 1);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (issued < steps) { issue_w(issued); ++issued; }
                X_GEMM_STEP(accq[nt], step, kt);
            }
        }
    }

    // ---- 3. q (f16) overwrites the panel: element (row, n) -> k-tile n / 64, chunk (n % 64) / 8, half (n % 8) / 4 ----
    // Attention bookkeeping first: the loads of the first head's K (registers) and V (LDS-DMA into the now idle ring) fly
    // while q is written back.
    const int grp8 = wid >> 2, w4 = wid & 3;            // wave group (heads 0-2 | 3-4), wave inside it (32 rows each)
    const int nh = grp8 == 0 ? 3 : 2;
    const int lr32 = lane & 31, hf = lane >> 5;
    const int qrow = w4 * 32 + lr32;                    // row of the panel
    const int ntiles = (p.nk + 63) >> 6;                // (host-checked: nk <= 128)
    const int srow = 8 * w4 + (lane >> 3);
    const int vch = (lane & 7) ^ (((srow >> 1) & 1) << 2);
    // V staging buffers (16 KiB: two 64-key tiles) per (group, parity of the head step), placed so that the last step
    // (group 0 alone, parity 0) leaves the first two ring stages free for Wo: g1p0 | g1p1 | g0p1 | g0p0
    const int vb_ch0 = grp8 == 0 ? 3 * X_VBUF_CH : 0, vb_ch1 = grp8 == 0 ? 2 * X_VBUF_CH : X_VBUF_CH;
    const unsigned d_vb0 = lds_base + (unsigned)__builtin_amdgcn_readfirstlane((W_RING0 + vb_ch0) * 16 + w4 * 1024);
    const unsigned d_vb1 = lds_base + (unsigned)__builtin_amdgcn_readfirstlane((W_RING0 + vb_ch1) * 16 + w4 * 1024);
    const unsigned short* const zsrc = reinterpret_cast<const unsigned short*>(sdmi_zeros);
    const unsigned short* const Kimg = p.k + ((int64_t)img * p.nk) * p.ldk + hf * 8;
    const unsigned short* const Vimg = p.v + ((int64_t)img * p.nk) * p.ldv;
    // V of head step hs -> LDS: wave w4 issues rows {8 w4 .. +7, 32 + 8 w4 ..} of each tile, rows past nk zero
    auto stage_v = [&](int hs) {
        const unsigned short* Vp = Vimg + (grp8 * 3 + hs) * 64 + vch * 8;
        const unsigned d0 = (hs & 1) ? d_vb1 : d_vb0;
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (t < ntiles) {
                const int r0 = t * 64 + srow, r1 = r0 + 32;
                const unsigned short* v0 = r0 < p.nk ? Vp + (int64_t)r0 * p.ldv : zsrc;
                const unsigned short* v1 = r1 < p.nk ? Vp + (int64_t)r1 * p.ldv : zsrc;
                glds_v2<0, 4096>(d0 + t * (X_TILE_CH * 16), v0, v1);
            }
    };
    // K fragments of (head h, half tile j = 32 keys) straight into registers: lane (key lr32, half hf), k-step s: chunk
    // 2 s + hf.  Keys past nk read the last key (masked below).
    const int nhalf = (p.nk + 31) >> 5;
    auto load_k = [&](int h, int j, uint4 (&kf)[4]) {
        const int r0 = min(j * 32 + lr32, p.nk - 1);
        const uint4* a0 = reinterpret_cast<const uint4*>(Kimg + (int64_t)r0 * p.ldk + h * 64);
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = a0[2 * s];
    };
    __syncthreads();      // every wave has read its last fragments of x and of the ring
    X_TS();
    uint4 kf[4];
    stage_v(0);
    load_k(grp8 * 3, 0, kf);
    {
        unsigned char* const pan = smem;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int row = wm * (MI * 16) + mi * 16 + frow;
                    const int n = nt * XBN + wn * (TN * 16) + ni * 16 + 4 * fq;
                    const f32x4 v = accq[nt][ni][mi];
                    uint2 w;
                    w.x = elem_pack2<DT>(v[0], v[1]);
                    w.y = elem_pack2<DT>(v[2], v[3]);
                    const int kt = n >> 6, ch = (n & 63) >> 3, half = (n & 7) >> 2;
                    *reinterpret_cast<uint2*>(pan + (kt * X_A_KT_CHUNKS + row * 8 + (ch ^ (row & 7))) * 16 + half * 8) = w;
                }
    }

    X_TS();
    // ---- 4. attention over the prompt's keys (sdmi_attn.hip's inner loop; Q from the panel, O back into it) ----
    // One barrier per head step: V of the step was issued a step earlier (double buffer), K of the next (head, tile)
    // is loaded while the current tile's softmax and P V run.
    Wcur = p.wo;
    {
        const int grp = lane >> 4;
        const int tq = (lane & 15) >> 2, tp = lane & 3;
        unsigned v_lane[2];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int row = 4 * hf + tq;
            const int chunk = (db * 4 + 2 * (grp & 1) + (tp >> 1)) ^ (((row >> 1) & 1) << 2);
            v_lane[db] = row * 128 + (chunk << 4) + ((tp & 1) << 3);
        }
        for (int hs = 0; hs < 3; ++hs) {
            const int h = grp8 * 3 + hs;
            const bool active = hs < nh;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // V of this step (and K of its first tile) have landed
            __syncthreads();                                      // ... for every wave; q / the previous step's buffers are done with
            if (hs + 1 < nh) stage_v(hs + 1);
            if (hs == 2) {                                        // [0, 40 KiB) of the ring is idle: the first two stages of Wo
#pragma unroll
                for (int s = 0; s < NSTW - 1; ++s) issue_w(s);
            }
            if (active) {
                // Q fragments of head h from the panel (k-tile h): lane (q = lr32, half hf), k-step s: chunk 2 s + hf
                uint4 qf[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) qf[s] = lds[h * X_A_KT_CHUNKS + qrow * 8 + ((2 * s + hf) ^ (qrow & 7))];
                const unsigned char* const vbuf = smem + (W_RING0 + ((hs & 1) ? vb_ch1 : vb_ch0)) * 16;
                f32x16 o0, o1;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
                float nm = 0.f;
                f32x2 l_run = {0.f, 0.f};
                // 32 keys a step (nk = 77: 3 steps, where 64-key tiles would pay for 4)
                for (int j = 0; j < nhalf; ++j) {
                    const unsigned char* Vb = vbuf + j * 4096;
                    f32x16 s0;
                    {
                        f32x16 zero;
#pragma unroll
                        for (int i = 0; i < 16; ++i) zero[i] = 0.f;
#pragma unroll
                        for (int s = 0; s < 4; ++s) s0 = mfma32<DT>(kf[s], qf[s], s == 0 ? zero : s0);
                    }
                    {       // K of the next (head, half tile) of this wave
                        int j2 = j + 1, h2 = h;
                        if (j2 == nhalf) { j2 = 0; ++h2; }
                        if (h2 < grp8 * 3 + nh) load_k(h2, j2, kf);
                    }
                    uint4 vf[2][2];       // [ks][db]
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int db = 0; db < 2; ++db) {
                            const unsigned char* va_p = Vb + v_lane[db] + (16 * ks) * 128;
                            const short4v va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (__attribute__((address_space(3))) short4v*)(va_p));
                            const short4v vc = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (__attribute__((address_space(3))) short4v*)(va_p + 8 * 128));
                            typedef __attribute__((ext_vector_type(8))) short short8v;
                            short8v vv;
                            vv[0] = va[0]; vv[1] = va[1]; vv[2] = va[2]; vv[3] = va[3];
                            vv[4] = vc[0]; vv[5] = vc[1]; vv[6] = vc[2]; vv[7] = vc[3];
                            vf[ks][db] = __builtin_bit_cast(uint4, vv);
                        }
#pragma unroll
                    for (int r = 0; r < 16; ++r) s0[r] = fmaf(s0[r], p.c, nm);
                    if ((j + 1) * 32 > p.nk) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
                            if (key >= p.nk) s0[r] = -INFINITY;
                        }
                    }
                    float mx = s0[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[r]);
                    {
                        float ma = mx, mb = mx;
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
                        mx = fmaxf(ma, mb);
                    }
                    if (j == 0 || __builtin_amdgcn_ballot_w64(mx > X_RESCALE_THR) != 0) {
                        const float delta = j == 0 ? mx : fmaxf(mx, 0.f);
                        const float alpha = j == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);
                        nm -= delta;
                        l_run *= alpha;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            s0[i] -= delta;
                            o0[i] *= alpha; o1[i] *= alpha;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) s0[r] = __builtin_amdgcn_exp2f(s0[r]);
#pragma unroll
                    for (int r = 0; r < 16; r += 2) l_run += (f32x2){s0[r], s0[r + 1]};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const uint4 pf = x_cvt8(s0, 8 * ks);
                        o0 = mfma32<DT>(vf[ks][0], pf, o0);
                        o1 = mfma32<DT>(vf[ks][1], pf, o1);
                    }
                }
                // normalised O_h over q_h in the panel: o_db[r] = O[q][d = db*32 + (r&3) + 8*(r>>2) + 4*hf]
                const float l_lane = l_run[0] + l_run[1];
                const float l_tot = l_lane + __shfl_xor(l_lane, 32);
                const float inv = 1.0f / l_tot;
                unsigned char* const prow = smem + (h * X_A_KT_CHUNKS + qrow * 8) * 16;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 w0, w1;
                    w0.x = elem_pack2<DT>(o0[4 * g] * inv, o0[4 * g + 1] * inv);
                    w0.y = elem_pack2<DT>(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
                    w1.x = elem_pack2<DT>(o1[4 * g] * inv, o1[4 * g + 1] * inv);
                    w1.y = elem_pack2<DT>(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
                    // d0 = 8 g + 4 hf (+ 32): chunk g (+ 4), half hf
                    *reinterpret_cast<uint2*>(prow + ((g ^ (qrow & 7)) << 4) + hf * 8) = w0;
                    *reinterpret_cast<uint2*>(prow + (((4 + g) ^ (qrow & 7)) << 4) + hf * 8) = w1;
                }
            }
        }
        __syncthreads();      // O is visible to every wave; the V buffers are free
    }
    X_TS();

    // ---- 5. out = O @ Wo^T + bias + x ----
    sdmi_gemm_args pe = {};
    pe.M = p.M; pe.N = 320; pe.bias = p.bias_o; pe.res = p.x; pe.ld_res = p.ldx; pe.out = p.out; pe.ld_out = p.ld_out;
    pe.alpha = 1.0f; pe.row_stats_out = p.row_stats_out; pe.row_stats_slots = 4;
    {       // (the first two stages of Wo were issued under the last head step)
        const int64_t mrow = m0 + wm * (MI * 16) + frow;
        // Both column tiles accumulate before anything is stored: vmcnt counts stores too, so stores of the first tile in
        // flight would stall every counted wait of the second tile's ring (measured: + 6000 cycles a tile).
        f32x4 acc[2][TN][MI];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b) acc[t][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int step = 0;
        issued = NSTW - 1;      // (a compile-time count again: the waits and ring slots of the unrolled loop fold)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt, ++step) {
                wait_w(issued - step - 1);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (issued < steps) { issue_w(issued); ++issued; }
                X_GEMM_STEP(acc[nt], step, kt);
            }
        }
        X_TS();
        // bias and residual of both column tiles in one trip (with one workgroup per CU nobody else hides a second one)
        f32x4 pre_b[2][TN];
        uint2 pre_r[2][MI][TN];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int64_t ncol_w = (int64_t)nt * XBN + wn * (TN * 16);
            if (wn & 1) epilogue_rows16_prefetch<DT, TN, MI, 1>(pe, mrow, ncol_w, fq, 0, pre_b[nt], pre_r[nt]);
            else        epilogue_rows16_prefetch<DT, TN, MI, 0>(pe, mrow, ncol_w, fq, 0, pre_b[nt], pre_r[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int64_t ncol_w = (int64_t)nt * XBN + wn * (TN * 16);
            const int slot = p.row_stats_out != nullptr ? nt * 2 + wn : -1;
            if (wn & 1) epilogue_rows16<DT, TN, MI, 1, true, true, 1, true>(pe, acc[nt], mrow, ncol_w, fq, 0, 0, nullptr, nullptr, slot, pre_b[nt], pre_r[nt]);
            else        epilogue_rows16<DT, TN, MI, 0, true, true, 1, true>(pe, acc[nt], mrow, ncol_w, fq, 0, 0, nullptr, nullptr, slot, pre_b[nt], pre_r[nt]);
        }
    }
#undef X_GEMM_STEP
    X_TS();
#ifdef SDMI_XATTN_TIMING
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && p.row_stats_out == nullptr) {
        long long* d = reinterpret_cast<long long*>(p.out) + (blockIdx.x == 0 ? 0 : 8);     // (debug build: clobbers out)
        for (int i = 0; i < nts; ++i) d[i] = tstamp[i];
    }
#endif
#undef X_TS
}

}  // namespace

extern "C" int sdmi_xattn_block_applies(const sdmi_xattn_args* a) {
    if (a == nullptr) return 0;
    if (a->dtype != SDMI_DTYPE_F16 || a->C != 320 || a->heads != 5) return 0;
    if (a->M <= 0 || a->M % XBM != 0 || a->rows_per_image <= 0 || a->rows_per_image % XBM != 0 || a->M % a->rows_per_image != 0) return 0;
    if (a->nk <= 0 || a->nk > 128) return 0;
    if (a->ldx % 8 || a->ld_out % 8 || a->ldk % 8 || a->ldv % 8 || a->ldk < 320 || a->ldv < 320 || a->ldx < 320 || a->ld_out < 320) return 0;
    if ((((uintptr_t)a->x | (uintptr_t)a->out | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->wq | (uintptr_t)a->wo) & 15) != 0) return 0;
    if ((a->M / XBM) > 0x7fffffffLL || a->ldx > (1 << 22)) return 0;      // (panel row offsets are 32-bit byte offsets)
    return 1;
}

extern "C" int sdmi_xattn_block(const sdmi_xattn_args* a, void* stream) {
    if (a == nullptr || a->x == nullptr || a->out == nullptr || a->wq == nullptr || a->wo == nullptr || a->k == nullptr ||
        a->v == nullptr || a->ln_gamma == nullptr || a->ln_beta == nullptr || !(a->ln_eps >= 0.f))
        return SDMI_ERR_BAD_ARG;
    if (!sdmi_xattn_block_applies(a)) return SDMI_ERR_BAD_ARG;
    if (a->row_stats_out != nullptr && a->row_stats_slots != 4) return SDMI_ERR_BAD_ARG;
    static SdmiPerDeviceOnce attr_once;
    const int st = attr_once.run([] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_c320_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   X_LDS_BYTES) == hipSuccess;
    });
    if (st != SDMI_OK) return st;
    XattnArgs p;
    p.x = reinterpret_cast<const unsigned short*>(a->x); p.ldx = a->ldx; p.M = a->M; p.rows_per_image = a->rows_per_image;
    p.ln_gamma = a->ln_gamma; p.ln_beta = a->ln_beta; p.ln_eps = a->ln_eps;
    p.wq = reinterpret_cast<const unsigned short*>(a->wq); p.wo = reinterpret_cast<const unsigned short*>(a->wo);
    p.bias_o = a->bias_o;
    p.k = reinterpret_cast<const unsigned short*>(a->k); p.v = reinterpret_cast<const unsigned short*>(a->v);
    p.ldk = a->ldk; p.ldv = a->ldv; p.nk = a->nk;
    p.c = a->scale * 1.4426950408889634f;
    p.out = reinterpret_cast<unsigned short*>(a->out); p.ld_out = a->ld_out;
    p.row_stats_out = a->row_stats_out;
    hipLaunchKernelGGL(xattn_c320_kernel, dim3((unsigned)(a->M / XBM)), dim3(512), X_LDS_BYTES, reinterpret_cast<hipStream_t>(stream), p);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}
