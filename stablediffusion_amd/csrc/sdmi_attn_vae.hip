// sdmi_attention_vae: the single-head attention of the VAE's AttnBlock (ldm/modules/diffusion/model.py:496-528;
// head dim D = channels = 512 for SD2) for gfx950, without the [B, N, N] score tensor the reference materialises.
//
//   W[i, j] = softmax_j( scale * q_i . k_j )                      (row-normalised over the key index j)
//   mode 0 "standard"  : O[i, :] = sum_j W[i, j] v[j, :]           upstream's softmax(QK^T) V
//   mode 1 "reference" : O[j, :] = sum_i W[i, j] v[i, :]           the reference's literal contraction over the QUERY
//                                                                  index (model.py:519-523, SURVEY.md defect B7)
// In reference mode the softmax axis is not the reduction axis, so an online softmax does not apply; both modes
// therefore run the same two passes:
//   pass A (stats): per query i, m_i = max_j and 1/l_i = 1 / sum_j exp over all keys (QK^T only, nothing stored but
//                   two floats per query - the caller's workspace);
//   pass B (apply): output rows r in blocks of 128 (32 per wave, row on the lane), contraction index c in tiles of 32:
//                   T^T[c, r] = Y_c . X_r (MFMA 32x32x16, Y rows from LDS, X fragments resident in registers),
//                   weight = exp2(T*scale*log2e - m) / l with the statistics of the row (standard: X = Q, Y = K) or of
//                   the column (reference: X = K, Y = Q), converted to the 16-bit operand type in place and fed as the
//                   B operand of O^T[d, r] += V^T[d, c] W^T[c, r] (V^T through ds_read_b64_tr_b16).
// 6 N^2 D FLOP instead of 4 N^2 D (QK^T twice), no rescaling, deterministic; 1.4 % of the decoder's work.
// Y / V tiles (and the 32 column statistics of reference mode) go global -> LDS by LDS-DMA into a 2-stage ring;
// swizzles are applied to the DMA source address (Y: chunk ^ row&15, V: chunk ^ (row&3)<<2 - conflict-free
// ds_read_b128 / transposed reads at a 1 KiB row pitch).  One 4-wave workgroup per CU (~420 VGPRs at D = 512).
#include <mutex>
#include "sdmi_common.h"

namespace {

constexpr int RB = 128;     // output rows per workgroup (32 per wave)
constexpr int CT = 32;      // contraction tile (keys / queries per step)

__device__ const uint4 vae_zero16[64] = {};     // 1 KiB of zeros: DMA source of rows past N

struct VaeAttnArgs {
    const unsigned short* x; const unsigned short* y; const unsigned short* v; unsigned short* out;
    float* stats;                      // [nb][N] float2 {m (log2 units), 1 / l}
    int N;
    int64_t ldx, ldy, ldv, ldo;
    float c;                           // scale * log2(e)
};

template <int DT>
__device__ __forceinline__ uint4 cvt8v(const f32x16& s, int base) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = s[base + j];
    return elem_pack8<DT>(f);
}

// MODE 0: statistics pass.  MODE 1: apply with the statistics of the output row.  MODE 2: with those of the column.
template <int DT, int D, int MODE>
__global__ __launch_bounds__(256, 1) void vae_attn_kernel(const VaeAttnArgs p) {
    constexpr int CPR = D / 8;                          // 16-byte chunks per row
    constexpr int TILE_B = CT * D * 2;                  // bytes of one Y or V tile
    constexpr int NDMA = TILE_B / 4096;                 // DMA instructions per wave and tile (1 KiB each, 4 waves)
    constexpr int SWY = CPR >= 16 ? 15 : CPR - 1;
    constexpr int STAGE_B = 2 * TILE_B + 4096;          // Y tile, V tile, 4 x 1 KiB statistics slots (slot 0 is used)
    static_assert(TILE_B % 4096 == 0, "a tile must be a whole number of 4 KiB (4 waves x 1 KiB DMA)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, hf = lane >> 5;
    // XCD-aware placement (as in sdmi_attn.hip): the row blocks of one image stream the same Y / V - keep them on one XCD
    int rblk = blockIdx.x, b = blockIdx.y;
    {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        if (lin < (int)((gridDim.y & ~7u) * gridDim.x)) {
            const int xcd = lin & 7, k = lin >> 3;
            b = (k / (int)gridDim.x) * 8 + xcd;
            rblk = k % (int)gridDim.x;
        }
    }
    const unsigned short* Xp = p.x + (int64_t)b * p.N * p.ldx;
    const unsigned short* Yp = p.y + (int64_t)b * p.N * p.ldy;
    const unsigned short* Vp = p.v + (int64_t)b * p.N * p.ldv;
    float* statp = p.stats + (int64_t)b * p.N * 2;

    const int row = rblk * RB + wid * 32 + lr;
    const int row_c = row < p.N ? row : p.N - 1;

    // ---- staging geometry: DMA instruction i of wave w covers 16-byte chunks g = (4i + w) * 64 + lane of a tile ----
    const int wave_u = __builtin_amdgcn_readfirstlane(wid);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned short* const zsrc = reinterpret_cast<const unsigned short*>(vae_zero16);
    // row R and physical chunk pc of chunk g: CPR <= 64 divides 64, so R = (4i + w) * (64 / CPR) + lane / CPR and
    // pc = lane % CPR - recomputed per instruction (a few VALU) instead of held in registers (the D = 512 apply kernel
    // needs every register it can get)
    constexpr int RPI = 64 / CPR;                       // rows per DMA instruction
    const int l_row = lane / CPR, l_pc = lane & (CPR - 1);
    const int ntiles = (p.N + CT - 1) / CT;

    // one tile: Y rows (and V rows, and the column statistics) of contraction indices [t*32, t*32 + 32) into stage ST
    auto stage_tile = [&](int t, int st) {
        const unsigned d0 = lds_base + st * STAGE_B + wave_u * 1024;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int R = (4 * i + wave_u) * RPI + l_row;
            const int cidx = t * CT + R;
            const bool ok = cidx < p.N;
            const int ych = l_pc ^ (R & SWY), vch = l_pc ^ (D >= 128 ? (R & 3) << 2 : ((R >> 1) & 1) << 2);
            const void* ys = ok ? (const void*)(Yp + (int64_t)cidx * p.ldy + ych * 8) : (const void*)(zsrc + (lane & 7) * 8);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(ys), "s"(d0 + i * 4096) : "memory");
            if (MODE != 0) {
                const void* vs = ok ? (const void*)(Vp + (int64_t)cidx * p.ldv + vch * 8) : (const void*)(zsrc + (lane & 7) * 8);
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(vs), "s"(d0 + TILE_B + i * 4096) : "memory");
            }
        }
        if (MODE == 2) {
            // 32 x {m, 1/l} = 256 bytes: lanes 0..15 of wave 0 fetch them, every other lane (and wave) fetches zeros
            const int cidx = t * CT + 2 * lane;      // lane l < 16 carries columns 2l, 2l + 1
            const bool ok = wave_u == 0 && lane < 16 && cidx < p.N;      // N is even (host-checked) -> pairs are whole
            const void* ss = ok ? (const void*)(statp + (int64_t)cidx * 2) : (const void*)(zsrc + (lane & 7) * 8);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(ss), "s"(d0 + 2 * TILE_B) : "memory");
        }
    };

    stage_tile(0, 0);
    // X fragments (B operand of T^T = Y X^T): lane (row lr, half hf), k-step s: X[row][16s + 8hf .. +7].  Consumed by an
    // empty asm so that hipcc's wait for these loads sits here and not inside the loop (see sdmi_attn.hip).
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 xf[D / 16];
#pragma unroll
    for (int s = 0; s < D / 16; ++s)
        xf[s] = *reinterpret_cast<const u32x4*>(Xp + (int64_t)row_c * p.ldx + 16 * s + 8 * hf);
#pragma unroll
    for (int s = 0; s < D / 16; ++s) asm volatile("" : "+v"(xf[s]));
    float my_m = 0.f, my_il = 0.f;
    if (MODE == 1) {
        const float2 st = *reinterpret_cast<const float2*>(statp + (int64_t)row_c * 2);
        my_m = st.x; my_il = st.y;
        asm volatile("" : "+v"(my_m), "+v"(my_il));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 o[MODE == 0 ? 1 : D / 32];
#pragma unroll
    for (int d = 0; d < (MODE == 0 ? 1 : D / 32); ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;      // MODE 0: this lane's half of the columns

    // transposed-read lane geometry (ds_read_b64_tr_b16, per 16-lane group): see sdmi_attn.hip
    const int grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int vrow = 4 * hf + tq;
    const int vswz = D >= 128 ? (vrow & 3) << 2 : ((vrow >> 1) & 1) << 2;
    const unsigned v_lane0 = vrow * (2 * D) + ((tp & 1) << 3);
    const int v_chunk0 = 2 * (grp & 1) + (tp >> 1);

    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) stage_tile(t + 1, cur ^ 1);       // the other stage: everyone passed the barrier of tile t-1
        const uint4* Yb = reinterpret_cast<const uint4*>(smem + cur * STAGE_B);
        const unsigned char* Vb = smem + cur * STAGE_B + TILE_B;
        const float* Sb = reinterpret_cast<const float*>(smem + cur * STAGE_B + 2 * TILE_B);

        // ---- T^T[c, r] = Y_c . X_r ----
        f32x16 s;
        {
            f32x16 zero;
#pragma unroll
            for (int i = 0; i < 16; ++i) zero[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < D / 16; ++ks) {
                const int ch = 2 * ks + hf;
                const uint4 yv = Yb[lr * CPR + (ch ^ (lr & SWY))];
                s = mfma32<DT>(yv, __builtin_bit_cast(uint4, xf[ks]), ks == 0 ? zero : s);
                // bound how far hipcc hoists the LDS reads: left alone it requests all D/16 fragments up front, which
                // at D = 512 costs 128 VGPRs the apply kernel does not have
                if (D >= 256 && (ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // accumulator register r holds column c = t*32 + (r&3) + 8*(r>>2) + 4*hf, row lr
        if (MODE == 0) {
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cidx = t * CT + (r & 3) + 8 * (r >> 2) + 4 * hf;
                s[r] = cidx < p.N ? s[r] * p.c : -INFINITY;
                mx = fmaxf(mx, s[r]);
            }
            if (mx > m_run) {                       // per-lane state: no cross-lane traffic until the end
                l_run *= __builtin_amdgcn_exp2f(m_run - mx);       // exp2(-inf) = 0 on the first tile
                m_run = mx;
            }
            if (m_run > -INFINITY) {
#pragma unroll
                for (int r = 0; r < 16; ++r) l_run += __builtin_amdgcn_exp2f(s[r] - m_run);
            }
        } else {
            if (MODE == 1) {
#pragma unroll
                // The exponent is clamped at 0 (for a real column it is <= 0 already: m is the row maximum).  In a ragged
                // last tile the columns past N have zero logits, so their exponent is -m: without the clamp a row whose real
                // logits are all strongly negative would give them the weight exp2(-m) / l = inf, and inf * 0 (their V rows
                // are zero) = NaN for the whole output row.  Clamped, the weight is at most 1 / l and the product is 0.
                for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fminf(fmaf(s[r], p.c, -my_m), 0.f)) * my_il;
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // columns 8g + 4hf + {0..3}: two 16-byte reads of {m, 1/l} pairs (same address in a half wave)
                    const f32x4 a = *reinterpret_cast<const f32x4*>(Sb + (8 * g + 4 * hf) * 2);
                    const f32x4 c2 = *reinterpret_cast<const f32x4*>(Sb + (8 * g + 4 * hf) * 2 + 4);
                    s[4 * g + 0] = __builtin_amdgcn_exp2f(fmaf(s[4 * g + 0], p.c, -a[0])) * a[1];
                    s[4 * g + 1] = __builtin_amdgcn_exp2f(fmaf(s[4 * g + 1], p.c, -a[2])) * a[3];
                    s[4 * g + 2] = __builtin_amdgcn_exp2f(fmaf(s[4 * g + 2], p.c, -c2[0])) * c2[1];
                    s[4 * g + 3] = __builtin_amdgcn_exp2f(fmaf(s[4 * g + 3], p.c, -c2[2])) * c2[3];
                }
            }
            // ---- O^T[d, r] += V^T[d, c] W^T[c, r] ----
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 pf = cvt8v<DT>(s, 8 * ks);
#pragma unroll
                for (int db = 0; db < D / 32; ++db) {
                    const unsigned char* va_p = Vb + v_lane0 + (((db * 4 + v_chunk0) ^ vswz) << 4) + (16 * ks) * (2 * D);
                    const short4v va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) short4v*)(va_p));
                    const short4v vc = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) short4v*)(va_p + 8 * (2 * D)));
                    typedef __attribute__((ext_vector_type(8))) short short8v;
                    short8v vv;
                    vv[0] = va[0]; vv[1] = va[1]; vv[2] = va[2]; vv[3] = va[3];
                    vv[4] = vc[0]; vv[5] = vc[1]; vv[6] = vc[2]; vv[7] = vc[3];
                    o[db] = mfma32<DT>(__builtin_bit_cast(uint4, vv), pf, o[db]);
                    if (D >= 256 && (db & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile t+1 landed (this wave's share)
        __syncthreads();                                      // ... everyone's; and everyone finished reading tile t
    }

    if (MODE == 0) {
        // merge the two column halves of a row (lanes lr and lr + 32)
        const float m_o = __shfl_xor(m_run, 32), l_o = __shfl_xor(l_run, 32);
        const float m = fmaxf(m_run, m_o);
        const float l = l_run * __builtin_amdgcn_exp2f(m_run - m) + l_o * __builtin_amdgcn_exp2f(m_o - m);
        if (row < p.N && hf == 0) *reinterpret_cast<float2*>(statp + (int64_t)row * 2) = make_float2(m, 1.0f / l);
        return;
    }
    // ---- store: o[db][r] = O[row][d = db*32 + (r&3) + 8*(r>>2) + 4*hf] ----
    if (row < p.N) {
        unsigned short* orow = p.out + ((int64_t)b * p.N + row) * p.ldo;
#pragma unroll
        for (int db = 0; db < (MODE == 0 ? 1 : D / 32); ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 w;
                w.x = elem_pack2<DT>(o[db][4 * g], o[db][4 * g + 1]);
                w.y = elem_pack2<DT>(o[db][4 * g + 2], o[db][4 * g + 3]);
                *reinterpret_cast<uint2*>(orow + db * 32 + 8 * g + 4 * hf) = w;
            }
    }
}

template <int DT, int D>
int launch_vae(const VaeAttnArgs& qk, const VaeAttnArgs& ap, int mode, dim3 grid, hipStream_t s) {
    constexpr size_t lds = 2 * (2 * (size_t)CT * D * 2 + 4096);
    static SdmiPerDeviceOnce once;       // one per instantiation, run once per device
    const int attr_status = once.run([] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(vae_attn_kernel<DT, D, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(vae_attn_kernel<DT, D, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(vae_attn_kernel<DT, D, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    });
    if (attr_status != SDMI_OK) return attr_status;
    hipLaunchKernelGGL((vae_attn_kernel<DT, D, 0>), grid, dim3(256), lds, s, qk);
    if (hipGetLastError() != hipSuccess) return SDMI_ERR_LAUNCH;
    if (mode == 0) hipLaunchKernelGGL((vae_attn_kernel<DT, D, 1>), grid, dim3(256), lds, s, ap);
    else           hipLaunchKernelGGL((vae_attn_kernel<DT, D, 2>), grid, dim3(256), lds, s, ap);
    if (hipGetLastError() != hipSuccess) return SDMI_ERR_LAUNCH;
    return SDMI_OK;
}

}  // namespace

extern "C" int64_t sdmi_attention_vae_workspace_bytes(int32_t nb, int32_t N) {
    return nb > 0 && N > 0 ? (int64_t)nb * N * 2 * (int64_t)sizeof(float) : 0;
}

extern "C" int sdmi_attention_vae(const void* q, const void* k, const void* v, void* out, int32_t nb, int32_t N, int32_t D,
                                  int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, int32_t mode,
                                  int32_t dtype, void* workspace, int64_t workspace_bytes, void* stream) {
    if (dtype != SDMI_DTYPE_BF16 && dtype != SDMI_DTYPE_F16) return SDMI_ERR_BAD_ARG;
    if (q == nullptr || k == nullptr || v == nullptr || out == nullptr) return SDMI_ERR_BAD_ARG;
    if (nb <= 0 || N <= 0 || (mode != 0 && mode != 1)) return SDMI_ERR_BAD_ARG;
    if (D != 64 && D != 128 && D != 256 && D != 512) return SDMI_ERR_BAD_ARG;
    if (N % 2 != 0) return SDMI_ERR_ALIGNMENT;                       // statistics travel in pairs
    if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4) return SDMI_ERR_ALIGNMENT;
    if (ldq < D || ldk < D || ldv < D || ldo < D) return SDMI_ERR_BAD_ARG;
    if (nb > 65535) return SDMI_ERR_BAD_ARG;
    if (workspace == nullptr || workspace_bytes < sdmi_attention_vae_workspace_bytes(nb, N)) return SDMI_ERR_WORKSPACE;
    VaeAttnArgs a;          // pass A: rows = queries, columns = keys
    a.x = reinterpret_cast<const unsigned short*>(q); a.ldx = ldq;
    a.y = reinterpret_cast<const unsigned short*>(k); a.ldy = ldk;
    a.v = reinterpret_cast<const unsigned short*>(v); a.ldv = ldv;
    a.out = reinterpret_cast<unsigned short*>(out); a.ldo = ldo;
    a.stats = reinterpret_cast<float*>(workspace);
    a.N = N;
    a.c = scale * 1.4426950408889634f;
    VaeAttnArgs bb = a;     // pass B: standard keeps (Q rows, K columns); reference swaps them (output row = key index)
    if (mode == 1) { bb.x = a.y; bb.ldx = ldk; bb.y = a.x; bb.ldy = ldq; }
    const dim3 grid((unsigned)((N + RB - 1) / RB), (unsigned)nb);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool f16 = dtype == SDMI_DTYPE_F16;
#define SDMI_VAE_D(DD)                                                                                   \
    case DD: return f16 ? launch_vae<1, DD>(a, bb, mode, grid, s) : launch_vae<0, DD>(a, bb, mode, grid, s);
    switch (D) {
        SDMI_VAE_D(64) SDMI_VAE_D(128) SDMI_VAE_D(256) SDMI_VAE_D(512)
    }
#undef SDMI_VAE_D
    return SDMI_ERR_BAD_ARG;
}
