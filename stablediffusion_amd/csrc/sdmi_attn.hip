// sdmi_attention_d64: flash-style self/cross attention, head dim 64, for gfx950 (MI355X).
//
// Workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 query rows and walks
// the keys in tiles of 64.  Everything is built on v_mfma_f32_32x32x16_bf16 with the QUERY index on
// the lane, so the online softmax is lane-local:
//   S^T[key, q]  = K_tile[64 x 64] * Q^T          A = K rows (ds_read_b128), B = Q fragments in registers
//   row max/sum  : over the 32 accumulator registers of the lane + one lane<->lane+32 exchange
//   O^T[d, q]   += V_tile^T[d, key] * P^T[key, q] A = V^T via ds_read_b64_tr_b16 (hardware transpose of
//                                                  the row-major V tile), B = P^T = the S^T accumulators
//                                                  converted to bf16 in place (no LDS round trip)
// The k-order of an accumulator-as-operand is permuted: element j of lane half h of k-step s is key
// 16s + 8(j>>2) + 4h + (j&3); the V^T fragment is read in exactly that order.
// K and V tiles are double-buffered in LDS (32 KiB), staged through registers with the global loads of
// tile t+1 issued before the MFMAs of tile t.  Swizzles: K chunk ^= (row>>1)&7 (conflict-free b128
// reads for 32 rows at one chunk), V chunk ^= ((row>>1)&1)<<2 (conflict-free transposed reads).
#include "sdmi_common.h"

#ifndef SDMI_ATTN_OCC
#define SDMI_ATTN_OCC 2
#endif

namespace {

constexpr int QB = 128;      // query rows per workgroup
constexpr int KB = 64;       // keys per tile
constexpr int D = 64;
constexpr int TILE_CH = KB * (D / 8);   // 16-byte chunks per K or V tile (512)

struct AttnArgs {
    const unsigned short* q; const unsigned short* k; const unsigned short* v; unsigned short* out;
    int heads, Nq, Nk;
    int64_t ldq, ldk, ldv, ldo;
    float c;   // scale * log2(e)
};

__device__ __forceinline__ uint4 zero_unless(uint4 v, bool ok) {
    v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
    return v;
}

template <int DT>
__device__ __forceinline__ uint4 cvt8(const f32x16& s, int base) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = s[base + j];
    return elem_pack8<DT>(f);
}

template <int DT>
__global__ __launch_bounds__(256, SDMI_ATTN_OCC) void attn_fwd_d64_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(16))) uint4 lds[4 * TILE_CH];   // K0 K1 V0 V1
    uint4* const Ks = lds;
    uint4* const Vs = lds + 2 * TILE_CH;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.y;
    const int b = bh / p.heads, h = bh - b * p.heads;

    const unsigned short* Qp = p.q + (int64_t)b * p.Nq * p.ldq + h * D;
    const unsigned short* Kp = p.k + (int64_t)b * p.Nk * p.ldk + h * D;
    const unsigned short* Vp = p.v + (int64_t)b * p.Nk * p.ldv + h * D;
    unsigned short* Op = p.out + (int64_t)b * p.Nq * p.ldo + h * D;

    // Q fragments (B operand of S^T = K Q^T): lane (q = lr, half hf), k-step s: Q[q][16s + 8hf .. +7]
    const int qrow = blockIdx.x * QB + wid * 32 + lr;
    const int qrow_c = qrow < p.Nq ? qrow : p.Nq - 1;
    uint4 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        qf[s] = *reinterpret_cast<const uint4*>(Qp + (int64_t)qrow_c * p.ldq + 16 * s + 8 * hf);

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = -1e30f, l_run = 0.f;

    // staging: thread owns chunk (row = id>>3, c = id&7) for id = tid and tid+256
    const int sc = tid & 7, sr = tid >> 3;   // rows sr and sr+32
    // Named staging registers + macros (arrays captured by lambdas end up in scratch).  Rows past Nk are
    // loaded from row 0 and zeroed by value when written to LDS.
    uint4 rk0, rk1, rv0, rv1;
    bool ok0 = false, ok1 = false;
#define ATTN_LOAD_TILE(T)                                                                      \
    do {                                                                                       \
        const int key0_ = (T) * KB + sr, key1_ = key0_ + 32;                                   \
        ok0 = key0_ < p.Nk; ok1 = key1_ < p.Nk;                                                \
        const int64_t r0_ = ok0 ? key0_ : 0, r1_ = ok1 ? key1_ : 0;                            \
        rk0 = *reinterpret_cast<const uint4*>(Kp + r0_ * p.ldk + sc * 8);                      \
        rk1 = *reinterpret_cast<const uint4*>(Kp + r1_ * p.ldk + sc * 8);                      \
        rv0 = *reinterpret_cast<const uint4*>(Vp + r0_ * p.ldv + sc * 8);                      \
        rv1 = *reinterpret_cast<const uint4*>(Vp + r1_ * p.ldv + sc * 8);                      \
    } while (0)
    const int k_idx0 = sr * 8 + (sc ^ ((sr >> 1) & 7));            // row sr
    const int k_idx1 = (sr + 32) * 8 + (sc ^ ((sr >> 1) & 7));     // row sr+32: ((sr+32)>>1)&7 == (sr>>1)&7
    const int v_idx0 = sr * 8 + (sc ^ (((sr >> 1) & 1) << 2));
    const int v_idx1 = (sr + 32) * 8 + (sc ^ (((sr >> 1) & 1) << 2));
#define ATTN_STORE_TILE(BUF, FULL)                                                             \
    do {                                                                                       \
        if (FULL) {   /* uniform: every key of the tile exists, nothing to zero */             \
            Ks[(BUF) * TILE_CH + k_idx0] = rk0;                                                \
            Ks[(BUF) * TILE_CH + k_idx1] = rk1;                                                \
            Vs[(BUF) * TILE_CH + v_idx0] = rv0;                                                \
            Vs[(BUF) * TILE_CH + v_idx1] = rv1;                                                \
        } else {                                                                               \
            Ks[(BUF) * TILE_CH + k_idx0] = zero_unless(rk0, ok0);                              \
            Ks[(BUF) * TILE_CH + k_idx1] = zero_unless(rk1, ok1);                              \
            Vs[(BUF) * TILE_CH + v_idx0] = zero_unless(rv0, ok0);                              \
            Vs[(BUF) * TILE_CH + v_idx1] = zero_unless(rv1, ok1);                              \
        }                                                                                      \
    } while (0)

    const int ntiles = (p.Nk + KB - 1) / KB;
    ATTN_LOAD_TILE(0);
    ATTN_STORE_TILE(0, KB <= p.Nk);
    __syncthreads();

    // transposed-read lane geometry (ds_read_b64_tr_b16, per 16-lane group):
    //   lane 4q'+p' of the group supplies the address of row q', columns 4p'..4p'+3 of a 4x16 block;
    //   lane i receives column i, rows 0..3.
    const int grp = lane >> 4;              // 0..3 ; d0 = 16*(grp&1), key half = grp>>1 == hf
    const int tq = (lane & 15) >> 2, tp = lane & 3;

    // V^T fragment addresses: row = (kb*32 + 16*ks) + 4*hf + tq (+8 for the second half), 16-byte chunk
    // db*4 + 2*(grp&1) + (tp>>1) swizzled by ((row>>1)&1)<<2, which only depends on tq
    unsigned v_lane[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int row = 4 * hf + tq;
        const int chunk = (db * 4 + 2 * (grp & 1) + (tp >> 1)) ^ (((row >> 1) & 1) << 2);
        v_lane[db] = row * 128 + (chunk << 4) + ((tp & 1) << 3);
    }

    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) ATTN_LOAD_TILE(t + 1);
        const uint4* Kb = Ks + cur * TILE_CH;
        const unsigned char* Vb = reinterpret_cast<const unsigned char*>(Vs + cur * TILE_CH);

        // ---- S^T = K Q^T : two 32-key blocks ----
        f32x16 s0, s1;
        {
            f32x16 zero;
#pragma unroll
            for (int i = 0; i < 16; ++i) zero[i] = 0.f;       // folds into the MFMA's inline-constant C operand
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int ch = 2 * s + hf;
                const int r0 = lr, r1 = 32 + lr;
                const uint4 k0 = Kb[r0 * 8 + (ch ^ ((r0 >> 1) & 7))];
                const uint4 k1 = Kb[r1 * 8 + (ch ^ ((r1 >> 1) & 7))];
                s0 = mfma32<DT>(k0, qf[s], s == 0 ? zero : s0);
                s1 = mfma32<DT>(k1, qf[s], s == 0 ? zero : s1);
            }
        }
        // accumulator register r of block kb holds key t*64 + kb*32 + (r&3) + 8*(r>>2) + 4*hf, query lr
        if ((t + 1) * KB > p.Nk) {
            asm volatile("; ragged last tile" ::: "memory");    // keeps this a (uniform) branch, not 96 selects per tile
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * KB + (r & 3) + 8 * (r >> 2) + 4 * hf;
                if (key >= p.Nk) s0[r] = -INFINITY;
                if (key + 32 >= p.Nk) s1[r] = -INFINITY;
            }
        }
        // ---- online softmax (per query = per lane pair (lane, lane^32)) ----
        float mx = s0[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s1[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.c);
        const float mc = m_new * p.c;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // raw v_exp_f32: arguments are <= 0, a flushed denormal is an exact-enough zero weight
            s0[r] = __builtin_amdgcn_exp2f(s0[r] * p.c - mc);
            s1[r] = __builtin_amdgcn_exp2f(s1[r] * p.c - mc);
            psum += s0[r] + s1[r];
        }
        l_run = l_run * alpha + psum;     // per-lane partial (this lane's 32 of the 64 keys)
        m_run = m_new;
        // rescale the running output only when some query of the wave saw a new maximum (rare after the first tiles)
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        }

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 pf = cvt8<DT>(kb == 0 ? s0 : s1, 8 * ks);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    // lane part of the address is loop-invariant (v_lane[db]); the rest is an immediate
                    const unsigned char* va_p = Vb + v_lane[db] + (kb * 32 + 16 * ks) * 128;
                    const short4v va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) short4v*)(va_p));
                    const short4v vc = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) short4v*)(va_p + 8 * 128));
                    typedef __attribute__((ext_vector_type(8))) short short8v;
                    short8v vv;
                    vv[0] = va[0]; vv[1] = va[1]; vv[2] = va[2]; vv[3] = va[3];
                    vv[4] = vc[0]; vv[5] = vc[1]; vv[6] = vc[2]; vv[7] = vc[3];
                    const uint4 vf = __builtin_bit_cast(uint4, vv);
                    if (db == 0) o0 = mfma32<DT>(vf, pf, o0);
                    else         o1 = mfma32<DT>(vf, pf, o1);
                }
            }
        }
        if (t + 1 < ntiles) ATTN_STORE_TILE(cur ^ 1, (t + 2) * KB <= p.Nk);
        __syncthreads();
    }

    // ---- normalise and store: o_db[r] = O[q = lr][d = db*32 + (r&3) + 8*(r>>2) + 4*hf] ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (qrow < p.Nq) {
        unsigned short* orow = Op + (int64_t)qrow * p.ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w0, w1;
            w0.x = elem_pack2<DT>(o0[4 * g] * inv, o0[4 * g + 1] * inv);
            w0.y = elem_pack2<DT>(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            w1.x = elem_pack2<DT>(o1[4 * g] * inv, o1[4 * g + 1] * inv);
            w1.y = elem_pack2<DT>(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(orow + 8 * g + 4 * hf) = w0;
            *reinterpret_cast<uint2*>(orow + 32 + 8 * g + 4 * hf) = w1;
        }
    }
}

}  // namespace

extern "C" int sdmi_attention_d64(const void* q, const void* k, const void* v, void* out, int32_t nb, int32_t heads,
                                  int32_t Nq, int32_t Nk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                                  float scale, int32_t dtype, void* stream) {
    if (dtype != SDMI_DTYPE_BF16 && dtype != SDMI_DTYPE_F16) return SDMI_ERR_BAD_ARG;
    if (q == nullptr || k == nullptr || v == nullptr || out == nullptr) return SDMI_ERR_BAD_ARG;
    if (nb <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0) return SDMI_ERR_BAD_ARG;
    if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4) return SDMI_ERR_ALIGNMENT;
    if (ldq < (int64_t)heads * D || ldk < (int64_t)heads * D || ldv < (int64_t)heads * D || ldo < (int64_t)heads * D)
        return SDMI_ERR_BAD_ARG;
    if ((int64_t)nb * heads > 65535) return SDMI_ERR_BAD_ARG;
    AttnArgs a;
    a.q = reinterpret_cast<const unsigned short*>(q);
    a.k = reinterpret_cast<const unsigned short*>(k);
    a.v = reinterpret_cast<const unsigned short*>(v);
    a.out = reinterpret_cast<unsigned short*>(out);
    a.heads = heads; a.Nq = Nq; a.Nk = Nk;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.c = scale * 1.4426950408889634f;
    const dim3 grid((unsigned)((Nq + QB - 1) / QB), (unsigned)(nb * heads));
    if (dtype == SDMI_DTYPE_F16)
        hipLaunchKernelGGL(attn_fwd_d64_kernel<1>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    else
        hipLaunchKernelGGL(attn_fwd_d64_kernel<0>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}
