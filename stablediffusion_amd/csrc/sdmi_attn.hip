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
// K and V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write) into a
// 3-stage ring (48 KiB): tile t+2 is issued before the MFMAs of tile t and retired with a counted vmcnt, so one
// younger tile stays in flight across the barrier.  Swizzles (applied to the SOURCE address, the DMA writes
// linearly): K chunk ^= (row>>1)&7 (conflict-free b128 reads for 32 rows at one chunk), V chunk ^= ((row>>1)&1)<<2
// (conflict-free transposed reads).  Keys past Nk are fetched from a zero constant.
// Softmax: one fma per logit puts it in log2 units relative to the running maximum (s * c - m), one v_exp_f32 makes
// the probability.  The running maximum is only moved when a tile exceeds it by more than 2^8 (probabilities stay
// <= 256: exact in fp32, in range for f16), which makes the rescale of O and l a rare, wave-uniform branch.
// Measured and dropped (MI355X, N = 4096): folding scale * log2(e) into the query projection and starting the QK^T
// accumulator at -m instead of 0, so that a probability is a bare exp2 - 32 fewer VALU per tile, yet 8 % SLOWER
// (846 against 920 TFLOP/s): an MFMA whose C operand is a VGPR block other than its destination costs more than
// the fmas it saves.  Also measured and dropped: a software-pipelined loop (QK^T MFMAs of tile t+1 issued under the
// softmax VALU of tile t inside one wave, 4-stage ring, 194 VGPRs -> two waves per SIMD): bit-identical results,
// 8-10 % slower than this kernel at three waves per SIMD (749 against 815 TFLOP/s on the same box) - co-resident
// waves in complementary phases already provide that overlap, and the third wave is worth more than the pipelining.
// What did pay (+24 %): keeping hipcc's wait for the Q fragment loads out of the loop (below).
#include "sdmi_common.h"

#ifndef SDMI_ATTN_OCC
#define SDMI_ATTN_OCC 2
#endif
#ifndef SDMI_ATTN_NST
#define SDMI_ATTN_NST 3
#endif
#ifndef SDMI_ATTN_HOIST
#define SDMI_ATTN_HOIST 0      // 1: pin every LDS fragment read of a tile ahead of its MFMA group (A/B: +-1 %, 22 more VGPRs)
#endif

namespace {

constexpr int QB = 128;      // query rows per workgroup
constexpr int KB = 64;       // keys per tile
constexpr int D = 64;
constexpr int TILE_CH = KB * (D / 8);   // 16-byte chunks per K or V tile (512)
constexpr int NST = SDMI_ATTN_NST;      // ring stages (3: tile t+2 in flight across the barrier; 2: tile t+1 only)

__device__ const uint4 attn_zero16 = {0u, 0u, 0u, 0u};

struct AttnArgs {
    const unsigned short* q; const unsigned short* k; const unsigned short* v; unsigned short* out;
    int heads, Nq, Nk;
    int rowstore;      // out and ldo allow 16-byte row-coalesced stores
    int64_t ldq, ldk, ldv, ldo;
    float c;   // scale * log2(e)
};

template <int DT>
__device__ __forceinline__ uint4 cvt8(const f32x16& s, int base) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = s[base + j];
    return elem_pack8<DT>(f);
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr float RESCALE_THR = 8.0f;     // log2 units: move the running maximum only when a tile tops it by 2^8

template <int DT>
__global__ __launch_bounds__(256, SDMI_ATTN_OCC) void attn_fwd_d64_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(16))) uint4 lds[2 * NST * TILE_CH];   // stage s: K tile at 2s, V tile at 2s+1

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, hf = lane >> 5;
    // XCD-aware placement: workgroups are dealt to the 8 XCDs round-robin by linear id, so the query blocks of one
    // (batch, head) - which all read the same K/V - would land on all eight L2s.  Remap the linear id so that they
    // share one XCD (K/V of the ~3 heads resident per XCD stay in its 4 MiB L2).
    int qblk = blockIdx.x, bh = blockIdx.y;
    {
        // heads are taken eight at a time (a bijection on every full group of 8; a tail of < 8 heads keeps its ids)
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        if (lin < (int)((gridDim.y & ~7u) * gridDim.x)) {
            const int xcd = lin & 7, k = lin >> 3;
            bh = (k / (int)gridDim.x) * 8 + xcd;
            qblk = k % (int)gridDim.x;
        }
    }
    const int b = bh / p.heads, h = bh - b * p.heads;

    const unsigned short* Qp = p.q + (int64_t)b * p.Nq * p.ldq + h * D;
    const unsigned short* Kp = p.k + (int64_t)b * p.Nk * p.ldk + h * D;
    const unsigned short* Vp = p.v + (int64_t)b * p.Nk * p.ldv + h * D;
    unsigned short* Op = p.out + (int64_t)b * p.Nq * p.ldo + h * D;

    const int qrow = qblk * QB + wid * 32 + lr;
    const int qrow_c = qrow < p.Nq ? qrow : p.Nq - 1;

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    float nm = 0.f;                    // nm = -(running maximum), log2 units
    f32x2 l_run = {0.f, 0.f}, l_run1 = {0.f, 0.f};   // running row sum: 2 x 2 partial sums per lane (v_pk_add_f32: 16 adds per
                                                     // tile instead of 32, in two independent chains)

    // ---- staging: wave w issues, per tile, K rows {8w..8w+7, 32+8w..} and the same V rows: 4 DMA instructions ----
    // lane l of an instruction owns physical chunk l&7 of row R = 32*i + 8*w + (l>>3) and fetches the logical chunk
    // the read side expects there (K: ^ (R>>1)&7, V: ^ ((R>>1)&1)<<2).
    const int wave_u = __builtin_amdgcn_readfirstlane(wid);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds;
    const unsigned d_wave = lds_base + wave_u * 1024;
    const int srow = 8 * wave_u + (lane >> 3);           // rows srow and srow + 32
    const int spc = lane & 7;
    const unsigned short* const zsrc = reinterpret_cast<const unsigned short*>(&attn_zero16);
    // ((srow+32)>>1)&7 == (srow>>1)&7, so both rows of a lane use the same logical chunk
    const int kch = spc ^ ((srow >> 1) & 7), vch = spc ^ (((srow >> 1) & 1) << 2);
    const unsigned short* kp0 = Kp + (int64_t)srow * p.ldk + kch * 8;
    const unsigned short* kp1 = kp0 + 32 * p.ldk;
    const unsigned short* vp0 = Vp + (int64_t)srow * p.ldv + vch * 8;
    const unsigned short* vp1 = vp0 + 32 * p.ldv;
    const int64_t kstep = (int64_t)KB * p.ldk, vstep = (int64_t)KB * p.ldv;
    // Issue tile T into ring stage ST; pointers advance by one tile.  Rows past Nk (last tile only) read zeros.
#define ATTN_STAGE_TILE(T, ST)                                                                 \
    do {                                                                                       \
        const unsigned d0_ = d_wave + (ST) * (2 * TILE_CH * 16);                               \
        if (((T) + 1) * KB <= p.Nk) {                                                          \
            glds_v2<0, 4096>(d0_, kp0, kp1);                                                   \
            glds_v2<TILE_CH * 16, 4096>(d0_, vp0, vp1);                                        \
        } else {                                                                               \
            const bool ok0_ = (T) * KB + srow < p.Nk, ok1_ = (T) * KB + srow + 32 < p.Nk;      \
            glds_v2<0, 4096>(d0_, ok0_ ? kp0 : zsrc, ok1_ ? kp1 : zsrc);                       \
            glds_v2<TILE_CH * 16, 4096>(d0_, ok0_ ? vp0 : zsrc, ok1_ ? vp1 : zsrc);            \
        }                                                                                      \
        kp0 += kstep; kp1 += kstep; vp0 += vstep; vp1 += vstep;                                \
    } while (0)

    const int ntiles = (p.Nk + KB - 1) / KB;
    ATTN_STAGE_TILE(0, 0);
    if (NST == 3 && ntiles > 1) ATTN_STAGE_TILE(1, 1);
    // Q fragments (B operand of S^T = K Q^T): lane (q = lr, half hf), k-step s: Q[q][16s + 8hf .. +7].
    // They are consumed (empty asm) right here so that hipcc retires its own wait for them BEFORE the loop: left to
    // itself it puts `s_waitcnt vmcnt(3..0)` ahead of the first uses inside the loop, where the counter also holds the
    // LDS-DMA of the tile just issued - every iteration would then wait for its own prefetch.
    u32x4 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        qf[s] = *reinterpret_cast<const u32x4*>(Qp + (int64_t)qrow_c * p.ldq + 16 * s + 8 * hf);
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // tiles 0 and 1 landed (this wave's share)
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

    int cur = 0, nxt = NST - 1;    // ring stage of tile t / of the tile issued in iteration t (t + NST - 1)
    for (int t = 0; t < ntiles; ++t) {
        const bool issued = t + NST - 1 < ntiles;
        const uint4* Kb = lds + cur * (2 * TILE_CH);
        const unsigned char* Vb = reinterpret_cast<const unsigned char*>(lds + cur * (2 * TILE_CH) + TILE_CH);

        // ---- S^T = K Q^T : two 32-key blocks ----
        // All eight K fragments are requested before the first MFMA (one LDS round trip exposed per tile instead
        // of one per MFMA pair), and all V^T fragments of the tile right behind the QK^T MFMAs, so that their latency
        // runs under the softmax; sched_barriers keep hipcc from sinking the reads back next to their uses.
        f32x16 s0, s1;
        uint4 kf[8];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ch = 2 * s + hf;
            const int r0 = lr, r1 = 32 + lr;
            kf[2 * s] = Kb[r0 * 8 + (ch ^ ((r0 >> 1) & 7))];
            kf[2 * s + 1] = Kb[r1 * 8 + (ch ^ ((r1 >> 1) & 7))];
        }
        if (SDMI_ATTN_HOIST) __builtin_amdgcn_sched_barrier(0);
        {
            f32x16 zero;
#pragma unroll
            for (int i = 0; i < 16; ++i) zero[i] = 0.f;       // folds into the MFMA's inline-constant C operand
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const uint4 qv = __builtin_bit_cast(uint4, qf[s]);
                s0 = mfma32<DT>(kf[2 * s], qv, s == 0 ? zero : s0);
                s1 = mfma32<DT>(kf[2 * s + 1], qv, s == 0 ? zero : s1);
            }
        }
        if (SDMI_ATTN_HOIST) __builtin_amdgcn_sched_barrier(0);
        uint4 vf[2][2][2];       // [kb][ks][db]
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
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
                    vf[kb][ks][db] = __builtin_bit_cast(uint4, vv);
                }
        if (SDMI_ATTN_HOIST) __builtin_amdgcn_sched_barrier(0);
        // tile t+2 into the stage tile t-1 vacated (everyone passed its barrier); issued here, behind the MFMAs, so
        // that the scalar set-up of the DMA fills the wait states before the accumulators can be read
        if (issued) ATTN_STAGE_TILE(t + NST - 1, nxt);
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = fmaf(s0[r], p.c, nm); s1[r] = fmaf(s1[r], p.c, nm); }
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
        // ---- online softmax (per query = per lane pair (lane, lane^32)); s = logit - running max, log2 units ----
        float mx = s0[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s1[r]);
        {   // the other half of the query's keys sits on lane ^ 32: v_permlane32_swap (lanes 32..63 of the first operand
            // <-> lanes 0..31 of the second), no LDS round trip on the critical path.  Inline asm on purpose: with the
            // builtin, hipcc folds max(result0, result1) of a value swapped with its own copy down to result0.
            float ma = mx, mb = mx;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
            mx = fmaxf(ma, mb);
        }
        // Move the running maximum (and rescale O, l) only on the first tile or when some query of the wave tops it
        // by more than 2^RESCALE_THR: a wave-uniform, rare branch.  Otherwise p = 2^s <= 2^RESCALE_THR as it stands.
        if (t == 0 || __builtin_amdgcn_ballot_w64(mx > RESCALE_THR) != 0) {
            asm volatile("; move the running maximum" ::: "memory");
            const float delta = t == 0 ? mx : fmaxf(mx, 0.f);
            const float alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);
            nm -= delta;
            l_run *= alpha; l_run1 *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s0[i] -= delta; s1[i] -= delta;
                o0[i] *= alpha; o1[i] *= alpha;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // raw v_exp_f32: a flushed denormal is an exact-enough zero weight
            s0[r] = __builtin_amdgcn_exp2f(s0[r]);
            s1[r] = __builtin_amdgcn_exp2f(s1[r]);
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) {     // per-lane partials (this lane's 32 of the 64 keys)
            l_run += (f32x2){s0[r], s0[r + 1]};
            l_run1 += (f32x2){s1[r], s1[r + 1]};
        }

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 pf = cvt8<DT>(kb == 0 ? s0 : s1, 8 * ks);
                o0 = mfma32<DT>(vf[kb][ks][0], pf, o0);
                o1 = mfma32<DT>(vf[kb][ks][1], pf, o1);
            }
        }
        if (NST == 3 && issued) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // tile t+1 landed, tile t+2 may still fly
        else                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur = cur == NST - 1 ? 0 : cur + 1;
        nxt = nxt == NST - 1 ? 0 : nxt + 1;
    }

    // ---- normalise and store: o_db[r] = O[q = lr][d = db*32 + (r&3) + 8*(r>>2) + 4*hf] ----
    const float l_lane = (l_run[0] + l_run[1]) + (l_run1[0] + l_run1[1]);
    const float l_tot = l_lane + __shfl_xor(l_lane, 32);
    const float inv = 1.0f / l_tot;
#ifndef SDMI_ATTN_ROWSTORE
#define SDMI_ATTN_ROWSTORE 1      // A/B: 0 = every lane stores its own 8-byte pieces
#endif
    if (SDMI_ATTN_ROWSTORE && p.rowstore) {
        // Row-coalesced store.  In the accumulator layout a lane holds sixteen 8-byte pieces of one output row, so a store
        // instruction touches 32 rows (32 cache lines for 512 bytes) and a wave needs 16 of them.  The wave's 32 x 64 tile
        // goes through LDS instead (its own 4.5 KiB of the K/V ring, which every wave has left behind the loop's last
        // barrier; LDS operations of one wave execute in order, so no barrier is needed): 8 lanes then store one whole
        // 128-byte row of the head - one cache line - and the wave issues 4 instructions of 8 lines each.  With 77 keys
        // (cross-attention: two key tiles per launch) the stores were a third of the kernel's time.
        unsigned char* ob = reinterpret_cast<unsigned char*>(lds) + wid * (32 * 144);      // row pitch 144 B
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w0, w1;
            w0.x = elem_pack2<DT>(o0[4 * g] * inv, o0[4 * g + 1] * inv);
            w0.y = elem_pack2<DT>(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            w1.x = elem_pack2<DT>(o1[4 * g] * inv, o1[4 * g + 1] * inv);
            w1.y = elem_pack2<DT>(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(ob + lr * 144 + (8 * g + 4 * hf) * 2) = w0;
            *reinterpret_cast<uint2*>(ob + lr * 144 + (32 + 8 * g + 4 * hf) * 2) = w1;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + (lane >> 3), ch = lane & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(ob + row * 144 + ch * 16);
            const int qr = qblk * QB + wid * 32 + row;
            if (qr < p.Nq) *reinterpret_cast<uint4*>(Op + (int64_t)qr * p.ldo + ch * 8) = v;
        }
        return;
    }
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
    a.rowstore = (ldo % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) ? 1 : 0;
    const dim3 grid((unsigned)((Nq + QB - 1) / QB), (unsigned)(nb * heads));
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SDMI_DTYPE_F16) hipLaunchKernelGGL(attn_fwd_d64_kernel<1>, grid, dim3(256), 0, hs, a);
    else                         hipLaunchKernelGGL(attn_fwd_d64_kernel<0>, grid, dim3(256), 0, hs, a);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}
