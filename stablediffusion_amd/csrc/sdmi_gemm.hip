// sdmi_gemm: bf16 MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
//   out[m, n] = epilogue(alpha * sum_k A[m, k] * W[n, k])          (see include/sdmi.h)
//
// One kernel serves nn.Linear, conv1x1 and conv3x3 (stride 1/2, optional fused nearest-x2
// gather, optional two-pointer channel concat): they differ only in how a 128-row x 64-k tile
// of A is gathered from HBM.  Both operands are K-contiguous, so both MFMA fragments are plain
// 16-byte LDS reads.
//
// Tiling: 128(M) x 128(N) x 64(K) per 256-thread workgroup; 4 waves as 2x2, each wave a 64x64
// sub-tile = 4x4 MFMA 16x16x32 tiles (64 fp32 accumulators).  The MFMA "A" operand is the W
// tile and the "B" operand is the activation tile, so an accumulator register quad holds 4
// CONSECUTIVE output columns of one output row: epilogue loads/stores are 8-16 byte vectors.
// LDS: double-buffered A and W tiles, 64 KiB, XOR-swizzled 16-byte chunks (conflict-free
// ds_read_b128), filled by global_load_lds_dwordx4 (LDS-DMA: no VGPR round trip, no ds_write -
// the ds_write_b128 path alone would cost ~0.8 of the MFMA time of a tile).  The gather of the
// implicit GEMM is expressed in the per-lane SOURCE address (zero taps read a zero constant);
// the DMA of tile k+1 is issued before the MFMAs of tile k and retired at the barrier.
// Workgroup ids are remapped so that each XCD (private L2) walks a contiguous, group-M-swizzled
// range of tiles.
#include "sdmi_common.h"

namespace {

// 16 bytes of zeros: the DMA source of padded rows / out-of-image taps
__device__ const uint4 sdmi_zero16 = {0u, 0u, 0u, 0u};

// Block tile BM x BN x BK with BN = 32*TN: TN = 4 (128 columns) or TN = 5 (160 columns; every UNet width is a
// multiple of 160, so those layers lose nothing to column padding).
constexpr int BM = 128, BK = 64;
constexpr int NTHREADS = 256;
constexpr int TILE_CHUNKS = BM * (BK / 8);          // 16-byte chunks per A tile (1024)
constexpr int lds_bytes(int tn) { return 2 * TILE_CHUNKS * 16 + 2 * (tn * 32) * (BK / 8) * 16; }   // A0 A1 W0 W1
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
// scaled and biased by the caller for the column-bias / GEGLU cases).
template <int DT>
__device__ __forceinline__ void store_quad(const sdmi_gemm_args& p, int64_t m, int64_t n, f32x4 v,
                                           int64_t zoff_out, int64_t zoff_res) {
    if (p.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = silu_f(v[j]);
    }
    if (p.bias != nullptr && p.bias_per_row) {
        const float b = p.bias[m];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += b;
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
    if (p.res != nullptr) {
        const unsigned short* r = reinterpret_cast<const unsigned short*>(p.res) + zoff_res + m * p.ld_res + n;
        if (full) {
            const uint2 rv = *reinterpret_cast<const uint2*>(r);
            float rf[4];
            elem_unpack4<DT>(rv, rf);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += rf[j];
        } else {
            for (int j = 0; j < 4 && n + j < p.N; ++j) v[j] += elem_to_f32<DT>(r[j]);
        }
    }
    if (p.out_kind == 0) {
        unsigned short* o = reinterpret_cast<unsigned short*>(p.out) + zoff_out + m * p.ld_out + n;
        if (full) {
            uint2 ov;
            ov.x = elem_pack2<DT>(v[0], v[1]);
            ov.y = elem_pack2<DT>(v[2], v[3]);
            *reinterpret_cast<uint2*>(o) = ov;
        } else {
            for (int j = 0; j < 4 && n + j < p.N; ++j) o[j] = elem_from_f32<DT>(v[j]);
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
}

template <int MODE, int DT, int TN>
__global__ __launch_bounds__(NTHREADS, 2) void sdmi_gemm_kernel(const sdmi_gemm_args p) {
    constexpr int BN = TN * 32;                 // 2 waves across N, TN 16-column blocks each
    constexpr int W_CHUNKS = BN * (BK / 8);     // 16-byte chunks per W tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* const As = reinterpret_cast<uint4*>(smem);
    uint4* const Ws = reinterpret_cast<uint4*>(smem) + 2 * TILE_CHUNKS;

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

    // ---- loader geometry: thread owns 16-byte chunk `lc` of rows lr + 32*i ----
    const int lr = tid >> 3;                    // tile rows lr + 32*i
    const int lc = (tid & 7) ^ (lr & 7);        // LOGICAL 16-byte chunk fetched into physical chunk tid&7

    int64_t a_off[4];      // MODE 0: m * lda (and m*lda2 recomputed);  MODE 1: pixel base b*H*W
    int iy0[4], ix0[4];    // MODE 1
    bool rvalid[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + lr + 32 * i;
        rvalid[i] = m < p.M;
        if (MODE == 0) {
            a_off[i] = m;
        } else {
            const int64_t hw = (int64_t)p.Ho * p.Wo;
            const int64_t b = m / hw;
            const int rem = (int)(m - b * hw);
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_off[i] = b * (int64_t)p.H * p.W_in;
            iy0[i] = rvalid[i] ? oy * p.stride - 1 : -(1 << 20);
            ix0[i] = ox * p.stride - 1;
        }
    }
    const int64_t k_valid = p.k_valid > 0 ? p.k_valid : p.K;
    const int Hl = p.H << p.upsample, Wl = p.W_in << p.upsample;
    // conv k-state for this thread's chunk: k = kt*64 + lc*8 -> (tap, ci)
    int c_tap = 0, c_ci = 0;
    if (MODE == 1) {
        const int k = kt_begin * BK + lc * 8;
        c_tap = k / p.Cin;
        c_ci = k - c_tap * p.Cin;
    }

    const unsigned short* w_ptr = W + (n0 + lr) * p.K + (int64_t)kt_begin * BK + lc * 8;
    const int64_t w_row32 = 32 * p.K;

    // ---- staging: global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction) ----
    // A wave-instruction writes LDS at (wave-uniform base) + lane*16, i.e. 8 consecutive tile rows x 8 physical
    // chunks.  The XOR swizzle therefore goes on the SOURCE address: lane l owns physical chunk l&7 of row
    // 32*i + 8*wave + (l>>3) and fetches logical chunk (l&7) ^ (row&7) (set up above as `lc`).  Rows / taps that
    // must read as zero fetch from a 16-byte zero constant instead (the DMA cannot mask).
    const unsigned short* const zsrc = reinterpret_cast<const unsigned short*>(&sdmi_zero16);
    const int wave_u = __builtin_amdgcn_readfirstlane(wid);
    // Issued through inline asm on purpose: with the builtin, hipcc orders every later ds_read of the tile
    // being computed behind the DMA of the NEXT tile (s_waitcnt vmcnt(0) ahead of the MFMAs), which serialises
    // load and compute.  The asm form is invisible to its wait-count pass; the DMA is retired by the explicit
    // s_waitcnt vmcnt(0) in front of the barrier that ends the k-step.  M0 (LDS destination) is written and
    // restored inside the statement (cdna_hip_programming.md §5.7).
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
#define SDMI_GLDS(SRC, LDSBYTE)                                                                            \
    do {                                                                                                   \
        unsigned keep_;                                                                                    \
        const void* src_ = (SRC);                                                                          \
        const unsigned dst_ = __builtin_amdgcn_readfirstlane(lds_base + (LDSBYTE));                        \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"                              \
                     "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"                                 \
                     : "=&s"(keep_) : "v"(src_), "s"(dst_) : "memory");                                    \
    } while (0)
    // LDS byte offset of instruction i of this wave inside one operand tile: (256*i + 64*wave) chunks
#define SDMI_TILE_OFF(I) (((256 * (I)) + 64 * wave_u) * 16)

#define SDMI_STAGE_A0(I, KK, BASE, LD, BUFBYTE)                                                            \
    SDMI_GLDS((rvalid[I] && kin_) ? (BASE) + a_off[I] * (LD) + (KK) : zsrc, (BUFBYTE) + SDMI_TILE_OFF(I))
#define SDMI_STAGE_A1(I, KY, KX, TV, BUFBYTE)                                                              \
    do {                                                                                                   \
        int iy_ = iy0[I] + (KY), ix_ = ix0[I] + (KX);                                                      \
        const bool ok_ = (TV) && (unsigned)iy_ < (unsigned)Hl && (unsigned)ix_ < (unsigned)Wl;             \
        iy_ >>= p.upsample; ix_ >>= p.upsample;                                                            \
        const int64_t off_ = (a_off[I] + (int64_t)iy_ * p.W_in + ix_) * p.Cin + c_ci;                      \
        SDMI_GLDS(ok_ ? A + off_ : zsrc, (BUFBYTE) + SDMI_TILE_OFF(I));                                    \
    } while (0)

    // Issue the DMA of k-tile KT into buffer BUF (A tile at BUF*16 KiB, W tile at 32 KiB + BUF*16 KiB).
#define SDMI_STAGE_TILE(KT, BUF)                                                                           \
    do {                                                                                                   \
        const int abyte_ = (BUF) * (TILE_CHUNKS * 16);                                                     \
        const int wbyte_ = 2 * (TILE_CHUNKS * 16) + (BUF) * (W_CHUNKS * 16);                               \
        if (MODE == 0) {                                                                                   \
            const int64_t k_ = (int64_t)(KT) * BK + lc * 8;                                                \
            const bool kin_ = k_ < k_valid;      /* columns past k_valid (K padding) read as zero */       \
            const bool second_ = (A2 != nullptr) && (k_ >= p.k_split);                                     \
            const unsigned short* base_ = second_ ? A2 : A;                                                \
            const int64_t ld_ = second_ ? p.lda2 : p.lda;                                                  \
            const int64_t kk_ = second_ ? k_ - p.k_split : k_;                                             \
            SDMI_STAGE_A0(0, kk_, base_, ld_, abyte_); SDMI_STAGE_A0(1, kk_, base_, ld_, abyte_);          \
            SDMI_STAGE_A0(2, kk_, base_, ld_, abyte_); SDMI_STAGE_A0(3, kk_, base_, ld_, abyte_);          \
        } else {                                                                                           \
            const int ky_ = c_tap / 3, kx_ = c_tap - 3 * ky_;                                              \
            const bool tapvalid_ = c_tap < 9;                                                              \
            SDMI_STAGE_A1(0, ky_, kx_, tapvalid_, abyte_); SDMI_STAGE_A1(1, ky_, kx_, tapvalid_, abyte_);  \
            SDMI_STAGE_A1(2, ky_, kx_, tapvalid_, abyte_); SDMI_STAGE_A1(3, ky_, kx_, tapvalid_, abyte_);  \
            c_ci += BK;                                                                                    \
            while (c_ci >= p.Cin) { c_ci -= p.Cin; ++c_tap; }                                              \
        }                                                                                                  \
        SDMI_GLDS(w_ptr, wbyte_ + SDMI_TILE_OFF(0));                                                       \
        SDMI_GLDS(w_ptr + w_row32, wbyte_ + SDMI_TILE_OFF(1));                                             \
        SDMI_GLDS(w_ptr + 2 * w_row32, wbyte_ + SDMI_TILE_OFF(2));                                         \
        SDMI_GLDS(w_ptr + 3 * w_row32, wbyte_ + SDMI_TILE_OFF(3));                                         \
        if (TN == 5) SDMI_GLDS(w_ptr + 4 * w_row32, wbyte_ + SDMI_TILE_OFF(4));                            \
        w_ptr += BK;                                                                                       \
    } while (0)

    f32x4 acc[TN][4];   // [ni][mi]
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fq = lane >> 4;
    const int a_base = (wm * 64 + frow) * 8;
    const int w_base = (wn * (TN * 16) + frow) * 8;

    if (kt_begin < kt_end) SDMI_STAGE_TILE(kt_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const bool more = kt + 1 < kt_end;
        if (more) SDMI_STAGE_TILE(kt + 1, cur ^ 1);     // DMA of the next tile flies under this tile's MFMAs
        const uint4* Ab = As + cur * TILE_CHUNKS;
        const uint4* Wb = Ws + cur * W_CHUNKS;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pc = (ks * 4 + fq) ^ (frow & 7);
            uint4 af[4], wf[TN];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = Ab[a_base + i * 128 + pc];
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[i] = Wb[w_base + i * 128 + pc];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = mfma16<DT>(wf[ni], af[mi], acc[ni][mi]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA of tile kt+1 has landed
        __syncthreads();                                     // ... and everyone's; buffer `cur` is free again
    }

    // ---- epilogue ----
    // accumulator quad acc[ni][mi][j]: output row m = m0 + wm*64 + mi*16 + (lane&15),
    // weight row (column of out) wcol = n0 + wn*16*TN + ni*16 + 4*(lane>>4) + j.
    if (p.split_k > 1) {
        // raw partial sums to the workspace slab of this slice: [split][M][tiles_n*BN] fp32
        float* ws = reinterpret_cast<float*>(p.workspace) + (int64_t)z * p.M * ((int64_t)tiles_n * BN);
        const int64_t ldw = (int64_t)tiles_n * BN;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int64_t m = m0 + wm * 64 + mi * 16 + frow;
            if (m >= p.M) continue;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int64_t wcol = n0 + wn * (TN * 16) + ni * 16 + 4 * fq;
                *reinterpret_cast<f32x4*>(ws + m * ldw + wcol) = acc[ni][mi];
            }
        }
        return;
    }

#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int64_t m = m0 + wm * 64 + mi * 16 + frow;
        if (m >= p.M) continue;
        if (p.geglu) {
#pragma unroll
            for (int ni = 0; ni + 1 < TN; ni += 2) {     // GEGLU runs with TN == 4 only (host-checked)
                const int64_t wcol = n0 + wn * (TN * 16) + ni * 16 + 4 * fq;
                const int64_t n = (wcol >> 5) * 16 + (wcol & 15);
                if (n >= p.N) continue;
                f32x4 v = acc[ni][mi] * p.alpha, g = acc[ni + 1][mi] * p.alpha;
                if (p.bias != nullptr) {
                    v += *reinterpret_cast<const f32x4*>(p.bias + wcol);
                    g += *reinterpret_cast<const f32x4*>(p.bias + wcol + 16);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= gelu_erf_f(g[j]);
                store_quad<DT>(p, m, n, v, zoff_out, zoff_res);
            }
        } else {
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
}

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
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= gelu_erf_f(g[j]);
        } else if (p.bias != nullptr && !p.bias_per_row) {
            for (int j = 0; j < 4 && n + j < p.N; ++j) v[j] += p.bias[n + j];
        }
        if (n >= p.N) continue;
        store_quad<DT>(p, m, n, v, 0, 0);
    }
}

}  // namespace

static int64_t slab_ld(int64_t N_w, int tile_n) { return (N_w + tile_n - 1) / tile_n * tile_n; }

extern "C" int64_t sdmi_gemm_workspace_bytes(int64_t M, int64_t N_w, int32_t split_k) {
    if (split_k <= 1) return 0;
    // enough for either column tile (128 or 160)
    const int64_t ldw = slab_ld(N_w, 128) > slab_ld(N_w, 160) ? slab_ld(N_w, 128) : slab_ld(N_w, 160);
    return (int64_t)split_k * M * ldw * (int64_t)sizeof(float);
}

extern "C" int sdmi_gemm(const sdmi_gemm_args* a, void* stream) {
    if (a == nullptr || a->A == nullptr || a->W == nullptr || a->out == nullptr) return SDMI_ERR_BAD_ARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return SDMI_ERR_BAD_ARG;
    if (a->K % BK != 0) return SDMI_ERR_ALIGNMENT;
    if (a->mode != 0 && a->mode != 1) return SDMI_ERR_BAD_ARG;
    if (a->out_kind < 0 || a->out_kind > 2) return SDMI_ERR_BAD_ARG;
    if (a->dtype != SDMI_DTYPE_BF16 && a->dtype != SDMI_DTYPE_F16) return SDMI_ERR_BAD_ARG;
    const int batch = a->batch > 0 ? a->batch : 1;
    const int split = a->split_k > 0 ? a->split_k : 1;
    if (batch > 1 && split > 1) return SDMI_ERR_BAD_ARG;
    if (a->geglu && (a->N % 16 != 0)) return SDMI_ERR_ALIGNMENT;
    if ((a->emb != nullptr || a->out_kind == 2) && a->rows_per_batch <= 0) return SDMI_ERR_BAD_ARG;
    if (a->emb != nullptr && (a->ld_emb % 4 != 0)) return SDMI_ERR_ALIGNMENT;
    if (a->res != nullptr && (a->ld_res % 4 != 0)) return SDMI_ERR_ALIGNMENT;
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
        if ((Hl + 2 - 3) / a->stride + 1 != a->Ho || (Wl + 2 - 3) / a->stride + 1 != a->Wo) return SDMI_ERR_BAD_ARG;
        if (a->A2 != nullptr) return SDMI_ERR_BAD_ARG;
    }
    sdmi_gemm_args p = *a;
    p.batch = batch;
    p.split_k = split;
    const int64_t n_w = p.geglu ? 2 * p.N : p.N;
    const int BN = p.tile_n == 0 ? 128 : p.tile_n;
    if (BN != 128 && BN != 160) return SDMI_ERR_BAD_ARG;
    if (BN == 160 && p.geglu) return SDMI_ERR_BAD_ARG;      // GEGLU pairs 16-column blocks inside a wave
    const int64_t tiles_m = (p.M + BM - 1) / BM, tiles_n = (n_w + BN - 1) / BN;
    if (tiles_m * tiles_n > 0x7fffffffLL) return SDMI_ERR_BAD_ARG;
    if (split > 1) {
        if (split > p.K / BK) return SDMI_ERR_BAD_ARG;
        if (p.workspace == nullptr || p.workspace_bytes < (int64_t)split * p.M * slab_ld(n_w, BN) * (int64_t)sizeof(float))
            return SDMI_ERR_WORKSPACE;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    typedef void (*kern_t)(const sdmi_gemm_args);
    static const kern_t kernels[2][2][2] = {     // [mode][dtype][tile_n == 160]
        {{sdmi_gemm_kernel<0, 0, 4>, sdmi_gemm_kernel<0, 0, 5>}, {sdmi_gemm_kernel<0, 1, 4>, sdmi_gemm_kernel<0, 1, 5>}},
        {{sdmi_gemm_kernel<1, 0, 4>, sdmi_gemm_kernel<1, 0, 5>}, {sdmi_gemm_kernel<1, 1, 4>, sdmi_gemm_kernel<1, 1, 5>}}};
    static bool attr_done = false;
    if (!attr_done) {
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                for (int c = 0; c < 2; ++c)
                    hipFuncSetAttribute(reinterpret_cast<const void*>(kernels[a][b][c]),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(c ? 5 : 4));
        attr_done = true;
    }
    const dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)(split > 1 ? split : batch));
    const bool f16 = p.dtype == SDMI_DTYPE_F16;
    const int wide = BN == 160;
    hipLaunchKernelGGL(kernels[p.mode][f16 ? 1 : 0][wide], grid, dim3(NTHREADS), lds_bytes(wide ? 5 : 4), s, p);
    SDMI_CHECK_LAUNCH();
    if (split > 1) {
        const int64_t quads = p.M * ((n_w + 3) / 4);
        const int64_t blocks = (quads + 255) / 256;
        const dim3 rgrid((unsigned)(blocks < 4096 ? blocks : 4096));
        if (f16) hipLaunchKernelGGL(sdmi_gemm_splitk_reduce<1>, rgrid, dim3(256), 0, s, p, tiles_n * BN);
        else     hipLaunchKernelGGL(sdmi_gemm_splitk_reduce<0>, rgrid, dim3(256), 0, s, p, tiles_n * BN);
        SDMI_CHECK_LAUNCH();
    }
    return SDMI_OK;
}
