// sdmi_conv_halo: implicit-GEMM conv3x3 (stride 1, pad 1) whose activation tile is staged ONCE per 64-channel chunk.
//
// The tiled kernel (sdmi_gemm_kernel.h, mode 1) walks K = 9 * Cin tap by tap and pulls the 256 x 64 A tile of every tap
// from L2 into LDS - the same pixels nine times, shifted.  Its k-loop is power-bound: time follows the
// work done, and moving 53 KB per k-tile is a fifth of it.  Here a block owns 256 consecutive output pixels = 256 / W whole
// image rows (W = 16 | 32 | 64); for a 64-channel chunk it stages those rows plus one above and one below - the "halo",
// (256 / W + 2) * W <= 384 pixels x 128 B = 48 KiB - and the nine taps of the chunk read it through tap-shifted
// fragment addresses: K runs chunk-major (k-tile j = chunk * 9 + tap; the weights stay packed tap-major, tile j reads
// columns tap * Cin + chunk * 64).  Left / right neighbours of the first / last pixel of an image row are zero: the
// halo has no pad columns - those lanes read a zero pixel kept in front of the buffers instead (a select on the address);
// rows above / below the image come from the zero page.  LDS: 1 KiB of pad (the zero pixel), two halo buffers (the next
// chunk lands under the nine taps of this one) and the 3-stage W ring = 1 + 96 + 60 KiB.  Bytes into LDS per k-tile:
// 20 (W) + 48 / 9 = 25.3 KB against 52 KB.
// Block tile 256 x 160 x 64, 8 waves, fragment layout / swizzle / MFMA roles / half-k-tile pipeline of sdmi_gemm_kernel.h;
// the accumulation order over K differs from mode 1 (chunk-major), so results agree with it to fp32 rounding, not bit for bit.
#include "sdmi_gemm_kernel.h"

namespace {

constexpr int H_TN = 5, H_MI = 4, H_NST = 3, H_BM = 256, H_BN = 160;
constexpr int H_W_CHUNKS = H_BN * 8;                    // one W ring stage (20 KiB)
constexpr int H_PIX = 384;                              // halo pixels staged per chunk (rows past the halo: don't care)
constexpr int H_CHUNKS = H_PIX * 8;                     // one halo buffer (48 KiB)
constexpr int H_PAD_PIX = 8;                            // in front of the halo buffers: pixel 0 = zeros (what a cleared lane reads), pixel 7 = the
                                                        // slot "one before the halo" that the top-left tap of the first pixel addresses
constexpr int H_HALO0 = H_PAD_PIX * 8;                  // first chunk of halo buffer 0; buffer 1 follows it
constexpr int H_RING0 = H_HALO0 + 2 * H_CHUNKS;
constexpr int H_LDS_BYTES = (H_RING0 + H_NST * H_W_CHUNKS) * 16;      // 160768
constexpr int H_NH = H_PIX / 64;                        // halo DMA instructions per thread (6)
static_assert(H_LDS_BYTES <= 160 * 1024, "LDS of a CU");

// at most `keep` (0 | 1 | 2) W stages of PW instructions, plus a halo (H_NH instructions) when `halo`, still in flight
template <int PW>
__device__ __forceinline__ void wait_halo(int keep, bool halo) {
    if (keep >= 2) {
        if (halo) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PW + H_NH) : "memory");
        else      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PW) : "memory");
    } else if (keep == 1) {
        if (halo) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PW + H_NH) : "memory");
        else      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PW) : "memory");
    } else {
        if (halo) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(H_NH) : "memory");
        else      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <int DT>
__global__ __launch_bounds__(512, 2) void sdmi_conv_halo_kernel(const sdmi_gemm_args p) {
    constexpr int TN = H_TN, MI = H_MI, NST = H_NST, BM = H_BM, BN = H_BN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint4* const lds = reinterpret_cast<const uint4*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    const int tiles_m = (int)(p.M / BM);                // (host-checked: M % 256 == 0)
    const int tiles_n = (int)((p.N + BN - 1) / BN);
    const TileCoord tc = tile_of_block(blockIdx.x, tiles_m, tiles_n);
    const int64_t m0 = (int64_t)tc.tm * BM;
    const int64_t n0 = (int64_t)tc.tn * BN;

    // split-K over whole chunks (grid.z = slice)
    const int z = blockIdx.z;
    const int nchunks = p.Cin / BK;
    int c_begin = 0, c_end = nchunks;
    if (p.split_k > 1) {
        const int per = (nchunks + p.split_k - 1) / p.split_k;
        c_begin = z * per;
        c_end = min(nchunks, c_begin + per);
    }
    const int nk = max(c_end - c_begin, 0) * 9;

    const unsigned short* const A = reinterpret_cast<const unsigned short*>(p.A);
    const unsigned short* const W = reinterpret_cast<const unsigned short*>(p.W);
    const unsigned short* const zsrc = reinterpret_cast<const unsigned short*>(sdmi_zeros);

    // image geometry: the block's 256 pixels are rows y0 .. y0 + R - 1 of image b (host-checked: W = 16 | 32 | 64, H * W % 256 == 0)
    const int Wi = p.W_in, wsh = 31 - __builtin_clz(Wi), R = BM >> wsh;
    const int64_t hw = (int64_t)p.H * Wi;
    const int64_t b = m0 / hw;
    const int y0 = (int)((m0 - b * hw) >> wsh);

    // loader geometry (sdmi_gemm_kernel.h): thread owns physical chunk tid & 7 of rows lr + 64 i and fetches the logical
    // chunk (tid & 7) ^ (row & 7)
    const int lr = tid >> 3;
    const int lc = (tid & 7) ^ (lr & 7);
    const int wave_u = __builtin_amdgcn_readfirstlane(wid);
    const bool w_tail_wave = wave_u < (BN - 2 * 64) / 8;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned d_wave = lds_base + wave_u * 1024;

    // halo sources: pixel q = 64 i + lr of the halo = image row y0 - 1 + q / W, column q % W
    const unsigned short* h_ptr[H_NH];
#pragma unroll
    for (int i = 0; i < H_NH; ++i) {
        const int q = 64 * i + lr;
        const int hy = q >> wsh, hx = q & (Wi - 1);
        const int iy = y0 - 1 + hy;
        const bool ok = hy < R + 2 && (unsigned)iy < (unsigned)p.H;
        h_ptr[i] = (ok ? A + ((b * p.H + iy) * (int64_t)Wi + hx) * p.Cin : zsrc) + (c_begin * BK + lc * 8);
    }
    const unsigned short* const w_tile = W + n0 * p.K;
    unsigned w_voff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) w_voff[i] = (unsigned)((((int64_t)lr + 64 * i) * p.K + lc * 8) * 2);

    // issue stream: W tile j = (chunk, tap) reads weight columns tap * Cin + chunk * 64
    int is_tap = 0;
    int64_t is_c64 = (int64_t)c_begin * BK;
#define H_STAGE_W(BUF)                                                                                     \
    do {                                                                                                   \
        const unsigned d0_ = d_wave + (H_RING0 + (BUF) * H_W_CHUNKS) * 16;                                 \
        const unsigned short* src_ = w_tile + ((int64_t)is_tap * p.Cin + is_c64);                          \
        if (w_tail_wave) glds_s3<0>(d0_, src_, w_voff[0], w_voff[1], w_voff[2]);                           \
        else             glds_s2<0>(d0_, src_, w_voff[0], w_voff[1]);                                      \
        if (++is_tap == 9) { is_tap = 0; is_c64 += BK; }                                                   \
    } while (0)
#define H_STAGE_HALO(HB)                                                                                   \
    do {                                                                                                   \
        const unsigned d0_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(d_wave + (H_HALO0 + (HB) * H_CHUNKS) * 16)); \
        glds_v4<0>(d0_, h_ptr[0], h_ptr[1], h_ptr[2], h_ptr[3]);                                           \
        glds_v2<4 * 8192>(d0_, h_ptr[4], h_ptr[5]);                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < H_NH; ++i_) h_ptr[i_] += BK;                               \
    } while (0)

    f32x4 acc[TN][MI];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int c = 0; c < MI; ++c) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment geometry: lane (frow, fq) reads rows wm * 64 + mi * 16 + frow of the block = halo pixel q0[mi] at the centre tap
    // (in units of LDS pixels = 128 B, counted from the pad in front of buffer 0; both buffers are one run of pixels)
    const int frow = lane & 15, fq = lane >> 4;
    int q0[MI];
    bool at_left[MI], at_right[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int r = wm * (MI * 16) + mi * 16 + frow;
        const int lx = r & (Wi - 1);
        q0[mi] = H_PAD_PIX + r + Wi;                    // pad + (ly + 1) * W + lx
        at_left[mi] = lx == 0;
        at_right[mi] = lx == Wi - 1;
    }
    const int fq16 = fq << 4;
    const int w_base = H_RING0 + (wn * (TN * 16) + frow) * 8;
    if (tid < 8) reinterpret_cast<uint4*>(smem)[tid] = make_uint4(0u, 0u, 0u, 0u);      // the zero pixel (visible behind the first barrier)

    // read stream.  Per tile and row block ONE byte address: pixel * 128 + ((chunk ^ pixel & 7) * 16) with pixel = q0 + tap
    // shift (+ 384 for halo buffer 1) and chunk = fq for the first k-half, fq ^ 4 (address ^ 64) for the second; the lanes
    // whose left / right neighbour is outside the image read the zero pixel instead.  The tap is a compile-time constant of
    // the unrolled loop below, and pixel & 7 only depends on the tap's dx (W and 384 are multiples of 8): the swizzle term
    // comes from a table per dx, and the address is one three-operand add.
    unsigned base7[MI], swz[3][MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        base7[mi] = (unsigned)q0[mi] << 7;
#pragma unroll
        for (int d = 0; d < 3; ++d) swz[d][mi] = (unsigned)((((q0[mi] + d - 1) << 4) & 0x70) ^ fq16);
    }
    int r_hbq = 0;                                      // pixel offset of the halo buffer being read (0 | 384)
    unsigned a_byte[MI];
    uint4 af0[MI], wf0[TN], af1[MI], wf1[TN];
#define H_TILE_ADDR(TAP)                                                                                   \
    do {                                                                                                   \
        constexpr int dy_ = (TAP) / 3 - 1, dx_ = (TAP) % 3 - 1;                                            \
        const unsigned sdq7_ = (unsigned)((dy_ * Wi + dx_ + r_hbq) << 7);                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_) {                                                \
            unsigned a_ = base7[i_] + sdq7_ + swz[dx_ + 1][i_];                                            \
            if (dx_ < 0) a_ = at_left[i_] ? (unsigned)fq16 : a_;                                           \
            if (dx_ > 0) a_ = at_right[i_] ? (unsigned)fq16 : a_;                                          \
            a_byte[i_] = a_;                                                                               \
        }                                                                                                  \
    } while (0)
#define H_READ(AF, WF, STAGE, KS)                                                                          \
    do {                                                                                                   \
        const uint4* Wp_ = lds + (STAGE) * H_W_CHUNKS;                                                     \
        const int pc_ = ((KS) * 4 + fq) ^ (frow & 7);                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < TN; ++i_) WF[i_] = Wp_[w_base + i_ * 128 + pc_];           \
        _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_)                                                  \
            AF[i_] = *reinterpret_cast<const uint4*>(smem + (a_byte[i_] ^ ((KS) ? 64u : 0u)));             \
    } while (0)
#define H_MFMA(AF, WF)                                                                                     \
    do {                                                                                                   \
        _Pragma("unroll") for (int ni_ = 0; ni_ < TN; ++ni_)                                               \
            _Pragma("unroll") for (int mi_ = 0; mi_ < MI; ++mi_)                                           \
                acc[ni_][mi_] = mfma16<DT>(WF[ni_], AF[mi_], acc[ni_][mi_]);                               \
    } while (0)
#define H_INTERLEAVE()                                                                                     \
    do {                                                                                                   \
        _Pragma("unroll") for (int g_ = 0; g_ < MI + TN; ++g_) {                                           \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                             \
        }                                                                                                  \
    } while (0)
#define H_WAIT(KEEP, HALO)                                                                                 \
    do {                                                                                                   \
        if (w_tail_wave) wait_halo<3>(KEEP, HALO); else wait_halo<2>(KEEP, HALO);                          \
    } while (0)

    if (nk > 0) {
        // prologue: both halo buffers are free - chunks c_begin and c_begin + 1 - then the first three W tiles
        H_STAGE_HALO(0);
        int h_next = c_begin + 1;                       // next chunk to stage
        if (h_next < c_end) { H_STAGE_HALO(1); ++h_next; }
        int kt_issue = 0;
#pragma unroll
        for (int st = 0; st < NST; ++st)
            if (kt_issue < nk) { H_STAGE_W(st); ++kt_issue; }
        H_WAIT(kt_issue - 1, false);                    // W tile 0 and everything older (both halos) have landed
        __syncthreads();
        H_TILE_ADDR(0);
        H_READ(af0, wf0, 0, 0);
        H_READ(af1, wf1, 0, 1);
        H_MFMA(af0, wf0);
        int halo_age = 1 << 20;                         // iterations since the last halo issue in the loop
        int kt = 0;
        // iteration (chunk, T) finishes tile kt = 9 chunk + T and starts tile kt + 1 = tap (T + 1) % 9; ring stage of a tile =
        // its tap % 3 (9 is a multiple of the ring depth), so taps, ring stages and the masked taps are all static here
#define H_ITER(T)                                                                                          \
        if (kt + 1 < nk) {                                                                                 \
            constexpr int tn_ = ((T) + 1) % 9;                                                             \
            /* W tile kt + 1 landed; younger: W tile kt + 2 and a halo issued in one of the last two iterations */ \
            H_WAIT(kt_issue - kt - 2, halo_age <= 2);                                                      \
            __syncthreads();            /* every wave has finished reading tile kt (its W stage and, at tap 8, its halo) */ \
            if ((T) == 8) r_hbq ^= H_PIX;                                                                  \
            H_TILE_ADDR(tn_);                                                                              \
            H_READ(af0, wf0, tn_ % 3, 0);                                                                  \
            H_MFMA(af1, wf1);           /* second half of tile kt */                                       \
            H_INTERLEAVE();                                                                                \
            if (kt_issue < nk) { H_STAGE_W((T) % 3); ++kt_issue; }                                         \
            ++halo_age;                                                                                    \
            if ((T) == 8 && h_next < c_end) {   /* the buffer of the chunk just finished takes the chunk after next */ \
                H_STAGE_HALO((r_hbq ^ H_PIX) != 0 ? 1 : 0);                                                \
                ++h_next;                                                                                  \
                halo_age = 1;                                                                              \
            }                                                                                              \
            H_READ(af1, wf1, tn_ % 3, 1);                                                                  \
            H_MFMA(af0, wf0);           /* first half of tile kt + 1 */                                    \
            H_INTERLEAVE();                                                                                \
            ++kt;                                                                                          \
        }
        for (int c = c_begin; c < c_end; ++c) {
            H_ITER(0) H_ITER(1) H_ITER(2) H_ITER(3) H_ITER(4) H_ITER(5) H_ITER(6) H_ITER(7) H_ITER(8)
        }
#undef H_ITER
        H_MFMA(af1, wf1);
    }
#undef H_STAGE_W
#undef H_STAGE_HALO
#undef H_READ
#undef H_TILE_ADDR
#undef H_MFMA
#undef H_INTERLEAVE
#undef H_WAIT

    // ---- epilogue (sdmi_gemm_kernel.h: the split-K slab or the fast 16-bit form; the host sends nothing else here) ----
    if (p.split_k > 1) {
        float* ws = reinterpret_cast<float*>(p.workspace) + (int64_t)z * p.M * ((int64_t)tiles_n * BN);
        const int64_t ldw = (int64_t)tiles_n * BN;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int64_t m = m0 + wm * (MI * 16) + mi * 16 + frow;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int64_t wcol = n0 + wn * (TN * 16) + ni * 16 + 4 * fq;
                *reinterpret_cast<f32x4*>(ws + m * ldw + wcol) = acc[ni][mi];
            }
        }
        return;
    }
    const int64_t mrow = m0 + wm * (MI * 16) + frow, ncol_w = n0 + wn * (TN * 16);
    if (wn & 1) epilogue_rows16<DT, TN, MI, 1>(p, acc, mrow, ncol_w, fq, 0, 0);
    else        epilogue_rows16<DT, TN, MI, 0>(p, acc, mrow, ncol_w, fq, 0, 0);
}

typedef void (*halo_kern_t)(const sdmi_gemm_args);
const halo_kern_t halo_kernels[2] = {sdmi_conv_halo_kernel<0>, sdmi_conv_halo_kernel<1>};

}  // namespace

// The launches this kernel is built for (everything else stays on the tiled kernel): conv3x3, stride 1, no upsample, symmetric
// padding, image width 16 | 32 | 64 with whole 256-pixel row strips, Cin a multiple of 64, the 256 x 160 tile, and an epilogue
// the fast 16-bit form (or the split-K slab) covers.
static bool halo_applies(const sdmi_gemm_args& p) {
    if (p.mode != 1 || p.algo != 0 || p.tile_n != 160 || p.tile_m != 256) return false;
    if (p.dtype != SDMI_DTYPE_BF16 && p.dtype != SDMI_DTYPE_F16) return false;
    if (p.stride != 1 || p.upsample != 0 || p.asym_pad != 0 || p.batch > 1 || p.geglu || p.bias_per_row) return false;
    if (p.H != p.Ho || p.W_in != p.Wo || (p.W_in != 16 && p.W_in != 32 && p.W_in != 64)) return false;
    if (((int64_t)p.H * p.W_in) % H_BM != 0 || p.M % H_BM != 0 || p.M <= 0) return false;
    if (p.Cin % BK != 0 || p.Cin > ZERO_ELEMS || p.K != 9 * (int64_t)p.Cin) return false;
    if (p.ln_colsum != nullptr || p.ln_gamma != nullptr || p.row_stats_out != nullptr || p.A2 != nullptr) return false;
    if (p.out_kind != 0 || (p.N & 7) != 0 || ((p.ld_out | p.ld_res) & 7) != 0 || p.res_f32) return false;
    if ((((uintptr_t)p.out | (uintptr_t)p.res | (uintptr_t)p.A | (uintptr_t)p.W) & 15) != 0) return false;
    if (p.split_k > 1 && p.split_k > p.Cin / BK) return false;
    return true;
}

// 1 = launched, 0 = not a launch this kernel is built for, < 0 = sdmi_status
int sdmi_conv_halo_launch(const sdmi_gemm_args& p, hipStream_t s) {
    if (!halo_applies(p)) return 0;
    static SdmiPerDeviceOnce once;      // per device: hipFuncSetAttribute acts on the current device
    const int attr_status = once.run([] {
        bool ok = true;
        for (int i = 0; i < 2; ++i)
            ok = (hipFuncSetAttribute(reinterpret_cast<const void*>(halo_kernels[i]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      H_LDS_BYTES) == hipSuccess) && ok;
        return ok;
    });
    if (attr_status != SDMI_OK) return attr_status;
    const int64_t tiles = (p.M / H_BM) * ((p.N + H_BN - 1) / H_BN);
    const dim3 grid((unsigned)tiles, 1, (unsigned)(p.split_k > 1 ? p.split_k : 1));
    hipLaunchKernelGGL(halo_kernels[p.dtype == SDMI_DTYPE_F16 ? 1 : 0], grid, dim3(512), H_LDS_BYTES, s, p);
    if (hipGetLastError() != hipSuccess) return SDMI_ERR_LAUNCH;
    return 1;
}
