// sdmi_pack: weight packing - fp32 weights in the reference's (PyTorch) layout -> the operand layout sdmi_gemm reads.
//
// Packed layout (include/sdmi.h): 16-bit [Npad][Kpad], K contiguous, Npad = rows rounded up to the column tile of the
// kernel (128 | 160), Kpad = K rounded up to 64, padding zero.
//   linear   [N, K]            : row n, column k as they stand
//   GEGLU    [2*inner, K]      : packed row r holds value (half 0) or gate (half 1) column blk*16 + i, with
//                                blk = r / 32, half = (r % 32) / 16, i = r % 16 - value and gate of one output column sit
//                                in the same lane of adjacent MFMA column blocks (attention.py:89: value half first)
//   conv3x3  [Cout, Cin, 3, 3] : column (ky*3 + kx) * Cin_p + ci, Cin_p = Cin rounded up to 8 (zero channels)
// fp32 mode ("planes"): every fp32 weight is split exactly into three bf16 pieces w = w0 + w1 + w2 and stored as the six
// planes (w0 w1 w0 w2 w1 w0) of width Kp = K (or Cin) rounded up to 32, inside every conv tap - the counterpart of the
// activation planes of sdmi_f32_split3.
//
// One element function serves the device kernel and the host loops (sdmi_pack_*_host: the same arithmetic compiled for
// the CPU, used where weights are prepared off the GPU and by the CPU tests), so the two cannot drift apart.
#include "sdmi_common.h"

namespace {

struct PackGeom {
    const float* w;
    int64_t N, K, ld_w;            // source rows of this call, source columns (linear) / Cin (conv), source row stride
    int64_t row_offset, rows_total, Npad, Kpad;
    int64_t kp;                    // plane width (planes) or padded Cin (conv 16-bit); unused for linear 16-bit
    int64_t inner;                 // GEGLU: gated width (rows_total == 2 * inner), else 0
    int32_t conv, planes, f16;
};

__host__ __device__ inline unsigned short to16(float f, int f16) {
    if (f16) return __builtin_bit_cast(unsigned short, (_Float16)f);
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}
__host__ __device__ inline float bf16_back(unsigned short b) {
    return __builtin_bit_cast(float, ((unsigned)b) << 16);
}

// value of packed element (row r of this call's row range, column c)
__host__ __device__ inline unsigned short pack_elem(const PackGeom& g, int64_t r, int64_t c) {
    if (r >= g.N) return 0;
    int64_t sr = r;
    if (g.inner > 0) sr = ((r % 32) / 16) * g.inner + (r / 32) * 16 + (r % 16);
    int plane = 0;
    int64_t k = c;
    float v = 0.f;
    if (!g.conv) {
        if (g.planes) {
            plane = (int)(c / g.kp);
            k = c - plane * g.kp;
            if (plane >= 6) return 0;
        }
        if (k >= g.K) return 0;
        v = g.w[sr * g.ld_w + k];
    } else {
        const int64_t tapw = g.planes ? 6 * g.kp : g.kp;
        const int64_t tap = c / tapw;
        if (tap >= 9) return 0;
        int64_t ci = c - tap * tapw;
        if (g.planes) {
            plane = (int)(ci / g.kp);
            ci -= plane * g.kp;
        }
        if (ci >= g.K) return 0;
        v = g.w[(sr * g.K + ci) * 9 + tap];         // [Cout, Cin, 3, 3]: tap = ky * 3 + kx
    }
    if (!g.planes) return to16(v, g.f16);
    const unsigned short w0 = to16(v, 0);
    const float r1 = v - bf16_back(w0);
    const unsigned short w1 = to16(r1, 0);
    const unsigned short w2 = to16(r1 - bf16_back(w1), 0);
    const int piece = plane == 0 || plane == 2 || plane == 5 ? 0 : (plane == 3 ? 2 : 1);      // (w0 w1 w0 w2 w1 w0)
    return piece == 0 ? w0 : (piece == 1 ? w1 : w2);
}

// rows this call writes: its own N rows, plus the zero rows up to Npad when it holds the last logical row
__host__ __device__ inline int64_t rows_written(const PackGeom& g) {
    return g.row_offset + g.N == g.rows_total ? g.Npad - g.row_offset : g.N;
}

__global__ __launch_bounds__(256) void pack_kernel(const PackGeom g, unsigned short* out) {
    const int64_t chunks = g.Kpad / 8;
    const int64_t total = rows_written(g) * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / chunks, c0 = (i - r * chunks) * 8;
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = pack_elem(g, r, c0 + j);
        uint4 v;
        v.x = e[0] | ((unsigned)e[1] << 16); v.y = e[2] | ((unsigned)e[3] << 16);
        v.z = e[4] | ((unsigned)e[5] << 16); v.w = e[6] | ((unsigned)e[7] << 16);
        *reinterpret_cast<uint4*>(out + (g.row_offset + r) * g.Kpad + c0) = v;
    }
}

__global__ __launch_bounds__(256) void pack_geglu_bias_kernel(const float* bias, float* out, int64_t inner) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < 2 * inner) out[r] = bias[((r % 32) / 16) * inner + (r / 32) * 16 + (r % 16)];
}

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

int make_geom(PackGeom& g, const float* w, int64_t N, int64_t K, int64_t ld_w, int64_t row_offset, int64_t rows_total,
              int32_t dtype, int32_t tile_n, int conv, int64_t inner) {
    if (w == nullptr || N <= 0 || K <= 0 || row_offset < 0 || rows_total < row_offset + N) return SDMI_ERR_BAD_ARG;
    if (dtype < 0 || dtype > SDMI_DTYPE_F32_PLANES) return SDMI_ERR_BAD_ARG;
    if (tile_n != 128 && tile_n != 160) return SDMI_ERR_BAD_ARG;
    if (!conv && ld_w < K) return SDMI_ERR_BAD_ARG;
    g.w = w; g.N = N; g.K = K; g.ld_w = ld_w; g.row_offset = row_offset; g.rows_total = rows_total;
    g.conv = conv; g.planes = dtype == SDMI_DTYPE_F32_PLANES; g.f16 = dtype == SDMI_DTYPE_F16; g.inner = inner;
    g.Npad = round_up(rows_total, tile_n);
    g.kp = conv ? sdmi_pack_conv3x3_cin(K, dtype) / (g.planes ? 6 : 1) : (g.planes ? round_up(K, 32) : K);
    g.Kpad = conv ? sdmi_pack_conv3x3_k(K, dtype) : sdmi_pack_linear_k(K, dtype);
    return SDMI_OK;
}

int run_device(const PackGeom& g, void* out, void* stream) {
    if (out == nullptr) return SDMI_ERR_BAD_ARG;
    const int64_t total = rows_written(g) * (g.Kpad / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g,
                       reinterpret_cast<unsigned short*>(out));
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

int run_host(const PackGeom& g, void* out) {
    if (out == nullptr) return SDMI_ERR_BAD_ARG;
    unsigned short* o = reinterpret_cast<unsigned short*>(out);
    const int64_t rows = rows_written(g);
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < g.Kpad; ++c) o[(g.row_offset + r) * g.Kpad + c] = pack_elem(g, r, c);
    return SDMI_OK;
}

}  // namespace

extern "C" int32_t sdmi_pack_tile_n(int64_t N, int32_t geglu) { return (N % 160 == 0 && !geglu) ? 160 : 128; }

extern "C" int64_t sdmi_pack_linear_k(int64_t K, int32_t dtype) {
    return dtype == SDMI_DTYPE_F32_PLANES ? round_up(6 * round_up(K, 32), 64) : round_up(K, 64);
}
extern "C" int64_t sdmi_pack_conv3x3_cin(int64_t Cin, int32_t dtype) {
    return dtype == SDMI_DTYPE_F32_PLANES ? 6 * round_up(Cin, 32) : round_up(Cin, 8);
}
extern "C" int64_t sdmi_pack_conv3x3_k(int64_t Cin, int32_t dtype) { return round_up(9 * sdmi_pack_conv3x3_cin(Cin, dtype), 64); }
extern "C" int64_t sdmi_pack_bytes(int64_t rows_total, int64_t k_packed, int32_t tile_n) {
    if (rows_total <= 0 || k_packed <= 0 || (tile_n != 128 && tile_n != 160)) return 0;
    return round_up(rows_total, tile_n) * k_packed * 2;
}

extern "C" int sdmi_pack_linear(const float* w, int64_t N, int64_t K, int64_t ld_w, int64_t row_offset, int64_t rows_total,
                                void* out, int32_t dtype, int32_t tile_n, void* stream) {
    PackGeom g;
    const int st = make_geom(g, w, N, K, ld_w, row_offset, rows_total, dtype, tile_n, 0, 0);
    return st != SDMI_OK ? st : run_device(g, out, stream);
}
extern "C" int sdmi_pack_linear_host(const float* w, int64_t N, int64_t K, int64_t ld_w, int64_t row_offset,
                                     int64_t rows_total, void* out, int32_t dtype, int32_t tile_n) {
    PackGeom g;
    const int st = make_geom(g, w, N, K, ld_w, row_offset, rows_total, dtype, tile_n, 0, 0);
    return st != SDMI_OK ? st : run_host(g, out);
}

extern "C" int sdmi_pack_geglu(const float* w, const float* bias, int64_t inner, int64_t K, int64_t ld_w, void* out,
                               float* bias_out, int32_t dtype, void* stream) {
    if (inner <= 0 || inner % 16 != 0 || (bias == nullptr) != (bias_out == nullptr)) return SDMI_ERR_BAD_ARG;
    PackGeom g;
    const int st = make_geom(g, w, 2 * inner, K, ld_w, 0, 2 * inner, dtype, 128, 0, inner);
    if (st != SDMI_OK) return st;
    if (bias != nullptr) {
        hipLaunchKernelGGL(pack_geglu_bias_kernel, dim3((unsigned)((2 * inner + 255) / 256)), dim3(256), 0,
                           reinterpret_cast<hipStream_t>(stream), bias, bias_out, inner);
        SDMI_CHECK_LAUNCH();
    }
    return run_device(g, out, stream);
}
extern "C" int sdmi_pack_geglu_host(const float* w, const float* bias, int64_t inner, int64_t K, int64_t ld_w, void* out,
                                    float* bias_out, int32_t dtype) {
    if (inner <= 0 || inner % 16 != 0 || (bias == nullptr) != (bias_out == nullptr)) return SDMI_ERR_BAD_ARG;
    PackGeom g;
    const int st = make_geom(g, w, 2 * inner, K, ld_w, 0, 2 * inner, dtype, 128, 0, inner);
    if (st != SDMI_OK) return st;
    if (bias != nullptr)
        for (int64_t r = 0; r < 2 * inner; ++r) bias_out[r] = bias[((r % 32) / 16) * inner + (r / 32) * 16 + (r % 16)];
    return run_host(g, out);
}

extern "C" int sdmi_pack_conv3x3(const float* w, int64_t Cout, int64_t Cin, void* out, int32_t dtype, int32_t tile_n,
                                 void* stream) {
    PackGeom g;
    const int st = make_geom(g, w, Cout, Cin, 0, 0, Cout, dtype, tile_n, 1, 0);
    return st != SDMI_OK ? st : run_device(g, out, stream);
}
extern "C" int sdmi_pack_conv3x3_host(const float* w, int64_t Cout, int64_t Cin, void* out, int32_t dtype, int32_t tile_n) {
    PackGeom g;
    const int st = make_geom(g, w, Cout, Cin, 0, 0, Cout, dtype, tile_n, 1, 0);
    return st != SDMI_OK ? st : run_host(g, out);
}
