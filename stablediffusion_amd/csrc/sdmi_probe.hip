// sdmi_probe: two measurement kernels behind bench.py's `peak_measured` fields ("report fractions
// against both nominal and measured peaks").  They compute nothing the path uses.
//   * mfma_probe_kernel: a register-only v_mfma_f32_16x16x32_f16 loop (no memory traffic at all) on four rotating sets
//     of pseudo-random operands, 8 waves per CU on every CU - the matrix rate the part sustains at the clock it holds
//     on random data (the nominal 2.5 PFLOP/s assumes 2.4 GHz, which no kernel working on random 16-bit operands is
//     clocked at: DESIGN.md section 3).
//   * copy_probe_kernel: a 16-byte-per-lane streaming copy - the HBM rate the part sustains (read + write).
#include "sdmi_common.h"

namespace {

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(512, 2) void mfma_probe_kernel(float* out, int iters) {
    f16x8 ra[4], rb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned h = hash32((threadIdx.x * 64 + s * 8 + j) * 2654435761u + blockIdx.x * 7919u + 8u);
            ra[s][j] = (_Float16)(((int)(h & 0xffff) - 32768) * (1.f / 32768.f));
            rb[s][j] = (_Float16)(((int)(h >> 16) - 32768) * (1.f / 32768.f));
        }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[i & 3], rb[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) out[threadIdx.x] = s;      // keeps the loop alive; practically never true
}

// four independent 16-byte loads in flight per lane and trip (one load per trip reaches ~4.6 TB/s, four ~6 TB/s)
__global__ __launch_bounds__(256) void copy_probe_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

}  // namespace

extern "C" int sdmi_probe_mfma(float* out, int32_t iters, double* flop, void* stream) {
    if (out == nullptr || iters <= 0) return SDMI_ERR_BAD_ARG;
    constexpr int BLOCKS = 256, THREADS = 512;
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(BLOCKS), dim3(THREADS), 0, reinterpret_cast<hipStream_t>(stream), out, iters);
    SDMI_CHECK_LAUNCH();
    if (flop != nullptr) *flop = (double)BLOCKS * (THREADS / 64) * (double)iters * 16.0 * 16384.0;
    return SDMI_OK;
}

extern "C" int sdmi_probe_copy(const void* src, void* dst, int64_t bytes, void* stream) {
    if (src == nullptr || dst == nullptr || bytes <= 0 || (bytes & 15) != 0) return SDMI_ERR_BAD_ARG;
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) != 0) return SDMI_ERR_ALIGNMENT;
    hipLaunchKernelGGL(copy_probe_kernel, dim3(256 * 32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), bytes / 16);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}
