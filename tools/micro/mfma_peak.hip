// Sustained MFMA rate of the part (development aid): register-only v_mfma loops, no memory traffic.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_peak tools/micro/mfma_peak.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// KIND 2: 16x16x32 with four sets of pseudo-random operands rotated from MFMA to MFMA (the multiplier inputs toggle
// the way they do in a GEMM on random data; KIND 0 / 1 feed every MFMA the same, nearly constant operands)
template <int KIND>
__global__ __launch_bounds__(512, 2) void mfma_loop(float* out, int iters, float seed) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed + threadIdx.x * 1e-3f); b[j] = (_Float16)(seed - j * 1e-3f); }
    if constexpr (KIND == 2) {
        f16x8 ra[4], rb[4];
        for (int s = 0; s < 4; ++s)
            for (int j = 0; j < 8; ++j) {
                const unsigned h = hash32((threadIdx.x * 64 + s * 8 + j) * 2654435761u + (unsigned)(seed * 16.f));
                ra[s][j] = (_Float16)(((int)(h & 0xffff) - 32768) * (1.f / 32768.f));
                rb[s][j] = (_Float16)(((int)(h >> 16) - 32768) * (1.f / 32768.f));
            }
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[i & 3], rb[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        if (s == 12345.678f) out[threadIdx.x] = s;
        return;
    }
    if constexpr (KIND == 0) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        if (s == 12345.678f) out[threadIdx.x] = s;
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
        if (s == 12345.678f) out[threadIdx.x] = s;
    }
}

template <int KIND>
static void run(const char* name, int blocks, int threads, int iters, int reps) {
    float* out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.5f);
    hipDeviceSynchronize();
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double waves = (double)blocks * threads / 64;
        const double flop = waves * iters * (KIND != 1 ? 16 * 16384.0 : 8 * 32768.0);
        printf("%s blocks=%d threads=%d iters=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks, threads, iters, ms, flop / ms / 1e9);
    }
    hipFree(out);
}

int main(int argc, char** argv) {
    if (argc > 1) {      // one variant only (for a counter pass): 0 constant operands, 2 random operands
        if (argv[1][0] == '2') run<2>("16x16x32 f16 random operands, 8 waves/CU, long", 256, 512, 40000, 3);
        else                   run<0>("16x16x32 f16, 8 waves/CU, long", 256, 512, 40000, 3);
        return 0;
    }
    run<2>("16x16x32 f16 random operands, 8 waves/CU", 256, 512, 4000, 3);
    run<2>("16x16x32 f16 random operands, 8 waves/CU, long", 256, 512, 40000, 3);
    run<0>("16x16x32 f16, 8 waves/CU", 256, 512, 4000, 3);
    run<1>("32x32x16 f16, 8 waves/CU", 256, 512, 4000, 3);
    run<0>("16x16x32 f16, 4 waves/CU", 256, 256, 8000, 3);
    run<1>("32x32x16 f16, 4 waves/CU", 256, 256, 8000, 3);
    run<0>("16x16x32 f16, 8 waves/CU, long", 256, 512, 40000, 3);
    return 0;
}
