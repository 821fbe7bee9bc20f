// Probe for the packed-fma operand form that lost products in the LayerNorm-folded GEMM epilogue:
//   v_pk_fma_f32 d, a, b, d op_sel:[0,1,0]     (low result takes the HIGH half of the second source)
// against the form the library now compiles to (broadcast on the FIRST source, op_sel:[1,0,0]), in isolation: every lane
// repeats the instruction on known values, optionally behind two v_rsq_f32 (the neighbours it had in the failing kernel),
// with 8-wave workgroups and 72 KiB of LDS each so that two of them share a CU.  Counts results that differ from fmaf.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/pk_fma_opsel_probe.hip -o /tmp/pk_probe && /tmp/pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int FORM, int TRANS>
__global__ __launch_bounds__(512) void probe(unsigned* bad, int iters) {
    extern __shared__ float lds[];
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    lds[threadIdx.x] = (float)t;
    __syncthreads();
    f32x2 a = {1.0f + (t & 255) * 0.25f, 2.0f + (t & 127) * 0.5f};
    f32x2 b = {3.0f + (t & 63), 0.5f + (t & 31) * 0.125f};       // the broadcast value is b.y
    float r0 = 1.0f + (t & 15), r1 = 2.0f + (t & 7);
    unsigned wrong = 0;
    for (int i = 0; i < iters; ++i) {
        f32x2 d = {(float)i, (float)(i + 1)};
        const f32x2 want = {__builtin_fmaf(-a.x, b.y, d.x), __builtin_fmaf(-a.y, b.y, d.y)};
        if (TRANS) asm volatile("v_rsq_f32 %0, %0\n\tv_rsq_f32 %1, %1" : "+v"(r0), "+v"(r1));
        if (FORM == 0)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(d) : "v"(a), "v"(b));
        else      // the same products with the broadcast on the first source
            asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(d) : "v"(a), "v"(b));
        wrong += (d.x != want.x) + (d.y != want.y);
        a.x += 0.25f; b.y += 0.125f;
        r0 = r0 * 0.5f + 3.0f; r1 = r1 * 0.25f + 5.0f;
    }
    if (wrong) atomicAdd(bad, wrong);
    if (r0 + r1 + lds[(threadIdx.x * 7) & 511] == -1.0f) atomicAdd(bad, 1u << 30);      // (keeps the values alive)
}

int main() {
    unsigned* bad;
    hipMalloc(&bad, sizeof(unsigned));
    const size_t lds = 72 * 1024;
#define RUN(F, T)                                                                                                   \
    do {                                                                                                            \
        hipFuncSetAttribute(reinterpret_cast<const void*>(probe<F, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipMemset(bad, 0, sizeof(unsigned));                                                                        \
        hipLaunchKernelGGL((probe<F, T>), dim3(2048), dim3(512), lds, 0, bad, 2000);                                \
        unsigned h = 0;                                                                                             \
        hipMemcpy(&h, bad, sizeof(unsigned), hipMemcpyDeviceToHost);                                                \
        printf("form %s%s: %u wrong results of %llu\n", F == 0 ? "op_sel:[0,1,0] (second source)" : "op_sel:[1,0,0] (first source)", \
               T ? " behind two v_rsq_f32" : "", h, 2ull * 2048 * 512 * 2000);                                     \
    } while (0)
    RUN(0, 0); RUN(0, 1); RUN(1, 0); RUN(1, 1);
    hipFree(bad);
    return 0;
}
