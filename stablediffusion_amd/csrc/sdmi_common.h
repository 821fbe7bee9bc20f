// Shared device helpers for the sdmi gfx950 kernels (bf16 packing, MFMA vector types).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "../../include/sdmi.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define SDMI_CHECK_LAUNCH()                                  \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return SDMI_ERR_LAUNCH; \
    } while (0)

// hipFuncSetAttribute acts on the CURRENT device's copy of a function, so "once" means once per device: a process
// that drives several GPUs runs `set` on each of them the first time it launches there.  Thread-safe; a refusal
// sticks and is reported on every later call (SDMI_ERR_LAUNCH).
struct SdmiPerDeviceOnce {
    static constexpr int MAX_DEVICES = 64;
    std::mutex m;
    signed char state[MAX_DEVICES] = {};      // 0 = not run yet, 1 = done, -1 = refused
    template <class F> int run(F&& set) {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return SDMI_ERR_LAUNCH;
        std::lock_guard<std::mutex> lock(m);
        if (state[dev] == 0) state[dev] = set() ? 1 : -1;
        return state[dev] > 0 ? SDMI_OK : SDMI_ERR_LAUNCH;
    }
};

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned)b) << 16);
}

// Round-to-nearest-even f32 -> bf16 through the hardware convert (keeps NaN a NaN).
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2v;

// one v_cvt_pk_bf16_f32 (round to nearest even)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}

__device__ __forceinline__ void unpack_bf16x8(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

__device__ __forceinline__ uint4 pack_bf16x8(const float* f) {
    uint4 v;
    v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
    return v;
}

// ---- element type policy: DT = 0 bf16, 1 f16 (both 16-bit MFMA operand formats run at the same rate) ----
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) float f32x8;

template <int DT> __device__ __forceinline__ float elem_to_f32(unsigned short b) {
    if constexpr (DT == 0) return bf16_bits_to_f32(b);
    else return (float)__builtin_bit_cast(_Float16, b);
}
template <int DT> __device__ __forceinline__ unsigned short elem_from_f32(float f) {
    if constexpr (DT == 0) return f32_to_bf16_bits(f);
    else return __builtin_bit_cast(unsigned short, (_Float16)f);
}
template <int DT> __device__ __forceinline__ unsigned elem_pack2(float lo, float hi) {
    if constexpr (DT == 0) {
        return pack_bf16x2(lo, hi);
    } else {
        const f32x2 v = {lo, hi};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2v));   // one v_cvt_pk_f16_f32
    }
}
template <int DT> __device__ __forceinline__ void elem_unpack8(const uint4& v, float* f) {
    if constexpr (DT == 0) {
        unpack_bf16x8(v, f);
    } else {
        const f16x8 h = __builtin_bit_cast(f16x8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (float)h[j];
    }
}
template <int DT> __device__ __forceinline__ uint4 elem_pack8(const float* f) {
    if constexpr (DT == 0) {
        return pack_bf16x8(f);
    } else {
        uint4 v;
        v.x = elem_pack2<1>(f[0], f[1]); v.y = elem_pack2<1>(f[2], f[3]);
        v.z = elem_pack2<1>(f[4], f[5]); v.w = elem_pack2<1>(f[6], f[7]);
        return v;
    }
}
template <int DT> __device__ __forceinline__ void elem_unpack4(const uint2& v, float* f) {
    f[0] = elem_to_f32<DT>((unsigned short)(v.x & 0xffffu)); f[1] = elem_to_f32<DT>((unsigned short)(v.x >> 16));
    f[2] = elem_to_f32<DT>((unsigned short)(v.y & 0xffffu)); f[3] = elem_to_f32<DT>((unsigned short)(v.y >> 16));
}
// The bf16 operand policy keeps its residual stream in f16 (include/sdmi.h, sdmi_gemm_args.out_f16 / res_f16 / a_f16):
// in a kernel instantiated for bf16 operands (DT == 0) a residual / output / LayerNorm input may be f16, chosen at run
// time by a wave-uniform flag.  Kernels instantiated for f16 operands (DT == 1) compile to exactly what they were.
template <int DT> __device__ __forceinline__ unsigned elem_pack2_sel(float lo, float hi, int f16) {
    if constexpr (DT == 0) { if (f16) return elem_pack2<1>(lo, hi); }
    return elem_pack2<DT>(lo, hi);
}
template <int DT> __device__ __forceinline__ void elem_unpack4_sel(const uint2& v, float* f, int f16) {
    if constexpr (DT == 0) { if (f16) { elem_unpack4<1>(v, f); return; } }
    elem_unpack4<DT>(v, f);
}
template <int DT> __device__ __forceinline__ void elem_unpack8_sel(const uint4& v, float* f, int f16) {
    if constexpr (DT == 0) { if (f16) { elem_unpack8<1>(v, f); return; } }
    elem_unpack8<DT>(v, f);
}
template <int DT> __device__ __forceinline__ float elem_to_f32_sel(unsigned short b, int f16) {
    if constexpr (DT == 0) { if (f16) return elem_to_f32<1>(b); }
    return elem_to_f32<DT>(b);
}
template <int DT> __device__ __forceinline__ unsigned short elem_from_f32_sel(float f, int f16) {
    if constexpr (DT == 0) { if (f16) return elem_from_f32<1>(f); }
    return elem_from_f32<DT>(f);
}
template <int DT> __device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
    if constexpr (DT == 0)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <int DT> __device__ __forceinline__ f32x16 mfma32(const uint4& a, const uint4& b, f32x16 c) {
    if constexpr (DT == 0)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the 16-bit output rounding): a dozen VALU
// instructions instead of the library routine's ~40, which matters in the GEGLU epilogue of small-K GEMMs.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}
// full-precision forms for fp32 outputs (the fp32 mode of the path)
__device__ __forceinline__ float silu_precise(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float gelu_erf_precise(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }
// The same GELU for the 16-bit epilogues, two values per instruction and no transcendental: erf(g / sqrt 2) = g Q(z) with
// z = g^2 / 3.5^2 - 1 on |g| <= 3.5 sqrt 2 (clamped beyond: Q is scaled so that the clamp gives exactly +-1), Q a
// degree-12 polynomial (Chebyshev fit, |erf error| <= 1e-6 evaluated in fp32; |GELU error| <= 2.3e-06 absolute - far
// below the fp16 / bf16 rounding of the output).  20 VALU issues per PAIR (two v_med3, 16 v_pk_fma / v_pk_mul + the
// gate product) against ~25 per value for the form above, whose rcp and exp run at quarter rate.
__device__ __forceinline__ f32x2 gelu_erf_pk(f32x2 g) {
#ifdef SDMI_DEBUG_NO_GELU      // timing experiment only (results wrong): what the GELU's VALU work costs a GEGLU launch
    return g * 0.5f;
#else
    constexpr float GC = 4.9497476f;
    f32x2 gc;
    gc[0] = __builtin_amdgcn_fmed3f(g[0], -GC, GC);
    gc[1] = __builtin_amdgcn_fmed3f(g[1], -GC, GC);
    const f32x2 one = {1.0f, 1.0f};
    const f32x2 z = __builtin_elementwise_fma(gc * gc, (f32x2){8.163265139e-02f, 8.163265139e-02f}, -one);
    constexpr float c[13] = {2.855814369e-01f, -1.419164662e-01f, 1.037636884e-01f, -8.105812545e-02f, 6.256822945e-02f, -4.570597125e-02f, 3.155011345e-02f, -2.147688549e-02f, 1.265742458e-02f, -4.842874289e-03f, 2.575485832e-03f, -2.926710946e-03f, 1.261159998e-03f};
    f32x2 q = {c[12], c[12]};
#pragma unroll
    for (int i = 11; i >= 0; --i) q = __builtin_elementwise_fma(q, z, (f32x2){c[i], c[i]});
    const f32x2 hg = g * 0.5f;
    return __builtin_elementwise_fma(hg * gc, q, hg);
#endif
}
__device__ __forceinline__ f32x4 gelu_erf_f4(const f32x4& g) {
    const f32x2 lo = gelu_erf_pk((f32x2){g[0], g[1]}), hi = gelu_erf_pk((f32x2){g[2], g[3]});
    return (f32x4){lo[0], lo[1], hi[0], hi[1]};
}

// ---- LDS-DMA issue (global_load_lds_dwordx4, 1 KiB per wave-instruction) ----
// Issued through inline asm on purpose: with the builtin, hipcc orders every later ds_read behind the DMA
// (s_waitcnt vmcnt(0) ahead of the MFMAs), which serialises load and compute.  The asm form is invisible to its
// wait-count pass; the ring is retired by explicit counted s_waitcnt vmcnt(N).  M0 carries the LDS destination:
// it is saved and restored inside each statement (cdna_hip_programming.md §5.7).  d0 = LDS byte address of this
// wave's slot in instruction 0 of the stage; instruction i lands O0 + STEP*i bytes further.
// "_s": scalar 64-bit base + one 32-bit byte offset per lane;  "_v": a 64-bit pointer per lane.
#define SDMI_M0_SAVE "s_mov_b32 %[keep], m0\n\t"
#define SDMI_M0_REST "s_mov_b32 m0, %[keep]"
#define SDMI_LD_S(N) "s_add_u32 m0, %[d0], %[o" #N "]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v" #N "], %[sb]\n\t"
#define SDMI_LD_V(N) "s_add_u32 m0, %[d0], %[o" #N "]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v" #N "], off\n\t"

template <int O0>
__device__ __forceinline__ void glds_s1(unsigned d0, const void* sb, unsigned v0) {
    unsigned keep;
    asm volatile(SDMI_M0_SAVE SDMI_LD_S(0) SDMI_M0_REST
                 : [keep] "=&s"(keep)
                 : [d0] "s"(d0), [sb] "s"(sb), [v0] "v"(v0), [o0] "n"(O0)
                 : "memory", "scc");
}
template <int O0>
__device__ __forceinline__ void glds_v1(unsigned d0, const void* v0) {
    unsigned keep;
    asm volatile(SDMI_M0_SAVE SDMI_LD_V(0) SDMI_M0_REST
                 : [keep] "=&s"(keep)
                 : [d0] "s"(d0), [v0] "v"(v0), [o0] "n"(O0)
                 : "memory", "scc");
}
template <int O0, int STEP = 8192>
__device__ __forceinline__ void glds_s2(unsigned d0, const void* sb, unsigned v0, unsigned v1) {
    unsigned keep;
    asm volatile(SDMI_M0_SAVE SDMI_LD_S(0) SDMI_LD_S(1) SDMI_M0_REST
                 : [keep] "=&s"(keep)
                 : [d0] "s"(d0), [sb] "s"(sb), [v0] "v"(v0), [v1] "v"(v1), [o0] "n"(O0), [o1] "n"(O0 + STEP)
                 : "memory", "scc");
}
template <int O0, int STEP = 8192>
__device__ __forceinline__ void glds_s3(unsigned d0, const void* sb, unsigned v0, unsigned v1, unsigned v2) {
    unsigned keep;
    asm volatile(SDMI_M0_SAVE SDMI_LD_S(0) SDMI_LD_S(1) SDMI_LD_S(2) SDMI_M0_REST
                 : [keep] "=&s"(keep)
                 : [d0] "s"(d0), [sb] "s"(sb), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [o0] "n"(O0),
                   [o1] "n"(O0 + STEP), [o2] "n"(O0 + 2 * STEP)
                 : "memory", "scc");
}
template <int O0, int STEP = 8192>
__device__ __forceinline__ void glds_s4(unsigned d0, const void* sb, unsigned v0, unsigned v1, unsigned v2, unsigned v3) {
    unsigned keep;
    asm volatile(SDMI_M0_SAVE SDMI_LD_S(0) SDMI_LD_S(1) SDMI_LD_S(2) SDMI_LD_S(3) SDMI_M0_REST
                 : [keep] "=&s"(keep)
                 : [d0] "s"(d0), [sb] "s"(sb), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [o0] "n"(O0),
                   [o1] "n"(O0 + STEP), [o2] "n"(O0 + 2 * STEP), [o3] "n"(O0 + 3 * STEP)
                 : "memory", "scc");
}
template <int O0, int STEP = 8192>
__device__ __forceinline__ void glds_v2(unsigned d0, const void* v0, const void* v1) {
    unsigned keep;
    asm volatile(SDMI_M0_SAVE SDMI_LD_V(0) SDMI_LD_V(1) SDMI_M0_REST
                 : [keep] "=&s"(keep)
                 : [d0] "s"(d0), [v0] "v"(v0), [v1] "v"(v1), [o0] "n"(O0), [o1] "n"(O0 + STEP)
                 : "memory", "scc");
}
template <int O0, int STEP = 8192>
__device__ __forceinline__ void glds_v4(unsigned d0, const void* v0, const void* v1, const void* v2, const void* v3) {
    unsigned keep;
    asm volatile(SDMI_M0_SAVE SDMI_LD_V(0) SDMI_LD_V(1) SDMI_LD_V(2) SDMI_LD_V(3) SDMI_M0_REST
                 : [keep] "=&s"(keep)
                 : [d0] "s"(d0), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [o0] "n"(O0),
                   [o1] "n"(O0 + STEP), [o2] "n"(O0 + 2 * STEP), [o3] "n"(O0 + 3 * STEP)
                 : "memory", "scc");
}
