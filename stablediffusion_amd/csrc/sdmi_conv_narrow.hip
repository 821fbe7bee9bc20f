// sdmi_conv_narrow: conv3x3 (stride 1, pad 1) with at most 16 output channels - the UNet's `out` conv (320 -> 4,
// openai_model.py:913-917) and the VAE decoder's conv_out (128 -> 3, model.py:365-369).  On the tiled kernel those run
// a 128-column tile for 3 or 4 real columns (2 % of the MFMA work is useful: 22-24 TFLOP/s); here the problem is what it
// is - a stream over the input: byte-bound.
//
//   * one 16-column MFMA block (v_mfma_f32_16x16x32: A = the weight fragment, B = 16 pixels x 32 channels of one tap);
//   * the weights (16 x 9 Cin, <= 147 KB) are staged once per workgroup into LDS in fragment order - every k-step is one
//     conflict-free 1 KiB read - and shared by all pixel groups the workgroup walks (grid-stride);
//   * activations never touch LDS memory: with one column block every element feeds exactly one MFMA, so the B
//     fragments come straight from global memory.  Loaded in the MFMA's own lane order (pixel = lane & 15) every
//     quarter-wave would touch 16 different cache lines for 256 bytes (measured: 1.0 TB/s, the tag look-ups bound it);
//     instead lane l loads chunk l & 3 of pixel l >> 2 - a quarter-wave reads 4 pixels x 64 contiguous bytes - and the
//     16-byte values are moved to the MFMA order (lane (c, p) <- lane 4 p + c) with four ds_bpermute_b32, which use the
//     LDS crossbar only.  The nine taps re-read a pixel from L1 / L2; out-of-image taps read a zero array through a
//     per-lane pointer select, so no load sits under a branch;
//   * two accumulators per wave (alternate k-steps) halve the dependent-MFMA chain; occupancy hides the load latency.
// Reads the packed weight buffer of sdmi_pack_conv3x3 as it is (rows 0..15).
#include <mutex>
#include "sdmi_common.h"

namespace {

constexpr int NARROW_MAX_CIN = 512;
__device__ const uint4 narrow_zeros[NARROW_MAX_CIN / 8 + 4] = {};

__device__ __forceinline__ uint4 to_mfma_order(const uint4& v, int perm) {
    uint4 r;
    r.x = (unsigned)__builtin_amdgcn_ds_bpermute(perm, (int)v.x);
    r.y = (unsigned)__builtin_amdgcn_ds_bpermute(perm, (int)v.y);
    r.z = (unsigned)__builtin_amdgcn_ds_bpermute(perm, (int)v.z);
    r.w = (unsigned)__builtin_amdgcn_ds_bpermute(perm, (int)v.w);
    return r;
}

template <int DT, int CS_T>      // CS_T = Cin / 32 at compile time (0: run-time)
__global__ __launch_bounds__(512) void conv3x3_narrow_kernel(const sdmi_gemm_args p) {
    extern __shared__ __attribute__((aligned(16))) uint4 wl[];      // [9 * CS][64] fragments
    const int CS = CS_T > 0 ? CS_T : p.Cin / 32;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned short* const W = reinterpret_cast<const unsigned short*>(p.W);
    const unsigned short* const X = reinterpret_cast<const unsigned short*>(p.A);
    // stage the weight fragments: chunk kc (8 k-values) of column col -> k-step kc / 4, lane (kc % 4) * 16 + col
    const int nchunks = 16 * 9 * CS * 4;
    for (int idx = tid; idx < nchunks; idx += 512) {
        const int col = idx & 15, kc = idx >> 4;
        wl[(kc >> 2) * 64 + (kc & 3) * 16 + col] = *reinterpret_cast<const uint4*>(W + (int64_t)col * p.K + 8 * kc);
    }
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    const int64_t hw = (int64_t)p.Ho * p.Wo;
    const int64_t groups = (p.M + 15) >> 4;
    f32x4 bq = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * fq + j < p.N) bq[j] = p.bias[4 * fq + j];
    }
    const unsigned short* const zsrc = reinterpret_cast<const unsigned short*>(narrow_zeros);
    const int perm = (fr * 4 + fq) * 4;      // byte index of the lane that loaded (pixel fr, chunk fq)
    for (int64_t g = (int64_t)blockIdx.x * 8 + wid; g < groups; g += (int64_t)gridDim.x * 8) {
        // output pixel of this lane's accumulator quad (MFMA order) ...
        const int64_t m = g * 16 + fr;
        const bool mok = m < p.M;
        const int64_t mm = mok ? m : p.M - 1;
        const int64_t b = (unsigned)mm / (unsigned)hw;          // (M < 2^31, host-checked: 32-bit divisions)
        const int rem = (int)(mm - b * hw);
        // ... and the pixel this lane LOADS for (lane >> 2), chunk lane & 3 of every 64-byte k-step
        const int lc = lane & 3;
        const int64_t lm = g * 16 + (lane >> 2) < p.M ? g * 16 + (lane >> 2) : p.M - 1;
        const int64_t lb = (unsigned)lm / (unsigned)hw;
        const int lrem = (int)(lm - lb * hw);
        const int oy = (int)((unsigned)lrem / (unsigned)p.Wo), ox = lrem - oy * p.Wo;
        const unsigned short* const xb = X + (lb * p.H * (int64_t)p.W_in) * p.Cin + 8 * lc;
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        auto tap_src = [&](int tap) {
            const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W_in;
            return ok ? xb + ((int64_t)iy * p.W_in + ix) * p.Cin : zsrc + 8 * lc;
        };
        if constexpr (CS_T > 0) {
            // double-buffered over the taps: the loads of tap t + 1 are in flight while the MFMAs of tap t run;
            // sched_barriers keep hipcc from hoisting all 9 * CS loads to the top (412 VGPRs at Cin = 320)
            uint4 xv[2][CS_T];
            {
                const unsigned short* src = tap_src(0);
#pragma unroll
                for (int cs = 0; cs < CS_T; ++cs) xv[0][cs] = *reinterpret_cast<const uint4*>(src + 32 * cs);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) {
                    const unsigned short* src = tap_src(tap + 1);
#pragma unroll
                    for (int cs = 0; cs < CS_T; ++cs) xv[(tap + 1) & 1][cs] = *reinterpret_cast<const uint4*>(src + 32 * cs);
                }
                __builtin_amdgcn_sched_barrier(0);
                const uint4* wt = wl + (tap * CS_T) * 64 + lane;
#pragma unroll
                for (int cs = 0; cs < CS_T; ++cs) {
                    const uint4 xm = to_mfma_order(xv[tap & 1][cs], perm);
                    if (cs & 1) acc1 = mfma16<DT>(wt[cs * 64], xm, acc1);
                    else        acc0 = mfma16<DT>(wt[cs * 64], xm, acc0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int tap = 0; tap < 9; ++tap) {
                const unsigned short* src = tap_src(tap);
                const uint4* wt = wl + (tap * CS) * 64 + lane;
                for (int cs = 0; cs < CS; cs += 2) {
                    const uint4 x0 = *reinterpret_cast<const uint4*>(src + 32 * cs);
                    const bool two = cs + 1 < CS;
                    const uint4 x1 = *reinterpret_cast<const uint4*>(two ? src + 32 * (cs + 1) : zsrc + 8 * lc);
                    acc0 = mfma16<DT>(wt[cs * 64], to_mfma_order(x0, perm), acc0);
                    acc1 = mfma16<DT>(wt[(two ? cs + 1 : cs) * 64], to_mfma_order(x1, perm), acc1);      // (zero activations: adds nothing)
                }
            }
        }
        // accumulator quad: pixel m (lane & 15), output channels 4 * fq .. 4 * fq + 3
        const f32x4 v = (acc0 + acc1) * p.alpha + bq;
        const int n0 = 4 * fq;
        if (!mok || n0 >= p.N) continue;
        if (p.out_kind == 2) {
            float* o = reinterpret_cast<float*>(p.out) + (b * p.N + n0) * hw + rem;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n0 + j < p.N) o[j * hw] = v[j];
        } else if (p.out_kind == 1) {
            float* o = reinterpret_cast<float*>(p.out) + m * p.ld_out + n0;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n0 + j < p.N) o[j] = v[j];
        } else {
            unsigned short* o = reinterpret_cast<unsigned short*>(p.out) + m * p.ld_out + n0;
            if (n0 + 3 < p.N) {
                uint2 ov;
                ov.x = elem_pack2<DT>(v[0], v[1]);
                ov.y = elem_pack2<DT>(v[2], v[3]);
                *reinterpret_cast<uint2*>(o) = ov;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (n0 + j < p.N) o[j] = elem_from_f32<DT>(v[j]);
            }
        }
    }
}

typedef void (*kern_t)(const sdmi_gemm_args);
template <int DT> kern_t pick(int cs) {
    switch (cs) {
        case 4:  return conv3x3_narrow_kernel<DT, 4>;       // Cin = 128 (decoder conv_out)
        case 10: return conv3x3_narrow_kernel<DT, 10>;      // Cin = 320 (UNet out)
        default: return conv3x3_narrow_kernel<DT, 0>;
    }
}

}  // namespace

// 1 = launched, 0 = not a problem this kernel is built for, < 0 = sdmi_status
int sdmi_conv_narrow_launch(const sdmi_gemm_args& p, hipStream_t s) {
    if (p.mode != 1 || p.N > 16 || p.N <= 0 || p.stride != 1 || p.upsample != 0 || p.asym_pad != 0) return 0;
    if (p.batch > 1 || p.split_k > 1 || p.emb != nullptr || p.res != nullptr || p.act != 0 || p.geglu || p.bias_per_row) return 0;
    if (p.out_f16 || p.dup_rows) return 0;      // (epilogue options of the tiled kernel only)
    if (p.Cin % 32 != 0 || p.Cin > NARROW_MAX_CIN || p.K < 9 * (int64_t)p.Cin || p.K % 8 != 0) return 0;
    if (p.H != p.Ho || p.W_in != p.Wo || p.M >= (1LL << 31)) return 0;
    if (p.out_kind == 0 && (p.ld_out % 4 != 0)) return 0;
    if ((((uintptr_t)p.A | (uintptr_t)p.W) & 15) != 0) return 0;
    static SdmiPerDeviceOnce once;      // per device: hipFuncSetAttribute acts on the current device
    const int attr_status = once.run([] {
        const kern_t all[6] = {conv3x3_narrow_kernel<0, 4>, conv3x3_narrow_kernel<0, 10>, conv3x3_narrow_kernel<0, 0>,
                               conv3x3_narrow_kernel<1, 4>, conv3x3_narrow_kernel<1, 10>, conv3x3_narrow_kernel<1, 0>};
        bool ok = true;
        for (int i = 0; i < 6; ++i)
            ok = (hipFuncSetAttribute(reinterpret_cast<const void*>(all[i]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      9 * (NARROW_MAX_CIN / 32) * 1024) == hipSuccess) && ok;
        return ok;
    });
    if (attr_status != SDMI_OK) return attr_status;
    const int cs = p.Cin / 32;
    const size_t lds = (size_t)9 * cs * 1024;
    const int64_t groups = (p.M + 15) / 16;
    // 8-wave workgroups; per CU as many as the LDS (160 KB) holds, at most 3 (24 waves per CU, what 84 VGPRs allow); each walks its share of
    // the pixel groups
    int per_cu = (int)(160 * 1024 / (lds + 1024));
    if (per_cu > 3) per_cu = 3;
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = (groups + 7) / 8;
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;
    const kern_t k = p.dtype == SDMI_DTYPE_F16 ? pick<1>(cs) : pick<0>(cs);
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(512), lds, s, p);
    if (hipGetLastError() != hipSuccess) return SDMI_ERR_LAUNCH;
    return 1;
}
