// sdmi_elem: layout conversion, timestep embedding, DDIM update, image quantisation and the
// row/column softmax used by the VAE single-head attention.  All HBM-bound, coalesced.
#include "sdmi_common.h"

namespace {

template <int DT>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, unsigned short* __restrict__ out,
                                                           int nb, int C, int Cpad, int64_t hw, float scale) {
    const int nvec = Cpad >> 3;
    const int64_t total = (int64_t)nb * hw * nvec;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        // consecutive threads -> consecutive pixels (coalesced fp32 reads per channel plane)
        const int64_t pix = idx % hw;
        const int64_t r = idx / hw;
        const int v = (int)(r % nvec);
        const int b = (int)(r / nvec);
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = v * 8 + j;
            f[j] = c < C ? x[((int64_t)b * C + c) * hw + pix] * scale : 0.f;
        }
        *reinterpret_cast<uint4*>(out + ((int64_t)b * hw + pix) * Cpad + v * 8) = elem_pack8<DT>(f);
    }
}

template <int DT>
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ out,
                                                          int64_t n) {
    const int64_t nv = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
        float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        *reinterpret_cast<uint4*>(out + i * 8) = elem_pack8<DT>(f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const int64_t i = (nv << 3) + threadIdx.x;
        out[i] = elem_from_f32<DT>(x[i]);
    }
}

template <int DT>
__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, unsigned short* __restrict__ out, int nb,
                                          int dim, float max_period) {
    const int half = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= half || b >= nb) return;
    // ldm/utils.py:304-314: freq = exp(-ln(max_period) * i / half); [cos | sin]
    const float freq = expf(-logf(max_period) * (float)i / (float)half);
    const float arg = (float)t[b] * freq;
    out[(int64_t)b * dim + i] = elem_from_f32<DT>(cosf(arg));
    out[(int64_t)b * dim + half + i] = elem_from_f32<DT>(sinf(arg));
}

__global__ __launch_bounds__(256) void ddim_step_kernel(const float* __restrict__ x, const float* __restrict__ mo,
                                                        const float* __restrict__ noise, float* __restrict__ x_prev,
                                                        float* __restrict__ pred_x0, int64_t n, int cfg, float gs,
                                                        float a_t, float a_prev, float sigma_t, float s1m, int v_param) {
    const float sqrt_a = sqrtf(a_t);
    const float sqrt_ap = sqrtf(a_prev);
    const float dir = sqrtf(1.0f - a_prev - sigma_t * sigma_t);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float o;
        if (cfg) {
            const float eu = mo[i], ec = mo[n + i];
            o = eu + gs * (ec - eu);
        } else {
            o = mo[i];
        }
        const float xv = x[i];
        float e_t, p0;
        if (v_param) {
            e_t = sqrt_a * o + s1m * xv;
            p0 = sqrt_a * xv - s1m * o;
        } else {
            e_t = o;
            p0 = (xv - s1m * e_t) / sqrt_a;
        }
        float xp = sqrt_ap * p0 + dir * e_t;
        if (noise != nullptr) xp += sigma_t * noise[i];
        x_prev[i] = xp;
        if (pred_x0 != nullptr) pred_x0[i] = p0;
    }
}

__global__ __launch_bounds__(256) void image_to_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int nb,
                                                          int C, int64_t hw) {
    const int64_t total = (int64_t)nb * hw;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = idx / hw, pix = idx - b * hw;
        for (int c = 0; c < C; ++c) {
            float v = (x[(b * C + c) * hw + pix] + 1.0f) * 0.5f;
            v = fminf(fmaxf(v, 0.f), 1.f) * 255.0f;
            out[idx * C + c] = (uint8_t)v;   // truncation, as numpy .astype(uint8) in upstream txt2img
        }
    }
}

// ---- softmax helpers (fp32 scores from sdmi_gemm out_kind 1) ----
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float u = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, u) : v + u;
    }
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
    return r;
}

template <int DT>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, unsigned short* __restrict__ P,
                                                           int64_t cols, int64_t ld_s, int64_t ld_p) {
    __shared__ float sh[4];
    const float* s = S + (int64_t)blockIdx.x * ld_s;
    unsigned short* p = P + (int64_t)blockIdx.x * ld_p;
    float mx = -INFINITY;
    for (int64_t j = threadIdx.x; j < cols; j += blockDim.x) mx = fmaxf(mx, s[j]);
    mx = block_reduce(mx, true, sh);
    float sum = 0.f;
    for (int64_t j = threadIdx.x; j < cols; j += blockDim.x) sum += __expf(s[j] - mx);
    sum = block_reduce(sum, false, sh);
    const float inv = 1.0f / sum;
    for (int64_t j = threadIdx.x; j < cols; j += blockDim.x) p[j] = elem_from_f32<DT>(__expf(s[j] - mx) * inv);
}

// column statistics of ST [batch][rows][cols]: thread per column, online (max, sum) over a row range.
__global__ __launch_bounds__(256) void softmax_colstats_kernel(const float* __restrict__ ST, float* __restrict__ part,
                                                               int64_t rows, int64_t cols, int nsplit) {
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int sp = blockIdx.y, b = blockIdx.z;
    if (col >= cols) return;
    const int64_t per = (rows + nsplit - 1) / nsplit;
    const int64_t r0 = sp * per, r1 = (r0 + per < rows) ? r0 + per : rows;
    const float* s = ST + (int64_t)b * rows * cols + col;
    float m = -INFINITY, l = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
        const float v = s[r * cols];
        const float mn = fmaxf(m, v);
        l = l * __expf(m - mn) + __expf(v - mn);
        m = mn;
    }
    float* o = part + (((int64_t)b * nsplit + sp) * 2) * cols + col;
    o[0] = m;
    o[cols] = l;
}

__global__ __launch_bounds__(256) void softmax_colmerge_kernel(const float* __restrict__ part, float* __restrict__ stats,
                                                               int64_t cols, int nsplit) {
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (col >= cols) return;
    float m = -INFINITY, l = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) {
        const float* o = part + (((int64_t)b * nsplit + sp) * 2) * cols + col;
        const float pm = o[0], pl = o[cols];
        if (pl == 0.f) continue;
        const float mn = fmaxf(m, pm);
        l = l * __expf(m - mn) + pl * __expf(pm - mn);
        m = mn;
    }
    stats[((int64_t)b * 2) * cols + col] = m;
    stats[((int64_t)b * 2 + 1) * cols + col] = 1.0f / l;
}

template <int DT>
__global__ __launch_bounds__(256) void softmax_colapply_kernel(const float* __restrict__ ST, unsigned short* __restrict__ PT,
                                                               const float* __restrict__ stats, int64_t rows,
                                                               int64_t cols) {
    // one thread handles 4 consecutive columns of one row
    const int b = blockIdx.y;
    const int64_t qpr = cols >> 2;
    const int64_t total = rows * qpr;
    const float* mrow = stats + ((int64_t)b * 2) * cols;
    const float* irow = mrow + cols;
    const float* s = ST + (int64_t)b * rows * cols;
    unsigned short* p = PT + (int64_t)b * rows * cols;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / qpr, c = (idx - r * qpr) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(s + r * cols + c);
        const f32x4 m = *reinterpret_cast<const f32x4*>(mrow + c);
        const f32x4 il = *reinterpret_cast<const f32x4*>(irow + c);
        uint2 o;
        o.x = elem_pack2<DT>(__expf(v[0] - m[0]) * il[0], __expf(v[1] - m[1]) * il[1]);
        o.y = elem_pack2<DT>(__expf(v[2] - m[2]) * il[2], __expf(v[3] - m[3]) * il[3]);
        *reinterpret_cast<uint2*>(p + r * cols + c) = o;
    }
}

constexpr int COL_SPLIT = 16;

inline unsigned grid_for(int64_t n, int per_block = 256, int64_t cap = 8192) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

}  // namespace

#define SDMI_DT_CHECK(dt) do { if ((dt) != SDMI_DTYPE_BF16 && (dt) != SDMI_DTYPE_F16) return SDMI_ERR_BAD_ARG; } while (0)

extern "C" int sdmi_nchw_f32_to_nhwc(const float* x, void* out, int32_t nb, int32_t C, int32_t Cpad, int64_t hw,
                                     float scale, int32_t dtype, void* stream) {
    SDMI_DT_CHECK(dtype);
    if (x == nullptr || out == nullptr || nb <= 0 || C <= 0 || hw <= 0) return SDMI_ERR_BAD_ARG;
    if (Cpad < C || Cpad % 8 != 0) return SDMI_ERR_ALIGNMENT;
    const int64_t total = (int64_t)nb * hw * (Cpad / 8);
    if (dtype == SDMI_DTYPE_F16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<1>, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           x, reinterpret_cast<unsigned short*>(out), nb, C, Cpad, hw, scale);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<0>, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           x, reinterpret_cast<unsigned short*>(out), nb, C, Cpad, hw, scale);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

extern "C" int sdmi_cast_f32(const float* x, void* out, int64_t n, int32_t dtype, void* stream) {
    if (x == nullptr || out == nullptr || n <= 0) return SDMI_ERR_BAD_ARG;
    SDMI_DT_CHECK(dtype);
    if (dtype == SDMI_DTYPE_F16)
        hipLaunchKernelGGL(f32_to_bf16_kernel<1>, dim3(grid_for((n + 7) / 8)), dim3(256), 0,
                           reinterpret_cast<hipStream_t>(stream), x, reinterpret_cast<unsigned short*>(out), n);
    else
        hipLaunchKernelGGL(f32_to_bf16_kernel<0>, dim3(grid_for((n + 7) / 8)), dim3(256), 0,
                           reinterpret_cast<hipStream_t>(stream), x, reinterpret_cast<unsigned short*>(out), n);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

extern "C" int sdmi_timestep_embedding(const int64_t* t, void* out, int32_t nb, int32_t dim, float max_period,
                                       int32_t dtype, void* stream) {
    if (t == nullptr || out == nullptr || nb <= 0 || dim <= 0) return SDMI_ERR_BAD_ARG;
    SDMI_DT_CHECK(dtype);
    if (dim % 2 != 0) return SDMI_ERR_ALIGNMENT;
    const int half = dim / 2;
    if (dtype == SDMI_DTYPE_F16)
        hipLaunchKernelGGL(timestep_embedding_kernel<1>, dim3((unsigned)((half + 63) / 64), (unsigned)nb), dim3(64), 0,
                           reinterpret_cast<hipStream_t>(stream), t, reinterpret_cast<unsigned short*>(out), nb, dim, max_period);
    else
        hipLaunchKernelGGL(timestep_embedding_kernel<0>, dim3((unsigned)((half + 63) / 64), (unsigned)nb), dim3(64), 0,
                           reinterpret_cast<hipStream_t>(stream), t, reinterpret_cast<unsigned short*>(out), nb, dim, max_period);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

extern "C" int sdmi_ddim_step(const float* x, const float* model_out, const float* noise, float* x_prev, float* pred_x0,
                              int64_t n, int32_t cfg, float guidance_scale, float a_t, float a_prev, float sigma_t,
                              float sqrt_one_minus_a_t, int32_t v_param, void* stream) {
    if (x == nullptr || model_out == nullptr || x_prev == nullptr || n <= 0) return SDMI_ERR_BAD_ARG;
    if (!(a_t > 0.f) || !(a_prev > 0.f)) return SDMI_ERR_BAD_ARG;
    hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                       model_out, noise, x_prev, pred_x0, n, cfg, guidance_scale, a_t, a_prev, sigma_t,
                       sqrt_one_minus_a_t, v_param);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

namespace {
// out = w0*x0 + w1*x1 + w2*x2 + w3*x3 (fp32; null inputs are skipped) - the PLMS multistep eps combination and
// the classifier-free-guidance combine  e_uc + s*(e_c - e_uc) = (1-s)*e_uc + s*e_c
__global__ __launch_bounds__(256) void lincomb4_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                       const float* __restrict__ x2, const float* __restrict__ x3,
                                                       float w0, float w1, float w2, float w3,
                                                       float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = w0 * x0[i];
        if (x1 != nullptr) v += w1 * x1[i];
        if (x2 != nullptr) v += w2 * x2[i];
        if (x3 != nullptr) v += w3 * x3[i];
        out[i] = v;
    }
}
}  // namespace

namespace {
__global__ __launch_bounds__(256) void diag_gaussian_kernel(const float* __restrict__ prm, float* __restrict__ mean,
                                                            float* __restrict__ log_var, float* __restrict__ stdv,
                                                            float* __restrict__ var, int nb, int C, int64_t hw, int det) {
    const int64_t per = (int64_t)C * hw, total = (int64_t)nb * per;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / per, r = i - b * per;
        const float m = prm[b * 2 * per + r];
        const float lv = fminf(fmaxf(prm[b * 2 * per + per + r], -30.0f), 20.0f);
        mean[i] = m;
        log_var[i] = lv;
        stdv[i] = det ? 0.f : expf(0.5f * lv);
        var[i] = det ? 0.f : expf(lv);
    }
}
__global__ __launch_bounds__(256) void mul_add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ c, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i] * c[i];
}
}  // namespace

extern "C" int sdmi_diag_gaussian(const float* parameters, float* mean, float* log_var, float* stdv, float* var, int32_t nb,
                                  int32_t C, int64_t hw, int32_t deterministic, void* stream) {
    if (parameters == nullptr || mean == nullptr || log_var == nullptr || stdv == nullptr || var == nullptr) return SDMI_ERR_BAD_ARG;
    if (nb <= 0 || C <= 0 || hw <= 0) return SDMI_ERR_BAD_ARG;
    hipLaunchKernelGGL(diag_gaussian_kernel, dim3(grid_for((int64_t)nb * C * hw)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), parameters, mean, log_var, stdv, var, nb, C, hw, deterministic);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

extern "C" int sdmi_mul_add(const float* a, const float* b, const float* c, float* out, int64_t n, void* stream) {
    if (a == nullptr || b == nullptr || c == nullptr || out == nullptr || n <= 0) return SDMI_ERR_BAD_ARG;
    hipLaunchKernelGGL(mul_add_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, b, c, out, n);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

extern "C" int sdmi_lincomb4(const float* x0, const float* x1, const float* x2, const float* x3, float w0, float w1,
                             float w2, float w3, float* out, int64_t n, void* stream) {
    if (x0 == nullptr || out == nullptr || n <= 0) return SDMI_ERR_BAD_ARG;
    hipLaunchKernelGGL(lincomb4_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x0, x1, x2,
                       x3, w0, w1, w2, w3, out, n);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

extern "C" int sdmi_image_to_u8(const float* x, uint8_t* out, int32_t nb, int32_t C, int64_t hw, void* stream) {
    if (x == nullptr || out == nullptr || nb <= 0 || C <= 0 || hw <= 0) return SDMI_ERR_BAD_ARG;
    hipLaunchKernelGGL(image_to_u8_kernel, dim3(grid_for((int64_t)nb * hw)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, out, nb, C, hw);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

extern "C" int sdmi_softmax_rows(const float* S, void* P, int64_t rows, int64_t cols, int64_t ld_s, int64_t ld_p,
                                 int32_t dtype, void* stream) {
    SDMI_DT_CHECK(dtype);
    if (S == nullptr || P == nullptr || rows <= 0 || cols <= 0 || ld_s < cols || ld_p < cols) return SDMI_ERR_BAD_ARG;
    if (rows > 0x7fffffffLL) return SDMI_ERR_BAD_ARG;
    if (dtype == SDMI_DTYPE_F16)
        hipLaunchKernelGGL(softmax_rows_kernel<1>, dim3((unsigned)rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), S,
                           reinterpret_cast<unsigned short*>(P), cols, ld_s, ld_p);
    else
        hipLaunchKernelGGL(softmax_rows_kernel<0>, dim3((unsigned)rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), S,
                           reinterpret_cast<unsigned short*>(P), cols, ld_s, ld_p);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}

// colstats scratch: [batch][COL_SPLIT + 1][2][cols] fp32 (partials then the merged stats)
extern "C" int64_t sdmi_softmax_cols_scratch_bytes(int32_t batch, int64_t cols) {
    return (int64_t)batch * (COL_SPLIT + 1) * 2 * cols * (int64_t)sizeof(float);
}

extern "C" int sdmi_softmax_cols(const float* ST, void* PT, float* colstats, int32_t batch, int64_t rows, int64_t cols,
                                 int32_t dtype, void* stream) {
    SDMI_DT_CHECK(dtype);
    if (ST == nullptr || PT == nullptr || colstats == nullptr || batch <= 0 || rows <= 0 || cols <= 0)
        return SDMI_ERR_BAD_ARG;
    if (cols % 4 != 0) return SDMI_ERR_ALIGNMENT;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* part = colstats;
    float* stats = colstats + (int64_t)batch * COL_SPLIT * 2 * cols;
    const unsigned cb = (unsigned)((cols + 255) / 256);
    hipLaunchKernelGGL(softmax_colstats_kernel, dim3(cb, COL_SPLIT, (unsigned)batch), dim3(256), 0, s, ST, part, rows,
                       cols, COL_SPLIT);
    SDMI_CHECK_LAUNCH();
    hipLaunchKernelGGL(softmax_colmerge_kernel, dim3(cb, (unsigned)batch), dim3(256), 0, s, part, stats, cols, COL_SPLIT);
    SDMI_CHECK_LAUNCH();
    if (dtype == SDMI_DTYPE_F16)
        hipLaunchKernelGGL(softmax_colapply_kernel<1>, dim3(grid_for(rows * (cols / 4), 256, 4096), (unsigned)batch), dim3(256),
                           0, s, ST, reinterpret_cast<unsigned short*>(PT), stats, rows, cols);
    else
        hipLaunchKernelGGL(softmax_colapply_kernel<0>, dim3(grid_for(rows * (cols / 4), 256, 4096), (unsigned)batch), dim3(256),
                           0, s, ST, reinterpret_cast<unsigned short*>(PT), stats, rows, cols);
    SDMI_CHECK_LAUNCH();
    return SDMI_OK;
}
