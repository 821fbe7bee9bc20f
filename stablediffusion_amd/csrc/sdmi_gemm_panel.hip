// sdmi_gemm_panel: instantiations and dispatch of the A-panel-stationary short-K GEMM (sdmi_gemm_panel.h).
#include <mutex>
#include "sdmi_gemm_panel.h"

namespace {

typedef void (*kern_t)(const sdmi_gemm_args);
struct PanelVariant { kern_t fn; int lds; int threads; int bm; };

// K = 320 (KT = 5).  [dtype][which]:
//   0  S : 64-row panels, 4 waves, two workgroups per CU, 160 columns, 2-stage ring            (N < 960)
//   1  L : 128-row panels, 8 waves, one workgroup per CU, 160 columns, 3-stage ring            (N >= 960)
//   2  LG: 128-row panels, 8 waves, GEGLU (128 weight columns), 4-stage ring
#define SDMI_PANEL_SET(DT)                                                                                         \
    {{sdmi_gemm_panel_kernel<DT, 5, 2, 5, 2, false>, panel_lds_bytes(5, 2, 5, 2), 256, 64},                         \
     {sdmi_gemm_panel_kernel<DT, 5, 4, 5, 3, false>, panel_lds_bytes(5, 4, 5, 3), 512, 128},                        \
     {sdmi_gemm_panel_kernel<DT, 4, 4, 5, 4, true>, panel_lds_bytes(4, 4, 5, 4), 512, 128}}
const PanelVariant k320[2][3] = {SDMI_PANEL_SET(0), SDMI_PANEL_SET(1)};
#undef SDMI_PANEL_SET

}  // namespace

// the problems the panel kernel is built for
static bool panel_applies(const sdmi_gemm_args& p) {
    if (p.dtype != SDMI_DTYPE_BF16 && p.dtype != SDMI_DTYPE_F16) return false;
    if (p.A == nullptr || p.W == nullptr || p.out == nullptr || p.algo == 1) return false;
    if (p.mode != 0 || p.batch > 1 || p.split_k > 1 || p.A2 != nullptr || p.ln_colsum != nullptr || p.row_stats_out != nullptr || p.emb != nullptr || p.bias_per_row || p.res_f32 ||
        p.out_kind != 0 || (p.k_valid != 0 && p.k_valid != p.K))
        return false;
    if (p.K != 320 || p.M < 16384) return false;      // fewer rows do not fill the chip with 64 / 128-row panels
    const int64_t n_w = p.geglu ? 2 * p.N : p.N;
    const int bn = p.geglu ? 128 : 160;
    if (n_w % bn != 0 || (p.N & 7) != 0) return false;
    if (((p.ld_out | p.ld_res | p.lda) & 7) != 0 || (((uintptr_t)p.out | (uintptr_t)p.res | (uintptr_t)p.A) & 15) != 0) return false;
    if (p.geglu && (p.res != nullptr || p.act != 0)) return false;
    if ((p.M + 127) / 128 * 2 > 0x7fffffffLL) return false;
    return true;
}

extern "C" int sdmi_gemm_panel_applies(const sdmi_gemm_args* a) {
    if (a == nullptr) return 0;
    sdmi_gemm_args p = *a;
    if (p.batch <= 0) p.batch = 1;
    if (p.split_k <= 0) p.split_k = 1;
    return panel_applies(p) ? 1 : 0;
}

// 1 = launched, 0 = this problem is not one the panel kernel is built for, < 0 = sdmi_status
int sdmi_gemm_panel_launch(const sdmi_gemm_args& p, hipStream_t s) {
    if (!panel_applies(p)) return 0;
    const int64_t n_w = p.geglu ? 2 * p.N : p.N;
#ifndef SDMI_PANEL_L_MIN_N
#define SDMI_PANEL_L_MIN_N 960
#endif
    const int which = p.geglu ? 2 : (n_w >= SDMI_PANEL_L_MIN_N ? 1 : 0);
    static SdmiPerDeviceOnce once;      // per device: hipFuncSetAttribute acts on the current device
    const int attr_status = once.run([] {
        bool ok = true;
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 3; ++b)
                ok = (hipFuncSetAttribute(reinterpret_cast<const void*>(k320[a][b].fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          k320[a][b].lds) == hipSuccess) && ok;
        return ok;
    });
    if (attr_status != SDMI_OK) return attr_status;
    const PanelVariant& v = k320[p.dtype == SDMI_DTYPE_F16 ? 1 : 0][which];
    const int64_t blocks = (p.M + v.bm - 1) / v.bm;
    hipLaunchKernelGGL(v.fn, dim3((unsigned)blocks), dim3(v.threads), v.lds, s, p);
    if (hipGetLastError() != hipSuccess) return SDMI_ERR_LAUNCH;
    return 1;
}
