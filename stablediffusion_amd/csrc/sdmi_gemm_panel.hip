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

// 1 = launched, 0 = this problem is not one the panel kernel is built for, < 0 = sdmi_status
int sdmi_gemm_panel_launch(const sdmi_gemm_args& p, hipStream_t s) {
    if (p.mode != 0 || p.batch > 1 || p.split_k > 1 || p.A2 != nullptr || p.emb != nullptr || p.bias_per_row || p.res_f32 ||
        p.out_kind != 0 || (p.k_valid != 0 && p.k_valid != p.K))
        return 0;
    if (p.K != 320 || p.M < 16384) return 0;      // fewer rows do not fill the chip with 64 / 128-row panels
    const int64_t n_w = p.geglu ? 2 * p.N : p.N;
    const int bn = p.geglu ? 128 : 160;
    if (n_w % bn != 0 || (p.N & 7) != 0) return 0;
    if (((p.ld_out | p.ld_res) & 7) != 0 || (((uintptr_t)p.out | (uintptr_t)p.res) & 15) != 0) return 0;
    if (p.geglu && (p.res != nullptr || p.act != 0)) return 0;
    int which;
    if (p.geglu) which = 2;
    else which = n_w >= 960 ? 1 : 0;
    if ((p.M * p.lda + 8) * 2 >= (int64_t)1 << 32) { /* per-lane offsets are relative to the row tile: always small */ }
    static std::once_flag once;
    static int attr_status = SDMI_OK;
    std::call_once(once, [] {
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 3; ++b)
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(k320[a][b].fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        k320[a][b].lds) != hipSuccess)
                    attr_status = SDMI_ERR_LAUNCH;
    });
    if (attr_status != SDMI_OK) return attr_status;
    const PanelVariant& v = k320[p.dtype == SDMI_DTYPE_F16 ? 1 : 0][which];
    const int64_t blocks = (p.M + v.bm - 1) / v.bm;
    if (blocks > 0x7fffffffLL) return 0;
    hipLaunchKernelGGL(v.fn, dim3((unsigned)blocks), dim3(v.threads), v.lds, s, p);
    if (hipGetLastError() != hipSuccess) return SDMI_ERR_LAUNCH;
    return 1;
}
