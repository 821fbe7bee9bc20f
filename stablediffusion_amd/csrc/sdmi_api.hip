// sdmi_api: status strings and ABI version of libsdmi_hip.so.
#include "sdmi_common.h"

extern "C" const char* sdmi_status_str(int status) {
    switch (status) {
        case SDMI_OK: return "ok";
        case SDMI_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size or unsupported combination)";
        case SDMI_ERR_ALIGNMENT: return "alignment: a leading dimension / channel count breaks the 16-byte vector rule";
        case SDMI_ERR_WORKSPACE: return "workspace missing or too small";
        case SDMI_ERR_LAUNCH: return "HIP kernel launch (or runtime call) failed";
        case SDMI_ERR_TOO_LARGE: return "problem exceeds the fixed scratch of the no-allocation entry point";
        default: return "unknown sdmi status";
    }
}

extern "C" int sdmi_abi_version(void) { return 13; }
