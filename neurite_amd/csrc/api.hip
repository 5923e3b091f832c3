// Library-level entry points of the C ABI (include/neurite_amd.h).
#include "nrt_common.h"

extern "C" const char *nrt_status_string(int status) {
    switch (status) {
        case NRT_OK: return "ok";
        case NRT_ERR_INVALID_ARG: return "invalid argument";
        case NRT_ERR_UNSUPPORTED: return "combination not supported by the HIP path";
        case NRT_ERR_LAUNCH: return "HIP kernel launch failed";
        case NRT_ERR_WORKSPACE: return "workspace missing or too small";
        default: return "unknown status";
    }
}

extern "C" int nrt_abi_version(void) { return 1; }

extern "C" const char *nrt_target_arch(void) { return "gfx950"; }
