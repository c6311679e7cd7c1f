// sjd_capi.hip -- version / error strings of libsjd_hip.so (the kernels' entry points live next to the kernels).
#include "../../include/sjd_hip.h"

extern "C" int sjd_version(void) { return SJD_VERSION; }

extern "C" const char *sjd_error_string(int code)
{
    switch (code) {
    case SJD_OK: return "ok";
    case SJD_ERR_BAD_ARG: return "bad argument";
    case SJD_ERR_UNSUPPORTED: return "unsupported configuration";
    case SJD_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown error";
    }
}
