// sjd_capi.hip -- version / error strings of libsjd_hip.so (the kernels' entry points live next to the kernels).
#include <hip/hip_runtime.h>

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

// The per-iteration control traffic of the decode loop without a framework in between (a torch `copy_` costs the host 6-8 us per call,
// these 2-3): the blob upload that opens an iteration and the stream wait that closes it.
extern "C" int sjd_upload_async(void *dst_device, const void *src_pinned_host, int64_t bytes, void *stream)
{
    if (!dst_device || !src_pinned_host || bytes < 0) return SJD_ERR_BAD_ARG;
    if (bytes == 0) return SJD_OK;
    return hipMemcpyAsync(dst_device, src_pinned_host, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream) == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH;
}

extern "C" int sjd_stream_synchronize(void *stream)
{
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH;
}
