// sjd_capi.hip -- version / error strings of libsjd_hip.so (the kernels' entry points live next to the kernels).
#include <hip/hip_runtime.h>
#include <time.h>

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

// Spin on a word of pinned host memory a kernel writes (the sequence flag behind K4's state mirror).  No HIP call: the runtime's stream
// wait falls back to an interrupt after a few microseconds of polling, and an SJD iteration is 3.6 ms long.
extern "C" int sjd_host_wait_u64(const volatile uint64_t *flag, uint64_t value, int64_t timeout_us)
{
    if (!flag) return SJD_ERR_BAD_ARG;
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == value) return SJD_OK;
        __builtin_ia32_pause();
        if ((spins & 1023u) == 1023u && timeout_us >= 0) {
            struct timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const int64_t us = (int64_t)(t1.tv_sec - t0.tv_sec) * 1000000 + (t1.tv_nsec - t0.tv_nsec) / 1000;
            if (us > timeout_us) return SJD_ERR_LAUNCH;
        }
    }
}
