// grid_barrier_probe.hip -- calibration only (not part of the product): what a dependent stage costs INSIDE one persistent kernel
// (one 512-thread workgroup per CU, a device-scope counter barrier between stages) compared with the same stage as a separate hipGraph
// node (tools/stage_floor_probe.hip: 1.6 us empty, 2.3 us with 1 MiB of cross-XCD data), and whether weight records requested BEFORE
// the barrier keep streaming while the workgroup waits.
//   barrier_only      N back-to-back barriers
//   barrier_copy      barrier + every thread loads 16 B another XCD's workgroup wrote in the previous stage and stores 16 B
//   stream            every wave streams `recs` 1-KiB records (G1's access pattern), barrier, copy stage, barrier, streams again ...
//   stream_prefetch   the same, but the first `pre` records of the NEXT stream phase are requested before the barrier
// build: hipcc -O3 --offload-arch=gfx950 -o tools/grid_barrier_probe tools/grid_barrier_probe.hip ;  run: tools/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// FENCE 0: no fences (raw arrive + poll), 1: every thread fences, 2: thread 0 fences (the others wait for their own stores only)
#ifndef FENCE
#define FENCE 2
#endif
#ifndef SLEEP
#define SLEEP 0
#endif
// BAR 0: one counter, everybody polls it.  1: one counter, the last arriver publishes the generation in a flag on another line, everybody polls
// the flag.  2: per-XCD counters (workgroup id % 8), the last of each XCD arrives at the device counter, the last of those publishes.
#ifndef BAR
#define BAR 0
#endif
__device__ __forceinline__ bool grid_barrier(unsigned *ctr, unsigned target)
{
    if (FENCE == 1) __threadfence();
    else __builtin_amdgcn_s_waitcnt(0x0F70);          // this thread's stores have been acknowledged by L2 (vmcnt(0); gfx9 counts stores there)
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        if (FENCE == 2) __threadfence();             // release: write the XCD's L2 back
        const unsigned nwg = gridDim.x, gen = target / nwg;
        int spins = 0;
        if (BAR == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && spins < (1 << 22)) { ++spins; if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); }
        } else {
            unsigned *flag = ctr + 64;
            bool last;
            if (BAR == 1) last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == target - 1;
            else {
                const unsigned x = blockIdx.x & 7, per = nwg / 8;       // (probe: nwg is a multiple of 8)
                last = false;
                if (__hip_atomic_fetch_add(ctr + 128 + 32 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen * per - 1)
                    last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen * 8 - 1;
            }
            if (last) __hip_atomic_store(flag, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen && spins < (1 << 22)) { ++spins; if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); }
        }
        ok = spins < (1 << 22);
        if (FENCE == 2) __threadfence();             // acquire: drop stale lines
    }
    __syncthreads();
    if (FENCE == 1) __threadfence();
    return ok != 0;
}

// mode 0: barriers only; 1: barrier + cross-XCD copy
__global__ __launch_bounds__(512) void k_barriers(unsigned *ctr, u32x4 *a, u32x4 *b, int n, int mode, int *err)
{
    const unsigned nwg = gridDim.x;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t j = (size_t)((blockIdx.x + 3) % nwg) * blockDim.x + threadIdx.x;
    for (int it = 0; it < n; ++it) {
        if (mode == 1) { u32x4 *src = (it & 1) ? b : a, *dst = (it & 1) ? a : b; u32x4 v = src[j]; v += 1u; dst[i] = v; }
        if (!grid_barrier(ctr, (unsigned)(it + 1) * nwg)) { if (threadIdx.x == 0) *err = 1; return; }
    }
}

// G1-like stream: wave w of workgroup g reads records [base, base + recs) of its own run, 8 loads in flight.
template <int PRE>
__global__ __launch_bounds__(512) void k_stream(unsigned *ctr, const u32x4 *__restrict__ w0, const u32x4 *__restrict__ w1, u32x4 *a, u32x4 *b,
                                                int recs, int n, int *err, unsigned *sink)
{
    const unsigned nwg = gridDim.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t run = ((size_t)blockIdx.x * 8 + wv) * recs * 64 + lane;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t j = (size_t)((blockIdx.x + 3) % nwg) * blockDim.x + threadIdx.x;
    u32x4 acc = {0u, 0u, 0u, 0u};
    u32x4 pre[PRE > 0 ? PRE : 1];
    unsigned bar = 0;
    for (int it = 0; it < n; ++it) {
        const u32x4 *w = ((it & 1) ? w1 : w0) + run;
        int s = 0;
        if (PRE > 0 && it > 0) {
#pragma unroll
            for (int u = 0; u < PRE; ++u) acc ^= pre[u];
            s = PRE;
        }
        for (; s + 8 <= recs; s += 8) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(w + (size_t)(s + u) * 64);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u];
        }
        for (; s < recs; ++s) acc ^= __builtin_nontemporal_load(w + (size_t)s * 64);
        // the small dependent stage between two projections: barrier, cross-XCD copy, barrier
        a[i] = acc;
        if (PRE > 0 && it + 1 < n) {
            const u32x4 *wn = ((it & 1) ? w0 : w1) + run;
#pragma unroll
            for (int u = 0; u < PRE; ++u) pre[u] = __builtin_nontemporal_load(wn + (size_t)u * 64);
        }
        if (!grid_barrier(ctr, ++bar * nwg)) { if (threadIdx.x == 0) *err = 1; return; }
        if (wv < 4 || PRE == 0) { u32x4 v = a[j]; v += 1u; b[i] = v; }          // with prefetch in flight only waves 0-3 would be free
        if (!grid_barrier(ctr, ++bar * nwg)) { if (threadIdx.x == 0) *err = 1; return; }
        acc ^= b[j];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && acc.x == 0x7f4a7c15u) sink[0] = acc.x;
}

// the same two stages as separate kernels (hipGraph chain): stream -> cross-XCD copy
__global__ __launch_bounds__(512) void k_stream_stage(const u32x4 *__restrict__ w0, u32x4 *a, const u32x4 *b, int recs, unsigned *sink)
{
    const unsigned nwg = gridDim.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32x4 *w = w0 + ((size_t)blockIdx.x * 8 + wv) * recs * 64 + lane;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t j = (size_t)((blockIdx.x + 3) % nwg) * blockDim.x + threadIdx.x;
    u32x4 acc = b[j];
    int s = 0;
    for (; s + 8 <= recs; s += 8) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(w + (size_t)(s + u) * 64);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    for (; s < recs; ++s) acc ^= __builtin_nontemporal_load(w + (size_t)s * 64);
    a[i] = acc;
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && acc.x == 0x7f4a7c15u) sink[0] = acc.x;
}
__global__ __launch_bounds__(512) void k_copy_stage(const u32x4 *a, u32x4 *b)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, j = (size_t)((blockIdx.x + 3) % gridDim.x) * blockDim.x + threadIdx.x;
    u32x4 v = a[j]; v += 1u; b[i] = v;
}

template <typename F> static float time_graph(hipStream_t s, int n, F launch)
{
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch(i);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return best * 1e3f / n;
}

template <typename F> static float time_it(hipStream_t s, unsigned *ctr, F launch)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        hipMemsetAsync(ctr, 0, 4096, s);
        hipEventRecord(e0, s); launch(); hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best * 1e3f;
}

int main()
{
    hipStream_t s; hipStreamCreate(&s);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int nwg = p.multiProcessorCount < 256 ? p.multiProcessorCount : 256;
    unsigned *ctr, *sink; int *err; u32x4 *a, *b, *w0, *w1;
    hipMalloc(&ctr, 4096); hipMalloc(&sink, 64); hipMalloc(&err, 4); hipMemset(err, 0, 4);
    hipMalloc(&a, (size_t)nwg * 512 * 16); hipMalloc(&b, (size_t)nwg * 512 * 16);
    hipMemset(a, 0, (size_t)nwg * 512 * 16); hipMemset(b, 0, (size_t)nwg * 512 * 16);
    const int max_recs = 96;                                    // 96 KiB per wave -> 192 MiB per phase
    const size_t wbytes = (size_t)nwg * 8 * max_recs * 1024;
    hipMalloc(&w0, wbytes); hipMalloc(&w1, wbytes); hipMemset(w0, 1, wbytes); hipMemset(w1, 2, wbytes);
    const int N = 200;
    float t_empty = time_it(s, ctr, [&] { hipLaunchKernelGGL(k_barriers, dim3(nwg), dim3(512), 0, s, ctr, a, b, 0, 0, err); });
    float t_bar = time_it(s, ctr, [&] { hipLaunchKernelGGL(k_barriers, dim3(nwg), dim3(512), 0, s, ctr, a, b, N, 0, err); });
    float t_copy = time_it(s, ctr, [&] { hipLaunchKernelGGL(k_barriers, dim3(nwg), dim3(512), 0, s, ctr, a, b, N, 1, err); });
    printf("{\"barrier\": %d, \"fence\": %d, \"sleep\": %d, \"workgroups\": %d, \"launch_us\": %.2f, \"barrier_only_us\": %.3f, \"barrier_plus_cross_xcd_copy_us\": %.3f}\n", BAR, FENCE, SLEEP, nwg, t_empty,
           (t_bar - t_empty) / N, (t_copy - t_empty) / N);
    if (getenv("BARRIER_ONLY")) return 0;
    const int n = 40;
    for (int recs : {24, 44, 56, 96}) {
        const double mb = (double)nwg * 8 * recs * 1024 / 1e6;
        float t0 = time_it(s, ctr, [&] { hipLaunchKernelGGL(k_stream<0>, dim3(nwg), dim3(512), 0, s, ctr, w0, w1, a, b, recs, n, err, sink); });
        float t8 = time_it(s, ctr, [&] { hipLaunchKernelGGL(k_stream<8>, dim3(nwg), dim3(512), 0, s, ctr, w0, w1, a, b, recs, n, err, sink); });
        float t16 = time_it(s, ctr, [&] { hipLaunchKernelGGL(k_stream<16>, dim3(nwg), dim3(512), 0, s, ctr, w0, w1, a, b, recs, n, err, sink); });
        float t24 = time_it(s, ctr, [&] { hipLaunchKernelGGL(k_stream<24>, dim3(nwg), dim3(512), 0, s, ctr, w0, w1, a, b, recs, n, err, sink); });
        float tg = time_graph(s, n, [&](int it) {
            hipLaunchKernelGGL(k_stream_stage, dim3(nwg), dim3(512), 0, s, (it & 1) ? w1 : w0, a, b, recs, sink);
            hipLaunchKernelGGL(k_copy_stage, dim3(nwg), dim3(512), 0, s, a, b); });
        printf("{\"phase_MB\": %.1f, \"recs_per_wave\": %d, \"two_graph_nodes_us\": %.2f, \"us_per_phase_no_prefetch\": %.2f, \"prefetch8\": %.2f, \"prefetch16\": %.2f, \"prefetch24\": %.2f, "
               "\"stream_only_at_6.3TBps_us\": %.2f}\n", mb, recs, tg, t0 / n, t8 / n, t16 / n, t24 / n, mb / 6.3);
    }
    int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    printf("{\"barrier_timeouts\": %d}\n", herr);
    return herr;
}
