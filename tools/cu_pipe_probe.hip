// cu_pipe_probe.hip -- calibration only: does an L2-resident side stream (G1's activation chunk: the same 1-2 MB read by every workgroup)
// cost a CU as much as the HBM weight stream it rides on?  Each wave streams `recs` 1-KiB weight records (non-temporal, 8 in flight) and,
// per 8 weight records, XR records of a small shared buffer (plain loads, L2 hits after the first touch).  G1 at 32 rows: XR = 1 (x bytes =
// 1/8 of the weight bytes per workgroup), at 128 rows: XR = 4.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/cu_pipe_probe tools/cu_pipe_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int XR>
__global__ __launch_bounds__(512) void k(const u32x4 *__restrict__ w, const u32x4 *__restrict__ x, u32x4 *out, int recs, int x_recs, unsigned *sink)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32x4 *wp = w + ((size_t)blockIdx.x * 8 + wv) * recs * 64 + lane;
    u32x4 acc = {0u, 0u, 0u, 0u};
    int xi = (blockIdx.x * 8 + wv) % x_recs;
    for (int s = 0; s + 8 <= recs; s += 8) {
        u32x4 v[8], xv[XR > 0 ? XR : 1];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(wp + (size_t)(s + u) * 64);
#pragma unroll
        for (int u = 0; u < XR; ++u) { xv[u] = x[(size_t)xi * 64 + lane]; xi = xi + 1 < x_recs ? xi + 1 : 0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
#pragma unroll
        for (int u = 0; u < XR; ++u) acc ^= xv[u];
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc;
    if ((acc.x ^ acc.y) == 0x9e3779b9u && acc.z == 0x7f4a7c15u) sink[0] = acc.x;
}

template <int XR> static float run(hipStream_t s, int wgs, int recs, const u32x4 *w0, const u32x4 *w1, const u32x4 *x, u32x4 *out, unsigned *sink)
{
    hipGraph_t g; hipGraphExec_t ge; const int n = 24;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<XR>, dim3(wgs), dim3(512), 0, s, (i & 1) ? w1 : w0, x, out, recs, 1024, sink);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) { hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return best * 1e3f / n;
}

int main()
{
    hipStream_t s; hipStreamCreate(&s);
    const int max_wgs = 256, max_recs = 128;
    const size_t wb = (size_t)max_wgs * 8 * max_recs * 1024;
    u32x4 *w0, *w1, *x, *out; unsigned *sink;
    hipMalloc(&w0, wb); hipMalloc(&w1, wb); hipMalloc(&x, 1024 * 1024); hipMalloc(&out, (size_t)max_wgs * 512 * 16); hipMalloc(&sink, 64);
    hipMemset(w0, 1, wb); hipMemset(w1, 2, wb); hipMemset(x, 3, 1024 * 1024);
    for (int wgs : {172, 208, 240, 256}) for (int recs : {56, 128}) {
        const double mb = (double)wgs * 8 * recs * 1024 / 1e6;
        float t0 = run<0>(s, wgs, recs, w0, w1, x, out, sink), t1 = run<1>(s, wgs, recs, w0, w1, x, out, sink), t2 = run<2>(s, wgs, recs, w0, w1, x, out, sink),
              t4 = run<4>(s, wgs, recs, w0, w1, x, out, sink);
        printf("{\"workgroups\": %d, \"recs_per_wave\": %d, \"weight_MB\": %.1f, \"us_x0\": %.2f, \"us_x1of8\": %.2f, \"us_x2of8\": %.2f, \"us_x4of8\": %.2f, "
               "\"GBps_per_CU_x0\": %.1f, \"GBps_per_CU_x4of8_incl_x\": %.1f}\n", wgs, recs, mb, t0, t1, t2, t4, mb / wgs / t0 * 1e3, mb * 1.5 / wgs / t4 * 1e3);
    }
    return 0;
}
