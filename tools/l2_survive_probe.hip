// l2_survive_probe.hip -- calibration only (not part of the product).  Question (round 5): does a line that kernel A READ into an XCD's L2
// survive the kernel boundary, i.e. does a dependent kernel B whose workgroup b (same linear id -> same XCD) reads the same slice hit in L2?
// If it does, a latency-bound glue kernel can pull the head of the next projection's weight stream into L2 while HBM idles.
//   cold   : after a 1 GiB flush read                         -> HBM
//   same   : B reads what A's workgroup of the same id read   -> L2 if lines survive the boundary, else Infinity Cache
//   shift  : B's workgroup b reads the slice of workgroup b+1 -> another XCD's slice: Infinity Cache at best
// Times are in-kernel: max(end) - min(start) over all workgroups on the 100 MHz wall clock.
//   build: hipcc -O3 --offload-arch=gfx950 -o tools/l2_survive_probe tools/l2_survive_probe.hip
//   run:   tools/l2_survive_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ void flush_k(const u32x4 *__restrict__ p, size_t n_vec, u32x4 *sink)
{
    const size_t nthreads = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (size_t i = tid; i < n_vec; i += nthreads) acc ^= p[i];
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[tid & 63] = acc;
}

__global__ void reset_k(unsigned long long *t) { t[0] = ~0ull; t[1] = 0ull; }

// workgroup b reads slice (b + shift) % grid: `per_wg` bytes, contiguous, 16 B per lane, U loads in flight per lane
template <int U, bool NT>
__global__ __launch_bounds__(512) void reader(const u32x4 *__restrict__ p, size_t vec_per_wg, int shift, u32x4 *sink, unsigned long long *t)
{
    const unsigned long long t0 = wall_clock64();
    const size_t slice = ((size_t)blockIdx.x + shift) % gridDim.x;
    const u32x4 *q = p + slice * vec_per_wg + threadIdx.x;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (size_t i = 0; i + (U - 1) * blockDim.x < vec_per_wg; i += (size_t)U * blockDim.x) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(q + i + u * blockDim.x) : q[i + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[threadIdx.x & 63] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&t[0], t0);
        atomicMax(&t[1], wall_clock64());
    }
}

// a small dependent kernel between A and B (what a glue launch is to the next projection)
__global__ void small_k(float *x) { x[blockIdx.x * blockDim.x + threadIdx.x] += 1.0f; }

static unsigned long long *g_t;
static u32x4 *g_sink;

template <bool NT>
static double timed_reader(const char *buf, size_t bytes, int grid, int shift)
{
    hipLaunchKernelGGL(reset_k, dim3(1), dim3(1), 0, 0, g_t);
    hipLaunchKernelGGL((reader<8, NT>), dim3(grid), dim3(512), 0, 0, (const u32x4 *)buf, bytes / 16 / grid, shift, g_sink, g_t);
    unsigned long long h[2];
    hipMemcpy(h, g_t, 16, hipMemcpyDeviceToHost);
    return (double)(h[1] - h[0]) * 0.01;       // us
}

int main(int argc, char **argv)
{
    const size_t big = (size_t)1 << 30;
    char *flushbuf, *buf;
    float *small;
    if (hipMalloc(&flushbuf, big) != hipSuccess || hipMalloc(&buf, (size_t)256 << 20) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&g_sink, 4096);
    hipMalloc(&g_t, 16);
    hipMalloc(&small, 1 << 20);
    hipMemset(flushbuf, 1, big);
    hipMemset(buf, 2, (size_t)256 << 20);
    hipMemset(small, 0, 1 << 20);
    hipDeviceSynchronize();
    const int mbs[] = {4, 8, 16, 24, 32, 64};
    const int grids[] = {256, 1024};
    printf("[\n");
    bool first = true;
    for (int gi = 0; gi < 2; ++gi)
        for (int mi = 0; mi < 6; ++mi)
            for (int rep = 0; rep < 2; ++rep) {
                const int grid = grids[gi];
                const size_t bytes = (size_t)mbs[mi] << 20;
                auto flush = [&]() { hipLaunchKernelGGL(flush_k, dim3(2048), dim3(256), 0, 0, (const u32x4 *)flushbuf, big / 16, g_sink); };
                // (reset_k sits between A and B in every sequence: itself a dependent launch, as a glue kernel would be)
                flush();
                const double cold = timed_reader<false>(buf, bytes, grid, 0);
                const double same = timed_reader<false>(buf, bytes, grid, 0);
                const double same_nt = timed_reader<true>(buf, bytes, grid, 0);
                const double shift = timed_reader<false>(buf, bytes, grid, 1);
                const double shift8 = timed_reader<false>(buf, bytes, grid, 8);        // another workgroup of the SAME XCD (ids congruent mod 8)
                flush();
                const double cold_nt = timed_reader<true>(buf, bytes, grid, 0);
                const double after_nt = timed_reader<false>(buf, bytes, grid, 0);      // does an nt read leave lines behind?
                flush();
                (void)timed_reader<false>(buf, bytes, grid, 0);
                hipLaunchKernelGGL(small_k, dim3(256), dim3(256), 0, 0, small);
                hipLaunchKernelGGL(small_k, dim3(256), dim3(256), 0, 0, small);
                const double same_after_two = timed_reader<true>(buf, bytes, grid, 0);
                printf("%s {\"MB\": %d, \"grid\": %d, \"cold_us\": %.2f, \"same_us\": %.2f, \"same_nt_us\": %.2f, \"shift1_us\": %.2f, \"shift8_us\": %.2f, "
                       "\"cold_nt_us\": %.2f, \"plain_after_nt_us\": %.2f, \"same_nt_after_two_small_kernels_us\": %.2f}",
                       first ? "" : ",\n", mbs[mi], grid, cold, same, same_nt, shift, shift8, cold_nt, after_nt, same_after_two);
                first = false;
            }
    printf("\n]\n");
    return 0;
}
