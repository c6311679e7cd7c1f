// stage_floor_probe.hip -- calibration only (not part of the product): what ONE dependent stage of the window forward costs inside a
// hipGraph on this box, as a function of what the stage does.  A chain of N kernels, each consuming what the previous one wrote:
//   empty        no memory traffic at all (pure dispatch + barrier between dependent graph nodes)
//   copy         every thread loads 16 B written by the previous kernel and stores 16 B (temporal / non-temporal variants)
//   reduce8      every thread loads 8 x 16 B (split-K partial style: 8 chunk planes) and stores 16 B
// build: hipcc -O3 --offload-arch=gfx950 -o tools/stage_floor_probe tools/stage_floor_probe.hip ;  run: tools/stage_floor_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ void k_empty() {}

template <int LD, int ST, int PLANES>     // LD/ST: 0 temporal, 1 non-temporal
__global__ void k_stage(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t plane, int shuffle = 0)
{
    // shuffle: read the block another workgroup (on another XCD: consecutive block ids go round-robin over the 8 XCDs) wrote
    const unsigned rb = shuffle ? (blockIdx.x + shuffle) % gridDim.x : blockIdx.x;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, j = (size_t)rb * blockDim.x + threadIdx.x;
    (void)i;
    u32x4 acc = {1u, 1u, 1u, 1u};
#pragma unroll
    for (int p = 0; p < PLANES; ++p) {
        const u32x4 v = LD ? __builtin_nontemporal_load(in + p * plane + j) : in[p * plane + j];
        acc += v;
    }
    if (ST) __builtin_nontemporal_store(acc, out + i); else out[i] = acc;
}

template <typename F> static float time_chain(hipStream_t s, int n, F launch)
{
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch(i);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return best * 1e3f / n;
}

int main()
{
    hipStream_t s; hipStreamCreate(&s);
    const int N = 128;
    const size_t max_vec = (size_t)4096 * 256;                 // 16 MB per plane at most
    u32x4 *a, *b; hipMalloc(&a, max_vec * 16 * 8); hipMalloc(&b, max_vec * 16 * 8);
    hipMemset(a, 0, max_vec * 16 * 8); hipMemset(b, 0, max_vec * 16 * 8);
    printf("{\"empty\": {");
    int first = 1;
    for (int grid : {1, 64, 256, 1024, 4096}) for (int block : {64, 256, 1024}) {
        float us = time_chain(s, N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), 0, s); });
        printf("%s\"g%d_b%d\": %.2f", first ? "" : ", ", grid, block, us); first = 0;
    }
    printf("}}\n");
    for (int grid : {32, 256, 1024, 4096}) {
        const int block = 256;
        const size_t plane = (size_t)grid * block;
        float c00 = time_chain(s, N, [&](int i) { hipLaunchKernelGGL((k_stage<0, 0, 1>), dim3(grid), dim3(block), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, plane); });
        float c01 = time_chain(s, N, [&](int i) { hipLaunchKernelGGL((k_stage<0, 1, 1>), dim3(grid), dim3(block), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, plane); });
        float c10 = time_chain(s, N, [&](int i) { hipLaunchKernelGGL((k_stage<1, 0, 1>), dim3(grid), dim3(block), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, plane); });
        float c11 = time_chain(s, N, [&](int i) { hipLaunchKernelGGL((k_stage<1, 1, 1>), dim3(grid), dim3(block), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, plane); });
        float r8 = time_chain(s, N, [&](int i) { hipLaunchKernelGGL((k_stage<0, 0, 8>), dim3(grid), dim3(block), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, plane); });
        float r8nt = time_chain(s, N, [&](int i) { hipLaunchKernelGGL((k_stage<1, 1, 8>), dim3(grid), dim3(block), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, plane); });
        float x1 = time_chain(s, N, [&](int i) { hipLaunchKernelGGL((k_stage<0, 0, 1>), dim3(grid), dim3(block), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, plane, 1); });
        float x3 = time_chain(s, N, [&](int i) { hipLaunchKernelGGL((k_stage<0, 0, 1>), dim3(grid), dim3(block), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, plane, 3); });
        float x8r = time_chain(s, N, [&](int i) { hipLaunchKernelGGL((k_stage<0, 0, 8>), dim3(grid), dim3(block), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, plane, 5); });
        float x1nt = time_chain(s, N, [&](int i) { hipLaunchKernelGGL((k_stage<1, 1, 1>), dim3(grid), dim3(block), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, plane, 1); });
        float same8 = time_chain(s, N, [&](int i) { hipLaunchKernelGGL((k_stage<0, 0, 1>), dim3(grid), dim3(block), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, plane, 8); });
        printf("{\"grid\": %d, \"block\": %d, \"KB_per_plane\": %zu, \"copy_us\": %.2f, \"copy_ntstore_us\": %.2f, \"copy_ntload_us\": %.2f, \"copy_nt_both_us\": %.2f, "
               "\"reduce8_us\": %.2f, \"reduce8_nt_us\": %.2f, \"copy_other_xcd_us\": %.2f, \"copy_other_xcd3_us\": %.2f, \"reduce8_other_xcd_us\": %.2f, "
               "\"copy_other_xcd_nt_us\": %.2f, \"copy_same_xcd_other_wg_us\": %.2f}\n", grid, block, plane * 16 / 1024, c00, c01, c10, c11, r8, r8nt, x1, x3, x8r, x1nt, same8);
    }
    return 0;
}
