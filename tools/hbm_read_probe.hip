// hbm_read_probe.hip -- calibration only (not part of the product): what a read-only 16-byte-per-lane stream reaches on
// this box, as a function of access pattern, loads in flight per lane, workgroup size and grid size.  Sets the practical
// ceiling kernel G1 (weight stream) is priced against next to the 8 TB/s datasheet peak.
//   build: hipcc -O3 --offload-arch=gfx950 -o tools/hbm_read_probe tools/hbm_read_probe.hip
//   run:   tools/hbm_read_probe [MB per launch = 180]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// pattern 0: grid-stride (the chip sweeps the buffer like a linear copy)
// pattern 1: every wave owns one contiguous run (G1 tile-major)
template <int U, bool NT>
__global__ void probe(const u32x4 *__restrict__ p, size_t n_vec, u32x4 *sink, int pattern)
{
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 acc = {0u, 0u, 0u, 0u};
    if (pattern == 0) {
        for (size_t i = tid; i + (U - 1) * nthreads < n_vec; i += U * nthreads) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * nthreads) : p[i + u * nthreads];
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
        }
    } else if (pattern == 2 || pattern == 3) {
        // G1's weight addressing: [2 chunks][steps][n_tiles] 1-KiB records (2: step-major, 3: tile-major), one tile per wave
        const size_t nwaves = nthreads >> 6, wave = tid >> 6, lane = tid & 63;
        const size_t n_tiles = nwaves / 2, chunk = wave / n_tiles, t = wave % n_tiles;
        const size_t steps = ((n_vec / 64) / nwaves) & ~(size_t)(U - 1);
        const u32x4 *q = p + (chunk * n_tiles * steps + (pattern == 2 ? t : t * steps)) * 64 + lane;
        const size_t rs = pattern == 2 ? n_tiles * 64 : 64;
        for (size_t i = 0; i < steps; i += U) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(q + (i + u) * rs) : q[(i + u) * rs];
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
        }
    } else {
        const size_t nwaves = nthreads >> 6, wave = tid >> 6, lane = tid & 63;
        const size_t per_wave = (n_vec / nwaves) & ~(size_t)(64 * U - 1);
        const u32x4 *q = p + wave * per_wave + lane;
        for (size_t i = 0; i < per_wave; i += 64 * U) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(q + i + 64 * u) : q[i + 64 * u];
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[tid & 63] = acc;
}

// Build-up towards G1 from the pure read: pattern-2 addressing (one tile per wave, step-major records), 8 records per group with the
// next group in flight, and per record   mode 1: an MFMA 32x32x16 with a register A operand   mode 2: the A operand read from LDS
// (ds_read_b128, as G1 does)   mode 3: + the fp32 tile written out at the end.
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int MODE>
__global__ __launch_bounds__(512) void g1like(const u32x4 *__restrict__ p, size_t n_vec, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    const size_t nthreads = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nwaves = nthreads >> 6, wave = tid >> 6, lane = tid & 63;
    const size_t n_tiles = nwaves / 2, chunk = wave / n_tiles, t = wave % n_tiles;
    const int steps = (int)(((n_vec / 64) / nwaves) & ~(size_t)7);
    const u32x4 *q = p + (chunk * n_tiles * steps + t) * 64 + lane;
    const size_t rs = n_tiles * 64;
    if (MODE >= 2) {
        for (int i = threadIdx.x; i < steps * 64; i += blockDim.x) lds[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
        __syncthreads();
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    u32x4 cur[8], nxt[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) cur[u] = __builtin_nontemporal_load(q + (size_t)u * rs);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const u32x4 areg = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    for (int g = 0; g < steps / 8; ++g) {
        const bool more = g + 1 < steps / 8;
        if (more) {
#pragma unroll
            for (int u = 0; u < 8; ++u) nxt[u] = __builtin_nontemporal_load(q + (size_t)((g + 1) * 8 + u) * rs);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const u32x4 a = MODE >= 2 ? lds[(g * 8 + u) * 64 + lane] : areg;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, cur[u]), acc, 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
        }
    }
    if (MODE >= 3) {
        float *o = sink + ((size_t)chunk * 32) * (n_tiles * 32) + t * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * (n_tiles * 32)] = acc[r];
    } else if (acc[0] == 1.2345e-30f) sink[tid & 63] = acc[1];
}

template <int MODE>
static float run_g1like(const char *base, size_t total, size_t bytes, float *sink, int blocks, int threads, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t slots = total / bytes, n_vec = bytes / 16, nwaves = (size_t)blocks * threads / 64;
    const int steps = (int)(((n_vec / 64) / nwaves) & ~(size_t)7);
    const size_t lds = MODE >= 2 ? (size_t)steps * 1024 : 0;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void *)g1like<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((g1like<MODE>), dim3(blocks), dim3(threads), lds, 0, (const u32x4 *)(base + (r % slots) * bytes), n_vec, sink);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((g1like<MODE>), dim3(blocks), dim3(threads), lds, 0, (const u32x4 *)(base + (r % slots) * bytes), n_vec, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (float)((size_t)steps * 1024 * nwaves / 1e9 / (ms / reps));
}

template <int U, bool NT>
static float run(const char *base, size_t total, size_t bytes, u32x4 *sink, int pattern, int blocks, int threads, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t slots = total / bytes;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((probe<U, NT>), dim3(blocks), dim3(threads), 0, 0, (const u32x4 *)(base + (r % slots) * bytes), bytes / 16, sink, pattern);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<U, NT>), dim3(blocks), dim3(threads), 0, 0, (const u32x4 *)(base + (r % slots) * bytes), bytes / 16, sink, pattern);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    // bytes one launch really reads (the loops drop the ragged remainder)
    const size_t n_vec = bytes / 16, nthreads = (size_t)blocks * threads;
    size_t vecs;
    if (pattern == 0) {
        size_t iters = 0;
        for (size_t i = 0; i + (U - 1) * nthreads < n_vec + 0 && i + (U - 1) * nthreads + nthreads - 1 < n_vec + nthreads; i += U * nthreads) ++iters;
        vecs = iters * U * nthreads;
        if (vecs > n_vec) vecs = n_vec;
    } else if (pattern >= 2) {
        vecs = ((((n_vec / 64) / (nthreads >> 6)) & ~(size_t)(U - 1)) * 64) * (nthreads >> 6);
    } else {
        vecs = ((n_vec / (nthreads >> 6)) & ~(size_t)(64 * U - 1)) * (nthreads >> 6);
    }
    return vecs == 0 ? 0.0f : (float)(vecs * 16 / 1e9 / (ms / reps));
}

int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 180;
    const size_t bytes = mb << 20, total = (size_t)3 << 30;
    char *buf;
    u32x4 *sink;
    if (hipMalloc(&buf, total) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, total);
    hipDeviceSynchronize();
    const int reps = 48;
    if (argc > 4 && atoi(argv[2]) == 9) {      // G1 build-up: MB 9 blocks threads
        const int g = atoi(argv[3]), t = atoi(argv[4]);
        float *fs;
        hipMalloc(&fs, (size_t)64 << 20);
        printf("{\"MB\": %zu, \"g1like\": true, \"blocks\": %d, \"threads\": %d, \"TBps_mfma_regA\": %.3f, \"TBps_mfma_ldsA\": %.3f, \"TBps_mfma_ldsA_store\": %.3f}\n",
               mb, g, t, run_g1like<1>(buf, total, bytes, fs, g, t, reps), run_g1like<2>(buf, total, bytes, fs, g, t, reps),
               run_g1like<3>(buf, total, bytes, fs, g, t, reps));
        return 0;
    }
    if (argc > 4) {      // one configuration: MB pattern blocks threads
        const int pattern = atoi(argv[2]), g = atoi(argv[3]), t = atoi(argv[4]);
        printf("{\"MB\": %zu, \"pattern\": %d, \"blocks\": %d, \"threads\": %d, \"TBps_u4\": %.3f, \"TBps_u8\": %.3f, \"TBps_u16\": %.3f, \"TBps_u8_temporal\": %.3f}\n", mb, pattern, g, t,
               run<4, true>(buf, total, bytes, sink, pattern, g, t, reps), run<8, true>(buf, total, bytes, sink, pattern, g, t, reps),
               run<16, true>(buf, total, bytes, sink, pattern, g, t, reps), run<8, false>(buf, total, bytes, sink, pattern, g, t, reps));
        return 0;
    }
    printf("{\"MB\": %zu, \"rows\": [\n", mb);
    const int grids[] = {256, 512, 1024, 2048, 4096};
    const int tpbs[] = {256, 512, 1024};
    bool first = true;
    for (int pattern = 0; pattern < 2; ++pattern)
        for (int gi = 0; gi < 5; ++gi)
            for (int ti = 0; ti < 3; ++ti) {
                const int g = grids[gi], t = tpbs[ti];
                if ((size_t)g * t > 2048 * 1024) continue;
                float a = run<4, true>(buf, total, bytes, sink, pattern, g, t, reps);
                float b = run<8, true>(buf, total, bytes, sink, pattern, g, t, reps);
                float c = run<16, true>(buf, total, bytes, sink, pattern, g, t, reps);
                float d = run<8, false>(buf, total, bytes, sink, pattern, g, t, reps);
                printf("%s {\"pattern\": %d, \"blocks\": %d, \"threads\": %d, \"TBps_u4\": %.3f, \"TBps_u8\": %.3f, \"TBps_u16\": %.3f, \"TBps_u8_temporal\": %.3f}",
                       first ? "" : ",\n", pattern, g, t, a, b, c, d);
                first = false;
            }
    printf("\n]}\n");
    return 0;
}
