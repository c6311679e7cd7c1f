// last_arriver_probe.hip -- calibration only (not part of the product): can the split-K reduction of G1-o / G1-down move into the
// PRODUCER's tail, so that the F1r stage (residual add + row statistics, its own graph node: 2.3 us of node boundary + one cold
// cross-XCD round trip) disappears?  VERDICT r2 "next #3": the experiment DESIGN.md 4.6 item 6 stopped short of -- per-access
// device-coherent (sc1) stores / loads for the exchanged planes instead of an L2 write-back + invalidate.
//
// Geometry of the real launches (Lumina-7B, 32 window rows): o = 22 column groups x 8 K-chunks (6 waves, 32 records per wave),
// down = 16 column groups x 13 K-chunks (8 waves, 56 records per wave); every workgroup streams its weight records like G1
// (1-KiB records, 8 in flight per wave, non-temporal) and owns a [32, 32 * waves] fp32 partial tile.
//   two_nodes     stream + plain partial stores | F1r-like kernel sums the chunk planes in order, adds the residual, writes h + row sums
//   last_arriver  partial stores sc1 -> vmcnt(0) -> device-scope ticket per column group; the LAST workgroup of a group loads all planes
//                 (sc1), sums them in chunk order, adds the residual, writes h + row sums, resets the ticket (no spin anywhere)
//   distributed   the same ticket, but every workgroup of the group waits (spins, bounded) for the count and reduces 32 / C rows each
// Every variant is a chain of `layers` dependent pairs in one hipGraph, weights cycling through > 256 MiB so they stream from HBM;
// values depend on the layer so that a stale (non-coherent) read cannot pass the check.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/last_arriver_probe tools/last_arriver_probe.hip ;  run: tools/last_arriver_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#ifndef VEC
#define VEC 1
#endif

__device__ __forceinline__ void st_sc1(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// row statistics stand-in: one atomic per wave (the product reduces inside the workgroup; same bytes out, no contention)
__device__ __forceinline__ void wave_add(float *dst, float v)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(dst, v);
}

// the stream of one G1 workgroup; returns a value that depends on the data only through a never-true comparison (keeps the loads alive)
__device__ __forceinline__ unsigned stream_records(const u32x4 *__restrict__ w, int recs)
{
    u32x4 acc = {0u, 0u, 0u, 0u};
    int s = 0;
    for (; s + 8 <= recs; s += 8) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(w + (size_t)(s + u) * 64);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    for (; s < recs; ++s) acc ^= __builtin_nontemporal_load(w + (size_t)s * 64);
    return ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && acc.x == 0x7f4a7c15u) ? 1u : 0u;
}

// partial value of (layer, chunk, row, col): exactly representable, sum over chunks in order is exact in fp32
__device__ __host__ __forceinline__ float pval(int layer, int chunk, int row, int col) { return (float)(((layer * 7 + chunk * 3 + row + col) & 63) - 31); }

// MODE 0: plain stores, no reduction (two_nodes producer).  1: last arriver.  2: distributed.
template <int MODE>
__global__ __launch_bounds__(512) void k_producer(const u32x4 *__restrict__ w, int recs, float *__restrict__ part, float *__restrict__ h,
                                                  float *__restrict__ rowsum, unsigned *__restrict__ ticket, int N, int C, int layer, int *err)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int g = blockIdx.x, c = blockIdx.y, G = gridDim.x;
    const int cols = 32 * waves, col0 = g * cols;
    const unsigned junk = stream_records(w + ((size_t)(c * G + g) * waves + wv) * recs * 64 + lane, recs);
    // D layout of the 32x32x16 MFMA: reg r of lane l -> row (r&3) + 8*(r>>2) + 4*(l>>5), column 32*wv + (l&31)
    const int col = col0 + 32 * wv + (lane & 31);
    if (col < N)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float *p = part + ((size_t)c * 32 + m) * N + col;
            const float v = pval(layer, c, m, col) + (float)junk;
            if (MODE == 0) *p = v; else st_sc1(p, v);
        }
    if (MODE == 0) return;
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): this thread's write-through stores are acknowledged
    __syncthreads();
    __shared__ unsigned s_ticket;
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(ticket + 32 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    int r_lo, r_hi;
    if (MODE == 1) {
        if (s_ticket != (unsigned)(C - 1)) return;      // not the last of its column group: done
        r_lo = 0; r_hi = 32;
    } else {
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(ticket + 32 * g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % (unsigned)C != 0u && spins < (1 << 20)) ++spins;
            if (spins >= (1 << 20)) *err = 1;
        }
        __syncthreads();
        const int per = (32 + C - 1) / C;
        r_lo = c * per; r_hi = min(32, r_lo + per);
    }
    // reduce rows [r_lo, r_hi) x cols of this group: chunk planes in order, + residual, write h, row sums of h^2 (one atomic-free slot
    // per (group, row): rowsum[g][row])
    float sq = 0.f;
#if VEC
    // four columns per thread, the C chunk planes of a quad requested back to back with 16-byte device-coherent loads, summed in order
    const int qpr = cols / 4, n_q = (r_hi - r_lo) * qpr;
    for (int e = threadIdx.x; e < n_q; e += blockDim.x) {
        const int m = r_lo + e / qpr, cc = col0 + 4 * (e % qpr);
        if (cc >= N) continue;
        typedef __attribute__((ext_vector_type(4))) float f4;
        f4 v[13];
#pragma unroll
        for (int ch = 0; ch < 13; ++ch)
            if (ch < C) __asm__ volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[ch]) : "v"(part + ((size_t)ch * 32 + m) * N + cc) : "memory");
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < 13; ++ch)
            if (ch < C) s4 += v[ch];
        f4 hv = *reinterpret_cast<f4 *>(h + (size_t)m * N + cc) + s4;
        *reinterpret_cast<f4 *>(h + (size_t)m * N + cc) = hv;
        sq += hv.x * hv.x + hv.y * hv.y + hv.z * hv.z + hv.w * hv.w;
    }
#else
    const int n_el = (r_hi - r_lo) * cols;
    for (int e = threadIdx.x; e < n_el; e += blockDim.x) {
        const int m = r_lo + e / cols, cc = col0 + e % cols;
        if (cc >= N) continue;
        float s = 0.f;
        for (int ch = 0; ch < C; ++ch) s += ld_sc1(part + ((size_t)ch * 32 + m) * N + cc);
        const float hv = h[(size_t)m * N + cc] + s;
        h[(size_t)m * N + cc] = hv;
        sq += hv * hv;
    }
#endif
    wave_add(rowsum + g * 32 + (r_lo & 31), sq);
    if (MODE == 1 && threadIdx.x == 0) __hip_atomic_store(ticket + 32 * g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the separate reduction stage of two_nodes (what F1r is): one thread per column and 4-row group, chunk planes summed in order
__global__ __launch_bounds__(256) void k_f1r(const float *__restrict__ part, float *__restrict__ h, float *__restrict__ rowsum, int N, int C)
{
    const int col = blockIdx.x * 256 + threadIdx.x, rg = blockIdx.y;
    float sq = 0.f;
    if (col < N)
    for (int m = 4 * rg; m < 4 * rg + 4; ++m) {
        float s = 0.f;
        for (int ch = 0; ch < C; ++ch) s += part[((size_t)ch * 32 + m) * N + col];
        const float hv = h[(size_t)m * N + col] + s;
        h[(size_t)m * N + col] = hv;
        sq += hv * hv;
    }
    wave_add(rowsum + (col / 512) * 32 + 4 * rg, sq);
}

struct Geo { const char *name; int G, C, waves, recs; };

int main()
{
    hipStream_t s; hipStreamCreate(&s);
    const int N = 4096, layers = 32;
    const Geo geos[2] = {{"o (22 x 8 workgroups, 6 waves, 32 records / wave, 34 MB)", 22, 8, 6, 32}, {"down (16 x 13 workgroups, 8 waves, 56 records / wave, 90 MB)", 16, 13, 8, 56}};
    float *part, *h, *rowsum; unsigned *ticket; int *err;
    hipMalloc(&part, (size_t)13 * 32 * N * 4); hipMalloc(&h, (size_t)32 * N * 4); hipMalloc(&rowsum, 64 * 32 * 4); hipMalloc(&ticket, 64 * 32 * 4); hipMalloc(&err, 4);
    hipMemset(err, 0, 4);
    for (const Geo &ge : geos) {
        const size_t wrec = (size_t)ge.G * ge.C * ge.waves * ge.recs * 64;         // u32x4 per layer
        const int wl = (int)((300u << 20) / (wrec * 16) + 1);                      // distinct weight buffers: > 256 MiB in rotation
        u32x4 *w; hipMalloc(&w, wrec * 16 * wl); hipMemset(w, 1, wrec * 16 * wl);
        float us[3]; int bad[3];
        for (int mode = 0; mode < 3; ++mode) {
            hipMemset(h, 0, (size_t)32 * N * 4); hipMemset(ticket, 0, 64 * 32 * 4); hipMemset(rowsum, 0, 64 * 32 * 4);
            hipGraph_t g; hipGraphExec_t gx;
            hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
            for (int l = 0; l < layers; ++l) {
                const u32x4 *wp = w + (size_t)(l % wl) * wrec;
                dim3 grid(ge.G, ge.C), blk(64 * ge.waves);
                if (mode == 0) {
                    hipLaunchKernelGGL(k_producer<0>, grid, blk, 0, s, wp, ge.recs, part, h, rowsum, ticket, N, ge.C, l, err);
                    hipLaunchKernelGGL(k_f1r, dim3((N + 255) / 256, 8), dim3(256), 0, s, part, h, rowsum, N, ge.C);
                } else if (mode == 1) hipLaunchKernelGGL(k_producer<1>, grid, blk, 0, s, wp, ge.recs, part, h, rowsum, ticket, N, ge.C, l, err);
                else hipLaunchKernelGGL(k_producer<2>, grid, blk, 0, s, wp, ge.recs, part, h, rowsum, ticket, N, ge.C, l, err);
            }
            hipStreamEndCapture(s, &g);
            hipGraphInstantiate(&gx, g, nullptr, nullptr, 0);
            hipGraphLaunch(gx, s); hipStreamSynchronize(s);
            // correctness of the first replay: h = sum over layers and chunks of pval (exact in fp32)
            std::vector<float> hh((size_t)32 * N);
            hipMemcpy(hh.data(), h, hh.size() * 4, hipMemcpyDeviceToHost);
            bad[mode] = 0;
            for (int m = 0; m < 32; ++m)
                for (int c = 0; c < N; c += 37) {
                    float want = 0.f;
                    for (int l = 0; l < layers; ++l) for (int ch = 0; ch < ge.C; ++ch) want += pval(l, ch, m, c);
                    if (hh[(size_t)m * N + c] != want) ++bad[mode];
                }
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e9f;
            for (int r = 0; r < 8; ++r) {
                hipEventRecord(e0, s); hipGraphLaunch(gx, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            us[mode] = best * 1e3f / layers;
            hipGraphExecDestroy(gx); hipGraphDestroy(g);
        }
        int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
        printf("{\"projection\": \"%s\", \"two_graph_nodes_us\": %.2f, \"last_arriver_sc1_us\": %.2f, \"distributed_sc1_us\": %.2f, "
               "\"wrong_elements\": [%d, %d, %d], \"spin_timeouts\": %d}\n", ge.name, us[0], us[1], us[2], bad[0], bad[1], bad[2], herr);
        hipFree(w);
    }
    return 0;
}
