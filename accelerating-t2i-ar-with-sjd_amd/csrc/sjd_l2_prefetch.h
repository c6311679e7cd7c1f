// sjd_l2_prefetch.h -- round 5: the head of the NEXT projection's weight stream pulled into the XCDs' L2 by spare workgroups of a
// latency-bound glue launch (F1r, F2, the K1 combine), so that the G1z / G1sz launch behind it opens on L2 hits while its waves' HBM requests for the
// rest of their units are already out.  This is the run-ahead weight loader of a persistent layer (guide: prefetch-credit) with the L2 as its
// ring and the kernel boundary as its hand-off:
//   * tools/l2_survive_probe.hip (profiles/r5_l2_survive_probe.json): lines a kernel READ with default-policy loads survive one and two
//     dependent kernel boundaries in the L2 of the XCD that read them (16 MB: 1.5 us against 3.7 cold, 32 MB: 2.1 against 6.2); lines read
//     with nt loads do not stay; a workgroup of ANOTHER XCD does not see them (Infinity Cache rate).
//   * so the pull has to be XCD-consistent: workgroup L of a launch is dispatched to XCD L mod 8 (round-robin over the linear workgroup id;
//     tests/test_gpu_glue.py::test_xcc_round_robin reads XCC_ID back), a pulling workgroup on XCD x walks the units of the consumer's
//     workgroups x, x + 8, x + 16, ... and takes the first `head_pairs` record pairs of each (every wave of the consumer streams its unit from
//     pair 0, all waves at once: the bytes the launch needs first are the heads of ALL units, not whole units).
// Nothing here changes a result: the pulled bytes are thrown away; a wrong mapping costs time, never correctness.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/sjd_hip_experimental.h"

typedef __attribute__((ext_vector_type(4))) unsigned sjd_pf_u32x4;

// record pair `job` of the head as XCD x reads it -> its address (wave-uniform arithmetic: one job per wave at a time); a pair that does not
// exist (ragged edges) reads the first one again, so that every load is unconditional (behind a conditional load the compiler drains the queue)
__device__ __forceinline__ const unsigned char *sjd_l2_head_pair(const sjd_l2_head &d, unsigned x, unsigned job)
{
    const unsigned char *base = reinterpret_cast<const unsigned char *>(d.wz);
    unsigned r = job, w, p;
    const unsigned waves = (unsigned)d.waves, F = (unsigned)d.head_pairs;
    if (d.step_major) { w = r % waves; r /= waves; p = r % F; r /= F; }
    else { p = r % F; r /= F; w = r % waves; r /= waves; }
    const unsigned L = x + 8u * r;                  // the consumer's workgroup (linear id)
    const unsigned bx = L % (unsigned)d.gx, by = L / (unsigned)d.gx;
    if (by >= (unsigned)d.gy) return base;
    size_t rec;
    if (d.kind == 0) {                              // g1z_skinny_gemm(_tiled): wave w of workgroup (bx, chunk by) owns column tile bx * waves + w
        const unsigned t_out = bx * waves + w;
        if (t_out >= (unsigned)d.n_out) return base;
        const unsigned t = (unsigned)d.tile0 + t_out, pairs = (by == (unsigned)d.gy - 1u) ? (unsigned)d.pairs_last : (unsigned)d.pairs_full;
        if (p >= pairs) return base;
        rec = (size_t)by * d.n_tiles * d.pairs_full + (d.step_major ? (size_t)p * d.n_tiles + t : (size_t)t * pairs + p);
    } else {                                        // g1z_gateup_silu(_tall): wave w = (K half w >> 2, gate tiles 2 bx, 2 bx + 1 | the up tiles of the same columns)
        const unsigned kh = w >> 2, q = w & 3u, n_gate = (unsigned)d.n_tiles / 2u;
        const unsigned t = (q < 2u ? 0u : n_gate) + 2u * bx + (q & 1u);
        if (p >= (unsigned)d.pairs_full) return base;
        rec = (size_t)kh * d.n_tiles * d.pairs_full + (d.step_major ? (size_t)p * d.n_tiles + t : (size_t)t * d.pairs_full + p);
    }
    return base + rec * 1536;
}

// The role of pulling workgroup j of P (P a multiple of 8) whose linear id in its launch is first_id + j (first_id a multiple of 8, so that
// workgroup j sits on XCD j mod 8).  A wave takes one record pair (1536 B) per job: 64 lanes x 16 B of low bytes + 32 x 16 B of codes (lanes
// 32..63 repeat the addresses of 0..31); four jobs = eight default-policy loads in flight per lane; results discarded.
__device__ __forceinline__ void sjd_l2_head_pull(const sjd_l2_head &d, int j, int P)
{
    const unsigned x = (unsigned)j & 7u, rank = (unsigned)j >> 3, per = (unsigned)P >> 3;
    const unsigned lane = threadIdx.x & 63u, nwv = blockDim.x >> 6;
    const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned n_wg = (unsigned)(d.gx * d.gy);
    if (n_wg <= x) return;
    const unsigned nL = (n_wg - x + 7u) >> 3;
    const unsigned n_jobs = nL * (unsigned)d.waves * (unsigned)d.head_pairs;
    const unsigned stride = per * nwv;
    sjd_pf_u32x4 acc = {0u, 0u, 0u, 0u};
    for (unsigned j0 = rank * nwv + wv; j0 < n_jobs; j0 += 4u * stride) {
        const unsigned char *b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned job = j0 + (unsigned)u * stride;
            const unsigned char *q = sjd_l2_head_pair(d, x, job < n_jobs ? job : 0u);
            const unsigned long long a = (unsigned long long)q;
            b[u] = (const unsigned char *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) << 32) |
                                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a));
        }
        sjd_pf_u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            typedef const __attribute__((address_space(1))) sjd_pf_u32x4 *gp;          // (global, not flat: the address went through integers)
            v[2 * u] = *(gp)(unsigned long long)(b[u] + lane * 16u);
            v[2 * u + 1] = *(gp)(unsigned long long)(b[u] + 1024u + (lane & 31u) * 16u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    // (an empty statement that "reads" the folded value keeps the loads alive without a store; a store into a static device variable nobody
    //  reads is dead code to the compiler -- the first build of this kernel was a lone s_endpgm)
    asm volatile("" : : "v"(acc.x), "v"(acc.y), "v"(acc.z), "v"(acc.w));
}
