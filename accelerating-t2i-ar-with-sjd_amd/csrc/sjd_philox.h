// sjd_philox.h -- the noise of the SJD sampler, generated where it is consumed.
//
// The reference draws three tensors per iteration from a device torch.Generator:
//     torch.multinomial(p[n, V])          == argmax(p / empty(n, V).exponential_(generator=g))        JL:118
//     torch.rand((1, n, V), generator=g)  -- of which it reads n - 1 elements                           JL:260-282
//     torch.multinomial(residual[1, V])   == argmax(r / empty(1, V).exponential_(generator=g))        JL:237 (on a rejection only)
// Rounds 1-2 filled [n, V] + [n, V] + [1, V] fp32 tensors with three ATen launches on a side stream.  This header restates what those
// ATen launches compute, so that K2 / K4 evaluate exactly the elements they read and nothing is written to HBM:
//
//   ATen `distribution_nullary_kernel` (aten/src/ATen/native/cuda/DistributionTemplates.h, torch 2.10; third-party -- not in the
//   reference tree, pinned by tests/test_gpu_philox.py against torch itself on the GPU):
//     block = 256 threads, grid = min(multiProcessorCount * (maxThreadsPerMultiProcessor / 256), ceil(numel / 256)), T = grid * 256;
//     thread idx owns Philox4x32-10 subsequence idx, starting at the generator's offset; loop l = 0, 1, ... draws one 4-vector r and
//     element  e = l * 4T + ii * T + idx  receives r[ii];   the generator's offset then advances by  (ceil(numel / 4T)) * 4.
//   Philox state as hipRAND / rocRAND set it up (rocrand_philox4x32_10.h): key = (seed lo, seed hi), counter = (offset / 4 as 64 bit,
//   subsequence as 64 bit); the l-th draw of a thread uses counter + l.
//   uint32 -> (0, 1]:  2^-32 + float(x) * 2^-32 (rocrand_uniform.h:67), contracted to ONE fma by the HIP default -ffp-contract=fast
//   under which torch's kernels are built (checked on the GPU against torch.rand: the fused form is the one that matches).
//   uniform_(0, 1):    u == 1 ? 0 : u                                      (ATen native/cuda/DistributionTemplates.h uniform_kernel)
//   exponential_(1):   u >= 1 - eps/2 ? eps/2 : -log(u)                    (ATen core/TransformationHelper.h exponential, device branch)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sjd_hip.h"

__device__ __forceinline__ void sjd_philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0, uint32_t k1)
{
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
}

// component `ii` of Philox4x32-10(counter = (ctr lo, ctr hi, subsequence, 0), key = seed)
__device__ __forceinline__ uint32_t sjd_philox4x32_10(uint64_t seed, uint64_t ctr, uint32_t subsequence, int ii)
{
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = subsequence, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        sjd_philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return ii == 0 ? c0 : ii == 1 ? c1 : ii == 2 ? c2 : c3;
}

// threads T of the ATen launch that would fill `numel` elements
__device__ __forceinline__ uint32_t sjd_philox_threads(uint64_t numel, uint32_t max_blocks)
{
    const uint64_t blocks = (numel + 255) / 256;
    return (uint32_t)(blocks < max_blocks ? blocks : max_blocks) * 256u;
}

// what the generator's offset advances by for a tensor of `numel` elements (host and device)
__host__ __device__ __forceinline__ uint64_t sjd_philox_offset_step(uint64_t numel, uint32_t max_blocks)
{
    if (numel == 0) return 0;
    const uint64_t blocks = (numel + 255) / 256;
    const uint64_t T = (blocks < max_blocks ? blocks : max_blocks) * 256ull;
    return ((numel - 1) / (T * 4) + 1) * 4;
}

// the uniform (0, 1] torch's kernel hands to its transform for element e of a tensor filled by T threads from `offset`
__device__ __forceinline__ float sjd_philox_uniform01(uint64_t seed, uint64_t offset, uint32_t T, uint64_t e)
{
    uint32_t idx, q;
    if (e < 0x100000000ull) { idx = (uint32_t)e % T; q = (uint32_t)e / T; }
    else { idx = (uint32_t)(e % T); q = (uint32_t)(e / T); }
    const uint32_t x = sjd_philox4x32_10(seed, offset / 4 + (q >> 2), idx, (int)(q & 3u));
    return __builtin_fmaf((float)x, 2.3283064e-10f, 2.3283064e-10f);
}

__device__ __forceinline__ float sjd_philox_rand(uint64_t seed, uint64_t offset, uint32_t T, uint64_t e)         // tensor.uniform_(0, 1)
{
    const float u = sjd_philox_uniform01(seed, offset, T, e);
    return u == 1.0f ? 0.0f : u;
}

__device__ __forceinline__ float sjd_philox_exponential(uint64_t seed, uint64_t offset, uint32_t T, uint64_t e)  // tensor.exponential_(1)
{
    const float u = sjd_philox_uniform01(seed, offset, T, e);
    const float lg = (u >= 1.0f - 1.1920929e-07f / 2) ? -1.1920929e-07f / 2 : logf(u);
    return (-1.0f / 1.0f) * lg;
}
