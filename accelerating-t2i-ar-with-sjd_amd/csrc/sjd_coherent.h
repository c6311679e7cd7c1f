// sjd_coherent.h -- device-coherent (sc1) accesses for data that workgroups of ONE launch exchange across XCDs (round 3: the reducing tail
// of G1, the split merge of K1).  An MI355X has eight XCDs with an L2 each; a plain store stays in the writer's L2 until the kernel
// boundary writes it back, and a plain load may hit a stale line.  Making the whole L2 coherent inside a kernel (__threadfence: write-back +
// invalidate) costs ~14 us; marking the few exchanged accesses device-scope instead writes them through / reads them
// from the memory side and costs one ordinary round trip.  Everything here is a compiler-visible atomic (relaxed, agent scope): the
// instruction selector emits global_load / global_store ... sc1 and keeps its own s_waitcnt bookkeeping -- hand-written asm loads would
// land in registers the register allocator believes it may already reuse.
//
// Ordering: the exchanges built on these accesses (g1_reduce_tail, k1_merge_publish, g1z_mlp_pair) use the guide's
// "drained sc1" form -- sc1 payload stores -> `s_waitcnt vmcnt(0)` (every store acknowledged by the memory side) -> workgroup barrier ->
// relaxed agent-scope counter add;  consumer: relaxed poll / ticket -> workgroup barrier -> sc1 loads -- which MI355X_MICROARCH.md lists under
// "Valid forms" ({sc0 sc1 stores and loads both sides}; "sc1 loads may replace the acquire only when the producer stored sc1").  No release /
// acquire FENCE is used on purpose: `fence(release, "agent")` lowers to buffer_wbl2 (a write-back of the XCD's whole L2, 1.7-6.5 us) and would
// cost more than the boundary these experiments tried to remove.  All three exchanges are off by default (measured no-go).
#pragma once
#include <hip/hip_runtime.h>

typedef __attribute__((ext_vector_type(4))) float sjd_f4;
typedef __attribute__((ext_vector_type(2))) float sjd_f2;

__device__ __forceinline__ sjd_f2 sjd_ld_coherent_f2(const float *p)
{
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return sjd_f2{__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32))};
}

__device__ __forceinline__ sjd_f4 sjd_ld_coherent_f4(const float *p)          // p 16-byte aligned: two 8-byte loads of one 16-byte piece
{
    const sjd_f2 a = sjd_ld_coherent_f2(p), b = sjd_ld_coherent_f2(p + 2);
    return sjd_f4{a.x, a.y, b.x, b.y};
}

__device__ __forceinline__ void sjd_st_coherent_f2(float *p, float x, float y)
{
    const unsigned long long v = (unsigned long long)__float_as_uint(x) | ((unsigned long long)__float_as_uint(y) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void sjd_st_coherent_f4(float *p, float x, float y, float z, float w)
{
    sjd_st_coherent_f2(p, x, y);
    sjd_st_coherent_f2(p + 2, z, w);
}
