// sjd_device.h -- device helpers shared by the sampling kernels (K2 / K4).
//
// Canonical fp32 numerics (specification; the CPU oracle restates the same arithmetic independently):
//   * no FMA contraction except the explicit __builtin_fmaf below  (file is built with -ffp-contract=off)
//   * exp: Cody-Waite range reduction + degree-5 polynomial, 0 below -87
//   * sum: thread T of a 1024-thread block owns columns {4T..4T+3} + 4096k, four running accumulators added in
//     increasing k; s_T = (a0+a1)+(a2+a3); xor-butterfly 32,16,8,4,2,1 inside each wave64; the 16 wave totals
//     are added in wave order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SJD_TPB 1024
#define SJD_WAVES (SJD_TPB / 64)
#define SJD_RADIX_BINS 2048

__device__ __forceinline__ float sjd_expf(float x)
{
    if (!(x >= -87.0f)) return 0.0f;
    const float LOG2E = 1.44269504088896341f;
    const float LN2_HI = 0.693359375f;
    const float LN2_LO = -2.12194440e-4f;
    const float MAGIC = 12582912.0f;
    float n = __builtin_fmaf(x, LOG2E, MAGIC) - MAGIC;
    float r = __builtin_fmaf(-n, LN2_HI, x);
    r = __builtin_fmaf(-n, LN2_LO, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r) + 1.0f;
    int ni = (int)n;
    return y * __uint_as_float((uint32_t)(ni + 127) << 23);
}

// Canonical natural logarithm of a positive finite float (musl / fdlibm logf: reduction to [sqrt(1/2), sqrt(2)), s = f / (2 + f),
// degree-4 even polynomial), every operation rounded separately and in the order written -- the CPU oracle restates it line by line.
// Only the temperature path of the residual resample uses it: softmax(log(d) / T) needs log d, which at T = 1 cancels (d / sum d).
__device__ __forceinline__ float sjd_logf(float x)
{
    const float LN2_HI = 6.9313812256e-01f, LN2_LO = 9.0580006145e-06f;
    const float LG1 = 0.66666662693f, LG2 = 0.40000972152f, LG3 = 0.28498786688f, LG4 = 0.24279078841f;
    uint32_t ix = __float_as_uint(x);
    int k = 0;
    if (ix < 0x00800000u) { x = x * 33554432.0f; k = -25; ix = __float_as_uint(x); }      // subnormal: scale by 2^25
    ix += 0x3f800000u - 0x3f3504f3u;
    k += (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    const float xr = __uint_as_float(ix);
    const float f = xr - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float w = z * z;
    const float t1 = w * (LG2 + w * LG4);
    const float t2 = z * (LG1 + w * LG3);
    const float R = t2 + t1;
    const float hfsq = (0.5f * f) * f;
    const float dk = (float)k;
    float r = s * (hfsq + R);
    r = r + dk * LN2_LO;
    r = r - hfsq;
    r = r + f;
    r = r + dk * LN2_HI;
    return r;
}

struct SjdShared {
    float wave_f[SJD_WAVES];
    int wave_i[SJD_WAVES];
    unsigned long long wave_u64[SJD_WAVES];
    unsigned hist[SJD_RADIX_BINS];
    int sel_bin;
    int sel_k;
    int misc[4];
};

// ---- exact reductions (order independent) -------------------------------------------------------------
__device__ __forceinline__ float block_max(float v, SjdShared &sh)
{
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.wave_f[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sh.wave_f[0];
    for (int w = 1; w < SJD_WAVES; ++w) r = fmaxf(r, sh.wave_f[w]);
    return r;
}

// max and count in ONE exchange (two barriers instead of four): the same values block_max / block_sum_int return
__device__ __forceinline__ float block_max_and_count(float v, int n, SjdShared &sh, int &n_total)
{
    for (int off = 32; off >= 1; off >>= 1) { v = fmaxf(v, __shfl_xor(v, off)); n += __shfl_xor(n, off); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh.wave_f[threadIdx.x >> 6] = v; sh.wave_i[threadIdx.x >> 6] = n; }
    __syncthreads();
    float r = sh.wave_f[0];
    int t = sh.wave_i[0];
    for (int w = 1; w < SJD_WAVES; ++w) { r = fmaxf(r, sh.wave_f[w]); t += sh.wave_i[w]; }
    n_total = t;
    return r;
}

__device__ __forceinline__ int block_sum_int(int v, SjdShared &sh)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.wave_i[threadIdx.x >> 6] = v;
    __syncthreads();
    int r = 0;
    for (int w = 0; w < SJD_WAVES; ++w) r += sh.wave_i[w];
    return r;
}

// ---- canonical (order-specified) fp32 sum --------------------------------------------------------------
__device__ __forceinline__ float block_canonical_sum(float a0, float a1, float a2, float a3, SjdShared &sh)
{
    float s = (a0 + a1) + (a2 + a3);
    for (int off = 32; off >= 1; off >>= 1) s = s + __shfl_xor(s, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.wave_f[threadIdx.x >> 6] = s;
    __syncthreads();
    float total = sh.wave_f[0];
    for (int w = 1; w < SJD_WAVES; ++w) total = total + sh.wave_f[w];
    return total;
}

// ---- argmax of (value, lowest index) -------------------------------------------------------------------
__device__ __forceinline__ unsigned long long pack_vi(float v, int idx)
{
    // v >= 0 here (ratios of non-negative numbers); larger value wins, then LOWER index
    return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(0x7fffffff - idx);
}

__device__ __forceinline__ int block_argmax(unsigned long long best, SjdShared &sh)
{
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned long long o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.wave_u64[threadIdx.x >> 6] = best;
    __syncthreads();
    unsigned long long r = sh.wave_u64[0];
    for (int w = 1; w < SJD_WAVES; ++w) r = sh.wave_u64[w] > r ? sh.wave_u64[w] : r;
    return 0x7fffffff - (int)(unsigned)(r & 0xffffffffu);
}

// two argmaxes in ONE exchange (the second candidates travel through the head of sh.hist, which no selection uses at that point)
__device__ __forceinline__ int block_argmax2(unsigned long long best, unsigned long long best2, SjdShared &sh, int &idx2)
{
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned long long o = __shfl_xor(best, off), o2 = __shfl_xor(best2, off);
        best = o > best ? o : best;
        best2 = o2 > best2 ? o2 : best2;
    }
    unsigned long long *second = reinterpret_cast<unsigned long long *>(sh.hist);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh.wave_u64[threadIdx.x >> 6] = best; second[threadIdx.x >> 6] = best2; }
    __syncthreads();
    unsigned long long r = sh.wave_u64[0], r2 = second[0];
    for (int w = 1; w < SJD_WAVES; ++w) { r = sh.wave_u64[w] > r ? sh.wave_u64[w] : r; r2 = second[w] > r2 ? second[w] : r2; }
    idx2 = 0x7fffffff - (int)(unsigned)(r2 & 0xffffffffu);
    return 0x7fffffff - (int)(unsigned)(r & 0xffffffffu);
}

// ---- radix select: exact k-th largest key among the keys a block-strided visitor yields -------------------
__device__ __forceinline__ unsigned f2key(float z)
{
    unsigned u = __float_as_uint(z);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// exclusive prefix (in thread order) of one int per thread
__device__ __forceinline__ int block_exclusive_scan(int v, SjdShared &sh)
{
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        int o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    __syncthreads();
    if (lane == 63) sh.wave_i[w] = inc;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < w; ++i) base += sh.wave_i[i];
    return base + inc - v;
}

// After the histogram of one radix digit is complete: find the bin holding the k-th largest (descending bins).
// Every thread returns the same (bin, remaining k).
__device__ __forceinline__ void radix_pick(int k, SjdShared &sh, int &bin, int &krem)
{
    __syncthreads();
    int t = threadIdx.x;
    int hi = SJD_RADIX_BINS - 1 - 2 * t, lo = hi - 1;
    int ch = (int)sh.hist[hi], cl = (int)sh.hist[lo];
    int before = block_exclusive_scan(ch + cl, sh);
    if (before < k && k <= before + ch) { sh.sel_bin = hi; sh.sel_k = k - before; }
    else if (before + ch < k && k <= before + ch + cl) { sh.sel_bin = lo; sh.sel_k = k - before - ch; }
    __syncthreads();
    bin = sh.sel_bin;
    krem = sh.sel_k;
    __syncthreads();
}

// Row visitor: thread T walks columns 4T + 4096*k (+0..3), the canonical ownership.
#define SJD_FOR_OWNED_COLS(V, c0)  for (int c0 = 4 * (int)threadIdx.x; c0 < (V); c0 += 4 * SJD_TPB)
// Same ownership restricted to the column window [lo, hi): only the 4-column groups that intersect it are visited
// (callers still test each column against the window / the rule).
__device__ __forceinline__ int sjd_first_owned_group(int lo)
{
    const int g0 = lo >> 2;
    return g0 + ((((int)threadIdx.x - g0) % SJD_TPB) + SJD_TPB) % SJD_TPB;
}
#define SJD_FOR_OWNED_COLS_IN(lo, hi, c0)  for (int c0 = 4 * sjd_first_owned_group(lo); c0 < (hi); c0 += 4 * SJD_TPB)

// k-th largest float of row[0..V) restricted to entries with `row[c] > floor_excl` (finite filter);
// requires 1 <= k <= count of such entries.
__device__ float block_kth_largest(const float *row, int lo, int V, int k, float floor_excl, SjdShared &sh)
{
    unsigned prefix = 0;
    int krem = k;
    const int shifts[3] = {21, 10, 0};
    const unsigned masks[3] = {0x7ffu, 0x7ffu, 0x3ffu};
    for (int pass = 0; pass < 3; ++pass) {
        for (int b = threadIdx.x; b < SJD_RADIX_BINS; b += SJD_TPB) sh.hist[b] = 0;
        __syncthreads();
        const int shift = shifts[pass];
        SJD_FOR_OWNED_COLS_IN(lo, V, c0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int c = c0 + j;
                if (c >= lo && c < V) {
                    float z = row[c];
                    if (z > floor_excl) {
                        unsigned key = f2key(z);
                        bool match = (pass == 0) || ((key >> (shift + (pass == 1 ? 11 : 10))) == prefix);
                        if (match) atomicAdd(&sh.hist[(key >> shift) & masks[pass]], 1u);
                    }
                }
            }
        }
        int bin;
        radix_pick(krem, sh, bin, krem);
        prefix = (pass == 0) ? (unsigned)bin : ((prefix << (pass == 1 ? 11 : 10)) | (unsigned)bin);
    }
    return key2f(prefix);
}

// the same over values a thread holds in registers: z[i][j] = the thread's j-th column of its i-th owned group (-inf where the window or the
// rule excludes the column: never > floor_excl).  Same keys into the same histograms as block_kth_largest over the staged row.
template <int NI>
__device__ __forceinline__ float block_kth_largest_regs(const float (&z)[NI][4], int k, float floor_excl, SjdShared &sh)
{
    unsigned prefix = 0;
    int krem = k;
    const int shifts[3] = {21, 10, 0};
    const unsigned masks[3] = {0x7ffu, 0x7ffu, 0x3ffu};
    for (int pass = 0; pass < 3; ++pass) {
        for (int b = threadIdx.x; b < SJD_RADIX_BINS; b += SJD_TPB) sh.hist[b] = 0;
        __syncthreads();
        const int shift = shifts[pass];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = z[i][j];
                if (v > floor_excl) {
                    const unsigned key = f2key(v);
                    const bool match = (pass == 0) || ((key >> (shift + (pass == 1 ? 11 : 10))) == prefix);
                    if (match) atomicAdd(&sh.hist[(key >> shift) & masks[pass]], 1u);
                }
            }
        int bin;
        radix_pick(krem, sh, bin, krem);
        prefix = (pass == 0) ? (unsigned)bin : ((prefix << (pass == 1 ? 11 : 10)) | (unsigned)bin);
    }
    return key2f(prefix);
}

// ---- exact k-th largest by BISECTION over the 32 key bits (round 4) -------------------------------------------------------
// Same answer as the radix select above -- the k-th largest key among the entries `> floor_excl`, returned as its float -- without the
// LDS histogram: that one funnels every entry of a row through atomicAdd on 2048 bins, of which a softmax row hits a dozen (the logits
// share their exponent), so the adds of 8192 .. 32768 entries serialise on a few LDS words: 25 us of K2's 35 at Lumina's shape, most of its
// 187 us at Emu3's (in-kernel stamps, round 3).  Here two key bits are fixed per round: the three candidates K | 1<<b, K | 2<<b, K | 3<<b are
// turned back into floats (key order == float order for everything that is not a NaN), every entry is compared against them with
// v_cmp -> wave ballot -> s_bcnt1 (the counts accumulate in SGPRs: no shuffle, no atomic), lane 0 of each wave posts its three counts,
// one barrier, every thread adds the 16 x 3 wave counts; 16 rounds.  Entries at or below the floor never count against a candidate above
// the floor, and the answer lies above it (k <= number of entries above the floor), so the decision of every round is the radix select's:
//   * -inf (K2's floor) is below every candidate that can be reached (a reached candidate has one of the bits 31..23 set);
//   * +0 (K4's floor) counts only against candidates <= key(+0), where the count is >= k either way.
// A candidate in the NaN range counts nothing, and nothing valid lies there.  -0 vs +0: equal as floats, adjacent as keys; the result is
// only ever used in `z < kth`, which cannot tell them apart.
// one round's exchange: the wave counts (three 21-bit fields of a u64: a row has fewer than 2^21 columns) are added into one of THREE LDS words
// used in rotation -- round r adds into word r % 3 and clears word (r + 1) % 3, which was last read two rounds (two barriers) ago -- so a
// round costs one 64-bit LDS atomic per wave, ONE barrier and one LDS read per thread.
__device__ __forceinline__ unsigned bisect_pick(unsigned K, int b, int k, int round, int n1, int n2, int n3, SjdShared &sh)
{
    unsigned long long *cnt = sh.wave_u64;             // (block_argmax's exchange buffer: not in use while a select runs)
    const int cur = round % 3, nxt = (round + 1) % 3;
    if ((threadIdx.x & 63) == 0)
        atomicAdd(&cnt[cur], (unsigned long long)(unsigned)n1 | ((unsigned long long)(unsigned)n2 << 21) | ((unsigned long long)(unsigned)n3 << 42));
    if (threadIdx.x == 0) cnt[nxt] = 0ull;
    __syncthreads();
    const unsigned long long t = cnt[cur];
    const int t1 = (int)(t & 0x1fffffu), t2 = (int)((t >> 21) & 0x1fffffu), t3 = (int)((t >> 42) & 0x1fffffu);
    return t3 >= k ? (K | (3u << b)) : t2 >= k ? (K | (2u << b)) : t1 >= k ? (K | (1u << b)) : K;
}

// the entries are what a thread holds in registers: z[i][j] (-inf, or anything at or below the floor, where the window or the rule excludes it)
template <int NI>
__device__ __forceinline__ float block_kth_largest_bisect_regs(const float (&z)[NI][4], int k, SjdShared &sh)
{
    __syncthreads();                                   // (whoever used the exchange words before is done with them)
    if (threadIdx.x == 0) sh.wave_u64[0] = 0ull;
    __syncthreads();
    unsigned K = 0;
    int round = 0;
    for (int b = 30; b >= 0; b -= 2, ++round) {
        const float f1 = key2f(K | (1u << b)), f2 = key2f(K | (2u << b)), f3 = key2f(K | (3u << b));
        int n1 = 0, n2 = 0, n3 = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = z[i][j];
                n1 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(v >= f1));
                n2 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(v >= f2));
                n3 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(v >= f3));
            }
        K = bisect_pick(K, b, k, round, n1, n2, n3, sh);
    }
    __syncthreads();                                   // (the last round's reads are done before the words serve block_argmax again)
    return key2f(K);
}

// the entries are staged in memory (LDS or global): column c of [lo, hi) at row[c - base] (a row staged in LDS starts at its window, not at
// column 0).  A window of at most NIW column groups per thread is taken into registers ONCE (entries at or below the floor as -inf) and
// selected there; a wider one (text rows over a whole vocabulary) is re-read every round.
#define SJD_BISECT_NIW 9
__device__ __forceinline__ float block_kth_largest_bisect(const float *row, int lo, int hi, int k, float floor_excl, SjdShared &sh, int base = 0)
{
    const int c_first = 4 * sjd_first_owned_group(lo);
    if ((hi - (lo & ~3) + 4 * SJD_TPB - 1) / (4 * SJD_TPB) <= SJD_BISECT_NIW) {         // (the same for every thread of the block)
        float v[SJD_BISECT_NIW][4];
#pragma unroll
        for (int i = 0; i < SJD_BISECT_NIW; ++i) {
            const int c0 = c_first + i * 4 * SJD_TPB;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + j;
                const bool in = c0 < hi && c >= lo && c < hi;
                const float x = row[(in ? c : lo) - base];
                v[i][j] = (in && x > floor_excl) ? x : -INFINITY;
            }
        }
        return block_kth_largest_bisect_regs<SJD_BISECT_NIW>(v, k, sh);
    }
    __syncthreads();
    if (threadIdx.x == 0) sh.wave_u64[0] = 0ull;
    __syncthreads();
    unsigned K = 0;
    int round = 0;
    for (int b = 30; b >= 0; b -= 2, ++round) {
        const float f1 = key2f(K | (1u << b)), f2 = key2f(K | (2u << b)), f3 = key2f(K | (3u << b));
        int n1 = 0, n2 = 0, n3 = 0;
        for (int c0 = c_first; __builtin_amdgcn_ballot_w64(c0 < hi) != 0ull; c0 += 4 * SJD_TPB) {     // (wave-uniform trip count: ballots inside)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + j;
                const bool in = c0 < hi && c >= lo && c < hi;
                const float v = row[(in ? c : lo) - base];
                const bool ok = in && v > floor_excl;
                n1 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(ok && v >= f1));
                n2 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(ok && v >= f2));
                n3 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(ok && v >= f3));
            }
        }
        K = bisect_pick(K, b, k, round, n1, n2, n3, sh);
    }
    __syncthreads();
    return key2f(K);
}

// ---- compaction of the entries that take part in a draw (round 4) ---------------------------------------------------------
// The draw needs one Philox4x32-10 evaluation (~130 instructions with the index arithmetic and the logarithm) per entry that carries
// probability mass -- top-k 2000 of 8192 / 2048 of 32768 columns -- but a wave runs a column slot as soon as ONE of its 64 lanes holds such an
// entry: practically every slot (36 per thread at Emu3's shape).  The kept entries {value, column} are therefore appended to a list in LDS
// (wave ballot -> one atomicAdd per wave and slot -> prefix of the lane) and the noise is evaluated DENSELY over the list.  The order of the
// list is arbitrary; the consumers are argmax reductions with an index tie-break, which do not depend on it.
__device__ __forceinline__ void wave_push(bool keep, int col, float v, unsigned long long *list, int cap, int *count)
{
    const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63, leader = __builtin_ctzll(m);
    int base = 0;
    if (lane == leader) base = atomicAdd(count, __builtin_popcountll(m));
    base = __shfl(base, leader);
    const int pos = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (keep && pos < cap) list[pos] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)col;
}

// the same list filled with ONE reservation per wave: every lane announces how many entries it will append, the wave's total is reserved with a
// single atomicAdd and each lane gets its first position (in-kernel stamps, round 4: one ballot + atomicAdd + broadcast per column slot was 8 -- Emu3:
// 32 -- dependent LDS round trips per wave, 128 -- 512 -- adds on one LDS word per row: 6.6 of K2's 31 us at Lumina's shape).  All 64 lanes call.
__device__ __forceinline__ int wave_reserve(int n_mine, int *count)
{
    const int lane = threadIdx.x & 63;
    int incl = n_mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
    int base = 0;
    if (lane == 63) base = atomicAdd(count, incl);
    base = __shfl(base, 63);
    return base + incl - n_mine;
}

// ---- exact k-th largest: a VALUE histogram first, the radix select over the survivors (round 4) ------------------------------
// The radix select funnels every score of a row through atomicAdd on the bins of its top 11 key bits (sign, exponent, two mantissa bits) --
// and the scores of one softmax row share their exponent: 8192 .. 32768 adds land on a dozen LDS words and serialise (in-kernel stamps,
// round 3: 25 of K2's 35 us at Lumina's shape).  A bisection over the key bits removes the atomics but needs 16 rounds of three compares
// per score: 2400 instructions per wave with four waves per SIMD -- measured no faster (19 us).  Here the FIRST histogram is over the
// scores' VALUES, 2048 equal bins across [zmax - 32, zmax] (what lies below carries < e^-32 of the mass and shares bin 0): the scores spread
// over hundreds of bins, a handful per bin, so the adds do not collide.  bin(z) is monotone in z (subtract, scale by a positive constant,
// clamp, truncate: each step is), hence every score above the selected bin is larger than every score in it: the k-th largest of the row
// is the k'-th largest of that bin, k' = k - (scores above).  The three radix passes then run as before but only the bin's few scores
// (typically < 20) reach an atomicAdd.  Same result as block_kth_largest: the k-th largest key, exactly.
// `visit(f)` calls f(z) for every entry this thread owns (entries at or below the floor / outside the window may be skipped or passed as -inf).
template <class Visit>
__device__ __forceinline__ float block_kth_largest_spread(Visit &&visit, int k, float zmax, SjdShared &sh)
{
    const float lo = zmax - 32.0f, inv = (float)(SJD_RADIX_BINS - 1) / 32.0f;
    auto bin_of = [&](float z) { return (int)fminf(fmaxf((z - lo) * inv, 0.0f), (float)(SJD_RADIX_BINS - 1)); };
    for (int b = threadIdx.x; b < SJD_RADIX_BINS; b += SJD_TPB) sh.hist[b] = 0;
    __syncthreads();
    visit([&](float z) { if (z > -INFINITY) atomicAdd(&sh.hist[bin_of(z)], 1u); });
    int bsel, krem;
    if (threadIdx.x == 0) sh.misc[2] = 0;          // (survivor count; the barriers of radix_pick lie between this and its first use)
    radix_pick(k, sh, bsel, krem);
    // The selected bin's scores (a handful .. a few hundred): their keys go to a list over the histogram's words (radix_pick has finished with
    // them) and every survivor ranks itself against the others -- the krem-th largest is the key with fewer than krem keys above it and at least
    // krem keys at or above it.  One barrier pair instead of the three radix passes' eighteen (in-kernel stamps, round 4: the select was 7.9 of
    // K2's 31 us at Lumina's shape, nearly all of it barriers).  More survivors than the rank loop is worth: the radix passes, as before.
    constexpr int RANK_CAP = 256;
    {
        unsigned *keys = sh.hist;
        visit([&](float z) {
            if (z > -INFINITY && bin_of(z) == bsel) {
                const int pos = atomicAdd(&sh.misc[2], 1);
                if (pos < RANK_CAP) keys[pos] = f2key(z);
            }
        });
        __syncthreads();
        const int n = sh.misc[2];
        if (n <= RANK_CAP) {
            for (int i = threadIdx.x; i < n; i += SJD_TPB) {
                const unsigned ki = keys[i];
                int gt = 0, ge = 0;
                for (int j = 0; j < n; ++j) { const unsigned kj = keys[j]; gt += kj > ki ? 1 : 0; ge += kj >= ki ? 1 : 0; }
                if (gt < krem && krem <= ge) sh.misc[3] = (int)ki;      // (equal keys all write the same word)
            }
            __syncthreads();
            const unsigned kk = (unsigned)sh.misc[3];
            __syncthreads();                       // (the histogram's words and misc[3] are free again)
            return key2f(kk);
        }
        __syncthreads();
    }
    unsigned prefix = 0;
    const int shifts[3] = {21, 10, 0};
    const unsigned masks[3] = {0x7ffu, 0x7ffu, 0x3ffu};
    for (int pass = 0; pass < 3; ++pass) {
        for (int b = threadIdx.x; b < SJD_RADIX_BINS; b += SJD_TPB) sh.hist[b] = 0;
        __syncthreads();
        const int shift = shifts[pass];
        visit([&](float z) {
            if (z > -INFINITY && bin_of(z) == bsel) {
                const unsigned key = f2key(z);
                const bool match = (pass == 0) || ((key >> (shift + (pass == 1 ? 11 : 10))) == prefix);
                if (match) atomicAdd(&sh.hist[(key >> shift) & masks[pass]], 1u);
            }
        });
        int bin;
        radix_pick(krem, sh, bin, krem);
        prefix = (pass == 0) ? (unsigned)bin : ((prefix << (pass == 1 ? 11 : 10)) | (unsigned)bin);
    }
    return key2f(prefix);
}

// ---- top-p cut (order-independent restatement of TopPLogitsWarper3d, see oracle/sjd_oracle.c header) ---------------------
// w[lo..hi): non-negative weights staged in global memory (e = exp(z - max) or the residual d); p_i = w_i / S.
// Returns K* = the largest uint32 key with canonical_sum{ p_i : w_i > 0, key(w_i) <= K* } <= thr (32 canonical sums).
// Ties at the cut: every entry whose weight EQUALS the threshold weight is removed or kept together (the set is a function of the values
// only), whereas the reference's sorted cumulative sum (logit_processor_3dim.py:406-419) splits a run of equal values by sort order --
// which torch.sort does not define for equal keys.  With fp32 softmax outputs a tie exactly at the cut is a measure-zero event; the oracle
// restates THIS rule (oracle/sjd_oracle.c) and is pinned against the reference's golden vectors, none of which hits one.
__device__ unsigned block_top_p_cut_key(const float *w, int lo, int hi, float S, float thr, SjdShared &sh, int base = 0)
{
    unsigned K = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = K | (1u << bit);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        SJD_FOR_OWNED_COLS_IN(lo, hi, c0) {
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + j;
                t[j] = 0.0f;
                if (c >= lo && c < hi) {
                    const float x = w[c - base];
                    if (x > 0.0f && f2key(x) <= cand) t[j] = x / S;
                }
            }
            a0 = a0 + t[0]; a1 = a1 + t[1]; a2 = a2 + t[2]; a3 = a3 + t[3];
        }
        const float T = block_canonical_sum(a0, a1, a2, a3, sh);
        if (T <= thr) K = cand;
    }
    return K;
}

// applies the cut in place (w_i = 0 for removed entries; the lowest-index maximum is always kept) and returns the new sum
__device__ float block_top_p_apply(float *w, int lo, int hi, float S, float thr, SjdShared &sh, int base = 0)
{
    unsigned long long best = 0ull;
    SJD_FOR_OWNED_COLS_IN(lo, hi, c0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j;
            if (c >= lo && c < hi) { unsigned long long cand = pack_vi(w[c - base], c); best = cand > best ? cand : best; }
        }
    }
    const int imax = block_argmax(best, sh);
    const unsigned K = block_top_p_cut_key(w, lo, hi, S, thr, sh, base);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    SJD_FOR_OWNED_COLS_IN(lo, hi, c0) {
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j;
            t[j] = 0.0f;
            if (c >= lo && c < hi) {
                float x = w[c - base];
                if (c != imax && x > 0.0f && f2key(x) <= K) { x = 0.0f; w[c - base] = 0.0f; }
                t[j] = x;
            }
        }
        a0 = a0 + t[0]; a1 = a1 + t[1]; a2 = a2 + t[2]; a3 = a3 + t[3];
    }
    return block_canonical_sum(a0, a1, a2, a3, sh);
}
