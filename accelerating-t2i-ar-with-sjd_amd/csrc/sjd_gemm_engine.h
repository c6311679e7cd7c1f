// sjd_gemm_engine.h -- round 5, VERDICT r4 next #1 stage A: kernel G1z in the LOADER / CONSUMER form (included by sjd_gemm.hip).
//
// One persistent 256-thread workgroup per CU: wave 0 is the LOADER -- it moves the 12-bit weight records HBM -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... offen lds, 1 KiB per instruction, no destination registers, counted with s_waitcnt vmcnt(N)) into three private
// rings of NS slots, one per CONSUMER wave; waves 1..3 read the records back from LDS (ds_read_b128 + ds_read_b64 per lane and pair), decode
// them with g1z_operand and feed the same MFMA sequence per (tile, K chunk) unit as g1z_skinny_gemm -- so the fp32 planes are BIT-IDENTICAL.
//   slot   = 1 KiB header area (the unit's exception header, fetched with every slot so that each slot is the same number of DMA
//            instructions: the vmcnt arithmetic stays a compile-time constant) + 8 record pairs (16 k-steps) x 1536 B = 13 KiB
//   rings  = 3 consumers x NS slots (NS = 3: 117 KiB) next to the staged activation chunk (KC <= 512: 32 KiB; NS = 2 for KC <= 1024)
//   sync   = two monotonic LDS counters per consumer: filled[c] (loader -> consumer, written after vmcnt says the slot landed) and
//            freed[c] (consumer -> loader, written after the consumer's last ds_read of the slot returned); polled with s_sleep; every poll is
//            BOUNDED (g1e_timeouts counts abandoned waits: wrong numbers, never a hang)
//   work   = workgroup b owns K chunk b / (P / n_chunks) and a contiguous run of column tiles there; its units go round-robin to the three
//            consumers; the loader runs DEPTH = 3 slots ahead of the landing front and up to 3 NS slots ahead of the consumers.
// Tile-major packing only (a unit's pairs are contiguous: a slot is ONE linear 12 KiB copy); M <= 32; bf16.
// What it is for: the run-ahead weight stream of a persistent layer (guide: ldsdma-fill, prefetch-credit).  Stage A's exit gate is this kernel
// as a launch of its own against g1z_skinny_gemm at the same shape: profiles/r5_engine_stageA.txt.
#pragma once

__device__ unsigned g1e_timeouts;
#ifdef G1E_TRACE          // (probe build: where the loader of workgroup 0 and its first consumer spend their cycles, s_memtime)
__device__ unsigned long long g1e_trace[16];
#define G1E_T() __builtin_readcyclecounter()
extern "C" int sjd_debug_engine_trace(unsigned long long *host_out)
{ return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g1e_trace), sizeof(g1e_trace), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }
#endif

#ifndef G1E_NT
#define G1E_NT 1              // nt (streamed-once) policy on the weight DMA (guide: nt-weights)
#endif

// one LDS-DMA instruction: 64 lanes x 16 B from (descriptor + voff) to LDS [lds_addr + 16 lane, +16)   (M0 saved / restored inside: guide 5.7)
__device__ __forceinline__ void g1e_dma16(u32x4 rsrc, unsigned voff, unsigned lds_addr)
{
    unsigned keep;
#if G1E_NT
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
#else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
#endif
}
// 64 lanes x 4 B (the header: 256 B per instruction)
__device__ __forceinline__ void g1e_dma4(u32x4 rsrc, unsigned voff, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
template <int N> __device__ __forceinline__ void g1e_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ u32x4 g1e_rsrc(const void *p, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)p;
    u32x4 r;
    r[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) & 0xffffu;       // stride 0: raw buffer
    r[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ unsigned g1e_lds_addr(const void *p)
{
    return (unsigned)(unsigned long)((__attribute__((address_space(3))) const unsigned char *)p);
}

constexpr int G1E_PAIRS = 8;                          // record pairs per slot
constexpr int G1E_HDR = 1024;                         // header area of a slot
constexpr int G1E_SLOT = G1E_HDR + G1E_PAIRS * 1536;  // 13312 B
constexpr int G1E_CONS = 3;                           // consumer waves
#ifndef G1E_DEPTH_N
#define G1E_DEPTH_N 4
#endif
#ifndef G1E_SLEEP
#define G1E_SLEEP 1
#endif
constexpr int G1E_DEPTH = G1E_DEPTH_N;                          // slots in flight per loader (4 x 13 DMA instructions: vmcnt is six bits)

// bounded poll of a monotonic LDS counter: returns once *flag + add >= need (or after ~2^20 polls: g1e_timeouts, wrong numbers, no hang)
__device__ __forceinline__ void g1e_poll(volatile unsigned *flag, unsigned add, unsigned need)
{
    int spins = 0;
    while (*flag + add < need) {
        __builtin_amdgcn_s_sleep(G1E_SLEEP);
        if (++spins > (1 << 20)) { if ((threadIdx.x & 63) == 0) atomicAdd(&g1e_timeouts, 1u); break; }
    }
}

template <int NS, int HD>          // HD = header DMA instructions per slot = cap / 32 (1, 2, 4)
__global__ __launch_bounds__(256, 1) void g1e_skinny_gemm(const unsigned short *__restrict__ x, const unsigned char *__restrict__ wz,
                                                          const u32x2 *__restrict__ exc, float *__restrict__ out, int M, int N, int K, int KC,
                                                          int n_tiles, int tile0, int wg_per_chunk)
{
    constexpr int DT = SJD_DTYPE_BF16;
    constexpr bool WIDE = HD == 4;
    constexpr int CAP = 32 * HD;
#if defined(G1E_NOHDR)
    constexpr int W = G1E_NDATA_W;
#else
    constexpr int W = 12 + HD;                        // DMA instructions per slot
#endif
    static_assert(G1E_DEPTH <= 4 && W * (G1E_DEPTH - 1) <= 63, "vmcnt is six bits");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int chunk = blockIdx.x / wg_per_chunk, g = blockIdx.x - chunk * wg_per_chunk;
    const int k0 = chunk * KC;
    const int steps = min(KC, K - k0) / 16;
    const int pairs = (steps + 1) / 2, pairs_full = (KC / 16 + 1) / 2;
    const int n_slots = (pairs + G1E_PAIRS - 1) / G1E_PAIRS;
    const int n_out = N / 32;
    const int t_lo = (int)(((long)g * n_out) / wg_per_chunk), t_hi = (int)(((long)(g + 1) * n_out) / wg_per_chunk);
    const int n_units = t_hi - t_lo;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // LDS: [x chunk: steps * 1 KiB][rings: 3 x NS slots][flags]
    u32x4 *xl = reinterpret_cast<u32x4 *>(smem);
    unsigned char *rings = smem + (size_t)(KC / 16) * 1024;
    volatile unsigned *flags = reinterpret_cast<volatile unsigned *>(rings + (size_t)G1E_CONS * NS * G1E_SLOT);
    volatile unsigned *filled = flags, *freed = flags + 4, *x_ready = flags + 8;
    if (threadIdx.x < 12) flags[threadIdx.x] = 0u;
    __syncthreads();              // the only workgroup barrier: the flags are armed

    if (w == 0) {
        // ------------------------------------------------------------------------------------------- LOADER
        const unsigned total = (unsigned)(n_units * n_slots);
        const size_t chunk_base = (size_t)chunk * n_tiles * pairs_full;
        // slot s of the launch = (unit i = s / n_slots, slot k of the unit), consumer c = i mod 3, the q-th slot (1-based) that consumer receives.
        // Two cursors walk that sequence -- one at the issue point, one at the landing front -- by increments: the first version derived
        // (i, k, c, q) from s with five integer divisions per step on the loader's one wave, ~900 cycles per slot, 76 % of its time
        // (cycle counters, profiles/r5_engine_stageA.txt).
        static_assert(G1E_CONS == 3, "three consumers");
        // plain scalars and macros: a cursor struct handed to lambdas by reference (and, before that, arrays indexed by c) went to SCRATCH memory
        unsigned i_i = 0, i_k = 0, i_c = 0, i_q0 = 1, i_q1 = 1, i_q2 = 1, i_r0 = 0, i_r1 = 0, i_r2 = 0;       // issue cursor (q: NEXT sequence number; r: ring position)
        unsigned p_k = 0, p_c = 0, p_q0 = 1, p_q1 = 1, p_q2 = 1;                                             // landing-front cursor
#define G1E_SEL(c_, a_, b_, d_) ((c_) == 0u ? (a_) : (c_) == 1u ? (b_) : (d_))
#define G1E_BUMP(r_) ((r_) + 1u == (unsigned)NS ? 0u : (r_) + 1u)
#define G1E_PUBLISH() do { \
            filled[p_c] = G1E_SEL(p_c, p_q0, p_q1, p_q2); \
            if (p_c == 0u) ++p_q0; else if (p_c == 1u) ++p_q1; else ++p_q2; \
            if (++p_k == (unsigned)n_slots) { p_k = 0; p_c = p_c + 1u == 3u ? 0u : p_c + 1u; } } while (0)
        auto issue_dma = [&](unsigned k, int t, unsigned slot) {
            const u32x4 wr = g1e_rsrc(wz + (chunk_base + (size_t)t * pairs) * 1536, (unsigned)pairs * 1536u);
            const u32x4 hr = g1e_rsrc(exc + ((size_t)chunk * n_tiles + t) * CAP, (unsigned)CAP * 8u);
#ifndef G1E_NODMA          // (timing probe: the loader publishes slots it never filled -- what the consumers alone sustain; results wrong)
#ifndef G1E_NOHDR          // (timing probe: no header DMA)
#pragma unroll
            for (int h = 0; h < HD; ++h) g1e_dma4(hr, (unsigned)(h * 256 + lane * 4), slot + (unsigned)(h * 256));
#endif
#ifndef G1E_NDATA
#define G1E_NDATA 12       // (timing probe: fewer data DMAs per slot)
#endif
#pragma unroll
            for (int p = 0; p < G1E_NDATA; ++p) g1e_dma16(wr, k * (unsigned)(G1E_PAIRS * 1536) + (unsigned)(p * 1024 + lane * 16), slot + (unsigned)(G1E_HDR + p * 1024));
#endif
        };
#define G1E_ISSUE() do { \
            const unsigned rpos_ = G1E_SEL(i_c, i_r0, i_r1, i_r2); \
            issue_dma((unsigned)__builtin_amdgcn_readfirstlane((int)i_k), tile0 + t_lo + __builtin_amdgcn_readfirstlane((int)i_i), \
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(rings_addr + (i_c * (unsigned)NS + rpos_) * (unsigned)G1E_SLOT))); \
            if (i_c == 0u) { ++i_q0; i_r0 = G1E_BUMP(i_r0); } else if (i_c == 1u) { ++i_q1; i_r1 = G1E_BUMP(i_r1); } else { ++i_q2; i_r2 = G1E_BUMP(i_r2); } \
            if (++i_k == (unsigned)n_slots) { i_k = 0; ++i_i; i_c = i_c + 1u == 3u ? 0u : i_c + 1u; } } while (0)
        const unsigned rings_addr = g1e_lds_addr(rings);
        unsigned issued = 0, published = 0;
        int idle = 0;
#ifdef G1E_TRACE
        unsigned long long t_issue = 0, t_wait = 0, t_idle = 0, t_begin = G1E_T(), n_wait = 0;
#endif
        while (published < total) {
            bool can = issued < total && (issued - published) < (unsigned)G1E_DEPTH;
#ifndef G1E_NOPOLL         // (timing probe, with G1E_NOCOMPUTE only: the loader never reads an LDS flag)
            if (can) can = (unsigned)__builtin_amdgcn_readfirstlane((int)freed[i_c]) + (unsigned)NS >= G1E_SEL(i_c, i_q0, i_q1, i_q2);      // its ring slot was read to its end
#endif
#ifdef G1E_TRACE
            if (can) { const unsigned long long a = G1E_T(); G1E_ISSUE(); ++issued; idle = 0; t_issue += G1E_T() - a; continue; }
#else
            if (can) { G1E_ISSUE(); ++issued; idle = 0; continue; }
#endif
            const unsigned inflight = issued - published;
            if (inflight == 0) {                                                    // nothing to wait for but a consumer
#ifdef G1E_TRACE
                t_idle += 100;
#endif
                __builtin_amdgcn_s_sleep(1);
                if (++idle > (1 << 20)) { if (lane == 0) atomicAdd(&g1e_timeouts, 1u); break; }
                continue;
            }
#ifdef G1E_TRACE
            const unsigned long long a_ = G1E_T();
#endif
            if (inflight >= 4) g1e_wait_vmcnt<3 * W>();
            else if (inflight == 3) g1e_wait_vmcnt<2 * W>();
            else if (inflight == 2) g1e_wait_vmcnt<W>();
            else g1e_wait_vmcnt<0>();
#ifdef G1E_TRACE
            t_wait += G1E_T() - a_; ++n_wait;
#endif
            G1E_PUBLISH();
            ++published;
        }
#ifdef G1E_TRACE
        if (blockIdx.x == 0 && lane == 0) {
            g1e_trace[0] = t_issue; g1e_trace[1] = t_wait; g1e_trace[2] = t_idle; g1e_trace[3] = G1E_T() - t_begin; g1e_trace[4] = total; g1e_trace[5] = n_wait;
        }
#endif
        return;
    }

    // ----------------------------------------------------------------------------------------------- CONSUMERS
    const int c = w - 1;
    {   // the activation chunk, staged by the 192 consumer threads (one batch of <= 12 pieces each at KC = 512) in A-fragment order
        const int tid = (int)threadIdx.x - 64, nth = 64 * G1E_CONS;
        const int ppr = 2 * steps, total = 32 * ppr;
        for (int v0 = tid; v0 < total; v0 += 12 * nth) {
            u32x4 val[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int v = min(v0 + i * nth, total - 1), m = v / ppr, j = v - m * ppr;
                val[i] = *reinterpret_cast<const u32x4 *>(x + (size_t)min(m, M - 1) * K + k0 + 8 * j);
                if (m >= M) val[i] = u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int v = v0 + i * nth;
                if (v < total) { const int m = v / ppr, j = v - m * ppr, s = j >> 1; xl[s * 64 + g1_slot(j & 1, m, s)] = val[i]; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) atomicAdd((unsigned *)x_ready, 1u);
        g1e_poll(x_ready, 0u, (unsigned)G1E_CONS);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    for (int i = c; i < n_units; i += G1E_CONS) {
        const int t_out = t_lo + i;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        g1z_hdr hd;
        for (int k = 0; k < n_slots; ++k) {
            const unsigned q = (unsigned)((i / G1E_CONS) * n_slots + k + 1);
            g1e_poll(&filled[c], 0u, q);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const unsigned char *slot = rings + ((size_t)c * NS + (q - 1u) % NS) * G1E_SLOT;
            if (k == 0) {
                const u32x2 *e = reinterpret_cast<const u32x2 *>(slot);
                g1z_hraw h;
                h.a = e[min(lane, CAP - 1)];
                if constexpr (WIDE) h.b = e[64 + lane]; else h.b = h.a;
                hd = g1z_header<WIDE>(h, lane, CAP);
            }
            const unsigned char *rec = slot + G1E_HDR;
#ifdef G1E_NOCOMPUTE       // (timing probe: the consumers free every slot unread -- what the loader alone sustains; results wrong)
            if (k < 0)
#endif
            if ((k + 1) * 2 * G1E_PAIRS <= steps) {
                // a full slot: sixteen k-steps, no branch -- all eight pairs are requested from LDS before the first decode
                u32x4 lo[G1E_PAIRS];
                u32x2 cc[G1E_PAIRS];
#pragma unroll
                for (int p = 0; p < G1E_PAIRS; ++p) {
                    lo[p] = *reinterpret_cast<const u32x4 *>(rec + p * 1536 + lane * 16);
                    cc[p] = *reinterpret_cast<const u32x2 *>(rec + p * 1536 + 1024 + lane * 8);
                }
                const u32x4 *xa = xl + (size_t)k * (2 * G1E_PAIRS) * 64;
                u32x4 a0 = xa[g1_slot(lane >> 5, lane & 31, 0)];
#pragma unroll
                for (int p = 0; p < G1E_PAIRS; ++p) {
                    const int sa = 2 * (k * G1E_PAIRS + p);
                    const u32x4 a1 = xa[(2 * p + 1) * 64 + g1_slot(lane >> 5, lane & 31, 2 * p + 1)];
                    const u32x4 b0 = g1z_operand<WIDE>(lo[p].x, lo[p].y, cc[p].x, (unsigned)sa, hd, lane);
                    acc = G1Mfma<DT>::mma(a0, b0, acc);
                    if (p + 1 < G1E_PAIRS) a0 = xa[(2 * p + 2) * 64 + g1_slot(lane >> 5, lane & 31, 2 * p + 2)];
                    const u32x4 b1 = g1z_operand<WIDE>(lo[p].z, lo[p].w, cc[p].y, (unsigned)(sa + 1), hd, lane);
                    acc = G1Mfma<DT>::mma(a1, b1, acc);
                }
            } else {
#pragma unroll
                for (int p = 0; p < G1E_PAIRS; ++p) {
                    // the chunk's last, partial slot: every LDS read is unconditional (a pair past the unit's end was written as zeros by the
                    // descriptor's range check, the activation index is clamped), the MFMAs of k-steps past the chunk are skipped by uniform branches
                    const int sa = 2 * (k * G1E_PAIRS + p), sb = sa + 1;
                    const u32x4 lo = *reinterpret_cast<const u32x4 *>(rec + p * 1536 + lane * 16);
                    const u32x2 cc = *reinterpret_cast<const u32x2 *>(rec + p * 1536 + 1024 + lane * 8);
                    const u32x4 a0 = xl[min(sa, steps - 1) * 64 + g1_slot(lane >> 5, lane & 31, sa)];
                    const u32x4 a1 = xl[min(sb, steps - 1) * 64 + g1_slot(lane >> 5, lane & 31, sb)];
                    if (sa < steps) {
                        const u32x4 b0 = g1z_operand<WIDE>(lo.x, lo.y, cc.x, (unsigned)sa, hd, lane);
                        acc = G1Mfma<DT>::mma(a0, b0, acc);
                    }
                    if (sb < steps) {
                        const u32x4 b1 = g1z_operand<WIDE>(lo.z, lo.w, cc.y, (unsigned)sb, hd, lane);
                        acc = G1Mfma<DT>::mma(a1, b1, acc);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                  // the slot's last ds_read has returned
            freed[c] = q;
        }
        float *o = out + ((size_t)chunk * 32) * N + (size_t)t_out * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            o[(size_t)m * N] = acc[r];
        }
    }
}

// x [M <= 32, K] bf16, wz / exc = ops.pack_weight_z(W, KC, step_major = False) -> out fp32 [n_chunks, 32, N]: what sjd_skinny_gemm_z writes, bit for
// bit, from one persistent workgroup per CU (n_wg = the CU count, a multiple of the K-chunk count).
extern "C" int sjd_skinny_gemm_engine_z(const void *x, const void *wz, const void *exc, int exc_cap, float *out, int M, int N, int K, int KC,
                                        int dtype, int N_packed, int tile0, int n_wg, void *stream)
{
    if (!(exc_cap == 32 || exc_cap == 64 || exc_cap == 128)) return SJD_ERR_BAD_ARG;
    if (!x || !wz || !exc || !out || M < 1 || N < 32 || (N % 32) != 0 || (N_packed % 32) != 0 || (K % 16) != 0 || KC < 16 || (KC % 16) != 0)
        return SJD_ERR_BAD_ARG;
    if (dtype != SJD_DTYPE_BF16 || M > 32 || KC > 1024) return SJD_ERR_UNSUPPORTED;
    const int n_out = N / 32, n_tiles = N_packed / 32, n_chunks = (K + KC - 1) / KC;
    if (tile0 < 0 || tile0 + n_out > n_tiles || n_wg < n_chunks || (n_wg % n_chunks) != 0) return SJD_ERR_BAD_ARG;
    const int wpc = n_wg / n_chunks;
    if (wpc > n_out) return SJD_ERR_BAD_ARG;
    const int NS = KC <= 512 ? 3 : 2;
    const size_t lds = (size_t)(KC / 16) * 1024 + (size_t)G1E_CONS * NS * G1E_SLOT + 64;
    if (lds > 160 * 1024) return SJD_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
#define SJD_G1E(NS_, HD_) do { \
        (void)hipFuncSetAttribute((const void *)g1e_skinny_gemm<NS_, HD_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((g1e_skinny_gemm<NS_, HD_>), dim3(n_wg), dim3(256), lds, s, (const unsigned short *)x, (const unsigned char *)wz, \
                           (const u32x2 *)exc, out, M, N, K, KC, n_tiles, tile0, wpc); } while (0)
    const int hd = exc_cap / 32;
    if (NS == 3) { if (hd == 1) SJD_G1E(3, 1); else if (hd == 2) SJD_G1E(3, 2); else SJD_G1E(3, 4); }
    else { if (hd == 1) SJD_G1E(2, 1); else if (hd == 2) SJD_G1E(2, 2); else SJD_G1E(2, 4); }
#undef SJD_G1E
    return hipGetLastError() == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH;
}

extern "C" int sjd_engine_timeouts(void)
{
    unsigned v = 0;
    return hipMemcpyFromSymbol(&v, HIP_SYMBOL(g1e_timeouts), sizeof(v), 0, hipMemcpyDeviceToHost) == hipSuccess ? (int)v : -1;
}
