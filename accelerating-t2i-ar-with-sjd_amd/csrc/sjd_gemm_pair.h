// sjd_gemm_pair.h -- the MLP of a window forward as ONE launch (round 4 experiment; VERDICT r3 "next #2": a run-ahead weight loader across a
// dependency edge).  Included at the end of sjd_gemm.hip.
//
// Today: G1sz (gate|up + SiLU * up, 25.3 us) -> kernel boundary -> G1z (down, 14.9 us); the boundary costs ~1.5-2 us and the down kernel then
// opens with a cold round trip for its activation, its header and its first weight records (~2.5 us) -- ~4.5 us of a 42 us pair in which
// no weight byte moves.  Here both projections are phases of one launch of max(I / 64, down's workgroups) 512-thread workgroups:
//   phase A  workgroup b < I / 64 runs the body of g1z_gateup_silu for gate / up tiles 2 b, 2 b + 1 (arithmetic, accumulation order and
//            rounding of G1sz: the activation y is bit-identical) and publishes its 32 x 64 slice of y WRITE-THROUGH (8-byte relaxed agent-scope
//            stores = global_store_dwordx2 sc1), drains them (vmcnt(0)), and adds 1 to the arrival counter of the down projection's K chunk
//            its 64 columns belong to (KC of down is a multiple of 64);
//   phase B  every workgroup is one (column group, K chunk) unit of g1z_skinny_gemm.  BEFORE it waits it requests its unit's header and the
//            first ring of weight records -- they do not depend on y -- so the weight stream of `down` is already in flight while the last
//            gate|up workgroups finish; then one lane polls the chunk's counter (relaxed agent-scope loads, s_sleep between polls, bounded),
//            the activation chunk is staged with sc1 loads (the producers stored sc1: guide, "Valid forms"), and the body of g1z_skinny_gemm
//            runs unchanged (planes bit-identical to G1z's).
//   The workgroup that finishes LAST (a completion counter) re-arms every counter: the launch is replayable from a hipGraph with static
//   arguments.  All workgroups are resident at once (<= 256, one per CU: 128 KB of LDS each), so a waiting workgroup can always be served;
//   the poll is bounded anyway (a timeout is counted in g1_pair_timeouts and the workgroup goes on: wrong numbers, never a hang).
#pragma once

__device__ unsigned g1_pair_timeouts;

template <bool WIDE>
__device__ __forceinline__ void g1zp_gateup(int bid, const unsigned short *__restrict__ x, const unsigned char *__restrict__ wz,
                                            const u32x2 *__restrict__ exc, unsigned short *__restrict__ y, int M, int I, int K, int stride_cap,
                                            const float *__restrict__ row_sumsq, int rs_slices, float rs_inv_hidden, float rs_eps,
                                            unsigned char *smem, float *rsc)
{
    constexpr int SP = 64;                                        // K = 4096: two K halves x two phases of 64 k-steps
    const int rec_stride = stride_cap & 0xffff, exc_cap = stride_cap >> 16;
    constexpr int DT = SJD_DTYPE_BF16;
    constexpr int D = G1Z_DEPTH, DP = D / 2, TL = D < 8 ? 8 : D;
    u32x4 *xl = reinterpret_cast<u32x4 *>(smem);
    constexpr int PPS = 2 * SP;
    constexpr int NPT = (32 * 2 * PPS) / 512;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int kh = w >> 2, q = w & 3;
    const int n_gate = I / 32, n_tiles = 2 * n_gate;
    const int t_act = 2 * bid + (q & 1);
    const int t = (q < 2 ? 0 : n_gate) + t_act;
    constexpr int pairs = SP;
    const size_t chunk_base = (size_t)kh * n_tiles * pairs;
    const size_t tile_off = (rec_stride == 1) ? (size_t)t * pairs : (size_t)t;
    const unsigned rsb = (unsigned)rec_stride * 1536u;
    const __amdgpu_buffer_rsrc_t wr = g1z_unit_rsrc(wz + (chunk_base + tile_off) * 1536, (unsigned)(pairs - 1) * rsb + 1536u);
    auto w_load = [&](int p) -> g1z_pair { return g1z_load(wr, (unsigned)lane, (unsigned)p * rsb); };
    auto x_load = [&](int ph, int i) -> u32x4 {
        const int v = i * 512 + threadIdx.x;
        const int m = v / (2 * PPS), hh = (v / PPS) & 1, j = v % PPS;
        return (m < M) ? *reinterpret_cast<const u32x4 *>(x + (size_t)m * K + hh * (K / 2) + ph * (K / 4) + 8 * j) : u32x4{0u, 0u, 0u, 0u};
    };
    auto x_store = [&](int i, u32x4 val) {
        const int v = i * 512 + threadIdx.x;
        const int m = v / (2 * PPS), hh = (v / PPS) & 1, j = v % PPS;
        xl[(hh * SP + (j >> 1)) * 64 + g1_slot(j & 1, m, j >> 1)] = val;
    };
    g1z_pair ring[DP];
    u32x4 val[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) val[i] = x_load(0, i);
    float ssv[8];
    {
        const float *ssp = row_sumsq ? row_sumsq : reinterpret_cast<const float *>(x);
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) ssv[qq] = ssp[(size_t)(row_sumsq ? min(qq, rs_slices - 1) : 0) * 32 + (threadIdx.x & 31)];
    }
    const g1z_hraw hraw = g1z_header_load<WIDE>(exc, (size_t)kh * n_tiles + t, lane, exc_cap);
#pragma unroll
    for (int u = 0; u < DP; ++u) ring[u] = w_load(u);
#pragma unroll
    for (int i = 0; i < NPT; ++i) x_store(i, val[i]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPT; ++i) val[i] = x_load(1, i);
    if (threadIdx.x < 32) {
        float tsum = 0.f;
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) tsum += (qq < rs_slices) ? ssv[qq] : 0.f;
        rsc[threadIdx.x] = row_sumsq ? rsqrtf(__builtin_fmaf(tsum, rs_inv_hidden, rs_eps)) : 1.0f;
    }
    const g1z_hdr hd = g1z_header<WIDE>(hraw, lane, exc_cap);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const u32x4 *xa = xl + (size_t)kh * SP * 64;
    auto trip = [&](int l0, int s0) {
        u32x4 a[2];
        a[0] = xa[l0 * 64 + g1_slot(lane >> 5, lane & 31, 0)];
#pragma unroll
        for (int u = 0; u < TL / 2; ++u) {
            g1z_pair &slot = ring[u % DP];
            a[1] = xa[(l0 + 2 * u + 1) * 64 + g1_slot(lane >> 5, lane & 31, 2 * u + 1)];
            const u32x4 b0 = g1z_operand<WIDE>(slot.lo.x, slot.lo.y, slot.c.x, (unsigned)(s0 + 2 * u), hd, lane);
            acc = G1Mfma<DT>::mma(a[0], b0, acc);
            if (u + 1 < TL / 2) a[0] = xa[(l0 + 2 * u + 2) * 64 + g1_slot(lane >> 5, lane & 31, 2 * u + 2)];
            const u32x4 b1 = g1z_operand<WIDE>(slot.lo.z, slot.lo.w, slot.c.y, (unsigned)(s0 + 2 * u + 1), hd, lane);
            slot = w_load(s0 / 2 + u + DP);
            acc = G1Mfma<DT>::mma(a[1], b1, acc);
        }
    };
    for (int g = 0; g < SP / TL; ++g) trip(g * TL, g * TL);                   // ---- phase 0
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPT; ++i) x_store(i, val[i]);
    __syncthreads();
    for (int g = 0; g < SP / TL; ++g) trip(g * TL, SP + g * TL);              // ---- phase 1
    __syncthreads();
    constexpr int RP = 36;
    float *red = reinterpret_cast<float *>(smem);
    {
        float *mine = red + (size_t)(kh * 4 + q) * 32 * RP + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * RP] = acc[r];
    }
    __syncthreads();
    {
        const int a = threadIdx.x >> 8, m = (threadIdx.x >> 3) & 31, c4 = (threadIdx.x & 7) * 4;
        auto plane = [&](int kh_, int q_) { return *reinterpret_cast<const float4 *>(red + ((size_t)(kh_ * 4 + q_) * 32 + m) * RP + c4); };
        const float4 g0 = plane(0, a), g1 = plane(1, a), u0 = plane(0, 2 + a), u1 = plane(1, 2 + a);
        const float gs[4] = {g0.x, g0.y, g0.z, g0.w}, gt[4] = {g1.x, g1.y, g1.z, g1.w};
        const float us_[4] = {u0.x, u0.y, u0.z, u0.w}, ut[4] = {u1.x, u1.y, u1.z, u1.w};
        const float rr = rsc[m];
        unsigned short o16[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gsum = 0.f, usum = 0.f;
            gsum += gs[j]; gsum += gt[j];
            usum += us_[j]; usum += ut[j];
            o16[j] = sjd_silu_mul_elem<DT>(gsum, usum, rr);
        }
        if (m < M) {         // write-through (sc1): the consumers of this slice sit on other CUs / XCDs of the SAME launch
            const unsigned long long pk = (unsigned long long)((unsigned)o16[0] | ((unsigned)o16[1] << 16)) |
                                          ((unsigned long long)((unsigned)o16[2] | ((unsigned)o16[3] << 16)) << 32);
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(y + (size_t)m * I + 32 * (2 * bid + a) + c4), pk, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// 16 bytes of the activation another workgroup of this launch published write-through: two device-coherent 8-byte loads
__device__ __forceinline__ u32x4 g1zp_ld16(const unsigned short *p)
{
    const unsigned long long a = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return u32x4{(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
}

template <bool WIDE>
__device__ __forceinline__ void g1zp_down(int bx, int chunk, const unsigned short *__restrict__ x, const unsigned char *__restrict__ wz,
                                          const u32x2 *__restrict__ exc, float *__restrict__ out, int M, int N, int K, int KC, int n_tiles,
                                          int rec_stride, int exc_cap, unsigned *__restrict__ ready, unsigned expected, unsigned char *smem)
{
    constexpr int MT = 1, n_waves = 8;
    constexpr int DT = SJD_DTYPE_BF16;
    constexpr int D = G1Z_DEPTH, DP = D / 2, TL = D < 8 ? 8 : D;
    u32x4 *xl = reinterpret_cast<u32x4 *>(smem);
    const int k0 = chunk * KC;
    const int steps = min(KC, K - k0) / 16;
    const int pairs = (steps + 1) / 2, pairs_full = (KC / 16 + 1) / 2;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int t_out = bx * n_waves + w;
    const int t = t_out;
    const bool has_tile = t_out < N / 32;
    const size_t chunk_base = (size_t)chunk * n_tiles * pairs_full;
    const size_t tile_off = (rec_stride == 1) ? (size_t)(has_tile ? t : 0) * pairs : (size_t)(has_tile ? t : 0);
    const unsigned rsb = (unsigned)rec_stride * 1536u;
    const __amdgpu_buffer_rsrc_t wr = g1z_unit_rsrc(wz + (chunk_base + tile_off) * 1536, has_tile ? (unsigned)(pairs - 1) * rsb + 1536u : 0u);
    auto w_load = [&](int p) -> g1z_pair { return g1z_load(wr, (unsigned)lane, (unsigned)p * rsb); };
    // ---- run ahead: the header and the first ring of weight records travel while the producers of this K chunk are still at work
    g1z_pair ring[DP];
    const g1z_hraw hraw = g1z_header_load<WIDE>(exc, (size_t)chunk * n_tiles + (has_tile ? t : 0), lane, exc_cap);
#pragma unroll
    for (int u = 0; u < DP; ++u) ring[u] = w_load(u);
    // ---- the dependency edge: all 64-column slices of y inside [k0, k0 + 16 steps) are published
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(ready + chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 22)) { atomicAdd(&g1_pair_timeouts, 1u); break; }
        }
    }
    __syncthreads();
    // ---- the activation chunk, device-coherently (its producers stored it write-through)
    const int ppr = 2 * steps;
#ifdef G1ZP_LD8                   // (A/B aid: two 8-byte agent-scope atomic loads per piece instead of one 16-byte sc1 buffer load)
    for (int v = threadIdx.x; v < 32 * ppr; v += 512) {
        const int m = v / ppr, j = v - m * ppr;
        const u32x4 val = (m < M) ? g1zp_ld16(x + (size_t)m * K + k0 + 8 * j) : u32x4{0u, 0u, 0u, 0u};
        xl[(j >> 1) * 64 + g1_slot(j & 1, m & 31, j >> 1)] = val;
    }
#else
    {   // 16-byte sc1 loads through a descriptor over the M valid rows (rows beyond read as zero); up to six pieces per thread in flight
        const __amdgpu_buffer_rsrc_t xr = g1z_unit_rsrc(reinterpret_cast<const unsigned char *>(x), (unsigned)M * (unsigned)K * 2u);
        constexpr int NX = 6;
        for (int v0 = threadIdx.x; v0 < 32 * ppr; v0 += 512 * NX) {
            u32x4 val[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int v = v0 + 512 * i, m = v / ppr, j = v - m * ppr;
                val[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, (unsigned)(((size_t)m * K + k0 + 8 * j) * 2u), 0, 16 /* sc1 */);
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int v = v0 + 512 * i, m = v / ppr, j = v - m * ppr;
                if (v < 32 * ppr) xl[(j >> 1) * 64 + g1_slot(j & 1, m & 31, j >> 1)] = val[i];
            }
        }
    }
#endif
    __syncthreads();
    if (!has_tile) return;
    const g1z_hdr hd = g1z_header<WIDE>(hraw, lane, exc_cap);
    f32x16 acc[MT];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = 0.0f;
    auto a_read = [&](u32x4 (&a)[MT], int s, int u) { a[0] = xl[min(s, steps - 1) * 64 + g1_slot(lane >> 5, lane & 31, u)]; };
    for (int s0 = 0; s0 < steps; s0 += TL) {
        u32x4 a0[MT], a1[MT];
        a_read(a0, s0, 0);
#pragma unroll
        for (int u = 0; u < TL / 2; ++u) {
            const int sa = s0 + 2 * u, sb = sa + 1;
            g1z_pair &slot = ring[u % DP];
            if (sa < steps) {
                a_read(a1, sb, 2 * u + 1);
                const u32x4 b0 = g1z_operand<WIDE>(slot.lo.x, slot.lo.y, slot.c.x, (unsigned)sa, hd, lane);
                acc[0] = G1Mfma<DT>::mma(a0[0], b0, acc[0]);
            }
            u32x4 b1 = {0u, 0u, 0u, 0u};
            if (sb < steps) {
                if (u + 1 < TL / 2) a_read(a0, sb + 1, 2 * u + 2);
                b1 = g1z_operand<WIDE>(slot.lo.z, slot.lo.w, slot.c.y, (unsigned)sb, hd, lane);
            }
            slot = w_load(s0 / 2 + u + DP);
            if (sb < steps) acc[0] = G1Mfma<DT>::mma(a1[0], b1, acc[0]);
        }
    }
    float *o = out + ((size_t)chunk * 32) * N + (size_t)t_out * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        o[(size_t)m * N] = acc[0][r];
    }
}

struct g1zp_args {
    const unsigned short *x;          // [M, hidden] the MLP's input (the residual stream h)
    const unsigned char *wz_gu; const u32x2 *exc_gu; int stride_cap_gu;
    unsigned short *y;                // [M, I] silu(gate) * up
    const float *row_sumsq; int rs_slices; float rs_inv_hidden, rs_eps;
    const unsigned char *wz_dn; const u32x2 *exc_dn; int exc_cap_dn, rec_stride_dn;
    float *out;                       // [n_chunks, 32, hidden] split-K planes of the down projection
    int M, I, hidden, KC_dn, n_gu, n_bx, n_chunks;
    unsigned *ready;                  // [n_chunks + 1]: arrivals per K chunk of down, then the completion counter
};

template <bool WIDE_GU, bool WIDE_DN>
__global__ __launch_bounds__(512) void g1z_mlp_pair(const g1zp_args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float rsc[32];
    const int bid = blockIdx.x;
    if (bid < a.n_gu) {
        g1zp_gateup<WIDE_GU>(bid, a.x, a.wz_gu, a.exc_gu, a.y, a.M, a.I, a.hidden, a.stride_cap_gu, a.row_sumsq, a.rs_slices, a.rs_inv_hidden,
                             a.rs_eps, smem, rsc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this thread's write-through stores have reached the memory side
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(a.ready + (64 * bid) / a.KC_dn, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int units = a.n_bx * a.n_chunks;
    if (bid < units) {
        const int chunk = bid / a.n_bx, bx = bid % a.n_bx;
        const unsigned expected = (unsigned)(min(a.KC_dn, a.I - chunk * a.KC_dn) / 64);
        g1zp_down<WIDE_DN>(bx, chunk, a.y, a.wz_dn, a.exc_dn, a.out, a.M, a.hidden, a.I, a.KC_dn, a.hidden / 32, a.rec_stride_dn, a.exc_cap_dn,
                           a.ready, expected, smem);
    }
    // ---- re-arm: the workgroup that finishes last clears every counter (all waits of this launch are over by then)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(a.ready + a.n_chunks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            for (int c = 0; c <= a.n_chunks; ++c) __hip_atomic_store(a.ready + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// y [M <= 32, I] = silu(r gate(x)) * (r up(x)) and out [n_chunks, 32, hidden] = the split-K planes of y @ W_down^T in ONE launch: what
// sjd_gateup_silu_z followed by sjd_skinny_gemm_z write, bit for bit.  ready: (K / KC_dn rounded up) + 1 zero-initialised uint32, private to
// launches that cannot overlap (they re-arm themselves).  bf16, hidden = 4096 (the 7B / 8B architectures), KC_dn a multiple of 64, 8 column
// tiles per down workgroup, both launches' workgroups resident at once (<= resident_limit).
extern "C" int sjd_mlp_pair_z(const void *x, const void *wz_gu, const void *exc_gu, int exc_cap_gu, int step_major_gu, void *y,
                              const void *wz_dn, const void *exc_dn, int exc_cap_dn, int step_major_dn, float *out, int M, int I, int hidden,
                              int KC_dn, const sjd_row_norm *row_norm, unsigned *ready, int resident_limit, void *stream)
{
    if (!x || !wz_gu || !exc_gu || !y || !wz_dn || !exc_dn || !out || !ready || M < 1 || I < 64) return SJD_ERR_BAD_ARG;
    if (M > 32 || hidden != 4096 || (I % 64) != 0 || KC_dn < 64 || (KC_dn % 64) != 0 || KC_dn > 2560) return SJD_ERR_UNSUPPORTED;
    if (row_norm && (!row_norm->sumsq || row_norm->slices < 1 || row_norm->slices > 8 || row_norm->hidden < 1)) return SJD_ERR_BAD_ARG;
    auto capok = [](int c) { return c == 32 || c == 64 || c == 128; };
    if (!capok(exc_cap_gu) || !capok(exc_cap_dn)) return SJD_ERR_BAD_ARG;
    g1zp_args a;
    a.x = (const unsigned short *)x;
    a.wz_gu = (const unsigned char *)wz_gu; a.exc_gu = (const u32x2 *)exc_gu;
    a.stride_cap_gu = (step_major_gu ? 2 * (I / 32) : 1) | (exc_cap_gu << 16);
    a.y = (unsigned short *)y;
    a.row_sumsq = row_norm ? row_norm->sumsq : nullptr; a.rs_slices = row_norm ? row_norm->slices : 0;
    a.rs_inv_hidden = row_norm ? 1.0f / (float)row_norm->hidden : 0.f; a.rs_eps = row_norm ? row_norm->eps : 0.f;
    a.wz_dn = (const unsigned char *)wz_dn; a.exc_dn = (const u32x2 *)exc_dn; a.exc_cap_dn = exc_cap_dn;
    a.rec_stride_dn = step_major_dn ? hidden / 32 : 1;
    a.out = out;
    a.M = M; a.I = I; a.hidden = hidden; a.KC_dn = KC_dn;
    a.n_gu = I / 64; a.n_bx = (hidden / 32 + 7) / 8; a.n_chunks = (I + KC_dn - 1) / KC_dn;
    a.ready = ready;
    const int grid = a.n_gu > a.n_bx * a.n_chunks ? a.n_gu : a.n_bx * a.n_chunks;
    if (grid > resident_limit) return SJD_ERR_UNSUPPORTED;          // a waiting workgroup must never keep a producer from being scheduled
    const size_t lds_a = (size_t)2 * 64 * 1024, lds_b = (size_t)(KC_dn / 16) * 1024;
    const size_t lds = lds_a > lds_b ? lds_a : lds_b;
    hipStream_t s = (hipStream_t)stream;
#define SJD_PAIR_CASE(WG_, WD_) \
    if ((exc_cap_gu > 64) == WG_ && (exc_cap_dn > 64) == WD_) { \
        (void)hipFuncSetAttribute((const void *)g1z_mlp_pair<WG_, WD_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((g1z_mlp_pair<WG_, WD_>), dim3(grid), dim3(512), lds, s, a); \
        return hipGetLastError() == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH; \
    }
    SJD_PAIR_CASE(false, false) SJD_PAIR_CASE(false, true) SJD_PAIR_CASE(true, false) SJD_PAIR_CASE(true, true)
#undef SJD_PAIR_CASE
    return SJD_ERR_UNSUPPORTED;
}

extern "C" int sjd_mlp_pair_timeouts(void)
{
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g1_pair_timeouts), sizeof(v), 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)v;
}
