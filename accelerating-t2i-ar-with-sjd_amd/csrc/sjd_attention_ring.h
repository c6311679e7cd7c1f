// sjd_attention_ring.h -- K1 for the shapes whose query rows SHARE key tiles (grouped-query heads and / or several 16-row chunks: Emu3's
// GQA 32 / 8 with a draft window of 32, two to four prompts per forward), round 4.  Included by sjd_attention.hip (same helpers, same split /
// workspace layout, k1_combine unchanged); replaces k1_partial_shared as the default for those shapes.
//
// Why a new structure ("restructure, do not trim"): k1_partial_shared moved a tile HBM -> VGPRs -> ds_write -> barrier with
// three register sets "in flight".  Its ISA shows the compiler draining them: an `s_waitcnt vmcnt(0)` sits on the back edge of the unrolled
// loop, so every third tile waited for the two loads issued just before it -- a full HBM round trip per three tiles, whatever travels
// behind (that is the "a tile costs 1.4-1.9 us whatever is in flight" of round 3, and why six register sets measured equal to three).
// Here the tiles never touch a VGPR on their way in:
//   * LDS-DMA (`buffer_load_dwordx4 ... offen lds`, 1 KiB per wave instruction) writes K and V tiles straight into a RING of R slots of
//     32 keys (16 KiB per slot at D = 128: K 8 KiB + V 8 KiB); R - 1 tiles are in flight per workgroup (R = 6: 80 KiB per CU), counted with
//     hand-placed `s_waitcnt vmcnt(N)` -- the DMA has no destination register, so the compiler neither counts nor drains it;
//   * the LDS image is lane-linear (a DMA cannot pad rows), so bank conflicts are avoided by permuting the SOURCE: 16-byte piece p of key
//     row k lands at position p ^ (k & 15) of the K slot (fragment reads "lane = key, 16 B at piece 4 ks + g": 16 lanes, 16 positions) and
//     at position p ^ 2 (k & 7) of the V slot (transposed reads: 8 rows x 32 B per half wave: 16 positions);
//   * rows at or beyond the valid cache length are zeroed by the buffer descriptor's range check (num_records = valid rows): no select;
//   * Q arrives by DMA too (each wave its own 16 rows), so the kernel has NO register load the compiler would wait for with vmcnt(0);
//   * one raw s_barrier per tile: "my pieces of tile t have landed" (counted wait) -> barrier (everybody's have, and everybody is done with
//     tile t - 1) -> refill the slot of tile t - 1 with tile t + R - 1 -> compute tile t.
// The MFMA operands and rounding points of a tile are k1_partial_shared's (scores and sums fp32, P rounded once to the 16-bit operand type);
// the exponentials are taken as 2^((s - m) c1) in one fma + v_exp_f32, so the last fp32 bits differ from the round-3 kernel's.
#pragma once

template <int N> __device__ __forceinline__ void k1r_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one LDS-DMA piece: 64 lanes x 16 B from (descriptor base + voff) to LDS bytes [lds_addr + 16 * lane, + 16).  M0 is the compiler's: it is
// saved and restored inside the statement (guide 5.7); the s_nop covers the M0 write -> LDS-DMA read hazard.
__device__ __forceinline__ void k1r_dma16(u32x4 rsrc, unsigned voff, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_addr)
                 : "memory");
}

// raw buffer descriptor over `bytes` bytes from `p` (wave-uniform: every dword goes through readfirstlane); loads past `bytes` return zero
__device__ __forceinline__ u32x4 k1r_rsrc(const void *p, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)p;
    u32x4 r;
    r[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) & 0xffffu;       // stride 0: raw buffer
    r[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000u;                                                                         // 32-bit data format, no swizzle
    return r;
}

// the same as a compiler-visible descriptor (for __builtin_amdgcn_raw_buffer_load_*: loads the compiler counts)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t k1r_rsrc_b(const void *p, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

__device__ __forceinline__ unsigned k1r_lds_addr(const void *p)
{
    return (unsigned)(unsigned long)((__attribute__((address_space(3))) const unsigned char *)p);
}

// x (op) x[lane ^ 16] (op) ... over the four 16-lane groups of a wave: what `v = op(v, __shfl_xor(v, 16)); v = op(v, __shfl_xor(v, 32))` computes
// (the operands of every step are the same two values, and max / + are commutative: identical bits), on gfx950's row / half swaps
// instead of two ds_bpermute round trips through the LDS.
__device__ __forceinline__ float k1r_max_across_groups(float v)
{
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float k1r_sum_across_groups(float v)
{
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// NWV = waves per workgroup = (q head of the group, 16-row chunk) pairs: 4 or 8.  R = ring slots.
template <int DT, int D, int NWV, int R>
__global__ __launch_bounds__(64 * NWV) void k1_partial_ring(
    const unsigned short *__restrict__ q, const unsigned short *__restrict__ kc, const unsigned short *__restrict__ vc,
    const sjd_iter_params *__restrict__ params, const int *__restrict__ key_start,        // (among the first 16 dwords: preloaded into SGPRs)
    float *__restrict__ ws_o, float *__restrict__ ws_ml, int n_rows, int H, int H_kv, int S_max, int kv_len_arg, int n_split, int n_chunks, int B,
    int n_parts = 1)
{
    // n_parts (experiment, VERDICT r3 #1b): the (head, chunk) pairs of a (batch, kv head, split) spread over n_parts workgroups of NWV waves --
    // two 4-wave workgroups per CU instead of one 8-wave one, each with its own ring (the K/V tiles then travel to LDS twice)
    typedef typename Frag<DT>::vec vec;
    static_assert(D == 128, "the ring kernel is written for head_dim 128 (16 pieces of 16 bytes per row)");
    constexpr int KS = D / 32, DB = D / 16;
    constexpr int ROWB = D * 2;                         // bytes per key row (256)
    constexpr int TENSOR = K1_KT * ROWB;                // bytes per K or V tile (8 KiB)
    constexpr int SLOT = 2 * TENSOR;                    // a ring slot: K tile, then V tile
    constexpr int NP = TENSOR / 1024 / NWV;             // DMA pieces per wave, tile and tensor (1 with 8 waves, 2 with 4)
    constexpr int NI = 2 * NP;                          // DMA instructions per wave and tile
    static_assert(NP * NWV * 1024 == TENSOR, "the waves cover a tile exactly");
    static_assert(NI * (R - 1) + 4 <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(1024))) unsigned char k1r_lds[];      // [R slots][K | V] then [NWV][16 rows x 256 B] of Q
    SJD_TR(0);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = lane >> 4, c = lane & 15;
    const int G = H / H_kv;
    const int split = blockIdx.x / n_parts, part = blockIdx.x % n_parts, hkv = blockIdx.y, b = blockIdx.z;
    const int pair = part * NWV + w;
    const int head_in_group = pair % G, chunk = pair / G;     // wave = (q head of the group, 16-row chunk)
    const int head = hkv * G + head_in_group;
    int kv_base, n_total, kstart;
    k1_entry(params, key_start, b, kv_len_arg, n_rows, kv_base, n_total, kstart);
    const float scale = rsqrtf((float)D);

    // this wave's rows and tile range (identical to what k1_partial / k1_combine derive for its chunk)
    const int row0 = chunk * K1_ROWS;
    const int n_c = min(K1_ROWS, n_total - row0);
    const int kv_len = kv_base + row0;
    const int total = kv_len + max(n_c, 0);
    int t_lo, t_hi, eff_split, tps;
    k1_tile_range(kstart, total, n_split, t_lo, t_hi, eff_split, tps);
    const bool wave_on = split < eff_split;
    const int wt0 = wave_on ? t_lo + split * tps : 0, wt1 = wave_on ? min(t_hi, wt0 + tps) : 0;
    // the workgroup walks the union of its waves' ranges (they differ by at most a tile between the chunks)
    int bt0 = 1 << 30, bt1 = 0, total_max = 0;
    for (int ch = 0; ch < n_chunks; ++ch) {
        const int nc = min(K1_ROWS, n_total - ch * K1_ROWS), tot = kv_base + ch * K1_ROWS + max(nc, 0);
        int a, e, es, tp;
        k1_tile_range(kstart, tot, n_split, a, e, es, tp);
        if (split < es) { bt0 = min(bt0, a + split * tp); bt1 = max(bt1, min(e, a + split * tp + tp)); }
        total_max = max(total_max, tot);
    }
    if (bt1 <= bt0) {                         // no wave of this workgroup has a tile in this split
        if (wave_on) k1_store_empty_partial<D>(ws_o, ws_ml, ((((size_t)b * H + head) * n_chunks + chunk) * n_split + split) * K1_ROWS, c, g);
        return;
    }
    SJD_TR(1);                    // tile ranges known

    const unsigned ring0 = k1r_lds_addr(k1r_lds);
    const unsigned qlds = ring0 + R * SLOT + (unsigned)w * (K1_ROWS * ROWB);
    // ---- Q: this wave's 16 rows, 4 pieces; row r, 16-byte piece p -> position p ^ r of LDS row r (the K swizzle)
    {
        const u32x4 qr = k1r_rsrc(q, (unsigned)B * (unsigned)n_rows * (unsigned)H * (unsigned)ROWB);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 4 * j + g;                                       // LDS row of this lane
            const int grow = min(row0 + r, n_rows - 1);                    // (rows beyond the window: any valid row, zeroed below)
            const unsigned voff = (unsigned)(((b * n_rows + grow) * H + head) * ROWB + ((c ^ r) * 16));
            k1r_dma16(qr, voff, qlds + j * 1024);
        }
    }
    // ---- K / V descriptors: rows >= total_max read as zero (0 * NaN inside the MFMA, see k1_partial)
    const size_t slab = ((size_t)b * H_kv + hkv) * (size_t)S_max * D;
    const u32x4 kr = k1r_rsrc(kc + slab, (unsigned)total_max * ROWB), vr = k1r_rsrc(vc + slab, (unsigned)total_max * ROWB);
    unsigned ck[NP], cv[NP];                  // lane-constant source offsets inside a tile
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int key = 4 * (w + NWV * i) + g;                             // key row (inside the tile) of this lane's piece
        ck[i] = (unsigned)(key * ROWB + ((c ^ (key & 15)) * 16));
        cv[i] = (unsigned)(key * ROWB + ((c ^ (2 * (key & 7))) * 16));
    }
    auto issue = [&](int t, int slot) {       // tile t -> ring slot; beyond the workgroup's range: out-of-range offsets (zeros, no traffic)
        const unsigned tb = t < bt1 ? (unsigned)t * TENSOR : 0x7fff0000u;
        const unsigned dst = ring0 + (unsigned)slot * SLOT + (unsigned)w * 1024;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            k1r_dma16(kr, tb + ck[i], dst + i * NWV * 1024);
            k1r_dma16(vr, tb + cv[i], dst + TENSOR + i * NWV * 1024);
        }
    };
#pragma unroll
    for (int i = 0; i < R - 1; ++i) issue(bt0 + i, i);
    k1r_wait_vmcnt<NI *(R - 1)>();            // this wave's Q pieces have landed (nobody else reads them)

    vec qf[KS];
    {
        const bool rv = (c < n_c);
        const unsigned char *qrow = k1r_lds + R * SLOT + w * (K1_ROWS * ROWB) + c * ROWB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4 x = *reinterpret_cast<const u32x4 *>(qrow + (((4 * ks + g) ^ c) * 16));
            qf[ks] = as_frag<vec>(rv ? x : u32x4{0u, 0u, 0u, 0u});
        }
    }
    // lane-constant read offsets inside a slot: K fragment (key 16 kb + c, piece 4 ks + g) and V transposed block (row 4 g + c / 4 (+ 16),
    // piece 2 db + (c & 3) / 2, half c & 1)
    unsigned kofs[KS], vofs[DB];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kofs[ks] = (unsigned)(c * ROWB + (((4 * ks + g) ^ c) * 16));
    {
        const int vrow = 4 * g + (c >> 2);
#pragma unroll
        for (int db = 0; db < DB; ++db)
            vofs[db] = (unsigned)(TENSOR + vrow * ROWB + (((2 * db + ((c & 3) >> 1)) ^ (2 * (vrow & 7))) * 16) + 8 * (c & 1));
    }

    float m_run = -INFINITY, l_run = 0.0f;            // m_run: the running maximum of the RAW (unscaled) scores
    const float c1 = scale * 1.4426950408889634f;     // log2(e) / sqrt(D)
    f32x4 o_acc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) o_acc[db] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (timing probes, results invalid: -DK1R_NO_COMPUTE the DMA / barrier pipeline alone, -DK1R_NO_DMA the arithmetic of a tile on whatever the
    //  LDS holds, -DK1R_NO_SOFTMAX LDS reads + MFMAs without the online-softmax VALU chain, -DK1R_NO_BARRIER no workgroup barrier)
    auto compute_tile = [&](int t, int slot) {
#ifdef K1R_NO_COMPUTE
        return;
#endif
        if (t >= wt0 && t < wt1) {
            const unsigned char *sl = k1r_lds + slot * SLOT;
            // every LDS read of the tile is issued up front -- 8 K fragments, then the 16 transposed V blocks, which land under the QK^T
            // MFMAs and the softmax arithmetic (the compiler, left alone, recycled one register quad per fragment: eight serial LDS round trips)
            u32x4 kf[2][KS];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = *reinterpret_cast<const u32x4 *>(sl + kb * 16 * ROWB + kofs[ks]);
            u32x2 vlo[DB], vhi[DB];
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                vlo[db] = lds_tr_read(reinterpret_cast<const unsigned short *>(sl + vofs[db]));
                vhi[db] = lds_tr_read(reinterpret_cast<const unsigned short *>(sl + vofs[db] + 16 * ROWB));
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 st[2];
            st[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            st[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {             // two independent accumulation chains, interleaved
                st[0] = Frag<DT>::mfma(as_frag<vec>(kf[0][ks]), qf[ks], st[0]);
                st[1] = Frag<DT>::mfma(as_frag<vec>(kf[1][ks]), qf[ks], st[1]);
            }
            // The online softmax works on the RAW scores: p = exp2((s - m) * c1) with c1 = log2(e) / sqrt(D) -- one fma and one v_exp_f32 per score
            // (v_exp_f32 IS 2^x) instead of scale, subtract, multiply by log2(e), v_exp; the running maximum m_run is kept in raw units
            // (max commutes with the positive scale) and converted once, when the partial is published.
            // interior tile (every row of the chunk sees every key of it): no visibility arithmetic
            const bool interior = (t * K1_KT >= kstart) && (t * K1_KT + K1_KT - 1 <= kv_len) && (t * K1_KT + K1_KT <= total);
            if (!interior) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = t * K1_KT + 16 * kb + 4 * g + r;
                        const bool vis = (key >= kstart) && (key <= kv_len + c) && (key < total);
                        st[kb][r] = vis ? st[kb][r] : -INFINITY;
                    }
            }
#ifdef K1R_NO_SOFTMAX
            const float alpha = 1.0f;
            u32x4 pw{__float_as_uint(st[0][0]), __float_as_uint(st[0][1]), __float_as_uint(st[1][2]), __float_as_uint(st[1][3])};
            const vec pfrag = as_frag<vec>(pw);
#else
            float mx = fmaxf(fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3])), fmaxf(fmaxf(st[1][0], st[1][1]), fmaxf(st[1][2], st[1][3])));
            mx = k1r_max_across_groups(mx);               // the four lane groups of a row: xor 16, xor 32 (v_permlane*_swap, no LDS round trip)
            const float m_new = fmaxf(m_run, mx);
            const float m_safe = (m_new == -INFINITY) ? 0.0f : m_new;
            const float nm = -m_safe * c1;
            const float alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m_run, c1, nm));       // (m_run = -inf: 2^-inf = 0)
            float pv[8];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[4 * kb + r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][r], c1, nm));
            float rs = ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
            rs = k1r_sum_across_groups(rs);
            l_run = __builtin_fmaf(l_run, alpha, rs);
            m_run = m_new;
            u32x4 pw;
#pragma unroll
            for (int i = 0; i < 4; ++i) pw[i] = k1_cvt_pk<DT>(pv[2 * i], pv[2 * i + 1]);
            const vec pfrag = as_frag<vec>(pw);
#endif
            // the running output is rescaled only when some row's maximum moved (alpha == 1 exactly otherwise: 2^0)
            const bool rescale = __builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0ull;
            if (rescale) {
#pragma unroll
                for (int db = 0; db < DB; ++db) { o_acc[db][0] *= alpha; o_acc[db][1] *= alpha; o_acc[db][2] *= alpha; o_acc[db][3] *= alpha; }
            }
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const u32x4 vv{vlo[db][0], vlo[db][1], vhi[db][0], vhi[db][1]};
                o_acc[db] = Frag<DT>::mfma(as_frag<vec>(vv), pfrag, o_acc[db]);
            }
        }
    };

    // ---- the ring: tile r (relative) lives in slot r % R; R - 1 tiles in flight
    int slot = 0, refill = R - 1;             // slot of tile t; slot that tile t + R - 1 goes into (= the one tile t - 1 has just left)
    for (int t = bt0; t < bt1; ++t) {
#ifndef K1R_NO_DMA
        k1r_wait_vmcnt<NI *(R - 2)>();        // this wave's pieces of tile t have landed (tiles t + 1 .. t + R - 2 may still travel)
#endif
#ifndef K1R_NO_BARRIER
        __builtin_amdgcn_s_barrier();         // ... and everybody's; everybody is done reading tile t - 1
#endif
        if (t == bt0) SJD_TR(2);              // first tile in LDS
#ifndef K1R_NO_DMA
        issue(t + R - 1, refill);
#endif
        compute_tile(t, slot);
        refill = slot;
        slot = slot + 1 == R ? 0 : slot + 1;
    }
    SJD_TR(3);                    // key loop done
    k1r_wait_vmcnt<0>();          // (the tail's out-of-range pieces: nothing may still be writing this workgroup's LDS when it exits)
    if (!wave_on) return;
    // the wave covered every tile of its split: its (m, l, O) is the split partial
    const size_t slot0 = ((((size_t)b * H + head) * n_chunks + chunk) * n_split + split) * K1_ROWS;
#pragma unroll
    for (int db = 0; db < DB; ++db) *reinterpret_cast<f32x4 *>(ws_o + (slot0 + c) * D + 16 * db + 4 * g) = o_acc[db];
    if (g == 0) { ws_ml[(slot0 + c) * 2] = m_run * scale; ws_ml[(slot0 + c) * 2 + 1] = l_run; }       // (m in the units k1_combine merges in)
#ifdef SJD_TRACE
    SJD_TR(4);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    SJD_TR(5);                    // partial stored
    SJD_TR(6); SJD_TR(7);
#endif
}

// ------------------------------------------------------------------------------------------------ K1 (D-split, no key split, no combine)
// Round 4, the multi-head 16-row window (Lumina / Anole: B * H = 64 (batch, head) pairs, nothing shared between them): k1_partial fills the
// chip by splitting the KEYS of a pair over four workgroups, and their partial (m, l, O) then need a second dependent launch (k1_combine,
// 5 us of a 14 us pair, 3 us of them launch cost) or an in-kernel exchange that measured the same (round 3).  Here the four workgroups of a
// pair split the OUTPUT COLUMNS instead: each computes the scores of ALL keys (the K stream is read by all four -- from the XCD's L2 after
// the first, the block -> XCD map puts the four siblings on one XCD, dispatched back to back) and multiplies them with ITS 32 columns of V
// only; its eight waves walk the key tiles as k1_partial's do (K rows straight into MFMA fragments, V slice through a per-wave LDS tile),
// merge their eight (m, l, O) through LDS and write the NORMALISED 16-bit output: one launch, no workspace, no exchange between
// workgroups.  Cost: QK^T and the softmax run four times (16-row windows: the MFMA pipe idles anyway) and every CU pulls K + V / 4
// (2.5 x the bytes through its L1, 1 x from HBM).
template <int DT, int D, int NW, int DS>
__global__ __launch_bounds__(64 * NW) void k1_dsplit(
    const unsigned short *__restrict__ q, const unsigned short *__restrict__ kc, const unsigned short *__restrict__ vc,
    const sjd_iter_params *__restrict__ params, const int *__restrict__ key_start, unsigned short *__restrict__ out,
    int n_rows, int H, int H_kv, int S_max, int kv_len_arg, int n_chunks, int B)
{
    typedef typename Frag<DT>::vec vec;
    constexpr int KS = D / 32;
    constexpr int DW = D / DS;                // output columns of this workgroup
    constexpr int DB = DW / 16;               // 16-wide d blocks of it
    constexpr int VROW = DW + 8;              // padded LDS row of the V slice (elements)
    constexpr int V_BYTES = NW * K1_KT * VROW * 2;
    constexpr int R_BYTES = NW * K1_ROWS * (DW + K1_RPAD + 2) * 4;
    __shared__ __attribute__((aligned(16))) unsigned char arena[V_BYTES > R_BYTES ? V_BYTES : R_BYTES];
    unsigned short (*v_lds)[K1_KT * VROW] = reinterpret_cast<unsigned short (*)[K1_KT * VROW]>(arena);
    float (*red_o)[K1_ROWS][DW + K1_RPAD] = reinterpret_cast<float (*)[K1_ROWS][DW + K1_RPAD]>(arena);
    float (*red_ml)[K1_ROWS][2] = reinterpret_cast<float (*)[K1_ROWS][2]>(arena + NW * K1_ROWS * (DW + K1_RPAD) * 4);

    // block -> (pair, column slice): blocks id, id + 8, id + 16, ... run on one XCD; the DS siblings of a pair are consecutive among them
    const int n_pairs = n_chunks * H * B;
    int pair, dq;
    if ((n_pairs & 7) == 0) {
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        dq = slot % DS;
        pair = (slot / DS) * 8 + xcd;
    } else {
        dq = blockIdx.x % DS;
        pair = blockIdx.x / DS;
    }
    const int chunk = pair % n_chunks, head = (pair / n_chunks) % H, b = pair / (n_chunks * H);
    const int G = H / H_kv, hkv = head / G;
    int kv_base, n_total, kstart;
    k1_entry(params, key_start, b, kv_len_arg, n_rows, kv_base, n_total, kstart);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int row0 = chunk * K1_ROWS;
    const int n_c = min(K1_ROWS, n_total - row0);
    const int kv_len = kv_base + row0;
    const int total = kv_len + max(n_c, 0);
    const float scale = rsqrtf((float)D);
    const int t_lo = kstart / K1_KT, t_hi = (total + K1_KT - 1) / K1_KT;

    vec qf[KS];
    {
        const bool rv = (c < n_c);
        const unsigned short *qp = q + (((size_t)b * n_rows + (row0 + (rv ? c : 0))) * H + head) * D + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 x = rv ? *reinterpret_cast<const u32x4 *>(qp + 32 * ks) : u32x4{0, 0, 0, 0};
            qf[ks] = as_frag<vec>(x);
        }
    }
    const unsigned short *kbase = kc + ((size_t)b * H_kv + hkv) * (size_t)S_max * D;
    const unsigned short *vbase = vc + ((size_t)b * H_kv + hkv) * (size_t)S_max * D + dq * DW;

    float m_run = -INFINITY, l_run = 0.0f;
    f32x4 o_acc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) o_acc[db] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int VP = K1_KT * DW / (64 * 8);    // 16-B pieces per lane for one V slice tile
    constexpr int LPR = DW / 8;                  // lanes per V row
    static_assert(VP >= 1 && VP * 64 * 8 == K1_KT * DW, "a wave covers the V slice tile exactly");
    u32x4 kreg[2][KS], kn[2][KS], vstage[VP], vstage2[VP];
    unsigned short *vl = v_lds[w];
    auto load_tile = [&](int t, u32x4 (&kd)[2][KS], u32x4 (&vd)[VP]) {
        const unsigned short *kt = kbase + (size_t)(t * K1_KT) * D;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                kd[kb][ks] = *reinterpret_cast<const u32x4 *>(kt + (size_t)(16 * kb + c) * D + 32 * ks + 8 * g);
        const unsigned short *vt = vbase + (size_t)(t * K1_KT) * D;
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            const int idx = i * 64 + lane;
            vd[i] = *reinterpret_cast<const u32x4 *>(vt + (size_t)(idx / LPR) * D + 8 * (idx % LPR));
        }
    };
    auto store_v = [&](int t, u32x4 (&vd)[VP]) {          // rows of keys >= total are zeroed (0 * NaN inside the MFMA, see k1_partial)
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            const int idx = i * 64 + lane;
            const bool live = (t * K1_KT + idx / LPR) < total;
            *reinterpret_cast<u32x4 *>(vl + (idx / LPR) * VROW + 8 * (idx % LPR)) = live ? vd[i] : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto compute_tile = [&](int t) {
        f32x4 st[2];
        st[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        st[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            st[0] = Frag<DT>::mfma(as_frag<vec>(kreg[0][ks]), qf[ks], st[0]);
            st[1] = Frag<DT>::mfma(as_frag<vec>(kreg[1][ks]), qf[ks], st[1]);
        }
        const bool interior = (t * K1_KT >= kstart) && (t * K1_KT + K1_KT - 1 <= kv_len) && (t * K1_KT + K1_KT <= total);
        float mx = -INFINITY;
        if (interior) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float sv = st[kb][r] * scale; st[kb][r] = sv; mx = fmaxf(mx, sv); }
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * K1_KT + 16 * kb + 4 * g + r;
                    const bool vis = (key >= kstart) && (key <= kv_len + c) && (key < total);
                    const float sv = vis ? st[kb][r] * scale : -INFINITY;
                    st[kb][r] = sv;
                    mx = fmaxf(mx, sv);
                }
        }
        mx = k1r_max_across_groups(mx);
        const float m_new = fmaxf(m_run, mx);
        const float m_safe = (m_new == -INFINITY) ? 0.0f : m_new;
        const float alpha = __expf(m_run - m_safe);
        float rs = 0.0f;
        unsigned short pb[8];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(st[kb][r] - m_safe);
                rs += pv;
                pb[4 * kb + r] = Frag<DT>::cvt(pv);
            }
        rs = k1r_sum_across_groups(rs);
        l_run = l_run * alpha + rs;
        m_run = m_new;
        u32x4 pw;
        pw[0] = pb[0] | ((unsigned)pb[1] << 16);
        pw[1] = pb[2] | ((unsigned)pb[3] << 16);
        pw[2] = pb[4] | ((unsigned)pb[5] << 16);
        pw[3] = pb[6] | ((unsigned)pb[7] << 16);
        const vec pfrag = as_frag<vec>(pw);
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const unsigned short *a0 = vl + (4 * g + (c >> 2)) * VROW + 16 * db + 4 * (c & 3);
            const u32x2 lo = lds_tr_read(a0), hi = lds_tr_read(a0 + 16 * VROW);
            const u32x4 vv{lo[0], lo[1], hi[0], hi[1]};
            f32x4 acc = o_acc[db];
            acc[0] *= alpha; acc[1] *= alpha; acc[2] *= alpha; acc[3] *= alpha;
            o_acc[db] = Frag<DT>::mfma(as_frag<vec>(vv), pfrag, acc);
        }
    };
    auto adopt_next = [&](int tn, u32x4 (&vd)[VP]) {
        store_v(tn, vd);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kreg[kb][ks] = kn[kb][ks];
    };
    // wave w walks tiles t_lo + w, + NW, ...; two of them are in flight before the first is waited for (k1_partial)
    int t = t_lo + w;
    if (t < t_hi) {
        load_tile(t, kreg, vstage);
        if (t + NW < t_hi) {
            load_tile(t + NW, kn, vstage2);
            store_v(t, vstage);
            compute_tile(t);
            adopt_next(t + NW, vstage2);
            t += NW;
        } else {
            store_v(t, vstage);
        }
    }
    for (; t < t_hi; t += NW) {
        const int tn = t + NW;
        const bool has_next = tn < t_hi;
        if (has_next) load_tile(tn, kn, vstage);
        compute_tile(t);
        if (has_next) adopt_next(tn, vstage);
    }
    // ---- merge the eight key parts through LDS and write the normalised rows of this column slice
    __syncthreads();
    if (g == 0) { red_ml[w][c][0] = m_run; red_ml[w][c][1] = l_run; }
#pragma unroll
    for (int db = 0; db < DB; ++db) *reinterpret_cast<f32x4 *>(&red_o[w][c][16 * db + 4 * g]) = o_acc[db];
    __syncthreads();
    constexpr int D4 = DW / 4;
    for (int u = threadIdx.x; u < K1_ROWS * D4; u += 64 * NW) {
        const int row = u / D4, d = (u % D4) * 4;
        const int grow = row0 + row;
        if (grow >= n_rows) continue;
        float mk[NW], lk[NW];
        float4 ok[NW];
#pragma unroll
        for (int kp = 0; kp < NW; ++kp) {                      // every LDS read of this thread is issued before the first use
            const float2 v = *reinterpret_cast<const float2 *>(&red_ml[kp][row][0]);
            ok[kp] = *reinterpret_cast<const float4 *>(&red_o[kp][row][d]);
            mk[kp] = v.x;
            lk[kp] = v.y;
        }
        float M = -INFINITY;
#pragma unroll
        for (int kp = 0; kp < NW; ++kp) M = fmaxf(M, mk[kp]);
        const float Ms = (M == -INFINITY) ? 0.0f : M;
        float L = 0.f, O0 = 0.f, O1 = 0.f, O2 = 0.f, O3 = 0.f;
#pragma unroll
        for (int kp = 0; kp < NW; ++kp) {
            const float wgt = __expf(mk[kp] - Ms);
            L += wgt * lk[kp];
            O0 += wgt * ok[kp].x; O1 += wgt * ok[kp].y; O2 += wgt * ok[kp].z; O3 += wgt * ok[kp].w;
        }
        const float inv = L > 0.f ? 1.0f / L : 0.0f;
        uint2 pk{0u, 0u};
        if (grow < n_total) {
            pk.x = (unsigned)Frag<DT>::cvt(O0 * inv) | ((unsigned)Frag<DT>::cvt(O1 * inv) << 16);
            pk.y = (unsigned)Frag<DT>::cvt(O2 * inv) | ((unsigned)Frag<DT>::cvt(O3 * inv) << 16);
        }
        *reinterpret_cast<uint2 *>(out + (((size_t)b * n_rows + grow) * H + head) * D + dq * DW + d) = pk;
    }
}

// ------------------------------------------------------------------------------------------------ K1 (D-split, deep prefetch)
// k1_dsplit with PF key tiles in flight per wave (VERDICT r3 #1c: ">= 4 tiles in flight per wave").  In k1_dsplit a wave walks ceil(tiles / 8)
// key tiles (5 at kv 1216, 10 at 2368) with ONE tile requested ahead: its time is (tiles per wave) x (a memory round trip / 2).  Here a wave
// keeps PF tiles in PF register sets: every load goes through a buffer descriptor over the valid rows and is UNCONDITIONAL (a tile beyond the
// wave's last one is requested at an out-of-range offset: zeros, no traffic), and the loop is unrolled by PF with no exit in between, so the
// compiler's s_waitcnt counts stay exact (vmcnt(10 (PF - 1)) in front of a tile's first use) -- no drain at the back edge.  Rows at or beyond
// the valid length read as zero through the descriptor (no select while staging V).  Arithmetic of a tile: k1_dsplit's.
template <int DT, int D, int NW, int DS, int PF>
__global__ __launch_bounds__(64 * NW) void k1_dsplit_pf(
    const unsigned short *__restrict__ q, const unsigned short *__restrict__ kc, const unsigned short *__restrict__ vc,
    const sjd_iter_params *__restrict__ params, const int *__restrict__ key_start, unsigned short *__restrict__ out,
    int n_rows, int H, int H_kv, int S_max, int kv_len_arg, int n_chunks, int B)
{
    typedef typename Frag<DT>::vec vec;
    constexpr int KS = D / 32;
    constexpr int DW = D / DS, DB = DW / 16, VROW = DW + 8;
    constexpr int V_BYTES = NW * K1_KT * VROW * 2;
    constexpr int R_BYTES = NW * K1_ROWS * (DW + K1_RPAD + 2) * 4;
    __shared__ __attribute__((aligned(16))) unsigned char arena[V_BYTES > R_BYTES ? V_BYTES : R_BYTES];
    unsigned short (*v_lds)[K1_KT * VROW] = reinterpret_cast<unsigned short (*)[K1_KT * VROW]>(arena);
    float (*red_o)[K1_ROWS][DW + K1_RPAD] = reinterpret_cast<float (*)[K1_ROWS][DW + K1_RPAD]>(arena);
    float (*red_ml)[K1_ROWS][2] = reinterpret_cast<float (*)[K1_ROWS][2]>(arena + NW * K1_ROWS * (DW + K1_RPAD) * 4);

    const int n_pairs = n_chunks * H * B;
    int pair, dq;
    if ((n_pairs & 7) == 0) {
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        dq = slot % DS;
        pair = (slot / DS) * 8 + xcd;
    } else {
        dq = blockIdx.x % DS;
        pair = blockIdx.x / DS;
    }
    const int chunk = pair % n_chunks, head = (pair / n_chunks) % H, b = pair / (n_chunks * H);
    const int G = H / H_kv, hkv = head / G;
    int kv_base, n_total, kstart;
    k1_entry(params, key_start, b, kv_len_arg, n_rows, kv_base, n_total, kstart);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = lane >> 4, c = lane & 15;
    const int row0 = chunk * K1_ROWS;
    const int n_c = min(K1_ROWS, n_total - row0);
    const int kv_len = kv_base + row0;
    const int total = kv_len + max(n_c, 0);
    const float scale = rsqrtf((float)D);
    const int t_lo = kstart / K1_KT, t_hi = (total + K1_KT - 1) / K1_KT;

    vec qf[KS];
    {
        const bool rv = (c < n_c);
        const unsigned short *qp = q + (((size_t)b * n_rows + (row0 + (rv ? c : 0))) * H + head) * D + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 x = rv ? *reinterpret_cast<const u32x4 *>(qp + 32 * ks) : u32x4{0, 0, 0, 0};
            qf[ks] = as_frag<vec>(x);
        }
    }
    const size_t slab = ((size_t)b * H_kv + hkv) * (size_t)S_max * D;
    const unsigned valid_bytes = (unsigned)max(total, 0) * (unsigned)(D * 2);
    const __amdgpu_buffer_rsrc_t kr = k1r_rsrc_b(kc + slab, valid_bytes), vr = k1r_rsrc_b(vc + slab, valid_bytes);

    float m_run = -INFINITY, l_run = 0.0f;
    f32x4 o_acc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) o_acc[db] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int VP = K1_KT * DW / (64 * 8), LPR = DW / 8;
    static_assert(VP >= 1 && VP * 64 * 8 == K1_KT * DW, "a wave covers the V slice tile exactly");
    unsigned short *vl = v_lds[w];
    // lane-constant byte offsets inside a tile: K fragment pieces (key 16 kb + c, d = 32 ks + 8 g), V slice pieces
    unsigned kof[2][KS], vof[VP];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kof[kb][ks] = (unsigned)(((16 * kb + c) * D + 32 * ks + 8 * g) * 2);
#pragma unroll
    for (int i = 0; i < VP; ++i) { const int idx = i * 64 + lane; vof[i] = (unsigned)(((idx / LPR) * D + dq * DW + 8 * (idx % LPR)) * 2); }

    u32x4 kq[PF][2][KS], vq[PF][VP];
    auto load_tile = [&](int t, u32x4 (&kd)[2][KS], u32x4 (&vd)[VP]) {
        const unsigned tb = t < t_hi ? (unsigned)t * (unsigned)(K1_KT * D * 2) : 0x7fff0000u;      // beyond the wave's last tile: out of range
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kd[kb][ks] = __builtin_amdgcn_raw_buffer_load_b128(kr, tb + kof[kb][ks], 0, 0);
#pragma unroll
        for (int i = 0; i < VP; ++i) vd[i] = __builtin_amdgcn_raw_buffer_load_b128(vr, tb + vof[i], 0, 0);
    };
    auto compute_tile = [&](int t, u32x4 (&kd)[2][KS], u32x4 (&vd)[VP]) {
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            const int idx = i * 64 + lane;
            *reinterpret_cast<u32x4 *>(vl + (idx / LPR) * VROW + 8 * (idx % LPR)) = vd[i];
        }
        f32x4 st[2];
        st[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        st[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            st[0] = Frag<DT>::mfma(as_frag<vec>(kd[0][ks]), qf[ks], st[0]);
            st[1] = Frag<DT>::mfma(as_frag<vec>(kd[1][ks]), qf[ks], st[1]);
        }
        const bool interior = (t * K1_KT >= kstart) && (t * K1_KT + K1_KT - 1 <= kv_len) && (t * K1_KT + K1_KT <= total);
        float mx = -INFINITY;
        if (interior) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float sv = st[kb][r] * scale; st[kb][r] = sv; mx = fmaxf(mx, sv); }
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * K1_KT + 16 * kb + 4 * g + r;
                    const bool vis = (key >= kstart) && (key <= kv_len + c) && (key < total);
                    const float sv = vis ? st[kb][r] * scale : -INFINITY;
                    st[kb][r] = sv;
                    mx = fmaxf(mx, sv);
                }
        }
        mx = k1r_max_across_groups(mx);
        const float m_new = fmaxf(m_run, mx);
        const float m_safe = (m_new == -INFINITY) ? 0.0f : m_new;
        const float alpha = __expf(m_run - m_safe);
        float rs = 0.0f;
        unsigned short pb[8];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(st[kb][r] - m_safe);
                rs += pv;
                pb[4 * kb + r] = Frag<DT>::cvt(pv);
            }
        rs = k1r_sum_across_groups(rs);
        l_run = l_run * alpha + rs;
        m_run = m_new;
        u32x4 pw;
        pw[0] = pb[0] | ((unsigned)pb[1] << 16);
        pw[1] = pb[2] | ((unsigned)pb[3] << 16);
        pw[2] = pb[4] | ((unsigned)pb[5] << 16);
        pw[3] = pb[6] | ((unsigned)pb[7] << 16);
        const vec pfrag = as_frag<vec>(pw);
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const unsigned short *a0 = vl + (4 * g + (c >> 2)) * VROW + 16 * db + 4 * (c & 3);
            const u32x2 lo = lds_tr_read(a0), hi = lds_tr_read(a0 + 16 * VROW);
            const u32x4 vv{lo[0], lo[1], hi[0], hi[1]};
            f32x4 acc = o_acc[db];
            acc[0] *= alpha; acc[1] *= alpha; acc[2] *= alpha; acc[3] *= alpha;
            o_acc[db] = Frag<DT>::mfma(as_frag<vec>(vv), pfrag, acc);
        }
    };
    // wave w walks tiles t_lo + w, + NW, ...: PF of them in flight at any time
    const int t0 = t_lo + w;
    // (the scheduling barriers pin the ISSUE ORDER of the register sets' loads: left alone the scheduler issued the set that is used first
    //  LAST -- closest to its use -- and the first tile of every trip then waited for vmcnt(0))
#pragma unroll
    for (int s = 0; s < PF; ++s) { load_tile(t0 + s * NW, kq[s], vq[s]); __builtin_amdgcn_sched_barrier(0); }
    const int n_mine = t0 < t_hi ? (t_hi - t0 + NW - 1) / NW : 0;
    for (int i0 = 0; i0 < n_mine; i0 += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int t = t0 + (i0 + s) * NW;
            if (t < t_hi) compute_tile(t, kq[s], vq[s]);             // (uniform; no vector-memory instruction inside: the waitcnt counts stay exact)
            __builtin_amdgcn_sched_barrier(0);
            load_tile(t + PF * NW, kq[s], vq[s]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- merge the eight key parts through LDS and write the normalised rows of this column slice (k1_dsplit)
    __syncthreads();
    if (g == 0) { red_ml[w][c][0] = m_run; red_ml[w][c][1] = l_run; }
#pragma unroll
    for (int db = 0; db < DB; ++db) *reinterpret_cast<f32x4 *>(&red_o[w][c][16 * db + 4 * g]) = o_acc[db];
    __syncthreads();
    constexpr int D4 = DW / 4;
    for (int u = threadIdx.x; u < K1_ROWS * D4; u += 64 * NW) {
        const int row = u / D4, d = (u % D4) * 4;
        const int grow = row0 + row;
        if (grow >= n_rows) continue;
        float mk[NW], lk[NW];
        float4 ok[NW];
#pragma unroll
        for (int kp = 0; kp < NW; ++kp) {
            const float2 v = *reinterpret_cast<const float2 *>(&red_ml[kp][row][0]);
            ok[kp] = *reinterpret_cast<const float4 *>(&red_o[kp][row][d]);
            mk[kp] = v.x;
            lk[kp] = v.y;
        }
        float M = -INFINITY;
#pragma unroll
        for (int kp = 0; kp < NW; ++kp) M = fmaxf(M, mk[kp]);
        const float Ms = (M == -INFINITY) ? 0.0f : M;
        float L = 0.f, O0 = 0.f, O1 = 0.f, O2 = 0.f, O3 = 0.f;
#pragma unroll
        for (int kp = 0; kp < NW; ++kp) {
            const float wgt = __expf(mk[kp] - Ms);
            L += wgt * lk[kp];
            O0 += wgt * ok[kp].x; O1 += wgt * ok[kp].y; O2 += wgt * ok[kp].z; O3 += wgt * ok[kp].w;
        }
        const float inv = L > 0.f ? 1.0f / L : 0.0f;
        uint2 pk{0u, 0u};
        if (grow < n_total) {
            pk.x = (unsigned)Frag<DT>::cvt(O0 * inv) | ((unsigned)Frag<DT>::cvt(O1 * inv) << 16);
            pk.y = (unsigned)Frag<DT>::cvt(O2 * inv) | ((unsigned)Frag<DT>::cvt(O3 * inv) << 16);
        }
        *reinterpret_cast<uint2 *>(out + (((size_t)b * n_rows + grow) * H + head) * D + dq * DW + d) = pk;
    }
}

// ------------------------------------------------------------------------------------------------ K1 (D-split over an LDS-DMA ring)
// The column split with its K / V traffic on the ring kernel's transport.  k1_dsplit loads K rows as MFMA fragments straight from memory:
// every wave instruction touches 16 rows x 64 B -- half cache lines -- and a CU that has to pull 2.5 x the bytes of the key split is bound by
// its own load path (deeper register prefetch measured slower, profiles/r4_k1_dsplit_pf_ab.txt).  Here the tiles of a (batch, head) arrive by
// LDS-DMA in FULL rows (1 KiB per wave instruction = four 256-byte key rows) into a ring of 16 tile slots (K 8 KiB swizzled as in
// k1_partial_ring + the workgroup's 64-byte V slices 2 KiB); a ROUND is eight tiles, one per wave; round j + 1 travels while round j is
// computed; one barrier per round.  Each tile is read from LDS once (by the wave that owns it).  Q is staged by DMA through the second
// round's first slot before that round is requested, so no register load exists that the compiler would drain the DMAs for.
template <int DT, int D, int DS>
__global__ __launch_bounds__(512) void k1_dsplit_ring(
    const unsigned short *__restrict__ q, const unsigned short *__restrict__ kc, const unsigned short *__restrict__ vc,
    const sjd_iter_params *__restrict__ params, const int *__restrict__ key_start, unsigned short *__restrict__ out,
    int n_rows, int H, int H_kv, int S_max, int kv_len_arg, int n_chunks, int B)
{
    typedef typename Frag<DT>::vec vec;
    static_assert(D == 128 && DS == 4, "written for head_dim 128 split four ways");
    constexpr int NW = 8, KS = D / 32, DW = D / DS, DB = DW / 16;
    constexpr int ROWB = D * 2, KT_BYTES = K1_KT * ROWB, VT_BYTES = K1_KT * DW * 2, SLOT = KT_BYTES + VT_BYTES;      // 8192 + 2048
    (void)0;                                                        // (two rounds of eight tiles: 2 * NW slots)
    constexpr int PPT = KT_BYTES / 1024 + VT_BYTES / 1024;         // DMA pieces per tile (8 + 2)
    constexpr int PPW = PPT;                                       // pieces per wave and round (8 tiles x 10 pieces / 8 waves)
    extern __shared__ __attribute__((aligned(1024))) unsigned char k1r_lds[];      // NSLOT x SLOT = 160 KiB; merge buffers alias it at the end
    float (*red_o)[K1_ROWS][DW + K1_RPAD] = reinterpret_cast<float (*)[K1_ROWS][DW + K1_RPAD]>(k1r_lds);
    float (*red_ml)[K1_ROWS][2] = reinterpret_cast<float (*)[K1_ROWS][2]>(k1r_lds + NW * K1_ROWS * (DW + K1_RPAD) * 4);

    const int n_pairs = n_chunks * H * B;
    int pair, dq;
    if ((n_pairs & 7) == 0) {
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        dq = slot % DS;
        pair = (slot / DS) * 8 + xcd;
    } else {
        dq = blockIdx.x % DS;
        pair = blockIdx.x / DS;
    }
    const int chunk = pair % n_chunks, head = (pair / n_chunks) % H, b = pair / (n_chunks * H);
    const int G = H / H_kv, hkv = head / G;
    int kv_base, n_total, kstart;
    k1_entry(params, key_start, b, kv_len_arg, n_rows, kv_base, n_total, kstart);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = lane >> 4, c = lane & 15;
    const int row0 = chunk * K1_ROWS;
    const int n_c = min(K1_ROWS, n_total - row0);
    const int kv_len = kv_base + row0;
    const int total = kv_len + max(n_c, 0);
    const float scale = rsqrtf((float)D);
    const int t_lo = kstart / K1_KT, t_hi = (total + K1_KT - 1) / K1_KT;
    const int n_tiles = max(t_hi - t_lo, 0), n_rounds = (n_tiles + NW - 1) / NW;

    const unsigned ring0 = k1r_lds_addr(k1r_lds);
    const size_t slab = ((size_t)b * H_kv + hkv) * (size_t)S_max * D;
    const unsigned valid_bytes = (unsigned)max(total, 0) * (unsigned)ROWB;
    const u32x4 kr = k1r_rsrc(kc + slab, valid_bytes), vr = k1r_rsrc(vc + slab, valid_bytes);
    // this wave's pieces of a round: piece id = w + 8 i (i < 10) -> tile id / 10 of the round, piece id % 10 of the tile (0..7: K, 8..9: V)
    unsigned pv_off[PPW], pl_off[PPW];        // lane's source offset inside the tile / LDS offset of the piece inside the round
    bool p_isv[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int id = w + NW * i, tl = id / PPT, pc = id % PPT;
        p_isv[i] = pc >= KT_BYTES / 1024;
        if (!p_isv[i]) {
            const int key = 4 * pc + g;
            pv_off[i] = (unsigned)(tl * KT_BYTES + key * ROWB + ((c ^ (key & 15)) * 16));
            pl_off[i] = (unsigned)(tl * SLOT + pc * 1024);
        } else {
            const int vp = pc - KT_BYTES / 1024, row = 16 * vp + (lane >> 2), part = (lane & 3) ^ (2 * ((row >> 2) & 1));
            pv_off[i] = (unsigned)(tl * KT_BYTES + row * ROWB + dq * (DW * 2) + part * 16);
            pl_off[i] = (unsigned)(tl * SLOT + KT_BYTES + vp * 1024);
        }
    }
    auto issue_round = [&](int j) {           // round j -> ring half j & 1; tiles beyond the last one: out-of-range offsets (zeros, no traffic)
        const unsigned half = ring0 + (unsigned)(j & 1) * (NW * SLOT);
        const int tbase = t_lo + j * NW;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int tl = (w + NW * i) / PPT;
            const unsigned tb = (tbase + tl < t_hi) ? (unsigned)tbase * KT_BYTES : 0x7fff0000u;
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(half + pl_off[i]));       // (wave-uniform by construction)
            if (p_isv[i]) k1r_dma16(vr, tb + pv_off[i], dst); else k1r_dma16(kr, tb + pv_off[i], dst);
        }
    };
    // ---- Q through the second half's first 4 KiB, round 0 behind it
    {
        const u32x4 qr = k1r_rsrc(q, (unsigned)B * (unsigned)n_rows * (unsigned)H * (unsigned)ROWB);
        if (w < 4) {
            const int r = 4 * w + g;
            const int grow = min(row0 + r, n_rows - 1);
            k1r_dma16(qr, (unsigned)(((b * n_rows + grow) * H + head) * ROWB + ((c ^ r) * 16)),
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(ring0 + NW * SLOT + w * 1024)));
        }
    }
    issue_round(0);
    if (w < 4) k1r_wait_vmcnt<PPW>(); else k1r_wait_vmcnt<PPW>();      // (waves 0..3: their Q piece has landed; everybody meets at the barrier)
    __builtin_amdgcn_s_barrier();
    vec qf[KS];
    {
        const bool rv = (c < n_c);
        const unsigned char *qrow = k1r_lds + NW * SLOT + c * ROWB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4 x = *reinterpret_cast<const u32x4 *>(qrow + (((4 * ks + g) ^ c) * 16));
            qf[ks] = as_frag<vec>(rv ? x : u32x4{0u, 0u, 0u, 0u});
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();             // everybody has its Q fragments: the second half may be overwritten by round 1

    unsigned kofs[KS], vofs[DB];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kofs[ks] = (unsigned)(c * ROWB + (((4 * ks + g) ^ c) * 16));
    {
        const int vrow = 4 * g + (c >> 2);    // rows vrow and vrow + 16 share ((row >> 2) & 1)
#pragma unroll
        for (int db = 0; db < DB; ++db)
            vofs[db] = (unsigned)(KT_BYTES + vrow * (DW * 2) + (((2 * db + ((c & 3) >> 1)) ^ (2 * ((vrow >> 2) & 1))) * 16) + 8 * (c & 1));
    }
    float m_run = -INFINITY, l_run = 0.0f;
    const float c1 = scale * 1.4426950408889634f;
    f32x4 o_acc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) o_acc[db] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int j = 0; j < n_rounds; ++j) {
        k1r_wait_vmcnt<0>();                  // this wave's pieces of round j have landed
        __builtin_amdgcn_s_barrier();         // ... everybody's; everybody is done with round j - 1
        issue_round(j + 1);
        const int t = t_lo + j * NW + w;
        if (t < t_hi) {
            const unsigned char *sl = k1r_lds + (j & 1) * (NW * SLOT) + w * SLOT;
            u32x4 kf[2][KS];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = *reinterpret_cast<const u32x4 *>(sl + kb * 16 * ROWB + kofs[ks]);
            u32x2 vlo[DB], vhi[DB];
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                vlo[db] = lds_tr_read(reinterpret_cast<const unsigned short *>(sl + vofs[db]));
                vhi[db] = lds_tr_read(reinterpret_cast<const unsigned short *>(sl + vofs[db] + 16 * (DW * 2)));
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 st[2];
            st[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            st[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                st[0] = Frag<DT>::mfma(as_frag<vec>(kf[0][ks]), qf[ks], st[0]);
                st[1] = Frag<DT>::mfma(as_frag<vec>(kf[1][ks]), qf[ks], st[1]);
            }
            const bool interior = (t * K1_KT >= kstart) && (t * K1_KT + K1_KT - 1 <= kv_len) && (t * K1_KT + K1_KT <= total);
            if (!interior) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = t * K1_KT + 16 * kb + 4 * g + r;
                        const bool vis = (key >= kstart) && (key <= kv_len + c) && (key < total);
                        st[kb][r] = vis ? st[kb][r] : -INFINITY;
                    }
            }
            float mx = fmaxf(fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3])), fmaxf(fmaxf(st[1][0], st[1][1]), fmaxf(st[1][2], st[1][3])));
            mx = k1r_max_across_groups(mx);
            const float m_new = fmaxf(m_run, mx);
            const float m_safe = (m_new == -INFINITY) ? 0.0f : m_new;
            const float nm = -m_safe * c1;
            const float alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m_run, c1, nm));
            float pv[8];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[4 * kb + r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][r], c1, nm));
            float rs = ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
            rs = k1r_sum_across_groups(rs);
            l_run = __builtin_fmaf(l_run, alpha, rs);
            m_run = m_new;
            u32x4 pw;
#pragma unroll
            for (int i = 0; i < 4; ++i) pw[i] = k1_cvt_pk<DT>(pv[2 * i], pv[2 * i + 1]);
            const vec pfrag = as_frag<vec>(pw);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const u32x4 vv{vlo[db][0], vlo[db][1], vhi[db][0], vhi[db][1]};
                f32x4 acc = o_acc[db];
                acc[0] *= alpha; acc[1] *= alpha; acc[2] *= alpha; acc[3] *= alpha;
                o_acc[db] = Frag<DT>::mfma(as_frag<vec>(vv), pfrag, acc);
            }
        }
    }
    k1r_wait_vmcnt<0>();                      // (the out-of-range pieces of the round after the last one)
    __builtin_amdgcn_s_barrier();             // every wave is done with the ring: it becomes the merge buffers
    if (g == 0) { red_ml[w][c][0] = m_run * scale; red_ml[w][c][1] = l_run; }
#pragma unroll
    for (int db = 0; db < DB; ++db) *reinterpret_cast<f32x4 *>(&red_o[w][c][16 * db + 4 * g]) = o_acc[db];
    __syncthreads();
    constexpr int D4 = DW / 4;
    for (int u = threadIdx.x; u < K1_ROWS * D4; u += 64 * NW) {
        const int row = u / D4, d = (u % D4) * 4;
        const int grow = row0 + row;
        if (grow >= n_rows) continue;
        float mk[NW], lk[NW];
        float4 ok[NW];
#pragma unroll
        for (int kp = 0; kp < NW; ++kp) {
            const float2 v = *reinterpret_cast<const float2 *>(&red_ml[kp][row][0]);
            ok[kp] = *reinterpret_cast<const float4 *>(&red_o[kp][row][d]);
            mk[kp] = v.x;
            lk[kp] = v.y;
        }
        float M = -INFINITY;
#pragma unroll
        for (int kp = 0; kp < NW; ++kp) M = fmaxf(M, mk[kp]);
        const float Ms = (M == -INFINITY) ? 0.0f : M;
        float L = 0.f, O0 = 0.f, O1 = 0.f, O2 = 0.f, O3 = 0.f;
#pragma unroll
        for (int kp = 0; kp < NW; ++kp) {
            const float wgt = __expf(mk[kp] - Ms);
            L += wgt * lk[kp];
            O0 += wgt * ok[kp].x; O1 += wgt * ok[kp].y; O2 += wgt * ok[kp].z; O3 += wgt * ok[kp].w;
        }
        const float inv = L > 0.f ? 1.0f / L : 0.0f;
        uint2 pk{0u, 0u};
        if (grow < n_total) {
            pk.x = (unsigned)Frag<DT>::cvt(O0 * inv) | ((unsigned)Frag<DT>::cvt(O1 * inv) << 16);
            pk.y = (unsigned)Frag<DT>::cvt(O2 * inv) | ((unsigned)Frag<DT>::cvt(O3 * inv) << 16);
        }
        *reinterpret_cast<uint2 *>(out + (((size_t)b * n_rows + grow) * H + head) * D + dq * DW + d) = pk;
    }
}
