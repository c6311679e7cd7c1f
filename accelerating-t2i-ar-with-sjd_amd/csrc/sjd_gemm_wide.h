// sjd_gemm_wide.h -- G1w: the weight-streaming projection for windows of 65..256 rows (three to eight prompts per forward), round 6.
// Included by sjd_gemm.hip (same packed weights, same fp32 split-K planes [n_chunks, 32 MT, N] as g1_skinny_gemm: bit for bit).
//
// Why a new structure (VERDICT r5 "next #1").  g1_skinny_gemm_tiled8 with eight row tiles ran at 0.27 of the HBM peak (48.5 us per launch at
// 256 rows): a wave owned ONE 32-column tile and read EIGHT activation fragments from LDS per weight record (8 KiB per 8 MFMAs: the LDS read
// port at its limit), the activation sub-tile travelled global -> VGPR -> ds_write_b128 -> barrier (64 KiB per eight k-steps through the
// 79 B/clk store path, interleaved loads do not hide it: MI355X_MICROARCH "LDS"), and a 128-column workgroup re-read the activation at
// TWICE the weight bytes.  Here:
//   * the activation never touches a VGPR on its way in: LDS-DMA (`buffer_load_dwordx4 ... offen lds`, 1 KiB per wave instruction) writes
//     stages of SUB k-steps x 32 MT rows straight into a ring of NS slots; NS - 1 stages are in flight per workgroup;
//   * the LDS image of a DMA is lane-linear (it cannot pad rows), so bank conflicts are avoided by permuting the SOURCE: 16-byte piece p of
//     row m lands at position p ^ swz(m) of the row, swz(m) = (m SUB / 8) mod 2 SUB -- the sixteen lanes of every ds_read_b128 lane group then
//     hit sixteen different 16-byte bank groups (checked for SUB = 2, 4, 8 against the lane groups of MI355X_MICROARCH's LDS table);
//   * a wave owns CT column tiles (CT = 2: every activation fragment read from LDS feeds two MFMAs -- half the LDS reads per flop) and all
//     MT row tiles; the weight records still go HBM -> MFMA B registers directly (the packed stream IS the operand order), through a ring of
//     R = (NS - 1) SUB k-steps per tile refilled in place;
//   * every load of the kernel is hand-written and hand-counted: the DMA has no destination register the compiler could wait for, and
//     the weight loads are `asm volatile` as well so that ONE in-order vmcnt sequence covers both.  The prologue issues exactly the
//     steady-state sequence (per k-step: the records of tiles 0 .. CT - 2 of the k-step R ahead, DPK DMA pieces of the stage NS - 1 ahead, the
//     record of the last tile -- each BETWEEN two MFMAs of the k-step, not in a burst behind the barrier), so the wait counts (N_W0, N_WL, N_A
//     in the kernel) are exact from the first stage on (a count larger than the number of younger operations would not wait at all);
//   * one raw s_barrier per stage: "my pieces of stage st landed" -> barrier (everybody's have; everybody is done with stage st - 1) ->
//     refill the slot stage st - 1 left -> SUB k-steps of MFMAs.
// Rows >= M and columns past the matrix read as zero through the buffer descriptor's range check; the columns past a ragged chunk's end are
// requested out of range (one compare + select per DMA piece), so the k-steps past the end multiply zero records by zero fragments: no branch.
// Same MFMA sequence per (column tile, K chunk, row tile) as g1_skinny_gemm: the planes are bit-identical to the 32-row kernel's
// (tests/test_gpu_glue.py::test_g1_skinny_gemm_five_to_eight_row_tiles compares them plane for plane).
#pragma once

template <int N> __device__ __forceinline__ void gw_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// the same, with the registers the wait makes valid passed THROUGH the statement: the compiler then cannot move a use above it
template <int N> __device__ __forceinline__ void gw_wait_regs(u32x4 &a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void gw_wait_regs(u32x4 &a, u32x4 &b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }

// raw buffer descriptor over `bytes` bytes from `p` (wave-uniform); loads at or past `bytes` return zero without touching memory
__device__ __forceinline__ u32x4 gw_rsrc(const void *p, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)p;
    u32x4 r;
    r[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) & 0xffffu;       // stride 0: raw buffer
    r[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000u;
    return r;
}

// one LDS-DMA piece: 64 lanes x 16 B from (descriptor base + voff + soff) to LDS bytes [lds_addr + 16 lane, + 16).  M0 is the compiler's: saved /
// restored inside the statement; the s_nop covers the M0 write -> LDS-DMA read hazard (sjd_attention_ring.h: k1r_dma16).
__device__ __forceinline__ void gw_dma16(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 2\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane((int)soff)), "s"(__builtin_amdgcn_readfirstlane((int)lds_addr))
                 : "memory");
}

// (12-bit form, late round 6) the waits that tie a record pair's registers / a unit header's registers
template <int N> __device__ __forceinline__ void gw_wait_regs(u32x4 &a, u32x2 &b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void gw_wait_regs(u32x2 &a, u32x2 &b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }
// the codes of a record pair (64 lanes x 8 B), and one header entry per lane: like gw_wload, issued by hand and counted by hand
__device__ __forceinline__ u32x2 gw_wload2(u32x4 rsrc, unsigned voff, unsigned soff)
{
    u32x2 r;
    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen nt" : "=v"(r) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return r;
}
__device__ __forceinline__ u32x2 gw_gload2(const void *p)
{
    u32x2 r;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}

// one weight record (1 KiB per wave) into B-operand registers, non-temporal (streamed once); NOT waited for by the compiler
__device__ __forceinline__ u32x4 gw_wload(u32x4 rsrc, unsigned voff, unsigned soff)
{
    u32x4 r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(r) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return r;
}

__device__ __forceinline__ unsigned gw_lds_addr(const void *p)
{
    return (unsigned)(unsigned long)((__attribute__((address_space(3))) const unsigned char *)p);
}

// (timing probes, results invalid: -DGW_NO_MFMA the streams and the LDS traffic without the matrix cores, -DGW_NO_W no weight stream, -DGW_NO_DMA no
//  activation stream, -DGW_NO_STORE no partial planes, -DGW_NO_BARRIER no workgroup barrier)
#ifdef GW_NO_MFMA
template <int DT> __device__ __forceinline__ f32x16 gw_fake_mma(u32x4 a, u32x4 b, f32x16 c) { asm volatile("" ::"v"(a), "v"(b)); return c; }      // (operands kept alive, no instruction)
#define GW_MMA(a_, b_, c_) gw_fake_mma<DT>(a_, b_, c_)
#else
#define GW_MMA(a_, b_, c_) G1Mfma<DT>::mma(a_, b_, c_)
#endif
// MT row tiles (rows = 32 MT), CT column tiles per wave, NW waves (column groups) per workgroup, stages of SUB k-steps, NS ring slots,
// WPS = waves per SIMD the register budget is cut for (1: up to 512 registers; 2: 256 -- two workgroups of <= 4 waves per CU)
// Z (late round 6): the weights arrive as the LOSSLESS 12-bit stream of kernels G1z / G1sz (csrc/sjd_gemm.hip: 1536-byte record PAIRS of two k-steps --
// 64 lanes x 16 B of low bytes, 64 lanes x 8 B of codes -- and a header of exceptions per (chunk, tile) unit) and are decoded into the SAME B operands
// in front of their MFMAs (g1z_operand): bit-identical planes, 25 % fewer weight bytes.  One vector-memory operation per tile and k-step as before -- the
// low bytes of the pair R k-steps ahead behind an even k-step's last MFMA, its codes behind the odd one's -- so the hand counts stand; an even k-step
// decodes BOTH operands of its pair (its registers are then free for the refill) and waits one group less deep (the codes are one k-step younger than the
// low bytes); an odd k-step waits for nothing.  The unit headers are requested in front of the prologue and waited for behind it without draining it.
// WIDE: headers of 128 entries.  Raw units (a header count of -1) decode to garbage here: the caller runs sjd_raw_units_fixup behind the launch.
template <int DT, int MT, int CT, int NW, int SUB, int NS, int RW, int WPS, bool Z = false, bool WIDE = false>
__global__ __launch_bounds__(64 * NW, WPS) void g1_wide(const unsigned short *__restrict__ x, const u32x4 *__restrict__ wp, float *__restrict__ out,
                                                        int M, int N, int K, int KC, int n_tiles, int rec_stride, int tile0, int ldx, int xmap,
                                                        const u32x2 *__restrict__ exc, int exc_cap)
{
    static_assert(!Z || (DT == SJD_DTYPE_BF16 && SUB % 2 == 0 && (RW * SUB) % 2 == 0), "the 12-bit stream: bf16, whole record pairs per stage and ring");
    static_assert(SUB == 2 || SUB == 4 || SUB == 8, "a stage is 2, 4 or 8 k-steps");
    static_assert(CT == 1 || CT == 2, "one or two column tiles per wave");
    // The weight ring is RW stages (R k-steps) deep, the activation ring LA = NS - 1 stages ahead.  LA > RW matters: vmcnt retires in order, so the
    // wait for the (L2-resident, fast) pieces of stage st also waits for every weight record issued before them -- with LA == RW those are the
    // records of stage st itself, i.e. the top of every stage drained the weight ring (the two streams alone cost +2.5 / +0.5 us over the
    // arithmetic, together +7.3: profiles/r6_g1w_probes.txt); with LA = RW + 1 they are the records of stage st - 1, consumed already.
    constexpr int LA = NS - 1, R = RW * SUB;
    static_assert(LA >= RW && RW >= 1, "the activation ring is at least as far ahead as the weight ring");
    constexpr int ROWB = SUB * 32;                // bytes per activation row and stage
    constexpr int LPR = SUB * 2;                  // 16-byte pieces (DMA lanes) per row
    constexpr int RPP = 64 / LPR;                 // rows per DMA piece (1 KiB)
    constexpr int P = MT * SUB;                   // DMA pieces per stage
    constexpr int DPK = (P + NW * SUB - 1) / (NW * SUB);      // ... per wave and k-step (the pieces of a stage are issued spread over the k-steps of a stage)
    constexpr int NA = DPK * SUB;                 // ... per wave and stage
    constexpr bool DUMMY = NA * NW != P;          // some wave's last pieces lie past the stage: issued out of range into a scratch KiB (uniform counts)
    constexpr int SLOT = P * 1024;
    constexpr int NM = MT * CT;                   // MFMAs per k-step, tile-major: MFMA i = (tile i / MT, row tile i % MT)
    // vector-memory operations of a k-step in issue order: [records of tiles 0 .. CT - 2] [DPK DMA pieces] [record of tile CT - 1]
    constexpr int GRP = CT + DPK;
    constexpr int D0 = MT * (CT - 1) + 1;         // the DMA pieces follow MFMA D0, D0 + DSTR, ...
    constexpr int DSTR = (D0 + 2 * (DPK - 1) < NM - 1) ? 2 : 1;
    // (the last piece may share the last MFMA's slot with the record reload of tile CT - 1: the piece is issued first there, the order the counts assume)
    static_assert(D0 + DSTR * (DPK - 1) <= NM - 1, "the DMA pieces fit between the record loads");
    // younger operations when tile c's records of k-step s are waited for (issued R k-steps earlier), and when the last piece of stage st is
    constexpr int N_WL = (R - 1) * GRP + (CT - 1);                       // the last tile's record closes its group; the earlier tiles' of THIS k-step are out again
    constexpr int N_W0 = (R - 1) * GRP + DPK + 1;                        // (CT = 2) tile 0: the pieces and the last record of its own group follow it
    constexpr int N_A = (LA - 1) * SUB * GRP + 1;
    static_assert(N_A <= 63 && N_WL <= 63 && N_W0 <= 63, "vmcnt is a 6-bit counter");
    static_assert(NS >= 2, "at least two slots");
    extern __shared__ __attribute__((aligned(1024))) unsigned char gw_lds[];       // NS slots, then 1 KiB of scratch
    SJD_TR(0);
    SJD_TR_CLK(4);             // (shader-clock stamps next to the 100 MHz wall-clock ones: the clock the launch actually ran at)
    // XCD-aware (column group, K chunk) of this workgroup.  Workgroup L = x + gridDim.x * y runs on XCD L mod 8, and every XCD has its own L2: with the
    // plain (x, y) = (column group, chunk) map the workgroups of one chunk are spread over all eight XCDs, so every XCD's L2 holds the activation
    // columns of EVERY chunk -- 5.6 MB for the down projection at 256 rows against 4 MB of L2: PMC showed 45 MB of activation re-reads going to HBM
    // (profiles/g1w_traffic.json).  With a power-of-two chunk count the chunk follows the XCD instead (chunk = XCD mod n_chunks; sixteen and more
    // chunks: XCD + 8 * ...), an XCD then re-reads 1 / min(8, n_chunks) of the activation.  Which workgroup computes which (tile, chunk) does not
    // change any result.  xmap = 0: the plain map (A/B aid, SJD_G1W_XMAP=0).
    int bx = blockIdx.x, chunk = blockIdx.y;
    {
        const int gx = gridDim.x, nc = gridDim.y;
        if (xmap && nc > 1 && (nc & (nc - 1)) == 0) {
            const int total = gx * nc, lin = bx + gx * chunk, full = total & ~7;
            if (nc <= 8) {
                const int per = 8 / nc, b0 = (full >> 3) * per;          // column groups covered by the whole rounds of eight workgroups
                if (lin < full) { const int xcd = lin & 7, slot = lin >> 3; chunk = xcd % nc; bx = slot * per + xcd / nc; }
                else { const int j = lin - full, rest = gx - b0; bx = b0 + j % rest; chunk = j / rest; }
            } else if (total == full) {
                const int q = nc >> 3, xcd = lin & 7, slot = lin >> 3;
                chunk = xcd + 8 * (slot % q); bx = slot / q;
            }
        }
    }
    const int k0 = chunk * KC;
    const int steps = min(KC, K - k0) / 16;
    const int n_stage = (steps + SUB - 1) / SUB;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned rsb = (unsigned)rec_stride * (Z ? 1536u : 1024u);
    const int pairs = (steps + 1) / 2, pairs_full = (KC / 16 + 1) / 2;          // (Z) record pairs of this unit / of a full chunk's unit
    const size_t chunk_base = (size_t)chunk * n_tiles * (Z ? pairs_full : KC / 16);
    u32x4 wr[CT];
    int t_out[CT];
    [[maybe_unused]] u32x2 hra[CT], hrb[CT];                                     // (Z) this lane's header entries of the tiles' units
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        t_out[c] = (bx * NW + w) * CT + c;
        const bool has = t_out[c] < N / 32;
        const int t = tile0 + (has ? t_out[c] : 0);
        if constexpr (Z) {
            const size_t tile_off = (rec_stride == 1) ? (size_t)t * pairs : (size_t)t;
            wr[c] = gw_rsrc(reinterpret_cast<const unsigned char *>(wp) + (chunk_base + tile_off) * 1536, has ? (unsigned)(pairs - 1) * rsb + 1536u : 0u);
            const u32x2 *e = exc + ((size_t)chunk * n_tiles + t) * (size_t)exc_cap;
            hra[c] = gw_gload2(e + min(lane, exc_cap - 1));
            if constexpr (WIDE) hrb[c] = gw_gload2(e + 64 + lane); else hrb[c] = u32x2{0xffffffffu, 0xffffffffu};
        } else {
            const size_t tile_off = (rec_stride == 1) ? (size_t)t * steps : (size_t)t;
#ifdef GW_NO_W            // (timing probe, results invalid: no weight stream -- every record out of range)
            wr[c] = gw_rsrc(wp + (chunk_base + tile_off) * 64, 0u);
#else
            wr[c] = gw_rsrc(wp + (chunk_base + tile_off) * 64, has ? (unsigned)(steps - 1) * rsb + 1024u : 0u);      // (no tile: every record reads as zero)
#endif
        }
    }
    const u32x4 xr = gw_rsrc(x, (unsigned)M * (unsigned)ldx * 2u);      // (ldx: row stride of x in elements, >= K)
    const unsigned lds0 = gw_lds_addr(gw_lds);
    // this wave's DMA pieces of a stage: piece j = w NA + i covers rows j RPP .. + RPP - 1; lane l -> row j RPP + l / LPR, position l % LPR of the
    // row, which receives source piece (l % LPR) ^ swz(row)
    unsigned xoff[NA], xcol[NA];          // byte offset of the lane's 16 bytes in x; its column inside the stage
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int j = w * NA + i;
        const int row = j * RPP + lane / LPR, pos = lane % LPR;
        const int p = pos ^ ((((row & 31) * SUB) >> 3) & (LPR - 1));
        xoff[i] = (!DUMMY || j < P) ? ((unsigned)row * (unsigned)ldx + (unsigned)(k0 + 8 * p)) * 2u : 0x7fff0000u;
        xcol[i] = (unsigned)(8 * p);
    }
    const int chunk_cols = steps * 16;
    // A-fragment read offsets inside a slot: lane (m = l & 31, h = l >> 5), k-step u of the stage: row m, piece (2 u + h) ^ swz(m)
    unsigned aoff[SUB];
    {
        const int m = lane & 31, h = lane >> 5;
        const int sw = ((m * SUB) >> 3) & (LPR - 1);
#pragma unroll
        for (int u = 0; u < SUB; ++u) aoff[u] = (unsigned)(m * ROWB + 16 * ((2 * u + h) ^ sw));
    }
    const unsigned wvoff = (unsigned)lane * 16u;
    const unsigned piece0 = lds0 + (unsigned)(w * NA) * 1024u;        // this wave's first piece inside slot 0
    const unsigned scratch = lds0 + (unsigned)(NS * SLOT);

    // piece d of k-step position u of activation stage `sa`, into the ring slot at LDS address `sb` (+ this wave's piece offset)
    auto dma_piece = [&](int sa, unsigned sb, int u, int d) {
#ifdef GW_NO_DMA          // (timing probe: the activation pieces are issued out of range -- same instructions, no traffic)
        const unsigned soff = 0x7fff0000u;
#else
        const unsigned soff = sa < n_stage ? (unsigned)sa * ROWB : 0x7fff0000u;       // past the chunk: out of range (zeros, no traffic)
#endif
        const int i = u * DPK + d;
        unsigned dst = sb + (unsigned)i * 1024u;
        if constexpr (DUMMY) dst = (w * NA + i < P) ? dst : scratch;
        // a column past a ragged chunk's end belongs to the next chunk (or the next row): it is fetched out of range, i.e. lands as zeros, and the
        // k-steps past the end multiply zero records by zero fragments (+0 added to an accumulator that is never -0 leaves every bit as it is)
        const unsigned voff = (int)xcol[i] < chunk_cols - sa * (SUB * 16) ? xoff[i] : 0x7fff0000u;
        gw_dma16(xr, voff, soff, dst);
    };
    u32x4 W[CT][Z ? R / 2 : R];              // plain: one record per k-step of the ring; Z: the low bytes of one record pair per two k-steps
    [[maybe_unused]] u32x2 WC[CT][R / 2];    // (Z) ... and its codes
    [[maybe_unused]] u32x4 Bn[CT];           // (Z) the odd k-step's operand, decoded with the even one's
    // (Z) record load of ring k-step ks: even -> the pair's low bytes, odd -> its codes
    auto z_load = [&](int c, int ks_ring, unsigned pair_abs) {
        if ((ks_ring & 1) == 0) W[c][ks_ring >> 1] = gw_wload(wr[c], wvoff, pair_abs * rsb);
        else WC[c][ks_ring >> 1] = gw_wload2(wr[c], 1024u + (unsigned)lane * 8u, pair_abs * rsb);
    };
    f32x16 acc[CT][MT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][mt][r] = 0.0f;

    // ---- prologue: the steady-state issue sequence of the virtual stages -LA .. -1.  A record load whose k-step would be negative is replaced by an
    // out-of-range DMA piece into the scratch KiB: the counts stay uniform, and NO register is written -- a dummy register load would land
    // asynchronously in a register the compiler considers dead and has handed to somebody else by then (it did: wrong planes, first version)
#pragma unroll
    for (int v = 0; v < LA; ++v)
#pragma unroll
        for (int u = 0; u < SUB; ++u) {
            const int ks = (v - (LA - RW)) * SUB + u;         // k-step of this group's records
#pragma unroll
            for (int c = 0; c < CT - 1; ++c) {
                if (ks >= 0) { if constexpr (Z) z_load(c, ks, (unsigned)(ks >> 1)); else W[c][ks] = gw_wload(wr[c], wvoff, (unsigned)ks * rsb); }
                else gw_dma16(xr, 0x7fff0000u, 0u, scratch);
            }
#pragma unroll
            for (int d = 0; d < DPK; ++d) dma_piece(v, piece0 + (unsigned)(v * SLOT), u, d);
            if (ks >= 0) { if constexpr (Z) z_load(CT - 1, ks, (unsigned)(ks >> 1)); else W[CT - 1][ks] = gw_wload(wr[CT - 1], wvoff, (unsigned)ks * rsb); }
            else gw_dma16(xr, 0x7fff0000u, 0u, scratch);
        }
    // (Z) the unit headers: requested in front of the prologue, complete once no more than the prologue's own LA * SUB * GRP operations are outstanding
    [[maybe_unused]] g1z_hdr hd[CT];
    if constexpr (Z) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            gw_wait_regs<LA * SUB * GRP>(hra[c], hrb[c]);
            hd[c] = g1z_header<WIDE>(g1z_hraw{hra[c], hrb[c]}, lane, exc_cap);
        }
    }
    SJD_TR(1);
    // one k-step: the MFMAs in tile-major order, ONE other instruction group behind each of them (a single wave per SIMD issues in order: what is
    // not placed between two MFMAs is not hidden by them -- the first version issued its 16 MFMAs back to back and then ~45 other instructions:
    // 33 % of the wave's cycles were spent issuing while the matrix pipe idled, profiles/r6_g1w_first_pmc.txt):
    //   behind MFMA i < MT: the A fragment of row tile i for the NEXT k-step;  behind the last MFMA of tile c: its record R k-steps ahead;
    //   behind MFMA D0 + DSTR d: DMA piece d of the stage NS - 1 ahead;  in front of the first MFMA of tile c: the wait for its records.
    auto k_step = [&](const unsigned char *sl, int u, int r, int st, unsigned sb_refill, u32x4 (&a)[2][MT]) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const int c = i / MT, mt = i % MT;
            u32x4 b;
            if constexpr (Z) {
                if (mt == 0 && (r & 1) == 0) {            // an even k-step: both loads of its pair have landed once one group fewer is outstanding
                    if (c == CT - 1) gw_wait_regs<N_WL - GRP>(W[c][r >> 1], WC[c][r >> 1]); else gw_wait_regs<N_W0 - GRP>(W[c][r >> 1], WC[c][r >> 1]);
                    const unsigned s_abs = (unsigned)(st * SUB + u);
                    Bn[c] = g1z_operand<WIDE>(W[c][r >> 1].z, W[c][r >> 1].w, WC[c][r >> 1].y, s_abs + 1u, hd[c], lane);
                    W[c][r >> 1] = g1z_operand<WIDE>(W[c][r >> 1].x, W[c][r >> 1].y, WC[c][r >> 1].x, s_abs, hd[c], lane);      // (held in the pair's own registers until its last MFMA)
                }
                b = (r & 1) == 0 ? W[c][r >> 1] : Bn[c];
            } else {
                if (mt == 0) {
                    if (c == CT - 1) gw_wait_regs<N_WL>(W[c][r]); else gw_wait_regs<N_W0>(W[c][r]);
                }
                b = W[c][r];
            }
            acc[c][mt] = GW_MMA(a[u & 1][mt], b, acc[c][mt]);
            if (i < MT && u + 1 < SUB) a[(u + 1) & 1][i] = *reinterpret_cast<const u32x4 *>(sl + i * (32 * ROWB) + aoff[u + 1]);
            if (i >= D0 && ((i - D0) % DSTR) == 0 && (i - D0) / DSTR < DPK) dma_piece(st + LA, sb_refill, u, (i - D0) / DSTR);
            if (mt == MT - 1) {
                if constexpr (Z) z_load(c, r, (unsigned)((st * SUB + u + R) >> 1));
                else W[c][r] = gw_wload(wr[c], wvoff, (unsigned)(st * SUB + u + R) * rsb);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int slot = 0, refill = LA;              // slot of stage st; slot stage st + LA goes into (= the one stage st - 1 has left)
    for (int st0 = 0; st0 < n_stage; st0 += RW) {
#pragma unroll
        for (int q = 0; q < RW; ++q) {
            const int st = st0 + q;
            if (st >= n_stage) break;
            gw_wait<N_A>();                       // this wave's pieces of stage st have landed
#ifndef GW_NO_BARRIER
            __builtin_amdgcn_s_barrier();         // ... everybody's; everybody is done reading stage st - 1
#endif
            const unsigned char *sl = gw_lds + slot * SLOT;
            const unsigned sb_refill = piece0 + (unsigned)(refill * SLOT);
            u32x4 a[2][MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[0][mt] = *reinterpret_cast<const u32x4 *>(sl + mt * (32 * ROWB) + aoff[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < SUB; ++u) k_step(sl, u, q * SUB + u, st, sb_refill, a);
            refill = slot;
            slot = slot + 1 == NS ? 0 : slot + 1;
        }
    }
    SJD_TR(3);
    SJD_TR_CLK(5);
    gw_wait<0>();          // (the tail's out-of-range pieces: nothing may still be writing this workgroup's LDS when it exits)
#ifdef GW_NO_STORE
    if (acc[0][0][0] != 12345.0f) return;
#endif
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (t_out[c] >= N / 32) continue;
        float *o = out + ((size_t)chunk * (32 * MT)) * N + (size_t)t_out[c] * 32 + (lane & 31);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                o[(size_t)m * N] = acc[c][mt][r];
            }
    }
#ifdef SJD_TRACE
    SJD_TR(2);                    // plane stores issued
    __builtin_amdgcn_s_waitcnt(0x0F70);
    SJD_TR(6);                    // acknowledged
#endif
}

// LDS bytes of a g1_wide launch
template <int MT, int SUB, int NS> constexpr size_t g1_wide_lds() { return (size_t)NS * MT * SUB * 1024 + 1024; }

template <int DT, int MT, int CT, int NW, int SUB, int NS, int RW, int WPS>
static int g1_wide_launch(const void *x, const void *w_packed, float *out, int M, int N, int K, int KC, int n_tiles, int step_major, int tile0, hipStream_t s, int ldx = 0)
{
    const int n_out = N / 32, n_chunks = (K + KC - 1) / KC;
    const dim3 grid((n_out + NW * CT - 1) / (NW * CT), n_chunks), block(64 * NW);
    constexpr size_t lds = g1_wide_lds<MT, SUB, NS>();
    static_assert(lds <= 160 * 1024, "the activation ring must fit in LDS");
    auto kern = g1_wide<DT, MT, CT, NW, SUB, NS, RW, WPS>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // (the XCD-aware map pays where the activation outgrows an XCD's L2 share: eight prompts 6.75 -> 6.58 ms per step, two alternations on one box; at 128
    //  rows and below it is neutral end to end and costs the gate|up launch 3 us alone: plain map there.  profiles/r6_g1w_xmap_ab.txt; SJD_G1W_XMAP=0 / 1 forces)
    static const int xenv = [] { const char *e = getenv("SJD_G1W_XMAP"); return !e ? -1 : (e[0] == '0' ? 0 : 1); }();
    const int xmap = xenv >= 0 ? xenv : (MT > 4 ? 1 : 0);
    hipLaunchKernelGGL(kern, grid, block, lds, s, (const unsigned short *)x, (const u32x4 *)w_packed, out, M, N, K, KC, n_tiles, step_major ? n_tiles : 1, tile0, ldx > 0 ? ldx : K, xmap,
                       (const u32x2 *)nullptr, 0);
    return hipGetLastError() == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH;
}

// the same over the 12-bit stream (wz: record pairs, exc: unit headers of exc_cap entries); bf16
template <int MT, int CT, int NW, int SUB, int NS, int RW, int WPS>
static int g1_wide_launch_z(const void *x, const void *wz, const void *exc, int exc_cap, float *out, int M, int N, int K, int KC, int n_tiles, int step_major, int tile0,
                            hipStream_t s)
{
    const int n_out = N / 32, n_chunks = (K + KC - 1) / KC;
    const dim3 grid((n_out + NW * CT - 1) / (NW * CT), n_chunks), block(64 * NW);
    constexpr size_t lds = g1_wide_lds<MT, SUB, NS>();
    static_assert(lds <= 160 * 1024, "the activation ring must fit in LDS");
    static const int xenv = [] { const char *e = getenv("SJD_G1W_XMAP"); return !e ? -1 : (e[0] == '0' ? 0 : 1); }();
    const int xmap = xenv >= 0 ? xenv : (MT > 4 ? 1 : 0);
#define SJD_G1WZ(WIDE_) do { \
        auto kern = g1_wide<SJD_DTYPE_BF16, MT, CT, NW, SUB, NS, RW, WPS, true, WIDE_>; \
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kern, grid, block, lds, s, (const unsigned short *)x, (const u32x4 *)wz, out, M, N, K, KC, n_tiles, step_major ? n_tiles : 1, tile0, K, xmap, \
                           (const u32x2 *)exc, exc_cap); } while (0)
    if (exc_cap > 64) SJD_G1WZ(true); else SJD_G1WZ(false);
#undef SJD_G1WZ
    return hipGetLastError() == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH;
}
