// sjd_sampling.hip -- kernels K2 (logits -> probs -> sample), K4 (verify/accept + residual resample),
// K5 (window assembly / re-guess) for gfx950.  Built with -ffp-contract=off (canonical numerics, see sjd_device.h).
//
// One 1024-thread workgroup (16 wave64) owns one row.  The row is staged in the caller's probs_out / scratch buffer
// (L2-resident between passes); all cross-lane reductions use wave64 shuffles + a 16-entry LDS exchange.
#include <hip/hip_runtime.h>
#include <math.h>
#include <type_traits>

#include "../../include/sjd_hip.h"
#include "sjd_device.h"
#include "sjd_philox.h"

// phase timestamps of K2 / K4 (tools/phase_trace.py, -DSJD_TRACE; compiled out otherwise): slot = row (K2) / 32 (K4)
#ifdef SJD_TRACE
__device__ unsigned long long g_k2_trace[64][16];
#define SJD_TRS(slot, i) do { if (threadIdx.x == 0) g_k2_trace[(slot) & 63][i] = wall_clock64(); } while (0)
extern "C" int sjd_debug_trace_k2(unsigned long long *host_out, int n)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_k2_trace), (size_t)n * 16 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#else
#define SJD_TRS(slot, i) do { } while (0)
#endif

// smallest column window [lo, hi) containing every allowed column of the rule
__device__ __forceinline__ void rule_window(const sjd_row_rule &r, int V, int &lo, int &hi)
{
    if (r.n_ranges == 0) { lo = 0; hi = V; return; }
    lo = V; hi = 0;
#pragma unroll
    for (int a = 0; a < SJD_MAX_RANGES; ++a)
        if (a < r.n_ranges) { lo = min(lo, r.lo[a]); hi = max(hi, r.hi[a]); }
    lo = max(lo, 0); hi = min(hi, V);
    if (hi < lo) hi = lo;
}

__device__ __forceinline__ bool rule_allows(const sjd_row_rule &r, int c)
{
    if (r.n_ranges == 0) return true;
    bool ok = false;
#pragma unroll
    for (int a = 0; a < SJD_MAX_RANGES; ++a) ok |= (a < r.n_ranges) && (c >= r.lo[a]) && (c < r.hi[a]);
    return ok;
}

// ------------------------------------------------------------------------------------------------ K2
// replaces sampling_logits2tokens (reference jacobi_iteration_lumina_mgpt.py:82-132)
// PART: the logits are not materialised -- K2 reads the fp32 split-K partials of the output-head projection (G1 over the packed
// lm_head, only the vocabulary columns the grammar allows), sums the chunks in order, applies the folded final-RMSNorm row scale and
// rounds to the activation dtype exactly where nn.Linear would (MC:1560-1561: 16-bit lm_head output, then .float()); the CFG combine,
// grammar mask, top-k, softmax and draw are the same code as the dense-logits form.
static_assert(sizeof(sjd_row_rule) == 52 && sizeof(sjd_iter_params) == 64 + 8 * SJD_MAX_WINDOW + 2 * 52 * SJD_MAX_WINDOW, "sjd_iter_params layout is mirrored by ctypes (sjd_amd/_lib.py::IterParams)");
static_assert(sizeof(sjd_head_partials) == 96, "sjd_head_partials layout is mirrored by ctypes (sjd_amd/_lib.py::HeadPartials)");

__device__ __forceinline__ float k2_round16(float x, int dt)
{
    if (dt == SJD_DTYPE_BF16) {
        unsigned u = __float_as_uint(x);
        u += 0x7fffu + ((u >> 16) & 1u);
        return __uint_as_float(u & 0xffff0000u);
    }
    if (dt == SJD_DTYPE_F16) return (float)((_Float16)x);
    return x;
}

__device__ __forceinline__ float k2_row_scale(const sjd_head_partials &hp, int tok)
{
    if (!hp.row_sumsq) return 1.0f;
    float t = 0.f;
    for (int s0 = 0; s0 < hp.slices; s0 += 8) {          // the fixed order of row_sumsq_total (sjd_glue.hip)
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (s0 + q < hp.slices) ? hp.row_sumsq[(size_t)(s0 + q) * hp.prows + tok] : 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += v[q];
    }
    return rsqrtf(t * hp.inv_hidden + hp.eps);
}

// the scales of the cond row `tok` and of the uncond row hp.urow_off + tok (1 without one): both rows' statistics in flight together
__device__ __forceinline__ void k2_row_scales(const sjd_head_partials &hp, int tok, float &rc, float &ru)
{
    if (!hp.row_sumsq) return;
    const bool two = hp.urow_off > 0;
    float t = 0.f, tu = 0.f;
    for (int s0 = 0; s0 < hp.slices; s0 += 8) {          // the fixed order of row_sumsq_total (sjd_glue.hip)
        float v[8], w[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v[q] = (s0 + q < hp.slices) ? hp.row_sumsq[(size_t)(s0 + q) * hp.prows + tok] : 0.f;
            w[q] = (two && s0 + q < hp.slices) ? hp.row_sumsq[(size_t)(s0 + q) * hp.prows + hp.urow_off + tok] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) { t += v[q]; tu += w[q]; }
    }
    rc = rsqrtf(t * hp.inv_hidden + hp.eps);
    if (two) ru = rsqrtf(tu * hp.inv_hidden + hp.eps);
}

// ------------------------------------------------------------------------------------------------ K2a (round 4)
// The first thing K2 does with a row -- sum the head's split-K planes of the cond and the uncond row, apply the folded norm's row scale and the
// 16-bit rounding of the lm_head output, combine with the guidance scale (JL:104) -- on the WHOLE chip instead of on the row's one CU: a row of
// Emu3's 32768-column window is 524 KB of planes and eight dependent trips for them (33 of K2's 89 us by its phase stamps).  Here every thread
// owns four columns of one row; the result z[row][col - col0] (fp32) is what K2 then reads as a head of ONE plane, no uncond row, no scale, no
// rounding (the grammar mask stays in K2).  Same operations in the same order as K2's own pass 1: bit-identical scores.  Used when the head's
// window is wide (ops.logits_to_probs_sample_part): Emu3 K2 89.9 -> 71.7 + 7.2 us.
__global__ __launch_bounds__(256) void k2a_head_combine(const sjd_head_partials hp, float guidance, int V, const sjd_iter_params *__restrict__ params,
                                                        float *__restrict__ zbuf)
{
    const int row = blockIdx.y;
    const int c4 = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 4;           // column offset inside the head's window (n_cols % 4 == 0)
    if (c4 >= hp.n_cols) return;
    const int n_rows_dev = params->n_rows, use_cfg_dev = params->use_cfg;
    const bool two = hp.urow_off > 0;
    const float *c = hp.part + (size_t)row * hp.row_stride + c4;
    const float *u = hp.part + (size_t)(hp.urow_off + row) * hp.row_stride + c4;
    float zc[4] = {0.f, 0.f, 0.f, 0.f}, zu[4] = {0.f, 0.f, 0.f, 0.f};
    if (hp.n_chunks <= 8) {                    // every plane of both rows in flight before the first add; summed in chunk order
        float4 a[8], b[8];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch)
            if (ch < hp.n_chunks) {
                a[ch] = *reinterpret_cast<const float4 *>(c + (size_t)ch * hp.chunk_stride);
                if (two) b[ch] = *reinterpret_cast<const float4 *>(u + (size_t)ch * hp.chunk_stride);
            }
#pragma unroll
        for (int ch = 0; ch < 8; ++ch)
            if (ch < hp.n_chunks) {
                zc[0] += a[ch].x; zc[1] += a[ch].y; zc[2] += a[ch].z; zc[3] += a[ch].w;
                if (two) { zu[0] += b[ch].x; zu[1] += b[ch].y; zu[2] += b[ch].z; zu[3] += b[ch].w; }
            }
    } else {
        for (int ch = 0; ch < hp.n_chunks; ++ch) {
            const float4 a = *reinterpret_cast<const float4 *>(c + (size_t)ch * hp.chunk_stride);
            zc[0] += a.x; zc[1] += a.y; zc[2] += a.z; zc[3] += a.w;
            if (two) { const float4 b = *reinterpret_cast<const float4 *>(u + (size_t)ch * hp.chunk_stride); zu[0] += b.x; zu[1] += b.y; zu[2] += b.z; zu[3] += b.w; }
        }
    }
    if (row >= n_rows_dev) return;
    float rc = 1.0f, ru = 1.0f;
    k2_row_scales(hp, row, rc, ru);
    const bool cfg = two && use_cfg_dev;
    float z[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        zc[j] = k2_round16(zc[j] * rc, hp.round_dtype);
        if (cfg) zu[j] = k2_round16(zu[j] * ru, hp.round_dtype);
        z[j] = zc[j];
        if (cfg) { float t = zc[j] - zu[j]; t = guidance * t; z[j] = t + zu[j]; }
        const int col = hp.col0 + c4 + j;
        if (hp.dbg_c && col < V) {               // observers (tests): the logits exactly as derived
            hp.dbg_c[(size_t)row * V + col] = zc[j];
            if (cfg && hp.dbg_u) hp.dbg_u[(size_t)row * V + col] = zu[j];
        }
    }
    *reinterpret_cast<float4 *>(zbuf + (size_t)row * hp.n_cols + c4) = float4{z[0], z[1], z[2], z[3]};
}

#define K2_NI 3               // column groups (of 4 x 1024) a thread keeps in registers: windows of up to 12288 columns (9 groups = Emu3's 32768-column rows no longer fit the 128 VGPRs of a 1024-thread workgroup: spills)
// SLOTS (round 6): blockIdx.y = the slot of a continuous batch (sjd_slots in include/sjd_hip.h) -- every per-slot pointer moves by the slot's
// stride before anything else happens, the rest is the one-slot kernel.  The one-slot instantiation is untouched (hp binds to the kernel argument).
template <bool PART, bool SLOTS = false>
__global__ __launch_bounds__(SJD_TPB) void k2_logits_to_probs_sample(
    const float *__restrict__ logits_c, const float *__restrict__ logits_u, long row_stride, float guidance, int V,
    const sjd_iter_params *__restrict__ params, const float *__restrict__ noise, float *__restrict__ probs_out,
    int64_t *__restrict__ tokens_out, const sjd_head_partials hp_arg, int64_t *__restrict__ amax_out, int lds_floats, const sjd_slots sb)
{
    __shared__ SjdShared sh;
    extern __shared__ __attribute__((aligned(16))) float sjd_dyn_lds[];      // round 4: the staged scores of a row too wide for registers
    sjd_head_partials hp_slot;
    if constexpr (SLOTS) {
        const size_t s = blockIdx.y;
        hp_slot = hp_arg;
        hp_slot.part += s * (size_t)sb.head_rows * (size_t)hp_arg.row_stride;
        if (hp_slot.row_sumsq) hp_slot.row_sumsq += s * (size_t)sb.head_rows;
        if (hp_slot.zero_state) hp_slot.zero_state += s * (size_t)sb.zero_state_stride;
        if (hp_slot.dbg_c) hp_slot.dbg_c += s * (size_t)sb.dbg_stride;
        if (hp_slot.dbg_u) hp_slot.dbg_u += s * (size_t)sb.dbg_stride;
        params = reinterpret_cast<const sjd_iter_params *>(reinterpret_cast<const char *>(params) + s * (size_t)sb.params_stride);
        probs_out += s * (size_t)sb.probs_stride;
        tokens_out = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(tokens_out) + s * (size_t)sb.state_stride);
        if (amax_out) amax_out = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(amax_out) + s * (size_t)sb.state_stride);
    }
    const sjd_head_partials &hp = SLOTS ? hp_slot : hp_arg;
    const int row = blockIdx.x;
    SJD_TRS(row, 0);
    // Round 4: everything the row needs before its first sum is requested in ONE round trip -- the row count, the rule, the folded norm's row
    // statistics of both batch rows -- and only then looked at (row < gridDim.x <= SJD_MAX_WINDOW: the rule's slot exists whatever n_rows is).
    // By its phase stamps the kernel used to make five dependent trips to cold memory before the first add: n_rows, the rule, the statistics
    // of the cond row, of the uncond row, the partial planes.
    asm volatile("" :: "s"(hp.zero_state), "s"(hp.row_sumsq), "s"(params));     // (these kernel arguments are wanted by the first batch: fetch them first)
    float sv[8], sw[8];
    const bool stats8 = PART && hp.row_sumsq && hp.slices <= 8;          // (hidden sizes up to 4096: one batch of scalar loads, no branch around any)
    {   // (unconditional loads from a valid address either way: a branch here would end the basic block and with it the overlap)
        const float *ss = stats8 ? hp.row_sumsq : reinterpret_cast<const float *>(params);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const size_t off = stats8 ? (size_t)min(q, hp.slices - 1) * hp.prows + row : 0;
            sv[q] = ss[off];
            sw[q] = ss[off + ((stats8 && hp.urow_off > 0) ? hp.urow_off : 0)];
        }
    }
    const int n_rows_dev = params->n_rows, use_cfg_dev = params->use_cfg;
    const uint32_t ph_blocks = (uint32_t)params->philox_blocks;
    const uint64_t ph_seed = params->philox_seed, ph_off = params->philox_offset[0];
    // what this row of probs_out is known to hold: zeros outside [zlo, zhi) (zlo < 0: unknown) -- see sjd_head_partials::zero_state
    int *zst = (PART && hp.zero_state) ? hp.zero_state + 2 * (row < SJD_MAX_WINDOW ? row : 0) : nullptr;
    const int *zp = zst ? zst : reinterpret_cast<const int *>(params);        // (unconditional loads, as above)
    const int zl = zp[0], zh = zp[1];
    const sjd_row_rule rule = params->rules[row < SJD_MAX_WINDOW ? row : 0];
    float rc = 1.0f, ru = 1.0f;
    // (the compiler sinks the rule's loads below the early return otherwise: a trip of their own; these empty statements need the values HERE,
    //  where the wait for n_rows stands anyway)
    asm volatile("" :: "s"(zl), "s"(zh), "s"(use_cfg_dev), "s"(ph_blocks), "s"(ph_seed), "s"(ph_off), "s"(n_rows_dev),
                 "s"(rule.n_ranges), "s"(rule.forced), "s"(rule.temperature), "s"(sv[0]), "s"(sv[1]), "s"(sv[2]), "s"(sv[3]), "s"(sv[4]),
                 "s"(sv[5]), "s"(sv[6]), "s"(sv[7]), "s"(sw[0]), "s"(sw[1]), "s"(sw[2]), "s"(sw[3]), "s"(sw[4]), "s"(sw[5]), "s"(sw[6]), "s"(sw[7]));
    if (row >= n_rows_dev) return;
    SJD_TRS(row, 1);              // first batch (row count, rule, statistics, zero state) arrived
    const int zlo = zst ? zl : -1, zhi = zst ? zh : -1;
    if (stats8) {
        float t = 0.f, tu = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) { t += (q < hp.slices) ? sv[q] : 0.f; tu += (q < hp.slices) ? sw[q] : 0.f; }      // row_sumsq_total's order
        rc = rsqrtf(t * hp.inv_hidden + hp.eps);
        if (hp.urow_off > 0) ru = rsqrtf(tu * hp.inv_hidden + hp.eps);
    } else if (PART) k2_row_scales(hp, row, rc, ru);
    if (threadIdx.x == 0) sh.misc[1] = 0;         // entries of the draw list (the first barrier lies far ahead of its first use)
    float *p = probs_out + (size_t)row * V;
    const float *e = noise + (size_t)row * V;
    // the Exp(1) noise of the draw: read from `noise`, or (params->philox_blocks > 0) generated here -- the elements torch's
    // empty(n_rows, V).exponential_(generator=g) would hold, for the columns that carry probability mass only (sjd_philox.h)
    const uint32_t ph_T = ph_blocks ? sjd_philox_threads((uint64_t)n_rows_dev * (uint64_t)V, ph_blocks) : 0u;

    if (rule.forced >= 0) {   // forced EOL / end-of-image row: softmax of (-inf,...,0,...,-inf) (LP:39-41)
        // (with a known state only the hull of the old window and the forced column needs rewriting)
        const int flo = zlo >= 0 ? min(zlo, rule.forced) : 0, fhi = zlo >= 0 ? max(zhi, rule.forced + 1) : V;
        SJD_FOR_OWNED_COLS_IN(flo, fhi, c0)
            for (int j = 0; j < 4; ++j)
                if (c0 + j >= flo && c0 + j < fhi) p[c0 + j] = (c0 + j == rule.forced) ? 1.0f : 0.0f;
        if (zst) {
            __syncthreads();                       // every thread has read the old state
            if (threadIdx.x == 0) { zst[0] = rule.forced; zst[1] = rule.forced + 1; }
        }
        if (threadIdx.x == 0) { tokens_out[row] = rule.forced; if (amax_out) amax_out[row] = rule.forced; }
        return;
    }

    const float *c, *u;
    if (PART) {                 // virtual column 0 of this row's first chunk; only columns [col0, col0 + n_cols) exist
        row_stride = hp.row_stride;
        c = hp.part + (size_t)row * row_stride - hp.col0;
        u = (hp.urow_off > 0 && use_cfg_dev) ? hp.part + (size_t)(hp.urow_off + row) * row_stride - hp.col0 : nullptr;
    } else {
        c = logits_c + (size_t)row * row_stride;
        u = (logits_u != nullptr && use_cfg_dev) ? logits_u + (size_t)row * row_stride : nullptr;
    }
    const bool vec = ((V & 3) == 0) && ((row_stride & 3) == 0) && (!PART || (hp.col0 & 3) == 0);
    // Only the window [wlo, whi) spanned by the rule's allowed ranges can hold probability mass (Lumina image rows: 8192 of
    // 65536 columns); everything outside is written as 0 once and never read again.
    int wlo, whi;
    rule_window(rule, V, wlo, whi);
    const int c0_first = 4 * sjd_first_owned_group(wlo);
    const bool clean_outside = zlo >= 0 && zlo >= wlo && zhi <= whi;      // the recorded window lies inside this one: outside is zero already
    if (!clean_outside && (wlo > 0 || whi < V)) {
        const bool v4 = (V & 3) == 0;              // (rows of probs_out are then 16-byte aligned)
        SJD_FOR_OWNED_COLS(V, c0) {
            if (v4 && (c0 + 3 < wlo || c0 >= whi)) *reinterpret_cast<float4 *>(p + c0) = float4{0.f, 0.f, 0.f, 0.f};
            else if (c0 < wlo || c0 + 3 >= whi)
                for (int j = 0; j < 4; ++j) { int col = c0 + j; if (col < V && (col < wlo || col >= whi)) p[col] = 0.0f; }
        }
    }

    // Where the scores of the window are STAGED between the passes: registers when they fit (K2_NI groups per thread), else the workgroup's
    // LDS (round 4: Emu3's 32768-column rows -- every pass used to re-read them from `p` through L2, a dependent round trip per column
    // group and pass), else `p` itself (text rows over a whole vocabulary).  `stg[col - sb]` addresses either.
    const bool in_lds = (whi - (wlo & ~3)) <= lds_floats;
    // The passes are instantiated twice, for a row staged in LDS and for one staged in `p`: with ONE pointer that may be either, every staged access
    // was a FLAT instruction (104 flat_load_dword + 28 flat_store_dword in this kernel's ISA, round 4) -- single dwords through the address-space check
    // instead of ds_read / ds_write.
    auto passes = [&](auto lds_tag) {
    constexpr bool STG_LDS = decltype(lds_tag)::value;
    float *stg;                                     // column `col` lives at stg[col - sb] (no pointer is ever moved below the LDS base)
    if constexpr (STG_LDS) stg = sjd_dyn_lds; else stg = p;
    const int sb = STG_LDS ? (wlo & ~3) : 0;
    SJD_TRS(row, 2);              // outside of the window zeroed
    // pass 1: CFG combine (JL:104) + grammar mask (LP:125-129); stage z; row max; finite count
    // Round 3: when the rule's window is at most K2_NI column groups per thread (Lumina's image rows: 3, Emu3's: 9) the staged scores are
    // taken back into REGISTERS once, behind this pass, and stay there to the final probabilities -- the radix select, the exponentials and
    // the draw read them there instead of re-reading `p` through L2 in every pass (each pass was a dependent round trip per column group:
    // 8 of them per pass for Emu3's 32768-column rows).  Same values, same visiting order, same accumulators: bit-identical.  Wider windows
    // (text rows over the whole vocabulary) keep the staged form.
    float tmax = -INFINITY;
    int cnt = 0;
    SJD_FOR_OWNED_COLS_IN(wlo, whi, c0) {
        float zc[4], zu[4];
        if (PART) {
            const bool in4 = vec && c0 >= hp.col0 && c0 + 3 < hp.col0 + hp.n_cols && c0 + 3 < V;
#pragma unroll
            for (int j = 0; j < 4; ++j) { zc[j] = 0.f; zu[j] = 0.f; }
            if (in4 && hp.n_chunks <= 8) {
                // all chunk planes of this thread's four columns are requested before the first add (in-kernel timestamps, round 2: one
                // plane at a time made this pass 8 dependent round trips to cold partials, 17 of K2's 44 us); summed in chunk order
                float4 a[8], b[8];
#pragma unroll
                for (int ch = 0; ch < 8; ++ch)
                    if (ch < hp.n_chunks) {
                        const size_t off = (size_t)ch * hp.chunk_stride + c0;
                        a[ch] = *reinterpret_cast<const float4 *>(c + off);
                        if (u) b[ch] = *reinterpret_cast<const float4 *>(u + off);
                    }
#pragma unroll
                for (int ch = 0; ch < 8; ++ch)
                    if (ch < hp.n_chunks) {
                        zc[0] += a[ch].x; zc[1] += a[ch].y; zc[2] += a[ch].z; zc[3] += a[ch].w;
                        if (u) { zu[0] += b[ch].x; zu[1] += b[ch].y; zu[2] += b[ch].z; zu[3] += b[ch].w; }
                    }
            } else
            for (int ch = 0; ch < hp.n_chunks; ++ch) {           // chunk order = the summation order of every G1 consumer
                const size_t off = (size_t)ch * hp.chunk_stride + c0;
                if (in4) {
                    const float4 a = *reinterpret_cast<const float4 *>(c + off);
                    zc[0] += a.x; zc[1] += a.y; zc[2] += a.z; zc[3] += a.w;
                    if (u) { const float4 b = *reinterpret_cast<const float4 *>(u + off); zu[0] += b.x; zu[1] += b.y; zu[2] += b.z; zu[3] += b.w; }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int col = c0 + j;
                        if (col >= hp.col0 && col < hp.col0 + hp.n_cols && col < V) { zc[j] += c[off + j]; if (u) zu[j] += u[off + j]; }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                zc[j] = k2_round16(zc[j] * rc, hp.round_dtype);
                if (u) zu[j] = k2_round16(zu[j] * ru, hp.round_dtype);
                const int col = c0 + j;
                if (hp.dbg_c && col >= wlo && col < whi) {       // observers (tests): the logits exactly as this kernel derived them
                    hp.dbg_c[(size_t)row * V + col] = zc[j];
                    if (u && hp.dbg_u) hp.dbg_u[(size_t)row * V + col] = zu[j];
                }
            }
        } else if (vec && c0 + 3 < V) {
            float4 a = *reinterpret_cast<const float4 *>(c + c0);
            zc[0] = a.x; zc[1] = a.y; zc[2] = a.z; zc[3] = a.w;
            if (u) { float4 b = *reinterpret_cast<const float4 *>(u + c0); zu[0] = b.x; zu[1] = b.y; zu[2] = b.z; zu[3] = b.w; }
        } else {
            for (int j = 0; j < 4; ++j) { zc[j] = (c0 + j < V) ? c[c0 + j] : 0.f; zu[j] = (u && c0 + j < V) ? u[c0 + j] : 0.f; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int col = c0 + j;
            if (col >= wlo && col < whi) {
                float z = zc[j];
                if (u) { float t = zc[j] - zu[j]; t = guidance * t; z = t + zu[j]; }
                if (!rule_allows(rule, col)) z = -INFINITY;
                stg[col - sb] = z;
                tmax = fmaxf(tmax, z);
                cnt += (z > -INFINITY) ? 1 : 0;
            }
        }
    }
    // the thread's own staged scores come back into registers in ONE round trip (all groups requested together) and stay there
    const bool fits = (whi - (wlo & ~3) + 4 * SJD_TPB - 1) / (4 * SJD_TPB) <= K2_NI;       // (the same for every thread of the block)
    float zr[K2_NI][4];
    if (fits) {
#pragma unroll
        for (int i = 0; i < K2_NI; ++i) {
            const int c0 = c0_first + i * 4 * SJD_TPB;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = c0 + j;
                const bool in = c0 < whi && col >= wlo && col < whi;
                const float v = stg[(in ? col : wlo) - sb];            // (unconditional load: every group of the thread in flight at once)
                zr[i][j] = in ? v : -INFINITY;                  // a column outside the window: never counted, never kept
            }
        }
    }
    SJD_TRS(row, 3);              // planes summed, scores staged
    int n_finite;
    const float zmax = block_max_and_count(tmax, cnt, sh, n_finite);       // (its barriers also make the staged z visible to the whole block)
    SJD_TRS(row, 4);              // logits staged, max known

    // top-k (LP:196-204): keep z >= k-th largest; k-th is -inf when fewer than k finite entries exist
    float kth = -INFINITY;
    if (rule.top_k > 0 && rule.top_k < V && rule.top_k < n_finite)
    {
        // (zmax is finite here: n_finite > top_k >= 1; a +inf score would make the value bins meaningless -> the plain radix select)
        if (!(zmax < INFINITY)) kth = block_kth_largest_bisect(stg, wlo, whi, rule.top_k, -INFINITY, sh, sb);
        else if (fits)
            kth = block_kth_largest_spread([&](auto &&f) {
#pragma unroll
                for (int i = 0; i < K2_NI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) f(zr[i][j]);
            }, rule.top_k, zmax, sh);
        else
            kth = block_kth_largest_spread([&](auto &&f) {
                SJD_FOR_OWNED_COLS_IN(wlo, whi, c0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int col = c0 + j; if (col >= wlo && col < whi) f(stg[col - sb]); }
                }
            }, rule.top_k, zmax, sh);
    }

    SJD_TRS(row, 5);              // top-k threshold known
    // pass A: e = exp(z - max) for kept entries, canonical sum.  TemperatureLogitsWarper (rule.temperature != 1): the kept scores are
    // divided by T first -- after the grammar mask and its top-k, before top-p and the softmax, where HF's generate() places the warper;
    // max(z / T) == max(z) / T because the division is monotone
    const bool tempered = rule.temperature > 0.0f && rule.temperature != 1.0f;
    const float zmax_t = tempered ? zmax / rule.temperature : zmax;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const bool top_p_on = rule.top_p_thr >= 0.0f;
    if (fits) {
#pragma unroll
        for (int i = 0; i < K2_NI; ++i) {
            const int c0 = c0_first + i * 4 * SJD_TPB;
            if (c0 < whi) {
                float ev[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = c0 + j;
                    ev[j] = 0.0f;
                    if (col >= wlo && col < whi) {
                        const float z = zr[i][j];
                        const float zt = tempered ? z / rule.temperature : z;
                        ev[j] = (z < kth) ? 0.0f : sjd_expf(zt - zmax_t);
                        if (top_p_on) stg[col - sb] = ev[j];           // (the top-p cut works on the staged weights)
                    }
                    zr[i][j] = ev[j];
                }
                a0 = a0 + ev[0]; a1 = a1 + ev[1]; a2 = a2 + ev[2]; a3 = a3 + ev[3];
            }
        }
    } else {
        SJD_FOR_OWNED_COLS_IN(wlo, whi, c0) {
            float ev[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int col = c0 + j;
                ev[j] = 0.0f;
                if (col >= wlo && col < whi) {
                    float z = stg[col - sb];
                    const float zt = tempered ? z / rule.temperature : z;
                    ev[j] = (z < kth) ? 0.0f : sjd_expf(zt - zmax_t);
                    stg[col - sb] = ev[j];
                }
            }
            a0 = a0 + ev[0]; a1 = a1 + ev[1]; a2 = a2 + ev[2]; a3 = a3 + ev[3];
        }
    }
    float S = block_canonical_sum(a0, a1, a2, a3, sh);
    if (top_p_on) {                                                                             // TopPLogitsWarper3d (LP:406-419)
        S = block_top_p_apply(stg, wlo, whi, S, rule.top_p_thr, sh, sb);
        if (fits) {                                  // the cut zeroed some staged weights: take them back into the registers
            __syncthreads();
#pragma unroll
            for (int i = 0; i < K2_NI; ++i) {
                const int c0 = c0_first + i * 4 * SJD_TPB;
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int col = c0 + j; if (c0 < whi && col >= wlo && col < whi) zr[i][j] = stg[col - sb]; }
            }
        }
    }

    SJD_TRS(row, 6);              // sum known
    // pass B: p = e / S ; multinomial == lowest-index argmax of p / Exp(1)   (JL:111-118)
    unsigned long long best = 0ull, best_p = 0ull;
    // in-kernel noise: the entries with mass are compacted into an LDS list (behind the staged row) and drawn densely below (wave_push)
    const int list_base = in_lds ? ((whi - (wlo & ~3) + 3) & ~3) : 0;
    unsigned long long *klist = reinterpret_cast<unsigned long long *>(sjd_dyn_lds + list_base);
    const int list_cap = (lds_floats - list_base) / 2;
    const bool compact = ph_blocks != 0 && list_cap >= 64;
    auto draw = [&](int col, float w) {
        float pv = w / S;
        p[col] = pv;
        unsigned long long cand = pack_vi(pv, col);
        best_p = cand > best_p ? cand : best_p;
        float r;
        if (ph_blocks) r = pv > 0.0f ? pv / sjd_philox_exponential(ph_seed, ph_off, ph_T, (uint64_t)row * (uint64_t)V + (uint64_t)col) : 0.0f;   // (0 / e == 0: e is finite and > 0)
        else r = pv / e[col];
        cand = pack_vi(r, col);
        best = cand > best ? cand : best;
    };
    if (compact) {
        // the list is filled with one reservation per wave (wave_reserve): a first walk stores the probabilities and counts the thread's entries
        // with mass, a second appends them from the thread's first position on
        int n_mine = 0;
        auto first = [&](int col, float w) -> float {
            const float pv = w / S;
            p[col] = pv;
            unsigned long long cand = pack_vi(pv, col);
            best_p = cand > best_p ? cand : best_p;
            if (pv > 0.0f) ++n_mine;
            else { cand = pack_vi(0.0f, col); best = cand > best ? cand : best; }       // (r = 0 for an entry without mass, as before)
            return pv;
        };
        if (fits) {
#pragma unroll
            for (int i = 0; i < K2_NI; ++i) {
                const int c0 = c0_first + i * 4 * SJD_TPB;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = c0 + j;
                    zr[i][j] = (c0 < whi && col >= wlo && col < whi) ? first(col, zr[i][j]) : 0.0f;
                }
            }
        } else {
            SJD_FOR_OWNED_COLS_IN(wlo, whi, c0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int col = c0 + j; if (col >= wlo && col < whi) stg[col - sb] = first(col, stg[col - sb]); }
            }
        }
        int pos = wave_reserve(n_mine, &sh.misc[1]);
        auto second = [&](int col, float pv) {
            if (pv > 0.0f) {
                if (pos < list_cap) klist[pos] = ((unsigned long long)__float_as_uint(pv) << 32) | (unsigned)col;
                ++pos;
            }
        };
        if (fits) {
#pragma unroll
            for (int i = 0; i < K2_NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) second(c0_first + i * 4 * SJD_TPB + j, zr[i][j]);
        } else {
            SJD_FOR_OWNED_COLS_IN(wlo, whi, c0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int col = c0 + j; if (col >= wlo && col < whi) second(col, stg[col - sb]); }
            }
        }
    } else if (fits) {
#pragma unroll
        for (int i = 0; i < K2_NI; ++i) {
            const int c0 = c0_first + i * 4 * SJD_TPB;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int col = c0 + j; if (c0 < whi && col >= wlo && col < whi) draw(col, zr[i][j]); }
        }
    } else {
        SJD_FOR_OWNED_COLS_IN(wlo, whi, c0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int col = c0 + j; if (col >= wlo && col < whi) draw(col, stg[col - sb]); }
        }
    }
    SJD_TRS(row, 7);              // probabilities stored, draw list filled
    if (compact) {
        __syncthreads();
        const int n_kept = sh.misc[1];
        if (n_kept <= list_cap) {
            for (int i = threadIdx.x; i < n_kept; i += SJD_TPB) {
                const unsigned long long en = klist[i];
                const int col = (int)(unsigned)(en & 0xffffffffull);
                const float pv = __uint_as_float((unsigned)(en >> 32));
                const unsigned long long cand = pack_vi(pv / sjd_philox_exponential(ph_seed, ph_off, ph_T, (uint64_t)row * (uint64_t)V + (uint64_t)col), col);
                best = cand > best ? cand : best;
            }
        } else {                                    // (more entries with mass than the list holds -- rows without a top-k: the sparse form)
            SJD_FOR_OWNED_COLS_IN(wlo, whi, c0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = c0 + j;
                    if (col >= wlo && col < whi) {
                        const float pv = p[col];
                        if (pv > 0.0f) {
                            const unsigned long long cand = pack_vi(pv / sjd_philox_exponential(ph_seed, ph_off, ph_T, (uint64_t)row * (uint64_t)V + (uint64_t)col), col);
                            best = cand > best ? cand : best;
                        }
                    }
                }
            }
        }
    }
    SJD_TRS(row, 8);              // probabilities written
    int am;                                          // by-product: the row's mode (lowest index among equal maxima)
    const int tok = block_argmax2(best, best_p, sh, am);
    if (threadIdx.x == 0) {
        tokens_out[row] = tok;
        if (amax_out) amax_out[row] = am;
        if (zst) { zst[0] = wlo; zst[1] = whi; }    // (every read of the old state lies behind several barriers)
    }
    SJD_TRS(row, 9);
    };
    if (in_lds) passes(std::true_type{}); else passes(std::false_type{});
}

// ------------------------------------------------------------------------------------------------ K4
// The per-iteration read-back without a copy engine: the last kernel of an iteration writes the state blob (sizeof(sjd_state)) into HOST memory the
// device can address (pinned, fine-grained), so the host's single sync of the iteration is a stream synchronize and nothing else.
// rocprofv3 kernel trace, round 2 (tools/step_gaps.py): the D2H copy that used to follow K4 started ~10 us after it (hand-over from
// the compute queue to the SDMA engine) on top of the copy itself.  All threads of the block call this (barrier inside); the words
// are read back with device-scope loads: every store of this block has reached L2 at the barrier, K2's came with the previous kernel.
__device__ __forceinline__ void k4_mirror_state(const sjd_state *state, sjd_state *host_mirror, int iter_seq)
{
    if (!host_mirror) return;
    __syncthreads();
    constexpr int WORDS = (int)(sizeof(sjd_state) / sizeof(unsigned long long));
    static_assert(sizeof(sjd_state) % sizeof(unsigned long long) == 0, "sjd_state is copied in 8-byte words");
    if (threadIdx.x < WORDS) {
        const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(state) + threadIdx.x, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
        reinterpret_cast<unsigned long long *>(host_mirror)[threadIdx.x] = v;
    }
    __threadfence_system();
    __syncthreads();                      // every word is out and fenced: publish the sequence number the host polls (sjd_host_wait_u64)
    if (threadIdx.x == 0)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(host_mirror) + WORDS, (unsigned long long)(unsigned)iter_seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

// replaces SpeculativeSampler.__call__ / find_first_misaligned_token_inds (reference JL:247-333)
template <bool SLOTS = false>
__global__ __launch_bounds__(SJD_TPB) void k4_verify_accept(
    const sjd_iter_params *__restrict__ params, sjd_state *__restrict__ state, const float *__restrict__ probs,
    const float *__restrict__ prev_probs, const float *__restrict__ rs, const float *__restrict__ noise2,
    float *__restrict__ scratch_global, int V, sjd_state *__restrict__ host_mirror, int lds_floats, const sjd_slots sb)
{
    __shared__ SjdShared sh;
    extern __shared__ __attribute__((aligned(16))) float sjd_dyn_lds[];      // round 4: the residual row, when its window fits (else `scratch`)
    if constexpr (SLOTS) {        // blockIdx.y = the slot (see k2_logits_to_probs_sample)
        const size_t s = blockIdx.y;
        params = reinterpret_cast<const sjd_iter_params *>(reinterpret_cast<const char *>(params) + s * (size_t)sb.params_stride);
        state = reinterpret_cast<sjd_state *>(reinterpret_cast<char *>(state) + s * (size_t)sb.state_stride);
        probs += s * (size_t)sb.probs_stride;
        prev_probs += s * (size_t)sb.probs_stride;
        scratch_global += s * (size_t)sb.scratch_stride;
        if (host_mirror) host_mirror = reinterpret_cast<sjd_state *>(reinterpret_cast<char *>(host_mirror) + s * (size_t)sb.mirror_stride);
    }
    SJD_TRS(32, 0);
    const int n = params->n_rows;
    if (n <= 1) {   // prefill / single-token phase short-circuit (JL:344-350)
        if (threadIdx.x == 0) { state->m = 1; state->rejected = 0; state->n_prev = n; }
        k4_mirror_state(state, host_mirror, params->iter_seq);
        return;
    }
    const int scheme = params->scheme;
    // phase 1: one lane per draft; ballot + find-first-zero = longest accepted prefix
    if (threadIdx.x < 64) {
        const int i = threadIdx.x;
        bool acc = true;
        long x = 0;
        if (i >= 1 && i < n) {
            x = state->win_tok[i];
            if (scheme == 0) {
                float pa = probs[(size_t)(i - 1) * V + x];
                int qs = state->q_src[i];
                float qd = (qs >= 0) ? prev_probs[(size_t)qs * V + x] : 1.0f;
                float ratio = pa / qd;                       // JL:286 (NaN compares false => reject)
                float uu;                                    // JL:282: rs[b, i, x] of torch.rand((1, n, V)); generated in place when
                if (params->philox_blocks > 0)               // the noise is not handed over as tensors (15 uniforms instead of n * V)
                    uu = sjd_philox_rand(params->philox_seed, params->philox_offset[1],
                                         sjd_philox_threads((uint64_t)n * (uint64_t)V, (uint32_t)params->philox_blocks), (uint64_t)i * (uint64_t)V + (uint64_t)x);
                else uu = rs[(size_t)i * V + x];
                acc = uu < (ratio > 1.0f ? 1.0f : ratio);
            } else {
                acc = (x == state->tokens[i - 1]);           // JL:325
            }
        }
        unsigned long long ok = __ballot(acc);
        unsigned long long window = (n >= 64) ? ~1ull : (((1ull << n) - 1ull) & ~1ull);
        unsigned long long rej = (~ok) & window;
        int m = rej ? (__ffsll((long long)rej) - 1) : n;
        if (scheme == 0 && i >= 1 && i < m) state->tokens[i - 1] = x;   // accepted drafts (JL:288)
        if (i == 0) sh.misc[0] = m;
    }
    __syncthreads();
    const int m = sh.misc[0];
    SJD_TRS(32, 1);               // accept tests done, prefix known
    const bool rejected = (scheme == 0) && (m < n);
    const uint32_t ph_blocks = (uint32_t)params->philox_blocks;
    const uint64_t ph_seed = params->philox_seed, ph_off2 = params->philox_offset[2];
    const uint32_t ph_T2 = ph_blocks ? sjd_philox_threads((uint64_t)V, ph_blocks) : 0u;
    bool degenerate = false;        // residual distribution empty under the residual rule: the reference's torch.multinomial raises
    if (rejected) {
        // phase 2: residual resample of position m-1 from norm(max(p - q, 0)) (JL:203-241)
        const int row = m - 1;
        const sjd_row_rule rule = params->resid_rules[row];
        if (rule.forced >= 0) {
            if (threadIdx.x == 0) state->tokens[row] = rule.forced;
        } else {
            const long x = state->win_tok[m];
            const int qs = state->q_src[m];
            const float *prow = probs + (size_t)row * V;
            const float *qrow = (qs >= 0) ? prev_probs + (size_t)qs * V : nullptr;
            int wlo, whi;
            rule_window(rule, V, wlo, whi);
            // the residual weights d are staged in the workgroup's LDS when the rule's window fits (image rows: 8192 .. 32768 columns) -- the
            // five passes below were five trips through L2 per column group (74 us at Emu3's shape, one workgroup) -- else in `scratch`
            // (instantiated twice -- staged in LDS / in `scratch` -- so that the staged accesses are ds_ / global_ instructions, not FLAT ones: see K2)
            const bool res_lds = (whi - (wlo & ~3)) <= lds_floats;
            auto resample = [&](auto lds_tag) {
            constexpr bool STG_LDS = decltype(lds_tag)::value;
            float *scratch;                  // column `col` of the staged row lives at scratch[col - sb]
            if constexpr (STG_LDS) scratch = sjd_dyn_lds; else scratch = scratch_global;
            const int sb = STG_LDS ? (wlo & ~3) : 0;
            int cnt = 0;
            SJD_FOR_OWNED_COLS_IN(wlo, whi, c0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int col = c0 + j;
                    if (col >= wlo && col < whi) {
                        float qv = qrow ? qrow[col] : ((long)col == x ? 1.0f : 0.0f);
                        float d = prow[col] - qv;
                        d = d > 0.0f ? d : 0.0f;
                        if (!rule_allows(rule, col)) d = 0.0f;
                        scratch[col - sb] = d;
                        cnt += d > 0.0f ? 1 : 0;
                    }
                }
            }
            SJD_TRS(32, 2);       // residual row staged
            const int n_pos = block_sum_int(cnt, sh);
            __syncthreads();
            float kth = 0.0f;
            if (rule.top_k > 0 && rule.top_k < V && rule.top_k < n_pos) kth = block_kth_largest_bisect(scratch, wlo, whi, rule.top_k, 0.0f, sh, sb);
            {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                SJD_FOR_OWNED_COLS_IN(wlo, whi, c0) {
                    float dv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int col = c0 + j;
                        dv[j] = 0.0f;
                        if (col >= wlo && col < whi) {
                            float d = scratch[col - sb];
                            dv[j] = (d < kth) ? 0.0f : d;
                            scratch[col - sb] = dv[j];
                        }
                    }
                    a0 = a0 + dv[0]; a1 = a1 + dv[1]; a2 = a2 + dv[2]; a3 = a3 + dv[3];
                }
                float S = block_canonical_sum(a0, a1, a2, a3, sh);
                if (rule.temperature > 0.0f && rule.temperature != 1.0f && S > 0.0f) {
                    // TemperatureLogitsWarper on the residual logits log(d): weights exp(log(d) / T - max) instead of d (JL:203-241 with the
                    // warper in the processor list); log d through the canonical sjd_logf the oracle restates
                    float dm = 0.f;
                    SJD_FOR_OWNED_COLS_IN(wlo, whi, c0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            int col = c0 + j;
                            if (col >= wlo && col < whi) dm = fmaxf(dm, scratch[col - sb]);
                        }
                    }
                    dm = block_max(dm, sh);
                    const float lm = sjd_logf(dm) / rule.temperature;
                    a0 = a1 = a2 = a3 = 0.f;
                    SJD_FOR_OWNED_COLS_IN(wlo, whi, c0) {
                        float dv[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            int col = c0 + j;
                            dv[j] = 0.0f;
                            if (col >= wlo && col < whi) {
                                const float d = scratch[col - sb];
                                dv[j] = d > 0.0f ? sjd_expf(sjd_logf(d) / rule.temperature - lm) : 0.0f;
                                scratch[col - sb] = dv[j];
                            }
                        }
                        a0 = a0 + dv[0]; a1 = a1 + dv[1]; a2 = a2 + dv[2]; a3 = a3 + dv[3];
                    }
                    S = block_canonical_sum(a0, a1, a2, a3, sh);
                }
                if (rule.top_p_thr >= 0.0f) S = block_top_p_apply(scratch, wlo, whi, S, rule.top_p_thr, sh, sb);
                SJD_TRS(32, 3);   // (top-k,) sum known
                degenerate = !(S > 0.0f);         // 0/0 below: flagged to the host (state->rejected = 2), never a silent arbitrary id
                unsigned long long best = 0ull;
                // in-kernel noise: the entries with mass go through a compacted LDS list (wave_push), the noise is evaluated densely
                const int list_base = STG_LDS ? ((whi - sb + 3) & ~3) : 0;           // behind the staged row, if it is staged in LDS
                unsigned long long *klist = reinterpret_cast<unsigned long long *>(sjd_dyn_lds + list_base);
                const int list_cap = (lds_floats - list_base) / 2;
                const bool compact = ph_blocks != 0 && list_cap >= 64;
                if (compact) {
                    if (threadIdx.x == 0) sh.misc[1] = 0;
                    __syncthreads();
                }
                if (compact) {                      // one reservation per wave (wave_reserve): count the thread's entries with mass, then append them
                    int n_mine = 0;
                    SJD_FOR_OWNED_COLS_IN(wlo, whi, c0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int col = c0 + j;
                            if (col >= wlo && col < whi) {
                                const float dv = scratch[col - sb] / S;
                                if (dv > 0.0f) ++n_mine;
                                else { const unsigned long long cand = pack_vi(dv, col); best = cand > best ? cand : best; }           // (0 or NaN, as before)
                            }
                        }
                    }
                    int pos = wave_reserve(n_mine, &sh.misc[1]);
                    SJD_FOR_OWNED_COLS_IN(wlo, whi, c0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int col = c0 + j;
                            if (col >= wlo && col < whi) {
                                const float dv = scratch[col - sb] / S;
                                if (dv > 0.0f) {
                                    if (pos < list_cap) klist[pos] = ((unsigned long long)__float_as_uint(dv) << 32) | (unsigned)col;
                                    ++pos;
                                }
                            }
                        }
                    }
                } else
                SJD_FOR_OWNED_COLS_IN(wlo, whi, c0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int col = c0 + j;
                        if (col >= wlo && col < whi) {
                            const float dv = scratch[col - sb] / S;
                            float r;
                            if (ph_blocks) r = dv > 0.0f ? dv / sjd_philox_exponential(ph_seed, ph_off2, ph_T2, (uint64_t)col) : dv;    // (dv is 0 or NaN here)
                            else r = dv / noise2[col];
                            unsigned long long cand = pack_vi(r, col);
                            best = cand > best ? cand : best;
                        }
                    }
                }
                if (compact) {
                    __syncthreads();
                    const int n_kept = sh.misc[1];
                    if (n_kept <= list_cap) {
                        for (int i = threadIdx.x; i < n_kept; i += SJD_TPB) {
                            const unsigned long long en = klist[i];
                            const int col = (int)(unsigned)(en & 0xffffffffull);
                            const float dv = __uint_as_float((unsigned)(en >> 32));
                            const unsigned long long cand = pack_vi(dv / sjd_philox_exponential(ph_seed, ph_off2, ph_T2, (uint64_t)col), col);
                            best = cand > best ? cand : best;
                        }
                    } else {
                        SJD_FOR_OWNED_COLS_IN(wlo, whi, c0) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int col = c0 + j;
                                if (col >= wlo && col < whi) {
                                    const float dv = scratch[col - sb] / S;
                                    if (dv > 0.0f) {
                                        const unsigned long long cand = pack_vi(dv / sjd_philox_exponential(ph_seed, ph_off2, ph_T2, (uint64_t)col), col);
                                        best = cand > best ? cand : best;
                                    }
                                }
                            }
                        }
                    }
                }
                SJD_TRS(32, 4);   // list drawn
                const int tok = block_argmax(best, sh);
                if (threadIdx.x == 0) state->tokens[row] = tok;
            }
            };
            if (res_lds) resample(std::true_type{}); else resample(std::false_type{});
        }
    }
    if (threadIdx.x == 0) { state->m = m; state->rejected = rejected ? (degenerate ? 2 : 1) : 0; state->n_prev = n; }
    SJD_TRS(32, 5);               // token written
    k4_mirror_state(state, host_mirror, params->iter_seq);
    SJD_TRS(32, 6);               // state mirrored into host memory
}

// ------------------------------------------------------------------------------------------------ K5
// replaces prepare_inputs_for_generation_jacobi / get_multi_token_for_preparation('random') (reference JL:470-514,
// 606-701): window = [last emitted | carried unverified samples | fresh random ids]
// positions_out (optional): the position ids of the window rows, kv_len + i + pos_offset[b] (reference JL:1062-1073 builds them on the
// host from cache_position; three ATen element-wise launches per iteration before round 2's end)
template <bool SLOTS = false>
__global__ void k5_reguess(const sjd_iter_params *__restrict__ params, sjd_state *__restrict__ state,
                           int64_t *__restrict__ input_ids_out, int n_batch, int max_rows, const int64_t *__restrict__ pos_offset,
                           int64_t *__restrict__ positions_out, const sjd_slots sb)
{
    if constexpr (SLOTS) {        // blockIdx.x = the slot: its n_batch rows of the shared [n_slots * n_batch, max_rows] id / position tensors
        const size_t s = blockIdx.x;
        params = reinterpret_cast<const sjd_iter_params *>(reinterpret_cast<const char *>(params) + s * (size_t)sb.params_stride);
        state = reinterpret_cast<sjd_state *>(reinterpret_cast<char *>(state) + s * (size_t)sb.state_stride);
        input_ids_out += s * (size_t)n_batch * (size_t)max_rows;
        if (pos_offset) pos_offset += s * (size_t)n_batch;
        if (positions_out) positions_out += s * (size_t)n_batch * (size_t)max_rows;
    }
    const int i = threadIdx.x;
    const int n = params->n_rows;
    const int m = state->m, n_prev = state->n_prev;
    int a = n_prev - m;
    if (a > n - 1) a = n - 1;
    if (a < 0) a = 0;
    int64_t tok = 0;
    int qs = -1;
    if (i < n) {
        if (i <= a) { tok = state->tokens[m - 1 + i]; qs = m - 1 + i; }     // JL:657-661, 688-695
        else { tok = params->fresh_tok[i - 1 - a]; qs = -1; }               // JL:505-514 (implicit one-hot row)
    }
    __syncthreads();
    if (i < max_rows) {
        if (i >= n) { tok = state->tokens[m - 1]; qs = -1; }                // padding rows: any valid id
        state->win_tok[i] = tok;
        state->q_src[i] = qs;
        for (int b = 0; b < n_batch; ++b) input_ids_out[(size_t)b * max_rows + i] = tok;
        if (positions_out) {
            const int64_t base = (int64_t)params->kv_len + i;
            for (int b = 0; b < n_batch; ++b) positions_out[(size_t)b * max_rows + i] = base + (pos_offset ? pos_offset[b] : 0);
        }
    }
}

// dynamic LDS (in floats) K2 / K4 ask for to stage a row's window: what `cols` columns need, capped at what a CU has left beside SjdShared
// (150 KiB: windows of up to 38400 columns -- Emu3's 32768 visual tokens fit, a whole text vocabulary does not and keeps the global staging)
#define SJD_DRAW_LIST 2304          // entries of the draw list K2 / K4 ask LDS room for behind a staged row (top-k 2000 / 2048 + ties)
static int sjd_stage_lds_floats(long cols)
{
    const long cap = 150 * 1024 / 4;
    return (int)(cols < cap ? ((cols + 3) & ~3L) : cap);
}

// ------------------------------------------------------------------------------------------------ C-ABI
extern "C" int sjd_reguess_ex(const sjd_iter_params *params, sjd_state *state, int64_t *input_ids_out, int n_batch, int max_rows,
                              const int64_t *pos_offset, int64_t *positions_out, void *stream)
{
    if (!params || !state || !input_ids_out || max_rows < 1 || max_rows > SJD_MAX_WINDOW || n_batch < 1) return SJD_ERR_BAD_ARG;
    hipLaunchKernelGGL(k5_reguess<false>, dim3(1), dim3(64), 0, (hipStream_t)stream, params, state, input_ids_out, n_batch, max_rows, pos_offset,
                       positions_out, sjd_slots{});
    return hipGetLastError() == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH;
}

extern "C" int sjd_reguess(const sjd_iter_params *params, sjd_state *state, int64_t *input_ids_out, int n_batch,
                           int max_rows, void *stream)
{
    return sjd_reguess_ex(params, state, input_ids_out, n_batch, max_rows, nullptr, nullptr, stream);
}

extern "C" int sjd_logits_to_probs_sample_ex(const float *logits_c, const float *logits_u, int64_t row_stride, float guidance,
                                             int max_rows, int V, const sjd_iter_params *params, const float *noise,
                                             float *probs_out, int64_t *tokens_out, int64_t *amax_out, void *stream);

extern "C" int sjd_logits_to_probs_sample(const float *logits_c, const float *logits_u, int64_t row_stride, float guidance,
                                          int max_rows, int V, const sjd_iter_params *params, const float *noise,
                                          float *probs_out, int64_t *tokens_out, void *stream)
{
    return sjd_logits_to_probs_sample_ex(logits_c, logits_u, row_stride, guidance, max_rows, V, params, noise, probs_out, tokens_out, nullptr, stream);
}

extern "C" int sjd_logits_to_probs_sample_ex(const float *logits_c, const float *logits_u, int64_t row_stride, float guidance,
                                             int max_rows, int V, const sjd_iter_params *params, const float *noise,
                                             float *probs_out, int64_t *tokens_out, int64_t *amax_out, void *stream)
{
    if (!logits_c || !params || !probs_out || !tokens_out || max_rows < 1 || max_rows > SJD_MAX_WINDOW || V < 1)
        return SJD_ERR_BAD_ARG;                    /* noise may be NULL when params->philox_blocks > 0 (the kernel generates it) */
    sjd_head_partials none = {};
    const int lds_floats = sjd_stage_lds_floats((long)V + 2 * SJD_DRAW_LIST);
    (void)hipFuncSetAttribute((const void *)k2_logits_to_probs_sample<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_floats * 4);
    hipLaunchKernelGGL(k2_logits_to_probs_sample<false>, dim3(max_rows), dim3(SJD_TPB), (size_t)lds_floats * 4, (hipStream_t)stream, logits_c, logits_u,
                       (long)row_stride, guidance, V, params, noise, probs_out, tokens_out, none, amax_out, lds_floats, sjd_slots{});
    return hipGetLastError() == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH;
}

extern "C" int sjd_logits_to_probs_sample_part(const sjd_head_partials *head, float guidance, int max_rows, int V,
                                               const sjd_iter_params *params, const float *noise, float *probs_out, int64_t *tokens_out,
                                               int64_t *amax_out, void *stream)
{
    if (!head || !head->part || head->n_chunks < 1 || head->n_cols < 1 || head->col0 < 0 || head->row_stride < head->n_cols) return SJD_ERR_BAD_ARG;
    if (!params || !probs_out || !tokens_out || max_rows < 1 || max_rows > SJD_MAX_WINDOW || V < 1) return SJD_ERR_BAD_ARG;
    if (head->row_sumsq && (head->slices < 1 || head->prows < 1)) return SJD_ERR_BAD_ARG;
    const int lds_floats = sjd_stage_lds_floats((long)head->n_cols + 8 + 2 * SJD_DRAW_LIST);     // (the rows' windows lie inside the head's column window; + the draw list)
    (void)hipFuncSetAttribute((const void *)k2_logits_to_probs_sample<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_floats * 4);
    hipLaunchKernelGGL(k2_logits_to_probs_sample<true>, dim3(max_rows), dim3(SJD_TPB), (size_t)lds_floats * 4, (hipStream_t)stream, (const float *)nullptr,
                       (const float *)nullptr, (long)0, guidance, V, params, noise, probs_out, tokens_out, *head, amax_out, lds_floats, sjd_slots{});
    return hipGetLastError() == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH;
}

extern "C" int sjd_head_combine(const sjd_head_partials *head, float guidance, int max_rows, int V, const sjd_iter_params *params, float *z_out,
                                void *stream)
{
    if (!head || !head->part || head->n_chunks < 1 || head->n_cols < 4 || (head->n_cols & 3) || head->col0 < 0 || head->row_stride < head->n_cols) return SJD_ERR_BAD_ARG;
    if ((head->row_stride & 3) || (head->chunk_stride & 3) || ((uintptr_t)head->part & 15) || ((uintptr_t)z_out & 15)) return SJD_ERR_BAD_ARG;
    if (!params || !z_out || max_rows < 1 || max_rows > SJD_MAX_WINDOW || V < 1) return SJD_ERR_BAD_ARG;
    if (head->row_sumsq && (head->slices < 1 || head->prows < 1)) return SJD_ERR_BAD_ARG;
    const dim3 grid((head->n_cols / 4 + 255) / 256, max_rows);
    hipLaunchKernelGGL(k2a_head_combine, grid, dim3(256), 0, (hipStream_t)stream, *head, guidance, V, params, z_out);
    return hipGetLastError() == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH;
}

extern "C" int sjd_verify_accept_ex(const sjd_iter_params *params, sjd_state *state, const float *probs, const float *prev_probs,
                                    const float *rs, const float *noise2, float *scratch, int max_rows, int V, sjd_state *host_mirror,
                                    void *stream)
{
    if (!params || !state || !probs || !prev_probs || !scratch || max_rows < 1 || max_rows > SJD_MAX_WINDOW || V < 1)
        return SJD_ERR_BAD_ARG;                    /* rs / noise2 may be NULL when params->philox_blocks > 0 */
    const int lds_floats = sjd_stage_lds_floats((long)V + 2 * SJD_DRAW_LIST);
    (void)hipFuncSetAttribute((const void *)k4_verify_accept<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_floats * 4);
    hipLaunchKernelGGL(k4_verify_accept<false>, dim3(1), dim3(SJD_TPB), (size_t)lds_floats * 4, (hipStream_t)stream, params, state, probs, prev_probs, rs,
                       noise2, scratch, V, host_mirror, lds_floats, sjd_slots{});
    return hipGetLastError() == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH;
}

extern "C" int sjd_verify_accept(const sjd_iter_params *params, sjd_state *state, const float *probs, const float *prev_probs,
                                 const float *rs, const float *noise2, float *scratch, int max_rows, int V, void *stream)
{
    return sjd_verify_accept_ex(params, state, probs, prev_probs, rs, noise2, scratch, max_rows, V, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------ every slot of a continuous batch per launch
static bool sjd_slots_ok(const sjd_slots *sl)
{
    return sl && sl->n_slots >= 1 && sl->n_slots <= 65535 && sl->params_stride >= (int64_t)sizeof(sjd_iter_params) && (sl->params_stride % 8) == 0 &&
           sl->state_stride >= (int64_t)sizeof(sjd_state) && (sl->state_stride % 8) == 0;
}

extern "C" int sjd_reguess_slots(const sjd_iter_params *params0, sjd_state *state0, int64_t *input_ids_out, int n_batch, int max_rows,
                                 const int64_t *pos_offset, int64_t *positions_out, const sjd_slots *slots, void *stream)
{
    if (!params0 || !state0 || !input_ids_out || max_rows < 1 || max_rows > SJD_MAX_WINDOW || n_batch < 1 || !sjd_slots_ok(slots)) return SJD_ERR_BAD_ARG;
    hipLaunchKernelGGL(k5_reguess<true>, dim3(slots->n_slots), dim3(64), 0, (hipStream_t)stream, params0, state0, input_ids_out, n_batch, max_rows,
                       pos_offset, positions_out, *slots);
    return hipGetLastError() == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH;
}

extern "C" int sjd_logits_to_probs_sample_part_slots(const sjd_head_partials *head, float guidance, int max_rows, int V, const sjd_iter_params *params0,
                                                     float *probs_out0, int64_t *tokens_out0, int64_t *amax_out0, const sjd_slots *slots, void *stream)
{
    if (!head || !head->part || head->n_chunks < 1 || head->n_cols < 1 || head->col0 < 0 || head->row_stride < head->n_cols) return SJD_ERR_BAD_ARG;
    if (!params0 || !probs_out0 || !tokens_out0 || max_rows < 1 || max_rows > SJD_MAX_WINDOW || V < 1) return SJD_ERR_BAD_ARG;
    if (head->row_sumsq && (head->slices < 1 || head->prows < 1)) return SJD_ERR_BAD_ARG;
    if (!sjd_slots_ok(slots) || slots->head_rows < max_rows || slots->probs_stride < (int64_t)max_rows * V) return SJD_ERR_BAD_ARG;
    if (head->zero_state && slots->zero_state_stride < 2 * (int64_t)max_rows) return SJD_ERR_BAD_ARG;
    if ((head->dbg_c || head->dbg_u) && slots->dbg_stride < (int64_t)max_rows * V) return SJD_ERR_BAD_ARG;
    const int lds_floats = sjd_stage_lds_floats((long)head->n_cols + 8 + 2 * SJD_DRAW_LIST);
    (void)hipFuncSetAttribute((const void *)k2_logits_to_probs_sample<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_floats * 4);
    hipLaunchKernelGGL((k2_logits_to_probs_sample<true, true>), dim3(max_rows, slots->n_slots), dim3(SJD_TPB), (size_t)lds_floats * 4, (hipStream_t)stream,
                       (const float *)nullptr, (const float *)nullptr, (long)0, guidance, V, params0, (const float *)nullptr, probs_out0, tokens_out0, *head,
                       amax_out0, lds_floats, *slots);
    return hipGetLastError() == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH;
}

extern "C" int sjd_verify_accept_slots(const sjd_iter_params *params0, sjd_state *state0, const float *probs0, const float *prev_probs0, float *scratch0,
                                       int max_rows, int V, sjd_state *host_mirror0, const sjd_slots *slots, void *stream)
{
    if (!params0 || !state0 || !probs0 || !prev_probs0 || !scratch0 || max_rows < 1 || max_rows > SJD_MAX_WINDOW || V < 1) return SJD_ERR_BAD_ARG;
    if (!sjd_slots_ok(slots) || slots->probs_stride < (int64_t)max_rows * V || slots->scratch_stride < V) return SJD_ERR_BAD_ARG;
    if (host_mirror0 && slots->mirror_stride < (int64_t)sizeof(sjd_state) + 8) return SJD_ERR_BAD_ARG;
    const int lds_floats = sjd_stage_lds_floats((long)V + 2 * SJD_DRAW_LIST);
    (void)hipFuncSetAttribute((const void *)k4_verify_accept<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_floats * 4);
    hipLaunchKernelGGL(k4_verify_accept<true>, dim3(1, slots->n_slots), dim3(SJD_TPB), (size_t)lds_floats * 4, (hipStream_t)stream, params0, state0, probs0,
                       prev_probs0, (const float *)nullptr, (const float *)nullptr, scratch0, V, host_mirror0, lds_floats, *slots);
    return hipGetLastError() == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------ noise, as a tensor (tests)
// What K2 / K4 generate in place, written out: element e of the tensor torch would have filled from a device generator at (seed, offset).
// kind 0: uniform_(0, 1) / torch.rand;  1: exponential_(1).  tests/test_gpu_philox.py compares it with torch bit for bit.
__global__ void philox_fill(float *__restrict__ out, uint64_t numel, uint64_t seed, uint64_t offset, uint32_t max_blocks, int kind)
{
    const uint32_t T = sjd_philox_threads(numel, max_blocks);
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < numel; e += (uint64_t)gridDim.x * blockDim.x)
        out[e] = kind == 0 ? sjd_philox_rand(seed, offset, T, e) : sjd_philox_exponential(seed, offset, T, e);
}

extern "C" int sjd_philox_fill(float *out, int64_t numel, uint64_t seed, uint64_t offset, int max_blocks, int kind, void *stream)
{
    if (!out || numel < 0 || max_blocks < 1 || kind < 0 || kind > 1) return SJD_ERR_BAD_ARG;
    if (numel == 0) return SJD_OK;
    hipLaunchKernelGGL(philox_fill, dim3(1024), dim3(256), 0, (hipStream_t)stream, out, (uint64_t)numel, seed, offset, (uint32_t)max_blocks, kind);
    return hipGetLastError() == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH;
}

extern "C" uint64_t sjd_philox_offset_increment(int64_t numel, int max_blocks)
{
    return (numel <= 0 || max_blocks < 1) ? 0 : sjd_philox_offset_step((uint64_t)numel, (uint32_t)max_blocks);
}
