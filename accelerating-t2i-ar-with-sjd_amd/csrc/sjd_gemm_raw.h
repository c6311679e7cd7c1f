// sjd_gemm_raw.h -- the per-unit escape of the 12-bit weight stream (round 6; VERDICT r5 "next #3").  Included by sjd_gemm.hip.
//
// sjd_amd.ops.pack_weight_z used to DECLINE a whole matrix when ONE (k-chunk, 32-column tile) unit needed more exceptions than a unit header
// holds (127): a checkpoint with zero rows, pruned blocks or a unit whose weights span more than sixteen binades then streamed that matrix
// uncompressed (it happened to Emu3's output head over one half-padded tile, round 5).  Now such a unit is marked RAW: its slot in the 12-bit
// stream is zero-filled and its weights travel verbatim (sjd_amd.ops.pack_weight's 1-KiB records, appended to the header array).
// WHERE the verbatim records are multiplied:
//   * g1z_skinny_gemm, g1z_gateup_silu and g1z_gateup_silu_tall (all but SP = 32 x two row tiles) run them IN the kernel: the wave that owns a
//     raw unit takes a plain loop over its records (csrc/sjd_gemm.hip: g1z_raw_records) -- 1 % raw units cost 0-3 % of a launch;
//   * the sub-tiled kernel (65..128 rows, or a 64-row window whose chunk does not fit LDS) and the one G1sz instantiation that sits at its
//     register limit run over the zeros and a FIX-UP launch behind them (this file) recomputes exactly the tiles the raw units feed -- the
//     same MFMA sequence per (tile, chunk, row tile) as g1_skinny_gemm.  (The fix-up was built first, for every kernel: +5-7 us per projection,
//     a unit's MFMAs are one dependent chain -- hence the in-kernel path for the 32 / 64-row kernels that carry the headline configurations.)
// Either way the planes / activations are bit-identical to the uncompressed kernels' (tests/test_gpu_glue.py::test_g1z_raw_units_*,
// test_g1sz_raw_pairs_*), and a matrix without raw units never leaves the old path.
//   g1_raw_units : planes of sjd_skinny_gemm_z.  out [n_chunks, prows, N] (the launch's column window starts at tile0)
//   g1_raw_gateup: activations of sjd_gateup_silu_z.  A raw unit anywhere in (gate tile t | up tile t) x (K half 0 | 1) makes the packer list
//                  the PAIR t: all four units travel verbatim and the wave redoes both accumulations and the SiLU epilogue of the pair.
#pragma once

template <int DT>
__global__ __launch_bounds__(64) void g1_raw_units(const unsigned short *__restrict__ x, const u32x4 *__restrict__ raw, const int *__restrict__ index,
                                                   float *__restrict__ out, int M, int N, int K, int KC, int tile0, int prows)
{
    const int unit = blockIdx.x, mt = blockIdx.y;
    const int chunk = index[2 * unit], t_out = index[2 * unit + 1] - tile0;
    if (t_out < 0 || t_out >= N / 32) return;                    // (a column window that does not hold this tile)
    const int k0 = chunk * KC, steps = min(KC, K - k0) / 16;
    const int lane = threadIdx.x, m = 32 * mt + (lane & 31), h = lane >> 5;
    const u32x4 *rec = raw + (size_t)unit * (KC / 16) * 64 + lane;
    const unsigned short *xr = x + (size_t)min(m, M - 1) * K + k0 + 8 * h;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    u32x4 a[8], b[8], an[8], bn[8];                               // sixteen k-steps in flight (see g1_raw_gateup)
    auto load = [&](u32x4 (&av)[8], u32x4 (&bv)[8], int s) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av[u] = *reinterpret_cast<const u32x4 *>(xr + 16 * min(s + u, steps - 1));
            bv[u] = __builtin_nontemporal_load(rec + (size_t)min(s + u, steps - 1) * 64);
        }
    };
    load(a, b, 0);
    for (int s = 0; s < steps; s += 8) {
        load(an, bn, s + 8);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s + u < steps) acc = G1Mfma<DT>::mma(m < M ? a[u] : u32x4{0u, 0u, 0u, 0u}, b[u], acc);
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[u] = an[u]; b[u] = bn[u]; }
    }
    float *o = out + ((size_t)chunk * prows + 32 * mt) * N + (size_t)t_out * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * N] = acc[r];
}

// raw: per listed pair [K half][gate | up][K / 32 records].  Four waves per (pair, row tile): wave w = 2 kh + gu runs the ONE accumulation chain of
// its unit (a unit's MFMAs are a dependent chain: four chains in one wave took 4 x as long), the four tiles meet in LDS for the epilogue.
template <int DT>
__global__ __launch_bounds__(256) void g1_raw_gateup(const unsigned short *__restrict__ x, const u32x4 *__restrict__ raw, const int *__restrict__ tiles,
                                                     unsigned short *__restrict__ y, int M, int I, int K, int prows,
                                                     const float *__restrict__ row_sumsq, int rs_slices, float rs_inv_hidden, float rs_eps)
{
    __shared__ float red[4][32][33];
    const int pair = blockIdx.x, mt = blockIdx.y, t = tiles[pair];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), kh = w >> 1;
    const int m = 32 * mt + (lane & 31), h = lane >> 5;
    const int steps = K / 32;                                     // k-steps of a K half
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    {
        const unsigned short *xr = x + (size_t)min(m, M - 1) * K + kh * (K / 2) + 8 * h;
        const u32x4 *rec = raw + ((size_t)pair * 4 + w) * steps * 64 + lane;
        // sixteen k-steps in flight: the group after the current one is requested before the current one's MFMAs (a group's loads are a cold
        // round trip of ~1.4 us: un-pipelined, the 128 k-steps of hidden 4096 took sixteen of them)
        u32x4 a[8], b[8], an[8], bn[8];
        auto load = [&](u32x4 (&av)[8], u32x4 (&bv)[8], int s) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                av[u] = *reinterpret_cast<const u32x4 *>(xr + 16 * min(s + u, steps - 1));
                bv[u] = __builtin_nontemporal_load(rec + (size_t)min(s + u, steps - 1) * 64);
            }
        };
        load(a, b, 0);
        for (int s = 0; s < steps; s += 8) {
            load(an, bn, s + 8);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s + u < steps) acc = G1Mfma<DT>::mma(m < M ? a[u] : u32x4{0u, 0u, 0u, 0u}, b[u], acc);
#pragma unroll
            for (int u = 0; u < 8; ++u) { a[u] = an[u]; b[u] = bn[u]; }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = acc[r];
    __syncthreads();
    // the epilogue of g1z_gateup_silu, element for element: the two K halves summed in order from zero, the folded RMSNorm's row scale from the
    // per-slice sums of squares in slice order (at most eight slices, as there), sjd_silu_mul_elem.  Thread -> (row = tid / 8, four columns)
    {
        const int rl = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4, row = 32 * mt + rl;
        float rr = 1.0f;
        if (row_sumsq) {
            float tsum = 0.f;
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) tsum += (qq < rs_slices) ? row_sumsq[(size_t)min(qq, rs_slices - 1) * prows + row] : 0.f;
            rr = rsqrtf(__builtin_fmaf(tsum, rs_inv_hidden, rs_eps));
        }
        unsigned short o16[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gsum = 0.f, usum = 0.f;
            gsum += red[0][rl][c4 + j]; gsum += red[2][rl][c4 + j];
            usum += red[1][rl][c4 + j]; usum += red[3][rl][c4 + j];
            o16[j] = sjd_silu_mul_elem<DT>(gsum, usum, rr);
        }
        if (row < M) {
            uint2 pk{(unsigned)o16[0] | ((unsigned)o16[1] << 16), (unsigned)o16[2] | ((unsigned)o16[3] << 16)};
            *reinterpret_cast<uint2 *>(y + (size_t)row * I + 32 * t + c4) = pk;
        }
    }
}

// index: int32 [n_raw, 2] = (k chunk, tile of the PACKED weight); raw: n_raw x (KC / 16) records of 1 KiB (a short last chunk is padded).
// out: the planes sjd_skinny_gemm_z(x, ..., N, K, KC, ..., N_packed, tile0) has just written on the same stream.
extern "C" int sjd_raw_units_fixup(const void *x, const void *raw, const int32_t *index, int n_raw, float *out, int M, int N, int K, int KC,
                                   int tile0, int dtype, void *stream)
{
    if (n_raw == 0) return SJD_OK;
    if (!x || !raw || !index || !out || n_raw < 0 || M < 1 || M > 256 || N < 32 || (N % 32) || (K % 16) || KC < 16 || (KC % 16) || tile0 < 0) return SJD_ERR_BAD_ARG;
    if (dtype != SJD_DTYPE_BF16) return SJD_ERR_UNSUPPORTED;
    const int mt = (M + 31) / 32;
    hipLaunchKernelGGL((g1_raw_units<SJD_DTYPE_BF16>), dim3(n_raw, mt), dim3(64), 0, (hipStream_t)stream, (const unsigned short *)x, (const u32x4 *)raw,
                       (const int *)index, out, M, N, K, KC, tile0, 32 * mt);
    return hipGetLastError() == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH;
}

// tiles: int32 [n_pairs] gate tiles t (columns [32 t, 32 t + 32) of y); raw: per pair [K half][gate | up][K / 32 records].  y: what
// sjd_gateup_silu_z(x, ..., M, I, K, ...) has just written on the same stream.
extern "C" int sjd_raw_gateup_fixup(const void *x, const void *raw, const int32_t *tiles, int n_pairs, void *y, int M, int I, int K, int dtype,
                                    const sjd_row_norm *row_norm, void *stream)
{
    if (n_pairs == 0) return SJD_OK;
    if (!x || !raw || !tiles || !y || n_pairs < 0 || M < 1 || M > 64 || I < 64 || (I % 64) || K < 64 || (K % 64)) return SJD_ERR_BAD_ARG;
    if (row_norm && (!row_norm->sumsq || row_norm->slices < 1 || row_norm->hidden < 1)) return SJD_ERR_BAD_ARG;
    if (dtype != SJD_DTYPE_BF16 || (row_norm && row_norm->slices > 8)) return SJD_ERR_UNSUPPORTED;
    const int mt = (M + 31) / 32;
    hipLaunchKernelGGL((g1_raw_gateup<SJD_DTYPE_BF16>), dim3(n_pairs, mt), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)x, (const u32x4 *)raw,
                       (const int *)tiles, (unsigned short *)y, M, I, K, 32 * mt, row_norm ? row_norm->sumsq : nullptr, row_norm ? row_norm->slices : 0,
                       row_norm ? 1.0f / (float)row_norm->hidden : 0.f, row_norm ? row_norm->eps : 0.f);
    return hipGetLastError() == hipSuccess ? SJD_OK : SJD_ERR_LAUNCH;
}
