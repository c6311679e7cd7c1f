// sjd_k1_split.h -- what the PRODUCER of K1's key-split partials (k1_partial*, sjd_attention.hip) and every CONSUMER that merges them
// (k1_combine and the in-kernel merge there; the staging of the output projection, g1_skinny_gemm_attn in sjd_gemm.hip) must agree on:
// the geometry of the splits and the arithmetic of one merge step.  One definition, so the consumers give the same bits.
//   workspace: ws_o  [B, H, n_chunks, n_split, 16 rows, D] fp32 (un-normalised O), then
//              ws_ml [B, H, n_chunks, n_split, 16 rows, 2] fp32 (running max m, sum l)   (sjd_attention_workspace_bytes)
#pragma once
#include <hip/hip_runtime.h>

#define K1_KT 32          // keys per wave tile
#define K1_ROWS 16        // query rows per chunk
#define K1_MIN_TILES_PER_SPLIT 4   // a key split is only opened when it gets at least one tile per wave

// Key-tile range [t_lo, t_hi) of a (batch row, chunk) and the number of splits actually used for it.  The launch grid
// is sized for n_split (static, hipGraph friendly); splits >= the effective count exit immediately and are skipped by
// the consumer, so short contexts do not pay for empty partials.
__device__ __forceinline__ void k1_tile_range(int kstart, int total, int n_split, int &t_lo, int &t_hi, int &eff_split, int &tps)
{
    t_lo = kstart / K1_KT;
    t_hi = (total + K1_KT - 1) / K1_KT;
    const int nt = max(t_hi - t_lo, 0);
    eff_split = min(n_split, max(1, (nt + K1_MIN_TILES_PER_SPLIT - 1) / K1_MIN_TILES_PER_SPLIT));
    tps = (nt + eff_split - 1) / eff_split;
}

// one step of the online merge of split partials (m, l, O[N]) into (M, L, acc[N]) -- written with explicit fma so that every consumer gives
// the same bits whatever the instruction selector makes of the code around it
template <int N>
__device__ __forceinline__ void k1_merge_step(float &M, float &L, float (&acc)[N], float m, float l, const float (&o)[N])
{
    const float Mn = fmaxf(M, m);
    const float Msafe = (Mn == -INFINITY) ? 0.0f : Mn;
    const float w0 = __expf(M - Msafe), w1 = __expf(m - Msafe);
    L = __builtin_fmaf(L, w0, l * w1);
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = __builtin_fmaf(acc[j], w0, o[j] * w1);
    M = Mn;
}
