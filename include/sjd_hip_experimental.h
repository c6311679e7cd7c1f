/*
 * sjd_hip_experimental.h -- entry points of libsjd_hip_exp.so that are NOT part of the product library (round 6; VERDICT r5 "next #6").
 *
 * libsjd_hip_exp.so is built from the same sources as libsjd_hip.so with -DSJD_EXPERIMENTAL (csrc/Makefile): it exports everything
 * include/sjd_hip.h declares PLUS the entry points below -- structures built in rounds 2-5, measured against the product path and not taken
 * (DESIGN.md sections 10, 10b; docs/LABBOOK_r1-r3.md), kept with their tests as the record of what was tried, and the tuning entry of kernel
 * G1w.  The product library is compiled WITHOUT their kernels and without their run-time switches inside the hot kernels (the reducing tail of
 * g1_skinny_gemm, the in-kernel split merge of k1_partial): engine.py / engine_batch.py / backbones.py reach none of them with default
 * switches; the opt-in switches of backbones.py (SJD_K1_FUSED, SJD_REDUCE_FUSED, SJD_MLP_PAIR, SJD_PREFETCH, ...) load this library.
 * Conventions as in sjd_hip.h.
 */
#ifndef SJD_HIP_EXPERIMENTAL_H
#define SJD_HIP_EXPERIMENTAL_H

#include "sjd_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* K1F -- F2 + K1 + the split combine in ONE launch, for the multi-head-attention draft window (H == H_kv, n_rows <= 16,
 * B * n_rows <= 32, D == 128, 16-bit cache): per (batch, head) one 512-thread workgroup sums the fp32 split-K partials `part`
 * [n_chunks, 32, 3 * H * D] of the q|k|v projection (G1), applies the folded-RMSNorm row scale (`row_norm`, may be NULL), the per-head
 * LayerNorm (qn_w/qn_b, kn_w/kn_b: NULL = none) and RoPE with the arithmetic of sjd_qknorm_rope_append, appends the k / v rows at cache
 * rows [kv_len, kv_len + n_rows), attends over [key_start[b], kv_len + i] and writes out [B, n_rows, H, D].  No workspace.
 * replaces, like F2 + K1: ChameleonLayerNorm, apply_rotary_pos_emb, DynamicCache.update, _update_causal_mask and
 * scaled_dot_product_attention (reference modeling_chameleon.py:198-219, 144-178, 547, 549-576; jacobi_iteration_lumina_mgpt.py:1256-1336). */
typedef struct sjd_row_norm sjd_row_norm;
int sjd_qkv_attention_fused(const float *part, int n_chunks, void *k_cache, void *v_cache, void *out, const void *qn_w, const void *qn_b,
                            const void *kn_w, const void *kn_b, const float *inv_freq, const int64_t *positions, int B, int n_rows, int H,
                            int D, int S_max, int dtype, const sjd_row_norm *row_norm, const int32_t *key_start,
                            const sjd_iter_params *params, int kv_len, void *stream);
/* K1Fs (round 3): the same with the key tiles of a (batch, head) split over n_split workgroups exactly as sjd_draft_window_attention splits
 * them (the effective count from the device-side kv_len), followed by the split combine: F2 (sjd_qknorm_rope_append) and the attention
 * partial pass in ONE launch that fills the chip.  Every workgroup derives q for itself; the workgroups whose tiles reach into the window's
 * own rows derive and append those K / V rows.  workspace: sjd_attention_workspace_bytes(B, H, n_rows, D, n_split) bytes.  n_split <= 1 is
 * sjd_qkv_attention_fused.  replaces, like it: q_norm / k_norm, apply_rotary_pos_emb, past_key_value.update and the attention of
 * ChameleonAttention.forward (reference modeling_chameleon.py:198-219, 144-178, 547, 499-581) for the window forward. */
int sjd_qkv_attention_fused_split(const float *part, int n_chunks, void *k_cache, void *v_cache, void *out, const void *qn_w, const void *qn_b,
                                  const void *kn_w, const void *kn_b, const float *inv_freq, const int64_t *positions, int B, int n_rows, int H,
                                  int D, int S_max, int dtype, const sjd_row_norm *row_norm, const int32_t *key_start,
                                  const sjd_iter_params *params, int kv_len, int n_split, void *workspace, void *stream);

/* Round 5 -- G1z in the LOADER / CONSUMER form (csrc/sjd_gemm_engine.h; VERDICT r4 next #1, stage A): one persistent 256-thread workgroup per CU,
 * wave 0 streams the 12-bit records HBM -> LDS by LDS-DMA into per-consumer rings, waves 1..3 decode them from LDS and run the MFMA sequence of
 * sjd_skinny_gemm_z -- the same fp32 planes [n_chunks, 32, N], bit for bit.  M <= 32, bf16, tile-major packing (step_major = 0), KC <= 1024;
 * n_wg = workgroups (the CU count), a multiple of the K-chunk count and <= N / 32 per chunk.  sjd_engine_timeouts: bounded LDS polls that gave
 * up since the library was loaded (0).  replaces, like G1: the nn.Linear calls of the window forward (reference modeling_chameleon.py:527-529,
 * 579, 637-643). */
int sjd_skinny_gemm_engine_z(const void *x, const void *wz, const void *exc, int exc_cap, float *out, int M, int N, int K, int KC,
                             int dtype, int N_packed, int tile0, int n_wg, void *stream);
int sjd_engine_timeouts(void);

/* Weight prefetch for G1: reads `nbytes` of packed weights with plain loads and discards them, so that the lines sit in the 256 MiB
 * Infinity Cache when the next sjd_skinny_gemm streams them.  Meant for a SIDE stream / parallel hipGraph branch while the
 * latency-bound kernels of the layer (F1r, F2, K1, F3) leave HBM idle.  No reference counterpart (the reference's nn.Linear calls,
 * modeling_chameleon.py:527-529, are library GEMMs); no effect on results.  sink: any 4-byte device scratch (never written in
 * practice).  blocks: workgroups of 256 threads (1..4096). */
int sjd_weight_prefetch(const void *w, int64_t nbytes, int blocks, void *sink, void *stream);

/* Round 5 -- the HEAD of the next projection's weight stream pulled into the XCDs' L2 by spare workgroups of the latency-bound launch in front
 * of it (csrc/sjd_l2_prefetch.h).  Lines read with default-policy loads stay in the L2 of the XCD that read them across a kernel boundary
 * (tools/l2_survive_probe.hip); workgroup L of a launch runs on XCD L mod 8; so pulling workgroup j (XCD j mod 8) reads the first `head_pairs`
 * record pairs of every unit the consumer's workgroups j mod 8, j mod 8 + 8, ... will stream.  sjd_l2_head describes the consumer launch; the
 * two constructors below derive it from the arguments that launch will be given, so the geometry lives next to the launchers.  No reference
 * counterpart (the reference's nn.Linear calls, modeling_chameleon.py:527-529, 579, 637-643, are library GEMMs); no effect on results. */
typedef struct sjd_l2_head {
    const void *wz;                 /* the 12-bit packed records (ops.pack_weight_z) */
    int32_t kind;                   /* 0: a sjd_skinny_gemm_z launch; 1: a sjd_gateup_silu_z launch */
    int32_t gx, gy, waves;          /* its grid and waves per workgroup */
    int32_t n_tiles, tile0, n_out;  /* packed column tiles, first tile and tile count of the launch (kind 0) */
    int32_t pairs_full, pairs_last; /* record pairs of a full K chunk's unit / of the last chunk's */
    int32_t step_major;
    int32_t head_pairs;             /* pairs per unit to pull */
} sjd_l2_head;
int sjd_l2_head_gemm_z(sjd_l2_head *out, const void *wz, int M, int N, int K, int KC, int waves, int step_major, int N_packed, int tile0, int head_pairs);
int sjd_l2_head_gateup_z(sjd_l2_head *out, const void *wz, int M, int I, int K, int step_major, int head_pairs);
int64_t sjd_l2_head_bytes(const sjd_l2_head *head);                 /* bytes the pull reads (the ragged edges counted exactly) */
/* the pull as a launch of its own (tools/l2_head_bench.py; the product hosts it in F1r / F2: sjd_residual_sumsq_pf, sjd_qknorm_rope_append_pf).
 * blocks: workgroups of 256 threads, a multiple of 8 */
int sjd_weight_prefetch_head(const sjd_l2_head *head, int blocks, void *stream);
/* sjd_residual_sumsq (F1r) hosting the pull of the projection that follows it: F1r's own workgroups first, `pf_blocks` pulling workgroups of
 * 128 threads behind them in the same launch (rounded up to whole grid rows).  head NULL / pf_blocks 0 = sjd_residual_sumsq.  Same results. */
int sjd_residual_sumsq_pf(void *h, const float *part, int n_chunks, int rows, int hidden, int dtype, float *out_sumsq,
                          const sjd_l2_head *head, int pf_blocks, void *stream);
/* XCC_ID of every workgroup of a (gx, gy) launch of 64-thread workgroups -> out[gx * gy] int32 (device): the dispatch rule the pull relies on */
int sjd_debug_xcc_map(int32_t *out, int gx, int gy, void *stream);

/* K1 in ONE launch (round 3): the key splits of a (batch, kv head, 16-row chunk) are merged by the last of their workgroups to finish
 * -- device-coherent exchange of the (m, l, O) partials, k1_combine's arithmetic in split order, same output bits -- instead of by a
 * second kernel.  tickets: B * H_kv * ceil(n_rows / 16) zero-initialised uint32, private to launches that cannot overlap (they re-arm
 * themselves).  Shapes served by the shared-tile kernel (grouped-query heads and / or two row chunks: Emu3) keep the two-kernel form.
 * Same arguments and call sites as sjd_draft_window_attention(_fp8) otherwise (modeling_chameleon.py:499-581). */
int sjd_draft_window_attention_merged(const void *q, const void *k_cache, const void *v_cache, void *out, int B, int n_rows, int H, int H_kv,
                                      int D, int S_max, int dtype, const int32_t *key_start, const sjd_iter_params *params, int kv_len,
                                      int n_split, void *workspace, uint32_t *tickets, void *stream, void *ev_start, void *ev_stop);
int sjd_draft_window_attention_fp8_merged(const void *q, const void *k_cache, const void *v_cache, void *out, int B, int n_rows, int H, int H_kv,
                                          int D, int S_max, int dtype, float k_scale, float v_scale, const int32_t *key_start,
                                          const sjd_iter_params *params, int kv_len, int n_split, void *workspace, uint32_t *tickets, void *stream);

/* The MLP of a window forward as ONE launch (round 4 experiment: a run-ahead weight loader across a dependency edge).  Replaces
 * sjd_gateup_silu_z followed by sjd_skinny_gemm_z (modeling_chameleon.py:637-643: down(silu(gate(x)) * up(x))): y [M <= 32, I] and the
 * split-K planes out [ceil(I / KC_dn), 32, hidden] of the down projection are what the two launches write, bit for bit.  Workgroup b < I / 64
 * computes gate / up tiles 2 b, 2 b + 1 and publishes its slice of y write-through; every workgroup is then one (column group, K chunk) unit
 * of the down projection: it requests its header and first weight records BEFORE it waits for the arrival counter of its K chunk, so the
 * down projection's weight stream starts under the tail of gate|up instead of behind a kernel boundary and a cold start.
 * ready: ceil(I / KC_dn) + 1 zero-initialised uint32 private to launches that cannot overlap (the last workgroup re-arms them: replayable
 * from a hipGraph).  bf16, hidden = 4096, KC_dn a multiple of 64 and <= 2560, eight column tiles per down workgroup; every workgroup of the
 * launch must be resident at once: grids above resident_limit (CUs of the device) are refused.  sjd_mlp_pair_timeouts: abandoned waits (0). */
int sjd_mlp_pair_z(const void *x, const void *wz_gu, const void *exc_gu, int exc_cap_gu, int step_major_gu, void *y, const void *wz_dn,
                   const void *exc_dn, int exc_cap_dn, int step_major_dn, float *out, int M, int I, int hidden, int KC_dn,
                   const sjd_row_norm *row_norm, uint32_t *ready, int resident_limit, void *stream);
int sjd_mlp_pair_timeouts(void);

/* G1 with stage F1r as its tail (round 3): h [M, N] += dtype(x @ W^T) in place and sumsq [N / 512, 32] = the per-slice sums of h^2, i.e.
 * sjd_skinny_gemm followed by sjd_residual_sumsq (the residual add + RMSNorm statistics of modeling_chameleon.py:59-73, 637, 643), bit for
 * bit, in one launch: the workgroups of a 512-column slice exchange their split-K planes device-coherently and reduce them in the
 * producer's tail (csrc/sjd_gemm.hip::g1_reduce_tail).  workspace: fp32 [n_chunks, 32, N]; ticket: N / 512 * 32 zero-initialised
 * uint32, private to launches that cannot overlap.  resident_limit: workgroups the device holds at once (the wait inside needs the whole
 * launch resident; larger launches are refused).  sjd_reduce_timeouts: waits that were abandoned since the library was loaded (0). */
int sjd_skinny_gemm_reduce(const void *x, const void *w_packed, float *workspace, void *h, float *sumsq, unsigned *ticket, int M, int N,
                           int K, int KC, int waves, int step_major, int dtype, int resident_limit, void *stream);
int sjd_reduce_timeouts(void);

/* G1w tuning entry (tools/g1w_bench.py): bf16; M <= 128 runs four row tiles, M <= 256 eight; tiles = column tiles per workgroup (2, 3, 4: one
 * per wave; 6, 8: two per wave); variant = (stage k-steps, ring slots, weight ring stages): 0 (4, 3, 2) = the product's (sjd_skinny_gemm with
 * more than 128 rows), 1 (4, 4, 2), 10: eight waves with one tile each, 20: four row tiles in the register budget of two workgroups per CU.
 * ldx: row stride of x in elements (0: K).  replaces, like G1: the nn.Linear calls of the decoder layer (reference
 * modeling_chameleon.py:527-529, 579, 193-195). */
int sjd_skinny_gemm_wide(const void *x, const void *w_packed, float *out, int M, int N, int K, int KC, int tiles, int step_major, int variant,
                         int ldx, void *stream);

/* Kernel G1w over the 12-bit weight stream (late round 6 experiment; csrc/sjd_gemm_wide.h, template parameter Z): sjd_skinny_gemm_z's planes, bit for bit,
 * for windows of 33..256 rows with `tiles` = 2, 3, 4, 6 or 8 column tiles per workgroup.  Measured slower than the product's choices at every row count but
 * for the o projection at 64 rows.  Raw units: run sjd_raw_units_fixup behind it.  replaces, like G1z: the nn.Linear calls of the decoder
 * layer (reference modeling_chameleon.py:527-529, 579, 193-195). */
int sjd_skinny_gemm_z_wide(const void *x, const void *wz, const void *exc, int exc_cap, float *out, int M, int N, int K, int KC, int tiles, int step_major,
                           int N_packed, int tile0, void *stream);

/* round-6 gate probe for "the o projection consumes K1's split partials" (VERDICT r5 next #4): the would-be staging prologue alone, on the o
 * projection's grid (csrc/sjd_gemm.hip::o_merge_prologue_probe).  part: fp32 [32 heads][n_split][rows][130]; mode 0 = empty body, 1 = merge. */
int sjd_o_merge_prologue_probe(const float *part, float *sink, int n_split, int rows, int mode, void *stream);

#ifdef __cplusplus
}
#endif
#endif
