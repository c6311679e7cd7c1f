/*
 * sjd_hip.h -- C-ABI of the MI355X-native Speculative Jacobi Decoding hot path (libsjd_hip.so).
 *
 * Drop-in boundary.  The reference (tyshiwo1/Accelerating-T2I-AR-with-SJD) has no FFI: its hot path is
 * Python on top of ATen ops.  Each entry point below replaces the reference Python function named next to
 * it; a maintainer binds them with ctypes and calls them from the reference's own
 * `_sample` hook.  Conventions:
 *   - every pointer is a DEVICE pointer unless the name starts with h_; no ownership is taken, nothing
 *     is allocated, the call is asynchronous on `stream` (a hipStream_t passed as void*);
 *   - return 0 on success, a negative SJD_ERR_* otherwise; never throws; no global mutable state;
 *   - per-iteration dynamic scalars live in a device-resident `sjd_iter_params` blob that the host fills
 *     and uploads once per iteration, so that the launch sequence is shape-static (hipGraph friendly).
 */
#ifndef SJD_HIP_H
#define SJD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SJD_VERSION 100
#define SJD_MAX_WINDOW 64      /* max draft-window length L (reference max_num_new_tokens: 16 / 32 by default, a free CLI argument of eval_model.py:76;
                                  64 = one wavefront of accept tests in K4, 128 forward rows with CFG) */
#define SJD_MAX_RANGES 4

#define SJD_OK 0
#define SJD_ERR_BAD_ARG (-1)
#define SJD_ERR_UNSUPPORTED (-2)
#define SJD_ERR_LAUNCH (-3)

#define SJD_DTYPE_BF16 0
#define SJD_DTYPE_F16 1
#define SJD_DTYPE_F32 2        /* K1/K3 only: exact-fp32 VALU variant for small parity runs (not a performance path) */

/* One row of the "3-dim" logits processors, reduced to what the kernels need.  Built on the host from
 * integer grammar state; replaces MultiTokensVLLogitsProcessor / MultiTokensInterleavedTopKLogitsWarper /
 * TopPLogitsWarper3d / EOLLogitProcessor3d / the Anole 3d processors
 * (reference scheduler/logit_processor_3dim.py:45-204, 207-419; scheduler/jacobi_iteration_emu3.py:44-128). */
typedef struct sjd_row_rule {
    int32_t n_ranges;               /* 0: every column allowed */
    int32_t lo[SJD_MAX_RANGES];     /* allowed columns = union of [lo,hi) */
    int32_t hi[SJD_MAX_RANGES];
    int32_t forced;                 /* >=0: p = one-hot(forced) (forced EOL / end-of-image rows); -1: none */
    int32_t top_k;                  /* <=0 or >=V: off */
    float   top_p_thr;              /* float32(1 - top_p); <0: off */
    float   temperature;            /* HF TemperatureLogitsWarper (scores / temperature, applied after the processors and the grammar's
                                     * own top-k, before top-p and the softmax -- where transformers' generate() puts it, third-party
                                     * 4.47.1 `_get_logits_processor`); 1 (or <= 0): off.  The residual call of the verify step sees the
                                     * same warper: softmax(log(max(p - q, 0)) / temperature) (JL:203-241) */
} sjd_row_rule;

/* Host-built, device-resident control blob for ONE SJD iteration. */
typedef struct sjd_iter_params {
    int32_t n_rows;                 /* window length n of this iteration (1..SJD_MAX_WINDOW) */
    int32_t kv_len;                 /* cache rows already valid before this window */
    int32_t use_cfg;                /* 1: z = g*(c-u)+u ; 0: z = c  (check_is_force_no_cfg, JL:70-80) */
    int32_t scheme;                 /* 0: speculative_jacobi ; 1: jacobi */
    int32_t n_fresh;                /* trailing window rows filled with fresh random ids */
    int32_t batch_rows;             /* several prompts per launch: 0 = this blob governs every batch row; > 0 = `params` points at a
                                       contiguous ARRAY of blobs and blob i governs batch rows [i*batch_rows, (i+1)*batch_rows) of K1 /
                                       K3 / F2 (each prompt has its own kv_len / n_rows); same value in every blob of the array */
    int32_t iter_seq;                             /* host's iteration counter: sjd_verify_accept_ex publishes it behind the mirrored state */
    int32_t philox_blocks;                        /* 0: K2 / K4 read the noise tensors they are handed.  > 0: they GENERATE the noise: the
                                                     elements torch's exponential_ / uniform_ would have written for a device generator
                                                     with the seed / offsets below (Philox4x32-10, csrc/sjd_philox.h); the value is the
                                                     grid cap of ATen's launch, multiProcessorCount * (maxThreadsPerMultiProcessor / 256) */
    uint64_t philox_seed;                         /* torch.Generator.initial_seed() */
    uint64_t philox_offset[3];                    /* the generator's offset before (0) the [n_rows, V] exponential_ of the multinomial
                                                     (JL:118), (1) the [1, n_rows, V] rand of the accept test (JL:260), (2) the [1, V]
                                                     exponential_ of the residual multinomial (JL:237) */
    int64_t fresh_tok[SJD_MAX_WINDOW];            /* random re-guess ids (host global RNG, JL:505-509), packed */
    sjd_row_rule rules[SJD_MAX_WINDOW];           /* rules of the sampling call, row j */
    sjd_row_rule resid_rules[SJD_MAX_WINDOW];     /* rule of the residual call if rejection happens at i=j+1 */
} sjd_iter_params;

#if defined(__HIPCC__)
/* blob that governs batch row b (see batch_rows) */
static inline __attribute__((device, always_inline)) const sjd_iter_params *sjdi_params_of(const sjd_iter_params *p, int b)
{
    return (p && p->batch_rows > 0) ? p + b / p->batch_rows : p;
}
/* kv_len and n_rows of batch row b in ONE scalar round trip when the launch carries one blob (or b belongs to the first of an array): the
 * three words are requested together instead of batch_rows first and the field behind it -- a kernel of the window forward opens with this
 * load, and there are ~130 such kernels per iteration.  A second, dependent load only for the later blobs of an array. */
static inline __attribute__((device, always_inline)) unsigned long long sjdi_uniform_u64(unsigned long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffull));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
static inline __attribute__((device, always_inline)) void sjdi_kv_rows(const sjd_iter_params *p, int b, int *kv_len, int *n_rows)
{
    /* {n_rows, kv_len} and batch_rows of the first blob, requested TOGETHER.  Written as instructions because the optimiser otherwise
     * re-creates the dependent form (it folds the two cases into one load from `p + blob`, behind the load of batch_rows): three scalar
     * round trips in a row opened k1_partial before this (kernel arguments, batch_rows, kv_len). */
    unsigned long long nk;
    int br;
    __asm__ volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dword %1, %2, 0x14\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(nk), "=&s"(br) : "s"(sjdi_uniform_u64((unsigned long long)p)) : "memory");
    const int n0 = (int)(unsigned)(nk & 0xffffffffull), kv0 = (int)(unsigned)(nk >> 32);
    const int blob = br > 0 ? b / br : 0;
    if (blob == 0) { *kv_len = kv0; *n_rows = n0; }
    else { *kv_len = p[blob].kv_len; *n_rows = p[blob].n_rows; }
}
#endif

/* Device-resident decode state carried between iterations (written by sjd_verify_accept, read by sjd_reguess). */
typedef struct sjd_state {
    int32_t m;                      /* tokens emitted by the last iteration (first_misaligned, 1..n) */
    int32_t rejected;               /* 1: a residual resample happened (the g-stream consumed noise2); 2: ... and the residual
                                     * distribution was EMPTY under the residual rule (sum 0): the host raises, as the
                                     * reference's torch.multinomial does on NaN probabilities (JL:237) */
    int32_t n_prev;                 /* window length of the last iteration */
    int32_t prob_buf;               /* which of the two prob buffers holds the last iteration's rows */
    int64_t tokens[SJD_MAX_WINDOW]; /* corrected samples Y of the last iteration: [0,m) emitted, [m,n) carried */
    int64_t win_tok[SJD_MAX_WINDOW];/* current window ids */
    int32_t q_src[SJD_MAX_WINDOW];  /* row of the previous prob buffer holding the draft distribution, -1: one-hot */
    int64_t amax[SJD_MAX_WINDOW];   /* K2 by-product: lowest-index argmax of each row's distribution p (the 'sample_horizon' draft
                                     * initialisation of JL:540-586 re-draws a draft from its left neighbour's distribution, which the
                                     * reference reduces to its top-1 entry) */
} sjd_state;

int sjd_version(void);
const char *sjd_error_string(int code);

/* K5 -- window assembly / re-guess.
 * replaces prepare_inputs_for_generation_jacobi + get_multi_token_for_preparation('random') +
 * gather_from_split_tensors (reference scheduler/jacobi_iteration_lumina_mgpt.py:470-514, 606-740;
 * scheduler/logit_processor_3dim.py:513-537).  window = [last emitted | carried samples | fresh ids];
 * draft rows of fresh ids are implicit one-hots (q_src = -1).  Also writes the ids, replicated for each of
 * the `n_batch` CFG rows, to `input_ids_out` [n_batch, max_rows] (feeds the embedding lookup). */
int sjd_reguess(const sjd_iter_params *params, sjd_state *state, int64_t *input_ids_out, int n_batch, int max_rows,
                void *stream);
/* same, and also writes the window's position ids positions_out[b][i] = params->kv_len + i + pos_offset[b] (int64 [n_batch, max_rows];
 * pos_offset int64 [n_batch] or NULL) -- what the reference derives from cache_position on the host (JL:1062-1073) */
int sjd_reguess_ex(const sjd_iter_params *params, sjd_state *state, int64_t *input_ids_out, int n_batch, int max_rows,
                   const int64_t *pos_offset, int64_t *positions_out, void *stream);

/* K2 -- logits -> CFG -> grammar -> top-k/top-p -> softmax -> multinomial.
 * replaces sampling_logits2tokens + the 3-dim processors (reference jacobi_iteration_lumina_mgpt.py:82-132).
 * logits_c/logits_u: fp32 rows of stride `row_stride` elements (logits_u may be NULL: never CFG);
 * noise: [max_rows, V] Exp(1) samples (torch.multinomial == argmax(p/noise));
 * probs_out: [max_rows, V] fp32; tokens_out: int64 [max_rows] (state->tokens works).  Rows >= params->n_rows
 * are skipped. */
int sjd_logits_to_probs_sample(const float *logits_c, const float *logits_u, int64_t row_stride, float guidance,
                               int max_rows, int V, const sjd_iter_params *params, const float *noise,
                               float *probs_out, int64_t *tokens_out, void *stream);
/* same; amax_out (int64 [max_rows], may be NULL; state->amax works) also receives the lowest-index argmax of every row of p */
int sjd_logits_to_probs_sample_ex(const float *logits_c, const float *logits_u, int64_t row_stride, float guidance,
                                  int max_rows, int V, const sjd_iter_params *params, const float *noise,
                                  float *probs_out, int64_t *tokens_out, int64_t *amax_out, void *stream);

/* K2 on an UNMATERIALISED output head: the logits of window row r are
 *     z[r, col] = dtype( row_scale[r] * sum_c part[c][r][col - col0] ),   col in [col0, col0 + n_cols),
 * i.e. the fp32 split-K partials of the lm_head projection (sjd_skinny_gemm_cols over the packed head, only the vocabulary columns the
 * rules allow), the folded final-RMSNorm row scale (row_sumsq from sjd_residual_sumsq; NULL = none) and the rounding nn.Linear applies
 * (reference modeling_chameleon.py:1560-1561 computes 16-bit logits, then .float()).  Cond rows are part rows [0, max_rows), uncond
 * rows [urow_off, urow_off + max_rows) (urow_off <= 0: never CFG).  Everything after the load -- CFG combine, grammar, top-k / top-p,
 * softmax, draw -- is sjd_logits_to_probs_sample's code, so results are bit-identical to feeding it those logits.
 * dbg_c / dbg_u (optional, [max_rows, V] fp32): receive the derived logits of the columns inside each row's rule window (tests). */
typedef struct sjd_head_partials {
    const float *part;              /* [n_chunks][prows][n_cols] fp32 */
    int32_t n_chunks;
    int64_t chunk_stride;           /* elements between chunks (= prows * row_stride) */
    int64_t row_stride;             /* elements between rows (>= n_cols) */
    int32_t col0, n_cols;           /* vocabulary columns covered by `part` */
    int32_t urow_off;               /* part row of uncond row 0 */
    int32_t round_dtype;            /* SJD_DTYPE_BF16 / SJD_DTYPE_F16: round like the 16-bit lm_head output; SJD_DTYPE_F32: none */
    const float *row_sumsq;         /* [slices][prows] per-slice sums of h^2 (folded final norm) or NULL */
    int32_t slices, prows;
    float inv_hidden, eps;
    float *dbg_c, *dbg_u;
    int32_t *zero_state;            /* round 4, optional ([max_rows][2] int32, -1 initially; ONE per probs_out buffer): per row the column window
                                     * [lo, hi) outside of which probs_out is known to hold zeros.  K2 skips the zero fill of the columns outside a
                                     * row's window when the recorded window lies inside it (image rows: the same window every iteration -- 57344
                                     * of Lumina's 65536, 151854 of Emu3's 184622 columns were rewritten with zeros per row and iteration) and
                                     * records the window it leaves behind.  Only K2 may write probs_out while a state is attached to it. */
} sjd_head_partials;
int sjd_logits_to_probs_sample_part(const sjd_head_partials *head /* host struct, passed by value to the kernel */, float guidance,
                                    int max_rows, int V, const sjd_iter_params *params, const float *noise, float *probs_out,
                                    int64_t *tokens_out, int64_t *amax_out /* may be NULL */, void *stream);

/* K2a (round 4) -- the first step of K2 on the whole chip, for heads whose column window is wide (Emu3: 32768 columns): z_out[row][col - col0]
 * (fp32, [max_rows][n_cols]) = the guided score of row `row` -- the chunk planes of its cond and uncond rows summed in chunk order, the folded norm's
 * row scale, the 16-bit rounding of the lm_head output, z = u + guidance (c - u) when params->use_cfg (JL:104) -- bit for bit what K2 derives from
 * the same `head`; the caller then hands K2 a head of ONE plane over z_out (n_chunks 1, urow_off 0, no row_sumsq, round_dtype SJD_DTYPE_F32).
 * part, z_out 16-byte aligned; row_stride, chunk_stride, n_cols multiples of 4.  dbg_c / dbg_u of `head` are written here. */
int sjd_head_combine(const sjd_head_partials *head /* host struct, passed by value to the kernel */, float guidance, int max_rows, int V,
                     const sjd_iter_params *params, float *z_out, void *stream);

/* K4 -- probabilistic verify-and-accept (longest accepted prefix) + residual resample of the first reject.
 * replaces SpeculativeSampler.__call__ / find_first_misaligned_token_inds / prefix_matching_next_tokens
 * (reference jacobi_iteration_lumina_mgpt.py:203-376).
 * probs: this iteration's rows [max_rows, V]; prev_probs: the previous iteration's rows (draft distributions,
 * indexed by state->q_src); rs: [max_rows, V] uniforms (only rs[i, win_tok[i]] is read, JL:282);
 * noise2: [V] Exp(1) for the residual multinomial; scratch: >= V floats.
 * Updates state->{m, rejected, n_prev, tokens}. */
int sjd_verify_accept(const sjd_iter_params *params, sjd_state *state, const float *probs, const float *prev_probs,
                      const float *rs, const float *noise2, float *scratch, int max_rows, int V, void *stream);

/* Round 6 -- K5 / K2 / K4 of EVERY slot of a continuous batch in ONE launch each (SJDBatchEngine: eight prompts paid 8 x (5 + 28 + 24 us) of
 * launches per iteration, a workgroup each).  The per-slot control blobs and buffers are laid out at fixed strides; slot s of the launch works on
 * <pointer of slot 0> + s * <stride> and is otherwise exactly the one-slot kernel (same code after the pointer shift: bit-identical results).
 * Noise is generated in the kernels (every slot's params->philox_blocks must be > 0). */
typedef struct sjd_slots {
    int32_t n_slots;
    int32_t head_rows;              /* K2: rows of `head->part` / `head->row_sumsq` between consecutive slots' cond row 0 (n_batch * max_rows) */
    int64_t params_stride;          /* bytes between consecutive slots' sjd_iter_params */
    int64_t state_stride;           /* bytes between consecutive slots' sjd_state (tokens_out / amax_out point INTO the state: same stride) */
    int64_t probs_stride;           /* floats between consecutive slots' probs_out / probs / prev_probs */
    int64_t zero_state_stride;      /* int32 elements between consecutive slots' head->zero_state */
    int64_t scratch_stride;         /* floats between consecutive slots' K4 scratch rows */
    int64_t mirror_stride;          /* bytes between consecutive slots' pinned host mirrors (K4) */
    int64_t dbg_stride;             /* floats between consecutive slots' head->dbg_c / dbg_u (observers) */
} sjd_slots;
/* sjd_reguess_ex per slot: input_ids_out / positions_out are [n_slots * n_batch, max_rows], pos_offset [n_slots * n_batch] */
int sjd_reguess_slots(const sjd_iter_params *params0, sjd_state *state0, int64_t *input_ids_out, int n_batch, int max_rows,
                      const int64_t *pos_offset, int64_t *positions_out, const sjd_slots *slots, void *stream);
/* sjd_logits_to_probs_sample_part per slot: `head` describes slot 0's rows of the one head launch all slots share */
int sjd_logits_to_probs_sample_part_slots(const sjd_head_partials *head, float guidance, int max_rows, int V, const sjd_iter_params *params0,
                                          float *probs_out0, int64_t *tokens_out0, int64_t *amax_out0 /* may be NULL */, const sjd_slots *slots,
                                          void *stream);
/* sjd_verify_accept_ex per slot (rs / noise2 generated in the kernel) */
int sjd_verify_accept_slots(const sjd_iter_params *params0, sjd_state *state0, const float *probs0, const float *prev_probs0, float *scratch0,
                            int max_rows, int V, sjd_state *host_mirror0 /* may be NULL */, const sjd_slots *slots, void *stream);

/* K3 -- KV append into the static cache (rollback is implicit: rejected rows are simply overwritten).
 * replaces DynamicCache.update's torch.cat + delete_false_key_value (reference modeling_chameleon.py:547;
 * jacobi_iteration_lumina_mgpt.py:47-54, 401-409) and KVCache.update (llamagen/llamagen.py:210-219).
 * k_new/v_new: [B, n_rows, H_kv, D]; k_cache/v_cache: [B, H_kv, S_max, D] of this layer; rows land at
 * [kv_len, kv_len + n_rows).  kv_len is read from params when params != NULL, else from `kv_len`. */
int sjd_kv_append(const void *k_new, const void *v_new, void *k_cache, void *v_cache, int B, int n_rows, int H_kv,
                  int D, int S_max, int dtype, const sjd_iter_params *params, int kv_len, void *stream);

/* K1 -- draft-window attention over the static cache (MFMA QK^T / PV, fp32 online softmax, split over keys).
 * replaces _update_causal_mask + repeat_kv + scaled_dot_product_attention
 * (reference jacobi_iteration_lumina_mgpt.py:1256-1336; modeling_chameleon.py:549-576; llamagen.py:264-273).
 * q: [B, n_rows, H, D]; out: [B, n_rows, H, D]; caches as in sjd_kv_append and ALREADY holding the window rows.
 * Row i of batch b sees key j iff key_start[b] <= j <= kv_len + i.  workspace: >= sjd_attention_workspace_bytes. */
int64_t sjd_attention_workspace_bytes(int B, int H, int n_rows, int D, int n_split);
int sjd_draft_window_attention(const void *q, const void *k_cache, const void *v_cache, void *out, int B, int n_rows,
                               int H, int H_kv, int D, int S_max, int dtype, const int32_t *key_start,
                               const sjd_iter_params *params, int kv_len, int n_split, void *workspace, void *stream);

/* Same as sjd_draft_window_attention; when ev_start/ev_stop (hipEvent_t) are non-NULL they are recorded on `stream`
 * immediately before / after the k1_partial launch (the dominant kernel), for live roofline measurement. */
int sjd_draft_window_attention_ex(const void *q, const void *k_cache, const void *v_cache, void *out, int B, int n_rows,
                                  int H, int H_kv, int D, int S_max, int dtype, const int32_t *key_start,
                                  const sjd_iter_params *params, int kv_len, int n_split, void *workspace, void *stream,
                                  void *ev_start, void *ev_stop);



/* F1-F3 -- fused element-wise glue of the draft-window forward (the "next" row of SURVEY.md 8f.1).
 * F1: h += delta (delta may be NULL; or delta = dtype(sum of the fp32 split-K partials `part`)); y = weight * dtype(h * rsqrt(mean(h^2) + eps)).  replaces ChameleonRMSNorm +
 *     the decoder layer's residual add (reference modeling_chameleon.py:59-73, 637, 643).  h, delta, y: [rows, hidden]. */
int sjd_add_rmsnorm(void *h, const void *delta, const void *weight, void *y, int rows, int hidden, float eps, int dtype,
                    const float *part, int n_chunks, void *stream);
/* F2: per (token, head) of a fused qkv projection [B*n, (H + 2*H_kv) * D]: optional per-head LayerNorm (gamma/beta of
 *     size D; NULL = none), RoPE with fp32 angles positions[t] * inv_freq[d], q -> q_out [B, n, H, D], k/v -> cache rows
 *     [kv_len + i] (kv_len from params when non-NULL).  replaces ChameleonLayerNorm, apply_rotary_pos_emb and
 *     DynamicCache.update (reference modeling_chameleon.py:198-219, 144-178, 547) -- K3 fused into the projection epilogue. */
int sjd_qknorm_rope_append(const void *qkv, void *q_out, void *k_cache, void *v_cache, const void *qn_w, const void *qn_b,
                           const void *kn_w, const void *kn_b, const float *inv_freq, const int64_t *positions, int B, int n,
                           int H, int H_kv, int D, int S_max, int dtype, const sjd_iter_params *params, int kv_len,
                           const float *part, int n_chunks, void *stream);
/* F3: y[rows, inter] = silu(gate_up[:, :inter]) * gate_up[:, inter:].  replaces ChameleonMLP's act_fn/mul (:193-195). */
int sjd_silu_mul(const void *gate_up, void *y, int rows, int inter, int dtype, const float *part, int n_chunks, void *stream);

/* Folded RMSNorm (optional, faster form of F1 -> projection -> F2/F3).  With the norm gain folded into the packed weight of the NEXT
 * projection (W' = W diag(gamma), done once at load time) the RMSNorm is one scale per row and commutes with the projection:
 *   gamma * (h r) W^T == r (h W'^T),  r = rsqrt(mean(h^2) + eps).
 * sjd_residual_sumsq (F1r) does only the residual half of F1 -- h[rows, hidden] += dtype(sum_c part[c]) in place (part may be NULL) --
 * over (row, 512-column slice) workgroups and writes out_sumsq[s, m] = sum of h[m, :]^2 over slice s ([ceil(hidden/512), R] floats,
 * R = rows rounded up to a multiple of 32, rows <= 128); the projection then runs on h itself and its consumer applies r through `row_norm`
 * (sjd_qknorm_rope_append_ex / sjd_silu_mul_ex, on the summed partials before they are rounded to `dtype`).
 * replaces ChameleonRMSNorm + the residual adds (reference modeling_chameleon.py:59-73, 637, 643); differs from F1 only in where bf16
 * rounding happens (x_norm is never rounded; W diag(gamma) is). */
typedef struct sjd_row_norm {
    const float *sumsq;       /* [slices, R] */
    int32_t slices, hidden;   /* hidden = the K of the projection whose input is normalised */
    float eps;
} sjd_row_norm;
int sjd_residual_sumsq(void *h, const float *part, int n_chunks, int rows, int hidden, int dtype, float *out_sumsq, void *stream);
int sjd_qknorm_rope_append_ex(const void *qkv, void *q_out, void *k_cache, void *v_cache, const void *qn_w, const void *qn_b,
                              const void *kn_w, const void *kn_b, const float *inv_freq, const int64_t *positions, int B, int n,
                              int H, int H_kv, int D, int S_max, int dtype, int kv_fp8, float k_scale, float v_scale,
                              const sjd_row_norm *row_norm, const sjd_iter_params *params, int kv_len, const float *part,
                              int n_chunks, void *stream);
int sjd_silu_mul_ex(const void *gate_up, void *y, int rows, int inter, int dtype, const float *part, int n_chunks,
                    const sjd_row_norm *row_norm, void *stream);

/* G1s -- the gate|up projection of the window with F3 as its epilogue, ONE launch: y [M, I] = silu(r (x Wg^T)) * (r (x Wu^T)), r = the
 * row scale of the folded RMSNorm (row_norm; NULL = 1), rounded to `dtype` where nn.Linear and the activation would round.
 * replaces gate_proj / up_proj / act_fn / the product of ChameleonMLP.forward (reference modeling_chameleon.py:193-195) for a window of
 * M <= 32 rows.  w_packed = the [2 I, K] weight [Wg; Wu] packed by sjd_amd.ops.pack_weight with KC = K / 2 (the copy sjd_skinny_gemm
 * streams).  The K split of G1 moves inside the workgroup (8 waves = 4 column tiles x 2 K halves, activation staged in two phases), so
 * the result is bit-identical to sjd_skinny_gemm(KC = K / 2) followed by sjd_silu_mul_ex on its two partial planes.
 * K in {512, 1024, 2048, 4096}, I % 64 == 0; SJD_ERR_UNSUPPORTED otherwise (the caller keeps G1 + F3). */
int sjd_gateup_silu(const void *x, const void *w_packed, void *y, int M, int I, int K, int step_major, int dtype,
                    const sjd_row_norm *row_norm, void *stream);

/* G1 -- weight-streaming projection of the draft window: out[c, m, n] = sum_{k in chunk c} x[m, k] * W[n, k], M <= 256 rows
 * (one to eight 32-row MFMA tiles: up to eight prompts' draft windows per forward), fp32 split-K partials [n_chunks, R, N] with R = M rounded up to
 * a multiple of 32 and n_chunks = ceil(K / KC); the consumer (F1/F2/F3 `part` argument) sums them.
 * replaces the nn.Linear calls of the decoder layer (reference modeling_chameleon.py:527-529, 579, 193-195) for the
 * window forward.  w_packed: the [N, K] weight re-ordered by sjd_amd.ops.pack_weight (MFMA 32x32x16 B-fragment order,
 * one contiguous run per (k-chunk, 32-column tile)).  N % 32 == 0, K % 16 == 0, KC % 16 == 0.  M <= 32: the activation chunk is staged
 * whole in LDS, min(KC, K) <= 2560.  M <= 64: whole while min(KC, K) <= 1280, otherwise -- and always for M > 64 -- in 256-column sub-tiles
 * double-buffered through LDS (no limit on KC; waves <= 8).  The partial planes do not depend on which of the two kernels ran;
 * waves (1..16) = column tiles per workgroup sharing one staged activation chunk; step_major selects the packed record
 * order (0: one contiguous run per tile, 1: the records of all tiles interleaved per k-step).
 * Round 6 -- kernel G1w (csrc/sjd_gemm_wide.h: activation stages by LDS-DMA into a ring shared by the workgroup, hand-counted waits, two to eight row
 * tiles of accumulators per wave) runs when `waves` is 2, 3, 4, 6 or 8 (then = column tiles per workgroup: one per wave up to 4, two per wave for 6 / 8)
 * and 32 < M <= 64 (bf16 or fp16), 64 < M <= 128 (bf16; M <= 96: not 2) or 128 < M <= 256 (bf16 only, these tile counts only): no limit on KC.  Same
 * chunking and accumulation order as the kernels above: the planes do not depend on which kernel ran (tests/test_gpu_glue.py::test_g1w_*,
 * test_g1_skinny_gemm_*_row_tiles).  SJD_G1_WIDE_64 / _128 / SJD_G1_WIDE = 0 in the environment keep the older kernels (A/B aids). */
int sjd_gemm_num_chunks(int K, int KC);

/* the same over the N = 32 n columns [32 * tile0, 32 * tile0 + N) of a weight packed with N_packed columns (the output head evaluated on
 * the grammar's column window out of ONE packed copy of lm_head): out [n_chunks, R, N] */
int sjd_skinny_gemm_cols(const void *x, const void *w_packed, float *out, int M, int N, int K, int KC, int waves, int step_major,
                         int dtype, int N_packed, int tile0, void *stream);
int sjd_skinny_gemm(const void *x, const void *w_packed, float *out, int M, int N, int K, int KC, int waves, int step_major,
                    int dtype, void *stream);

/* G1z / G1sz -- sjd_skinny_gemm_cols / sjd_gateup_silu over a LOSSLESS 12-bit re-encoding of the packed bf16 weight (sjd_amd.ops.pack_weight_z):
 * every weight travels as its low byte (exponent lsb + mantissa) plus a 4-bit code of its high byte -- (sign, offset 0..7 from the base of its
 * (k-chunk, 32-column tile) unit) -- in 768-byte records instead of 1-KiB ones; weights outside the unit's 16-binade window ("exceptions", ~1e-4
 * of a Gaussian matrix, a few 1e-4 of a heavy-tailed one) travel verbatim in `exc` (exc_cap = 32 / 64 / 128 entries of 8 bytes per unit, chosen
 * per matrix by the packer, which declines a matrix that needs more than 127 in some unit: {base, count}, then {k-step << 9 | lane << 3 | element,
 * 16 bits}) and are patched into the MFMA operand registers.  The operands, the accumulation order and the
 * result are BIT-IDENTICAL to the uncompressed kernels on the same weight: no change of precision, 25 % fewer bytes through the fabric that
 * bounds the window forward.  bf16 only (SJD_ERR_UNSUPPORTED otherwise), M <= 64, KC <= 4096.
 * replaces, like G1 / G1s: the nn.Linear calls of the decoder layer (reference modeling_chameleon.py:527-529, 579, 193-195) and lm_head
 * (modeling_chameleon.py:1560-1561) for the window forward. */
int sjd_skinny_gemm_z(const void *x, const void *wz, const void *exc, int exc_cap, float *out, int M, int N, int K, int KC, int waves, int step_major,
                      int dtype, int N_packed, int tile0, void *stream);
int sjd_gateup_silu_z(const void *x, const void *wz, const void *exc, int exc_cap, void *y, int M, int I, int K, int step_major, int dtype,
                      const sjd_row_norm *row_norm, void *stream);

/* Round 6 -- the per-unit escape of the 12-bit stream (csrc/sjd_gemm_raw.h).  A (k-chunk, 32-column tile) unit with more out-of-window weights
 * than a header holds (zero rows, pruned blocks, more than sixteen binades in one unit: a real checkpoint, reference IS:287-289, ML:83-140) no
 * longer makes sjd_amd.ops.pack_weight_z decline the matrix: the unit is zero-filled in the stream, travels verbatim in `raw` (1-KiB records in
 * sjd_skinny_gemm's order, KC / 16 per unit) and ONE small launch behind the stream kernel, on the same stream, recomputes the tiles it feeds --
 * the same MFMA sequence per (tile, chunk, row tile), bit-identical to the uncompressed kernels.  n_raw == 0: no launch.
 *   sjd_raw_units_fixup : behind sjd_skinny_gemm_z(x, ..., out, M, N, K, KC, ..., N_packed, tile0); index int32 [n_raw, 2] = (chunk, packed tile)
 *   sjd_raw_gateup_fixup: behind sjd_gateup_silu_z(x, ..., y, M, I, K, ..., row_norm); tiles int32 [n_pairs] gate tiles, raw per pair
 *                         [K half][gate | up][K / 32 records] (the packer lists the whole pair when any of its four units is raw)
 * replaces, like G1z / G1sz: the nn.Linear calls of the decoder layer (reference modeling_chameleon.py:527-529, 579, 193-195) and lm_head. */
int sjd_raw_units_fixup(const void *x, const void *raw, const int32_t *index, int n_raw, float *out, int M, int N, int K, int KC, int tile0,
                        int dtype, void *stream);
int sjd_raw_gateup_fixup(const void *x, const void *raw, const int32_t *tiles, int n_pairs, void *y, int M, int I, int K, int dtype,
                         const sjd_row_norm *row_norm, void *stream);


/* K1 / K3 over an fp8 KV cache (BASELINE config 5; there is no fp8 in the reference -- the parity target is the bf16 result
 * within tolerance): the cache holds OCP e4m3 bytes, value = fp8 * scale with one scale per tensor; q / out keep `dtype` (bf16/f16).
 * Half the KV bytes of sjd_draft_window_attention, both contractions on v_mfma_f32_16x16x32_fp8_fp8.  Other arguments as the
 * 16-bit entry points (replaces the same reference call sites, modeling_chameleon.py:499-581 / DynamicCache.update). */
int sjd_kv_append_fp8(const void *k_new, const void *v_new, void *k_cache, void *v_cache, int B, int n_rows, int H_kv, int D, int S_max,
                      int dtype, float k_scale, float v_scale, int head_major /* new rows [B,H_kv,n,D] instead of [B,n,H_kv,D] */,
                      const sjd_iter_params *params, int kv_len, void *stream);
/* F2 writing into an fp8 cache: k/v rows are rounded to `dtype` exactly as sjd_qknorm_rope_append does, then stored as fp8(x / scale). */
int sjd_qknorm_rope_append_fp8(const void *qkv, void *q_out, void *k_cache, void *v_cache, const void *qn_w, const void *qn_b,
                               const void *kn_w, const void *kn_b, const float *inv_freq, const int64_t *positions, int B, int n,
                               int H, int H_kv, int D, int S_max, int dtype, float k_scale, float v_scale,
                               const sjd_iter_params *params, int kv_len, const float *part, int n_chunks, void *stream);
int sjd_draft_window_attention_fp8(const void *q, const void *k_cache, const void *v_cache, void *out, int B, int n_rows, int H, int H_kv,
                                   int D, int S_max, int dtype, float k_scale, float v_scale, const int32_t *key_start,
                                   const sjd_iter_params *params, int kv_len, int n_split, void *workspace, void *stream);

/* K4 with the read-back folded in: when host_mirror != NULL (SJD_STATE_MIRROR_BYTES of HOST memory the device can address: hipHostMalloc /
 * a pinned torch tensor) the kernel ends by copying *state into it and publishing params->iter_seq behind it, so the host's one sync per iteration (reference: the ~25-40 implicit syncs of
 * jacobi_iteration_lumina_mgpt.py:1107-1208) is sjd_stream_synchronize and no D2H copy is enqueued. */
int sjd_verify_accept_ex(const sjd_iter_params *params, sjd_state *state, const float *probs, const float *prev_probs,
                         const float *rs, const float *noise2, float *scratch, int max_rows, int V, sjd_state *host_mirror, void *stream);
/* the iteration's parameter blob: pinned host -> device, asynchronous on `stream`; and the wait that closes an iteration */
int sjd_upload_async(void *dst_device, const void *src_pinned_host, int64_t bytes, void *stream);
int sjd_stream_synchronize(void *stream);
/* The mirror is sizeof(sjd_state) + 8 bytes: behind the state K4 stores (uint64) params->iter_seq, after a system-scope fence, as its very
 * last action.  A host that polls that word (sjd_host_wait_u64: a spin in C, no HIP call, returns SJD_ERR_LAUNCH after timeout_us) sees
 * the iteration's result a microsecond after the kernel wrote it instead of after the runtime's interrupt-driven stream wait. */
#define SJD_STATE_MIRROR_BYTES (sizeof(sjd_state) + 8)
int sjd_host_wait_u64(const volatile uint64_t *flag, uint64_t value, int64_t timeout_us);


/* K1 without key splits (round 4): the four workgroups of a (batch, head) split the OUTPUT COLUMNS -- each scores all keys of the pair
 * and multiplies them with its 32 columns of V -- merge their eight waves' states in LDS and write the normalised 16-bit output: one
 * launch, no workspace, no combine kernel.  Multi-head attention only (H == H_kv), D = 128 (SJD_ERR_UNSUPPORTED otherwise).  Same call
 * sites and argument meaning as sjd_draft_window_attention(_fp8) (modeling_chameleon.py:499-581); faster than key splits + k1_combine while
 * the context is short (16-bit caches: below ~750 keys, fp8: the whole 512 x 512 image; profiles/r4_k1_dsplit_ab.txt) -- the caller picks
 * per launch. */
int sjd_draft_window_attention_colsplit(const void *q, const void *k_cache, const void *v_cache, void *out, int B, int n_rows, int H, int H_kv,
                                        int D, int S_max, int dtype, const int32_t *key_start, const sjd_iter_params *params, int kv_len,
                                        void *stream);
int sjd_draft_window_attention_fp8_colsplit(const void *q, const void *k_cache, const void *v_cache, void *out, int B, int n_rows, int H,
                                            int H_kv, int D, int S_max, int dtype, float k_scale, float v_scale, const int32_t *key_start,
                                            const sjd_iter_params *params, int kv_len, void *stream);


/* In-kernel noise (SURVEY.md section 7 "reproduce torch's Philox offsets", 8-K4 "K4 needs L-1 uniforms").  The reference draws
 * torch.multinomial / torch.rand / torch.multinomial from a device torch.Generator (jacobi_iteration_lumina_mgpt.py:118, 260, 237).  With
 * params->philox_blocks > 0, K2 and K4 compute the very elements those ATen launches would have written (Philox4x32-10 at the generator's
 * seed and offsets, ATen's thread -> element layout; csrc/sjd_philox.h) and their noise / rs / noise2 arguments may be NULL; the host
 * advances the generator's offset by sjd_philox_offset_increment(numel, philox_blocks) per tensor the reference would have drawn.
 * sjd_philox_fill writes such a tensor out (kind 0: uniform_(0, 1) == torch.rand; 1: exponential_(1)) -- the handle of the parity test
 * against torch itself (tests/test_gpu_philox.py). */
int sjd_philox_fill(float *out, int64_t numel, uint64_t seed, uint64_t offset, int max_blocks, int kind, void *stream);
uint64_t sjd_philox_offset_increment(int64_t numel, int max_blocks);

/* HIP event helpers so that a ctypes host can time kernels on the stream they run on. */
void *sjd_event_create(void);
void sjd_event_destroy(void *ev);
int sjd_event_synchronize(void *ev);
float sjd_event_elapsed_ms(void *ev_start, void *ev_stop);

#ifdef __cplusplus
}
#endif
#endif /* SJD_HIP_H */
