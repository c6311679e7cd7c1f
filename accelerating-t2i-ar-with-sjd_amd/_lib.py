"""ctypes binding of libsjd_hip.so (the C-ABI declared in include/sjd_hip.h).

There is no fallback: if the shared library is missing or does not load, importing the hot path raises.
Build it with ``python __graft_entry__.py`` / ``make -C accelerating-t2i-ar-with-sjd_amd/csrc``.
"""
import ctypes
import os

MAX_WINDOW = 64
MAX_RANGES = 4
DTYPE_BF16, DTYPE_F16, DTYPE_F32 = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("SJD_HIP_LIB") or os.path.join(_HERE, "libsjd_hip.so")      # SJD_HIP_LIB: an instrumented build (tools/phase_trace.py)


class RowRule(ctypes.Structure):
    _fields_ = [("n_ranges", ctypes.c_int32), ("lo", ctypes.c_int32 * MAX_RANGES), ("hi", ctypes.c_int32 * MAX_RANGES),
                ("forced", ctypes.c_int32), ("top_k", ctypes.c_int32), ("top_p_thr", ctypes.c_float), ("temperature", ctypes.c_float)]


class IterParams(ctypes.Structure):
    _fields_ = [("n_rows", ctypes.c_int32), ("kv_len", ctypes.c_int32), ("use_cfg", ctypes.c_int32),
                ("scheme", ctypes.c_int32), ("n_fresh", ctypes.c_int32), ("batch_rows", ctypes.c_int32), ("iter_seq", ctypes.c_int32), ("philox_blocks", ctypes.c_int32),
                ("philox_seed", ctypes.c_uint64), ("philox_offset", ctypes.c_uint64 * 3),
                ("fresh_tok", ctypes.c_int64 * MAX_WINDOW), ("rules", RowRule * MAX_WINDOW),
                ("resid_rules", RowRule * MAX_WINDOW)]


class State(ctypes.Structure):
    _fields_ = [("m", ctypes.c_int32), ("rejected", ctypes.c_int32), ("n_prev", ctypes.c_int32),
                ("prob_buf", ctypes.c_int32), ("tokens", ctypes.c_int64 * MAX_WINDOW),
                ("win_tok", ctypes.c_int64 * MAX_WINDOW), ("q_src", ctypes.c_int32 * MAX_WINDOW), ("amax", ctypes.c_int64 * MAX_WINDOW)]


class RowNorm(ctypes.Structure):
    """sjd_row_norm: per-slice sums of squares of the residual stream -> the RMSNorm scale of a row (folded-norm path)."""
    _fields_ = [("sumsq", ctypes.c_void_p), ("slices", ctypes.c_int32), ("hidden", ctypes.c_int32), ("eps", ctypes.c_float)]


class Slots(ctypes.Structure):
    """sjd_slots: the strides between the slots of a continuous batch (K5 / K2 / K4 of every slot in one launch each)"""
    _fields_ = [("n_slots", ctypes.c_int32), ("head_rows", ctypes.c_int32), ("params_stride", ctypes.c_int64), ("state_stride", ctypes.c_int64),
                ("probs_stride", ctypes.c_int64), ("zero_state_stride", ctypes.c_int64), ("scratch_stride", ctypes.c_int64),
                ("mirror_stride", ctypes.c_int64), ("dbg_stride", ctypes.c_int64)]


class HeadPartials(ctypes.Structure):
    """sjd_head_partials: the unmaterialised output head K2 reads (split-K partials of the lm_head projection)"""
    _fields_ = [("part", ctypes.c_void_p), ("n_chunks", ctypes.c_int32), ("chunk_stride", ctypes.c_int64), ("row_stride", ctypes.c_int64),
                ("col0", ctypes.c_int32), ("n_cols", ctypes.c_int32), ("urow_off", ctypes.c_int32), ("round_dtype", ctypes.c_int32),
                ("row_sumsq", ctypes.c_void_p), ("slices", ctypes.c_int32), ("prows", ctypes.c_int32), ("inv_hidden", ctypes.c_float),
                ("eps", ctypes.c_float), ("dbg_c", ctypes.c_void_p), ("dbg_u", ctypes.c_void_p), ("zero_state", ctypes.c_void_p)]


class L2Head(ctypes.Structure):
    """sjd_l2_head: the head of a G1z / G1sz launch's weight stream as that launch will read it (round 5, csrc/sjd_l2_prefetch.h)"""
    _fields_ = [("wz", ctypes.c_void_p), ("kind", ctypes.c_int32), ("gx", ctypes.c_int32), ("gy", ctypes.c_int32), ("waves", ctypes.c_int32),
                ("n_tiles", ctypes.c_int32), ("tile0", ctypes.c_int32), ("n_out", ctypes.c_int32), ("pairs_full", ctypes.c_int32),
                ("pairs_last", ctypes.c_int32), ("step_major", ctypes.c_int32), ("head_pairs", ctypes.c_int32)]


# what include/sjd_hip.h declares = what libsjd_hip.so exports = what the product (engine.py, engine_batch.py, backbones.py with default switches) reaches
EXPORTS = ["sjd_version", "sjd_error_string", "sjd_reguess", "sjd_logits_to_probs_sample", "sjd_verify_accept",
           "sjd_kv_append", "sjd_attention_workspace_bytes", "sjd_draft_window_attention", "sjd_draft_window_attention_ex",
           "sjd_event_create", "sjd_event_destroy", "sjd_event_synchronize", "sjd_event_elapsed_ms",
           "sjd_add_rmsnorm", "sjd_qknorm_rope_append", "sjd_silu_mul", "sjd_gemm_num_chunks", "sjd_skinny_gemm",
           "sjd_kv_append_fp8", "sjd_draft_window_attention_fp8", "sjd_qknorm_rope_append_fp8",
           "sjd_residual_sumsq", "sjd_qknorm_rope_append_ex", "sjd_silu_mul_ex", "sjd_skinny_gemm_cols",
           "sjd_logits_to_probs_sample_part", "sjd_logits_to_probs_sample_ex", "sjd_reguess_ex",
           "sjd_verify_accept_ex", "sjd_upload_async", "sjd_stream_synchronize", "sjd_gateup_silu", "sjd_host_wait_u64",
           "sjd_philox_fill", "sjd_philox_offset_increment", "sjd_skinny_gemm_z", "sjd_gateup_silu_z",
           "sjd_draft_window_attention_colsplit", "sjd_draft_window_attention_fp8_colsplit",
           "sjd_head_combine", "sjd_raw_units_fixup", "sjd_raw_gateup_fixup",
           "sjd_reguess_slots", "sjd_logits_to_probs_sample_part_slots", "sjd_verify_accept_slots"]
# what include/sjd_hip_experimental.h adds: libsjd_hip_exp.so only (the measured no-go structures of rounds 2-5 and the G1w tuning entry)
EXP_EXPORTS = ["sjd_weight_prefetch", "sjd_qkv_attention_fused", "sjd_qkv_attention_fused_split", "sjd_skinny_gemm_reduce", "sjd_reduce_timeouts",
               "sjd_draft_window_attention_merged", "sjd_draft_window_attention_fp8_merged", "sjd_mlp_pair_z", "sjd_mlp_pair_timeouts",
               "sjd_l2_head_gemm_z", "sjd_l2_head_gateup_z", "sjd_l2_head_bytes", "sjd_weight_prefetch_head", "sjd_debug_xcc_map", "sjd_residual_sumsq_pf",
               "sjd_skinny_gemm_engine_z", "sjd_engine_timeouts", "sjd_skinny_gemm_wide", "sjd_o_merge_prologue_probe", "sjd_skinny_gemm_z_wide"]
EXP_SO_PATH = os.environ.get("SJD_HIP_EXP_LIB") or os.path.join(_HERE, "libsjd_hip_exp.so")

_lib = None


class SjdLibraryError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise SjdLibraryError(f"{SO_PATH} is missing: build the HIP extension first (python __graft_entry__.py). "
                              "There is no CPU/torch fallback for the SJD hot path.")
    lib = ctypes.CDLL(SO_PATH)
    _bind_product(lib)
    for name in EXPORTS:
        getattr(lib, name)
    assert ctypes.sizeof(RowRule) == 52 and ctypes.sizeof(IterParams) == 64 + 8 * MAX_WINDOW + 2 * 52 * MAX_WINDOW
    _lib = lib
    return lib


def _bind_product(lib):
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    lib.sjd_version.restype = i32
    lib.sjd_error_string.restype = ctypes.c_char_p
    lib.sjd_error_string.argtypes = [i32]
    lib.sjd_reguess.argtypes = [vp, vp, vp, i32, i32, vp]
    lib.sjd_reguess_ex.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
    lib.sjd_logits_to_probs_sample.argtypes = [vp, vp, i64, f32, i32, i32, vp, vp, vp, vp, vp]
    lib.sjd_verify_accept.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    lib.sjd_verify_accept_ex.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]
    lib.sjd_upload_async.argtypes = [vp, vp, i64, vp]
    lib.sjd_gateup_silu.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, ctypes.POINTER(RowNorm), vp]
    lib.sjd_stream_synchronize.argtypes = [vp]
    lib.sjd_host_wait_u64.argtypes = [vp, ctypes.c_uint64, i64]
    lib.sjd_kv_append.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp]
    lib.sjd_attention_workspace_bytes.restype = i64
    lib.sjd_attention_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    lib.sjd_draft_window_attention.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp, vp]
    lib.sjd_draft_window_attention_ex.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp]
    lib.sjd_draft_window_attention_colsplit.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp]
    lib.sjd_draft_window_attention_fp8_colsplit.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, f32, vp, vp, i32, vp]
    lib.sjd_add_rmsnorm.argtypes = [vp, vp, vp, vp, i32, i32, f32, i32, vp, i32, vp]
    lib.sjd_qknorm_rope_append.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp]
    lib.sjd_silu_mul.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp]
    lib.sjd_gemm_num_chunks.argtypes = [i32, i32]
    lib.sjd_skinny_gemm.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.sjd_qknorm_rope_append_fp8.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, f32, vp, i32, vp, i32, vp]
    lib.sjd_residual_sumsq.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    lib.sjd_qknorm_rope_append_ex.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, f32,
                                              ctypes.POINTER(RowNorm), vp, i32, vp, i32, vp]
    lib.sjd_silu_mul_ex.argtypes = [vp, vp, i32, i32, i32, vp, i32, ctypes.POINTER(RowNorm), vp]
    lib.sjd_kv_append_fp8.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, f32, i32, vp, i32, vp]
    lib.sjd_draft_window_attention_fp8.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, f32, vp, vp, i32, i32, vp, vp]
    lib.sjd_skinny_gemm_cols.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.sjd_logits_to_probs_sample_part.argtypes = [ctypes.POINTER(HeadPartials), f32, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.sjd_head_combine.argtypes = [ctypes.POINTER(HeadPartials), f32, i32, i32, vp, vp, vp]
    lib.sjd_reguess_slots.argtypes = [vp, vp, vp, i32, i32, vp, vp, ctypes.POINTER(Slots), vp]
    lib.sjd_logits_to_probs_sample_part_slots.argtypes = [ctypes.POINTER(HeadPartials), f32, i32, i32, vp, vp, vp, vp, ctypes.POINTER(Slots), vp]
    lib.sjd_verify_accept_slots.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, ctypes.POINTER(Slots), vp]
    lib.sjd_logits_to_probs_sample_ex.argtypes = [vp, vp, i64, f32, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.sjd_skinny_gemm_z.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.sjd_gateup_silu_z.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, ctypes.POINTER(RowNorm), vp]
    lib.sjd_raw_units_fixup.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.sjd_raw_gateup_fixup.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32, i32, ctypes.POINTER(RowNorm), vp]
    lib.sjd_philox_fill.argtypes = [vp, i64, ctypes.c_uint64, ctypes.c_uint64, i32, i32, vp]
    lib.sjd_philox_offset_increment.restype = ctypes.c_uint64
    lib.sjd_philox_offset_increment.argtypes = [i64, i32]
    lib.sjd_event_create.restype = vp
    lib.sjd_event_destroy.argtypes = [vp]
    lib.sjd_event_synchronize.argtypes = [vp]
    lib.sjd_event_elapsed_ms.restype = f32
    lib.sjd_event_elapsed_ms.argtypes = [vp, vp]


_exp = None


def load_exp():
    """libsjd_hip_exp.so: the product's entry points plus the experimental ones (include/sjd_hip_experimental.h).  Only the opt-in switches of
    backbones.py, the experiments' own tests and the tools call this; the product path never does."""
    global _exp
    if _exp is not None:
        return _exp
    if not os.path.exists(EXP_SO_PATH):
        raise SjdLibraryError(f"{EXP_SO_PATH} is missing: build it first (python __graft_entry__.py / make -C accelerating-t2i-ar-with-sjd_amd/csrc)")
    lib = ctypes.CDLL(EXP_SO_PATH)
    _bind_product(lib)
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    lib.sjd_draft_window_attention_merged.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    lib.sjd_draft_window_attention_fp8_merged.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, f32, vp, vp, i32, i32, vp, vp, vp]
    lib.sjd_o_merge_prologue_probe.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.sjd_skinny_gemm_z_wide.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.sjd_mlp_pair_z.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, i32, i32, vp, i32, i32, i32, i32, ctypes.POINTER(RowNorm), vp, i32, vp]
    lib.sjd_weight_prefetch.argtypes = [vp, i64, i32, vp, vp]
    lib.sjd_qkv_attention_fused.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, ctypes.POINTER(RowNorm),
                                            vp, vp, i32, vp]
    lib.sjd_qkv_attention_fused_split.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, ctypes.POINTER(RowNorm),
                                                  vp, vp, i32, i32, vp, vp]
    lib.sjd_skinny_gemm_reduce.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.sjd_l2_head_gemm_z.argtypes = [ctypes.POINTER(L2Head), vp, i32, i32, i32, i32, i32, i32, i32, i32, i32]
    lib.sjd_l2_head_gateup_z.argtypes = [ctypes.POINTER(L2Head), vp, i32, i32, i32, i32, i32]
    lib.sjd_l2_head_bytes.restype = i64
    lib.sjd_l2_head_bytes.argtypes = [ctypes.POINTER(L2Head)]
    lib.sjd_weight_prefetch_head.argtypes = [ctypes.POINTER(L2Head), i32, vp]
    lib.sjd_debug_xcc_map.argtypes = [vp, i32, i32, vp]
    lib.sjd_skinny_gemm_engine_z.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.sjd_residual_sumsq_pf.argtypes = [vp, vp, i32, i32, i32, i32, vp, ctypes.POINTER(L2Head), i32, vp]
    lib.sjd_skinny_gemm_wide.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    for name in EXPORTS + EXP_EXPORTS:
        getattr(lib, name)
    _exp = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise SjdLibraryError(f"{what} failed: {load().sjd_error_string(rc).decode()} ({rc})")
