"""ctypes/numpy front-end of the CPU oracle (oracle/sjd_oracle.c) + host-side grammar restatements.

TEST INFRASTRUCTURE ONLY -- see the header of sjd_oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package.

Grammar restatements (stateless functions of the accepted context):
  lumina_rules   <- MultiTokensVLLogitsProcessor + MultiTokensInterleavedTopKLogitsWarper
                    (reference scheduler/logit_processor_3dim.py:25-43, 84-155, 190-204)
  llamagen_rules <- TopKLogitsWarper + TopPLogitsWarper3d (llamagen/llamagen_solver.py:458-470)
  emu3_rules     <- EOLLogitProcessor3d (scheduler/jacobi_iteration_emu3.py:44-128) + TopK(2048)
  anole_rules    <- the five 3d processors wired at scheduler/jacobi_iteration_anhole.py:194-232
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsjd_oracle.so")
MAX_RANGES = 4


class RowRule(ctypes.Structure):
    _fields_ = [("n_ranges", ctypes.c_int32), ("lo", ctypes.c_int32 * MAX_RANGES), ("hi", ctypes.c_int32 * MAX_RANGES),
                ("forced", ctypes.c_int32), ("top_k", ctypes.c_int32), ("top_p_thr", ctypes.c_float), ("temperature", ctypes.c_float)]


def build():
    src = os.path.join(_HERE, "sjd_oracle.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        f32p, i64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
        _lib.sjd_o_logits_to_probs_sample.argtypes = [f32p, f32p, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                                      ctypes.POINTER(RowRule), f32p, f32p, i64p]
        _lib.sjd_o_verify_accept.argtypes = [ctypes.c_int, ctypes.c_int, i64p, i64p, f32p, ctypes.POINTER(f32p), f32p,
                                             ctypes.POINTER(RowRule), f32p, ctypes.POINTER(ctypes.c_int)]
        _lib.sjd_o_first_mismatch.argtypes = [ctypes.c_int, i64p, i64p]
        _lib.sjd_o_expf.argtypes = [ctypes.c_float]
        _lib.sjd_o_expf.restype = ctypes.c_float
        _lib.sjd_o_logf.argtypes = [ctypes.c_float]
        _lib.sjd_o_logf.restype = ctypes.c_float
        _lib.sjd_o_sum.argtypes = [f32p, ctypes.c_int]
        _lib.sjd_o_sum.restype = ctypes.c_float
        _lib.sjd_o_max_threads.restype = ctypes.c_int
        _lib.sjd_o_set_threads.argtypes = [ctypes.c_int]
    return _lib


def set_threads(n):
    """OpenMP threads of the K2 restatement (rows of a window are independent); returns the count in effect."""
    lib().sjd_o_set_threads(int(n))
    return int(lib().sjd_o_max_threads())


def top_p_threshold(top_p):
    """float32(1 - top_p) as compared at logit_processor_3dim.py:411; -1 disables (top_p >= 1 is a no-op on p)."""
    if top_p is None or top_p >= 1.0:
        return -1.0
    return float(np.float32(1.0 - float(top_p)))


def rule(ranges=(), forced=-1, top_k=0, top_p=None, temperature=1.0):
    r = RowRule()
    r.temperature = float(temperature or 1.0)
    ranges = list(ranges)
    assert len(ranges) <= MAX_RANGES, ranges
    r.n_ranges = len(ranges)
    for i, (lo, hi) in enumerate(ranges):
        r.lo[i], r.hi[i] = int(lo), int(hi)
    r.forced, r.top_k, r.top_p_thr = int(forced), int(top_k or 0), top_p_threshold(top_p)
    return r


def tempered(rules_fn, temperature, top_p=None):
    """rules_fn(ctx, n) -> the same rules with HF's TemperatureLogitsWarper(temperature) -- and, with top_p, HF's TopPLogitsWarper(top_p) --
    appended to the processor list (the warpers follow the user's processors in transformers' generate(): temperature, top-k, top-p; see
    sjd_oracle.c)"""
    def fn(ctx, n):
        out = []
        for r in rules_fn(ctx, n):
            c = RowRule.from_buffer_copy(r)
            c.temperature = float(temperature)
            if top_p is not None:
                c.top_p_thr = top_p_threshold(top_p)
            out.append(c)
        return out
    return fn


def rules_array(rules):
    arr = (RowRule * len(rules))()
    for i, r in enumerate(rules):
        arr[i] = r
    return arr


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def logits_to_probs_sample(logits_c, logits_u, guidance, rules, noise):
    """logits_c/u [n,V] fp32 (u None => no CFG); noise [n,V] Exp(1).  -> (tokens int64 [n], probs fp32 [n,V])"""
    c = _f32(logits_c)
    n, V = c.shape
    u = None if logits_u is None else _f32(logits_u)
    e = _f32(noise)
    probs = np.empty((n, V), dtype=np.float32)
    toks = np.empty((n,), dtype=np.int64)
    rc = lib().sjd_o_logits_to_probs_sample(
        _p(c, ctypes.c_float), None if u is None else _p(u, ctypes.c_float), float(guidance), n, V,
        rules_array(rules), _p(e, ctypes.c_float), _p(probs, ctypes.c_float), _p(toks, ctypes.c_int64))
    assert rc == 0
    return toks, probs


def verify_accept(win_tok, tokens, p, q_rows, rs, resid_rules, noise2):
    """q_rows: list of fp32 [V] arrays or None (one-hot at win_tok[i]).  -> (m, tokens(corrected), rejected)"""
    win = np.ascontiguousarray(win_tok, dtype=np.int64)
    tok = np.ascontiguousarray(tokens, dtype=np.int64).copy()
    p = _f32(p)
    n, V = p.shape
    rs = _f32(rs)
    e2 = _f32(noise2)
    keep = [None if q is None else _f32(q) for q in q_rows]
    ptrs = (ctypes.POINTER(ctypes.c_float) * n)()
    for i, q in enumerate(keep):
        ptrs[i] = None if q is None else _p(q, ctypes.c_float)
    rej = ctypes.c_int(0)
    m = lib().sjd_o_verify_accept(n, V, _p(win, ctypes.c_int64), _p(tok, ctypes.c_int64), _p(p, ctypes.c_float), ptrs,
                                  _p(rs, ctypes.c_float), rules_array(resid_rules), _p(e2, ctypes.c_float),
                                  ctypes.byref(rej))
    return int(m), tok, bool(rej.value)


def first_mismatch(win_tok, tokens):
    win = np.ascontiguousarray(win_tok, dtype=np.int64)
    tok = np.ascontiguousarray(tokens, dtype=np.int64)
    return int(lib().sjd_o_first_mismatch(len(win), _p(win, ctypes.c_int64), _p(tok, ctypes.c_int64)))


def expf(x):
    return float(lib().sjd_o_expf(float(x)))


def canonical_sum(v):
    v = _f32(v)
    return float(lib().sjd_o_sum(_p(v, ctypes.c_float), v.size))


# ------------------------------------------------------------------------------------------------
# grammar restatements: context (1-D int sequence of ACCEPTED ids) + n rows -> n RowRules
# ------------------------------------------------------------------------------------------------
def _forced_rows(tokenlen, n, line_len):
    """rows j in [0,n) with (tokenlen + 1 + j) % line_len == 0  (logit_processor_3dim.py:25-43)"""
    return [j for j in range(n) if (tokenlen + 1 + j) % line_len == 0]


def lumina_rules(ctx, n, image_top_k=2000, text_top_k=10, start=8197, end=8196, eol=8803, img_lo=4, img_hi=8196):
    ctx = [int(t) for t in ctx]
    n_start, n_end = ctx.count(start), ctx.count(end)
    k = image_top_k if n_start == n_end + 1 else text_top_k          # LP:195-198
    rules = [dict(ranges=(), forced=-1) for _ in range(n)]
    if n_start == n_end + 1:                                         # LP:94
        idx = len(ctx) - 1 - ctx[::-1].index(start)                  # last start token (LP:96-97)
        new_token_num = len(ctx) - (idx + 1)
        if new_token_num >= 2:                                       # LP:102
            h = (ctx[idx + 1] - 8804) * 2                            # LP:107-111
            w = (ctx[idx + 2] - 8804) * 2
            T = len(ctx) - (idx + 3)
            for r in rules:
                r["ranges"] = ((img_lo, img_hi),)                    # LP:125-129
            for j in _forced_rows(T, n, w + 1):                      # LP:132-137
                rules[j]["forced"] = eol
            for j in _forced_rows(T, n, (w + 1) * h + 1):            # LP:140-145
                rules[j]["forced"] = end
    return [rule(r["ranges"], r["forced"], k) for r in rules]


def lumina_force_no_cfg(ctx, start=8197, end=8196):
    """check_is_force_no_cfg (jacobi_iteration_lumina_mgpt.py:70-80)"""
    ctx = [int(t) for t in ctx]
    return ctx.count(start) == ctx.count(end)


def llamagen_rules(ctx, n, top_k, top_p):
    return [rule((), -1, top_k, top_p) for _ in range(n)]


def emu3_rules(ctx, n, H, W, vis_lo, vis_n, img_token, eoi_token, eos_token, eol_token, eof_token, pad_token,
               top_k=2048):
    ctx = [int(t) for t in ctx]
    pos = ctx.index(img_token)                                       # first occurrence, cached (JE:50-52)
    T = len(ctx) - (pos + 1)
    L1 = W + 1
    base = (W + 1) * H
    forced = [-1] * n
    for j in _forced_rows(T, n, L1):
        forced[j] = eol_token
    for j in _forced_rows(T, n, base + 1):
        forced[j] = eof_token
    for j in _forced_rows(T, n, base + 2):
        forced[j] = eoi_token
    for j in _forced_rows(T, n, base + 3):
        forced[j] = eos_token
    if T + n > base + 3:                                             # JE:118-123 (python slice semantics kept)
        s = base + 3 - T
        for j in range(n)[s:]:
            forced[j] = pad_token
    return [rule(((vis_lo, vis_lo + vis_n),), f, top_k) for f in forced]


def _mask_to_ranges(allowed):
    idx = np.flatnonzero(allowed)
    if idx.size == 0:
        raise ValueError("grammar masks every token")
    cuts = np.flatnonzero(np.diff(idx) > 1)
    los = np.concatenate([[idx[0]], idx[cuts + 1]])
    his = np.concatenate([idx[cuts] + 1, [idx[-1] + 1]])
    return list(zip(los.tolist(), his.tolist()))


def anole_rules(ctx, n, V, prompt_len, max_length, image_seq_length, boi=8197, eoi=8196, eos=2, img_lo=4, img_hi=8196,
                top_k=2000, mode="image-only"):
    """The restricted modes of the Anole pipeline (jacobi_iteration_anhole.py:178-260): "image-only" = processors 1-5 below,
    "interleaved-text-image" = 1-3, "text-only" = one SuppressTokens(image ids + boi + eoi).  NB: the reference evaluates these on the
    accepted prefix only, so every window row gets the same mask (logit_processor_3dim.py:242-256, 280-286, 323-338:
    input_ids.shape[1], no per-row offset)."""
    ctx = [int(t) for t in ctx]
    cur = len(ctx)
    masked = np.zeros(V, dtype=bool)
    img = np.zeros(V, dtype=bool)
    img[img_lo:img_hi] = True
    if mode == "text-only":
        masked |= img
        masked[[boi, eoi]] = True
        return [rule(_mask_to_ranges(~masked), -1, top_k) for _ in range(n)]
    assert mode in ("image-only", "interleaved-text-image"), mode
    # 1. AllowOnlyTokensAtRelativeOffset(trigger=boi, allowed=[eoi], offset=L+1, exclusive)
    offset = image_seq_length + 1
    only_eoi = np.zeros(V, dtype=bool)
    only_eoi[eoi] = True
    if cur < offset:
        masked |= only_eoi
    elif ctx[-offset] == boi:
        masked |= ~only_eoi
    else:
        masked |= only_eoi
    # 2. AllowOnlyTokensInRelativeWindow(trigger=boi, allowed=image ids, width=L, exclusive)
    window = min(image_seq_length, cur)
    if boi in ctx[-window:]:
        masked |= ~img
    else:
        masked |= img
    # 3. SuppressTokensInIndexRange([boi], start=max_length-L-1, end=inf)
    if not (max_length - image_seq_length - 1 > cur):
        masked[boi] = True
    if mode == "image-only":
        # 4. SuppressTokens(everything but image ids, eos, boi, eoi)
        allowed4 = img.copy()
        allowed4[[eos, boi, eoi]] = True
        masked |= ~allowed4
        # 5. SuppressTokensAtBegin([eos], begin_index=prompt_len): active for cur in {begin, begin+1}
        if prompt_len <= cur <= prompt_len + 1:
            masked[eos] = True
    ranges = _mask_to_ranges(~masked)
    return [rule(ranges, -1, top_k) for _ in range(n)]


def lumina_grid(ctx, start=8197, end=8196, img_lo=4, img_hi=8196):
    """(index of the first image token, w_latent, img_lo, img_hi) while an image is open (#start == #end + 1 and the two grid tokens are
    known), else None -- the geometry the spatial init schemes use (reference JL:531-537: img_width = w_latent_dim, one pad per row)."""
    ctx = list(ctx)
    if ctx.count(start) != ctx.count(end) + 1:
        return None
    i = len(ctx) - 1 - ctx[::-1].index(start)
    if len(ctx) - i < 3:
        return None
    w = (ctx[i + 2] - 8804) * 2
    return (i + 3, w, img_lo, img_hi) if w > 0 else None


def emu3_grid(ctx, H, W, vis_lo, vis_n, img_token):
    ctx = list(ctx)
    if img_token not in ctx:
        return None
    i = ctx.index(img_token)
    if len(ctx) - 1 - i >= (W + 1) * H:
        return None
    return (i + 1, W, vis_lo, vis_lo + vis_n)
