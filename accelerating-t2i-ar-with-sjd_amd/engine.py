"""SJD decode engine: the reference's `_sample` loop (scheduler/jacobi_iteration_lumina_mgpt.py:912-1249) as a
shape-static launch sequence over device-resident state.

Per window iteration (reference line numbers in brackets):
   host   n, a', fresh random ids (torch.randint on the GLOBAL CPU generator, JL:505), window rules, residual rules, the device
          generator's seed / offsets -> ONE pinned sjd_iter_params blob, one async H2D
   ---- ONE hipGraph, captured once per (output-head column window, prob-buffer parity) ----
   K5     window = [last emitted | carried samples | fresh ids]                              [JL:606-701]
   fwd    backbone.forward_window over the static L rows (G1 / F1r-F3 / K1 per layer, kv_len and n_rows read from the
          device blob; output head on the vocabulary columns the rules allow)                [JL:1107]
   K2     CFG + grammar + top-k + softmax + multinomial -> p[L,V], Y[L]                      [JL:82-132]
   K4     accept scan + residual resample -> m, corrected Y, written to pinned host memory   [JL:247-376]
   ----
   host   ONE sync, reads {m, rejected, Y}, appends Y[:m], kv_len += m                       [JL:378-430]
The draft distributions of the next window are rows (m-1 ...) of this iteration's p buffer, so nothing is copied
(two p buffers alternate); fresh random drafts are implicit one-hots; KV rollback is the kv_len update (rows of
rejected drafts are overwritten by the next window).

RNG streams mirror the reference: device generator g for the multinomial / rand / residual multinomial, global CPU
generator for the fresh ids.  Round 3: the device noise is no longer materialised -- K2 / K4 compute the elements torch's exponential_ /
rand WOULD have written for g's seed and offset (Philox4x32-10 in ATen's layout, csrc/sjd_philox.h; bit-identical to torch,
tests/test_gpu_philox.py) and the host advances g's offset by what torch would have consumed; the residual draw consumes its part of the
stream only when a rejection happened, as in the reference.  Observers (the parity tests' hook) still get the three tensors, drawn by
torch from the same generator state -- that is what the teacher-forced oracle replays consume.  `noise_device="cpu"` (replay of the
reference's CPU runs) keeps the tensor path: the noise is drawn on the host and uploaded.

When the grammar cannot name the residual rules before the launch (fast_residual_rules -> None: Anole's context-dependent processors, the
first rows after <start>, windows that hold an end token) the iteration is launched in two stages instead -- {K5, forward}, residual
rules computed on the host meanwhile and uploaded behind it on the same stream, {K2, K4}.
"""
import contextlib
import ctypes
import gc
import os
import random
import time
import warnings
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import _lib as L
from . import ops
from .grammar import spatial_fresh_tokens


# The iteration's one wait: a stream synchronize (default), or SJD_MIRROR_SPIN=1: the host spins on the sequence word K4 publishes behind the
# mirrored state (sjd_host_wait_u64, no HIP call).  Measured equal within the box noise (3.584 / 3.569 against 3.586 ms/step): the runtime's
# stream wait already polls.  The synchronize also surfaces a GPU fault as an error instead of a timeout, so it stays the default.
_MIRROR_STREAM_WAIT = os.environ.get("SJD_MIRROR_SPIN", "0") != "1"
_TWO_STAGE = os.environ.get("SJD_TWO_STAGE", "0") == "1"            # A/B: always launch {K5, forward} and {K2, K4} as two graphs (round 2's shape)


@dataclass
class SJDConfig:
    jacobi_loop_interval_l: int = 1
    jacobi_loop_interval_r: int = (768 // 16) ** 2 + 768 // 16
    max_num_new_tokens: int = 16
    guidance_scale: float = 3.0
    seed: Optional[int] = 42
    do_cfg: bool = True
    prefix_token_sampler_scheme: str = "speculative_jacobi"
    multi_token_init_scheme: str = "random"   # 'repeat_horizon' / 'sample_horizon': PARITY UNPINNED -- the reference raises IndexError at
    #                                           JL:577 for both, so no reference run exists; implemented from the evident intent
    #                                           (spatial init).  Differences a fixed upstream might show: 'sample_horizon' draws its
    #                                           top-1 re-sample with torch.multinomial(generator=g) (JL:495), advancing the device generator --
    #                                           here the mode is a K2 by-product and nothing is drawn; the reference clamps the left-neighbour
    #                                           index to the last KNOWN token (JL:574) -- here drafts chain through the fresh ones.  Grammars
    #                                           without grid() (LlamaGen, the 3d-processor Anole path) fall back to 'random' with a warning.
    img_vocab_lo: int = 4
    img_vocab_n: int = 8192
    max_length: int = 1 << 30
    eos_token_ids: tuple = ()
    do_sample: bool = True               # False: GenerationConfig(do_sample=False) -- sampling_logits2tokens takes the argmax of the processed scores
    #                                      and draws nothing (JL:127-129); the verify step's uniform and residual draws stay
    noise_device: Optional[str] = None   # None: the engine's device (what the reference does on a GPU);  "cpu": draw the noise
    #                                      from a CPU generator and upload it -- replays the reference's CPU run bit-exactly


@dataclass
class WindowSpec:
    """What differs between model families at the `_sample` boundary."""
    first_tokens: torch.Tensor            # [B_cfg, P0] ids fed by the first (prefill) iteration
    first_positions: torch.Tensor         # [B_cfg, P0]
    key_start: torch.Tensor               # int32 [B_cfg]: first visible cache row per batch row
    pos_offset: torch.Tensor              # int64 [B_cfg]: RoPE position = cache row + pos_offset
    kv_base: int = 0                      # cache rows valid before `_sample` starts (LlamaGen: cond tokens)


@dataclass
class DecodeStats:
    nfe: int = 0
    tokens: int = 0
    seconds: float = 0.0
    wall_seconds: float = 0.0
    timed_nfe: int = 0
    timed_region_reached: bool = True   # False: bench mode, but the decode ended before the timed region opened (stats cover the whole decode)
    kv_len: int = 0                # KV length at the end of the timed region (whole decode when none)
    kv_len_start: int = 0          # ... at its start
    total_tokens: int = 0          # whole decode (lead-in + warm-up + timed + continuation)
    total_seconds: float = 0.0
    timed_host_seconds: float = 0.0    # host_seconds / sync_seconds at the end of the timed region
    timed_sync_seconds: float = 0.0
    host_seconds: float = 0.0      # host bookkeeping + RNG launches before the window step is enqueued
    sync_seconds: float = 0.0      # time blocked in the per-iteration state read-back
    matched: List[int] = field(default_factory=list)


def set_seed(seed):
    """reference set_seed (JL:36-45)"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


_REDUCE_TIMEOUTS_SEEN = 0
_PAIR_TIMEOUTS_SEEN = 0


def check_reduce_timeouts():
    """The o / down projections of the window forward reduce their split-K planes in their own tail (sjd_skinny_gemm_reduce): their
    workgroups wait for each other, which needs the whole launch resident on the device.  On a partitioned or shared GPU a wait can be
    abandoned (bounded spin); the forward's result is then wrong, so a decode that saw one raises instead of returning tokens."""
    global _REDUCE_TIMEOUTS_SEEN, _PAIR_TIMEOUTS_SEEN
    if L._exp is None:                        # (both structures are experiments since round 6: libsjd_hip_exp.so, loaded only by their opt-in switches --
        return                                #  a process that never loaded it ran neither, and the product path must not load it)
    n_pair = ops.mlp_pair_timeouts()          # the opt-in one-launch MLP (SJD_MLP_PAIR=1) waits the same bounded way
    if n_pair > _PAIR_TIMEOUTS_SEEN:
        _PAIR_TIMEOUTS_SEEN = n_pair
        raise RuntimeError(f"{n_pair} workgroup(s) of a one-launch MLP (sjd_mlp_pair_z) gave up waiting for their activation chunk: the forward's "
                           "result is wrong; unset SJD_MLP_PAIR / model.mlp_pair to run the MLP as two launches")
    n = ops.reduce_timeouts()
    if n > _REDUCE_TIMEOUTS_SEEN:
        _REDUCE_TIMEOUTS_SEEN = n
        raise RuntimeError(f"{n} workgroup(s) of a reducing projection launch gave up waiting for their slice (the GPU does not hold the whole "
                           "launch at once: partitioned / shared device?); set SJD_REDUCE_FUSED=0 to run the projections with a separate F1r stage")


@contextlib.contextmanager
def capture_graph(g):
    """`with torch.cuda.graph(g, capture_error_mode="thread_local")` with the cyclic garbage collector held off for the length of the capture.
    A collection that happens to start while the stream is capturing runs the destructors of whatever garbage the process has piled up -- an
    earlier engine's hipGraphs, events, pinned blobs -- and a runtime call from such a destructor (graph / event destruction, a pinned free)
    in the middle of a capture aborts the process (seen once in the full GPU suite, round 4: `Fatal Python error: Aborted ... Garbage-collecting`
    under `_launch_window`).  The collector is only HELD OFF: a `gc.collect()` in front of every capture cost 27 ms each -- 0.13 s of a 3.1 s image
    (bench.py whole_image 3.05 -> 3.18 ms per step) -- and buys nothing, the garbage can wait.  Other host threads may still use the GPU
    (thread-local capture mode)."""
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            yield
    finally:
        if was_enabled:
            gc.enable()


def warn_no_grid(cfg, grammar):
    """a spatial init scheme on a grammar that has no image geometry (LlamaGen, the 3d-processor Anole path) degrades to 'random'"""
    from .grammar import _Grammar
    if cfg.multi_token_init_scheme != "random" and type(grammar).grid is _Grammar.grid:
        warnings.warn(f"multi_token_init_scheme={cfg.multi_token_init_scheme!r}: {type(grammar).__name__} has no image grid (img_width); "
                      "the fresh drafts stay uniformly random", stacklevel=3)


class SJDEngine:
    def __init__(self, backbone, vocab_size, device, max_window=16, n_batch=2, use_graph=False, narrow_head=True, head_partials=True):
        L.load()                                   # fail loudly if the HIP extension is missing
        if max_window > L.MAX_WINDOW:
            raise ValueError(f"max_window {max_window} > {L.MAX_WINDOW}")
        self.backbone, self.V, self.device, self.Lmax, self.B = backbone, int(vocab_size), torch.device(device), max_window, n_batch
        dev = self.device
        self.params = ops.DeviceBlob(L.IterParams, dev)
        self.state = ops.DeviceBlob(L.State, dev)
        self.probs = torch.zeros(2, self.Lmax, self.V, dtype=torch.float32, device=dev)
        # per probs buffer and row: the column window outside of which the row is known to hold zeros (K2 then skips that zero fill); the
        # dense-logits K2 of the prefill iteration does not keep it, so the state of a buffer it wrote is reset to "unknown" (-1)
        self.zero_state = torch.full((2, self.Lmax, 2), -1, dtype=torch.int32, device=dev)
        self.noise = self.rs = self.noise2 = None   # [L,V], [L,V], [1,V] fp32: allocated only for observers / host-drawn noise (_noise_tensors)
        self._philox = True                         # this decode's K2 / K4 generate their noise (False: they read the tensors)
        self.scratch = torch.empty(self.V, dtype=torch.float32, device=dev)
        self.input_ids = torch.zeros(self.B, self.Lmax, dtype=torch.int64, device=dev)
        self.arange = torch.arange(self.Lmax, device=dev)
        self.positions = torch.zeros(self.B, self.Lmax, dtype=torch.int64, device=dev)      # window position ids, written by K5
        self.tokens_ptr = self.state.field_ptr("tokens")
        self.amax_ptr = self.state.field_ptr("amax")
        self.key_start = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.pos_offset = torch.zeros(self.B, dtype=torch.int64, device=dev)
        off = L.IterParams.kv_len.offset
        self.kv_len_dev = self.params.dev[off:off + 4].view(torch.int32)      # device view of params->kv_len
        self.hook = None                            # test hook: called with per-iteration device tensors
        self.use_graph = use_graph                  # capture the window step (K5 -> forward -> K2 -> K4) in hipGraphs
        self.narrow_head = narrow_head              # evaluate the output head only for the columns the grammar allows
        # K2 reads the output head's split-K partials (no fp32 logits tensor) when the backbone can produce them
        self.head_partials = bool(head_partials) and getattr(backbone, "supports_head_partials", False)
        self._dbg = None                            # [2, L, V] logits as K2 derived them; allocated only for observers (hook)
        self._guidance = 3.0
        self._rule_bytes, self._rule_keep, self._cols_cache = {}, [], {}
        self._seq = 0
        self.reset_graphs()

    def reset_graphs(self):
        """Call after the backbone's cache / weights were re-allocated."""
        self._graphs, self._graph_logits, self._eager_runs = {}, {}, {}
        self.__dict__.pop("_graph_sig", None)

    def _check_graph_buffers(self):
        """The captured graphs hold raw addresses of the K1 workspace, of the KV cache (K and V) and of the packed weights: if any was
        re-allocated since the capture (HipWindowAttention.ws_version; the backbone's buffers_version, bumped by setup_cache /
        enable_fused; the cache tensors' data_ptr as a belt for backbones without the counter) the graphs are dropped and captured again
        instead of replaying into freed memory.  The G1 launch shapes are baked into the packed weights, so changing G1_CFG means
        calling enable_fused again -- which bumps the counter."""
        attn = getattr(self.backbone, "attn", None)
        cache = getattr(self.backbone, "cache", None)
        sig = (getattr(attn, "ws_version", 0), getattr(self.backbone, "buffers_version", 0),
               cache.k.data_ptr() if cache is not None else 0, cache.v.data_ptr() if cache is not None else 0)
        if sig != getattr(self, "_graph_sig", sig):
            self.reset_graphs()
        self._graph_sig = sig

    # ------------------------------------------------------------------------------------------------
    def _noise_tensors(self):
        if self.noise is None:
            dev = self.device
            self.noise = torch.ones(self.Lmax, self.V, dtype=torch.float32, device=dev)
            self.rs = torch.zeros(self.Lmax, self.V, dtype=torch.float32, device=dev)
            self.noise2 = torch.ones(1, self.V, dtype=torch.float32, device=dev)
        return self.noise, self.rs, self.noise2

    def _fill_params(self, n, kv_len, use_cfg, scheme, fresh, rules, resid, head_only=False, philox=None):
        """philox = (blocks, seed, off0, off1, off2): K2 / K4 generate the noise of a device generator at that state (None: tensors).
        head_only: upload everything in FRONT of resid_rules only -- the residual rules follow on the same stream once the host has
        computed them (_upload_resid), as a disjoint byte range of the blob."""
        p = self.params.view
        p.n_rows, p.kv_len, p.use_cfg, p.scheme, p.n_fresh = n, kv_len, int(use_cfg), scheme, len(fresh)
        self._seq = (self._seq % 0x7FFFFFF0) + 1          # K4 publishes it behind the mirrored state: what the host's spin waits for
        p.iter_seq = self._seq
        if philox is None:
            p.philox_blocks = 0
        else:
            p.philox_blocks, p.philox_seed = philox[0], philox[1]
            p.philox_offset[0], p.philox_offset[1], p.philox_offset[2] = philox[2], philox[3], philox[4]
        if fresh:
            p.fresh_tok[:len(fresh)] = fresh
        self._write_rules(L.IterParams.rules.offset, rules)
        if resid:
            self._write_rules(L.IterParams.resid_rules.offset, resid)
        self.params.upload(L.IterParams.resid_rules.offset if head_only else None)

    def _write_rules(self, offset, rules):
        """rules (interned sjd_row_rule structs, ops.make_rule) -> the blob, as ONE block copy: the packed bytes of a rule sequence are
        cached by the structs' identities (an image body asks for the same two or three sequences over and over)."""
        key = tuple(map(id, rules))
        blk = self._rule_bytes.get(key)
        if blk is None:
            blk = b"".join(bytes(r) for r in rules)
            if len(self._rule_bytes) < 4096:
                self._rule_bytes[key] = blk
                self._rule_keep.append(list(rules))            # the ids stay valid while the structs are alive
        ctypes.memmove(self.params.host.data_ptr() + offset, blk, len(blk))

    def _upload_resid(self, resid):
        """two-stage iterations: the residual rules (read by K4 only) were computed while the forward runs; their slice of the blob goes
        up on the SAME stream, behind the forward and in front of K2 / K4"""
        off = L.IterParams.resid_rules.offset
        self._write_rules(off, resid)
        L.check(L.load().sjd_upload_async(self.params.dev.data_ptr() + off, self.params.host.data_ptr() + off, self.params.nbytes - off,
                                          ops._stream()), "sjd_upload_async")

    def _forward_body(self, cols=None):
        """Shape-static launch sequence, part 1: K5 + transformer forward (every dynamic scalar is read from device blobs).
        cols: vocabulary column window of the output head (None = all columns)."""
        ops.reguess(self.params, self.state, self.input_ids, pos_offset=self.pos_offset, positions_out=self.positions)      # ids + position ids
        positions = self.positions
        if self.head_partials:
            return self.backbone.forward_window(self.input_ids, positions, -1, self.key_start, cols=cols, head_partials=True)
        if cols is None:
            return self.backbone.forward_window(self.input_ids, positions, -1, self.key_start)
        return self.backbone.forward_window(self.input_ids, positions, -1, self.key_start, cols=cols)

    def _calibrate_fp8(self, tokens, positions, key_start):
        """an fp8 KV cache whose scales nobody set: calibrate them on THIS prompt, once per backbone (static calibration; round 6)"""
        bb = self.backbone
        cache, attn = getattr(bb, "cache", None), getattr(bb, "attn", None)
        if (cache is None or cache.k.dtype != ops.FP8 or attn is None or getattr(attn, "layer_scales", 1) is not None
                or getattr(attn, "kv_scale", (1.0, 1.0)) != (1.0, 1.0) or not hasattr(bb, "calibrate_kv_scales")
                or os.environ.get("SJD_FP8_CALIBRATE", "1") == "0"):
            return
        bb.calibrate_kv_scales(tokens, positions, key_start)

    def _k2_out(self):
        """(tokens_out, amax_out) of K2: the draw goes into state.tokens and the mode of p into state.amax -- or, greedy (do_sample=False), the mode
        into state.tokens, which is what K4 verifies and the host appends (the draw lands in state.amax and is ignored).
        One documented deviation: the reference takes torch.argmax of the processed SCORES (JL:128), K2 the lowest-index mode of the
        softmaxed p.  They differ only when two maximal scores are distinct floats whose exponentials round to the same fp32 (exp(z - zmax) == 1
        for z < zmax): the gap must be below 2^-25, i.e. below an ulp of z unless |zmax| < 0.5 -- near-tied logits of magnitude below 0.5 at the
        top of a row.  tests/test_oracle_golden.py::test_greedy_mode_of_p_vs_argmax_of_scores pins exactly that boundary."""
        return (self.amax_ptr, self.tokens_ptr) if getattr(self, "_greedy", False) else (self.tokens_ptr, self.amax_ptr)

    def _sample_body(self, cur, logits, cols=None):
        """part 2: K2 + K4.  In-kernel noise (self._philox): the kernels are handed NULL instead of the noise tensors."""
        noise, rs, noise2 = (None, None, None) if self._philox else (self.noise, self.rs, self.noise2[0])
        if isinstance(logits, ops.HeadOut):
            dbg = None
            if self.hook is not None:
                if self._dbg is None:
                    self._dbg = torch.zeros(2, self.Lmax, self.V, dtype=torch.float32, device=self.device)
                dbg = self._dbg
                dbg.zero_()
            tok_out, amax_out = self._k2_out()
            ops.logits_to_probs_sample_part(logits, self._guidance, self.params, noise, self.probs[cur], tok_out, dbg=dbg,
                                            amax_out_ptr=amax_out, zero_state=self.zero_state[cur])
        else:
            lu = logits[1] if self.B > 1 else None
            tok_out, amax_out = self._k2_out()
            ops.logits_to_probs_sample(logits[0], lu, self._guidance, self.params, noise, self.probs[cur], tok_out,
                                       col0=cols[0] if cols else 0, amax_out_ptr=amax_out)
            self.zero_state[cur].fill_(-1)            # (the dense-logits K2 does not keep the rows' zero state)
        ops.verify_accept(self.params, self.state, self.probs[cur], self.probs[1 - cur], rs, noise2, self.scratch, mirror=True)

    def logit_columns(self, rules):
        """Vocabulary window [lo, hi) (32-aligned) that holds every id the non-forced rows of this iteration may emit, or None when
        it is not at most half of the vocabulary (text rows, unrestricted rules).  The output head is then evaluated for these
        columns only: the ids outside are masked by the grammar before the softmax (reference logit_processor_3dim.py:125-129,
        jacobi_iteration_emu3.py:44-128), so not computing them changes nothing downstream."""
        if not self.narrow_head:
            return None
        key = tuple(map(id, rules))
        if key in self._cols_cache:
            return self._cols_cache[key]
        cols = SJDEngine._logit_columns(self, rules)
        if len(self._cols_cache) < 4096:
            self._cols_cache[key] = cols
            self._rule_keep.append(list(rules))
        return cols

    def _logit_columns(self, rules):
        lo, hi = self.V, 0
        for r in rules:
            if r.forced >= 0:
                continue
            if r.n_ranges == 0:
                return None
            lo = min([lo] + [r.lo[i] for i in range(r.n_ranges)])
            hi = max([hi] + [r.hi[i] for i in range(r.n_ranges)])
        if hi <= lo:
            return None
        lo, hi = (lo // 32) * 32, min(self.V, ((hi + 31) // 32) * 32)
        return (lo, hi) if 2 * (hi - lo) <= self.V else None

    def _k1_regime(self):
        """the attention backend's launch form for this iteration (part of every graph key: the two forms are different kernels)"""
        return getattr(getattr(self.backbone, "attn", None), "regime", None)

    def _launch_forward(self, cols=None):
        """part 1 (K5 + transformer forward): eager, or a hipGraph captured once per output-head column window.  Returns the logits
        tensor part 2 will read (static across replays of the same graph)."""
        if not self.use_graph:
            return self._forward_body(cols)
        self._check_graph_buffers()
        fkey = ("fwd", cols, self._k1_regime())
        if fkey not in self._graphs:
            if self._eager_runs.get(fkey, 0) < 1:   # one eager run warms up allocations / hipBLASLt before capture
                self._eager_runs[fkey] = 1
                return self._forward_body(cols)
            g = torch.cuda.CUDAGraph()
            with capture_graph(g):
                self._graph_logits[fkey] = self._forward_body(cols)
            self._graphs[fkey] = g
        self._graphs[fkey].replay()
        return self._graph_logits[fkey]

    def _launch_sample(self, cur, logits, cols=None):
        """part 2 (K2 + K4) of a two-stage iteration; a hipGraph per (prob-buffer parity, column window), captured the second time that
        combination runs on graph-owned logits."""
        # (the K1 regime is part of the key: every regime's forward graph owns ITS static logits / head partials, and a sample graph bakes in
        #  the buffer it was captured on -- ADVICE r4: without it the graph of one regime replayed on the other regime's stale buffer.  The
        #  captured buffer's identity is kept beside the graph as a belt: a mismatch drops the graph and captures again.)
        regime = self._k1_regime()
        key = ("smp", cur, self._guidance, cols, self.hook is not None, self._philox, regime, getattr(self, "_greedy", False))
        if not self.use_graph or ("fwd", cols, regime) not in self._graphs:
            self._sample_body(cur, logits, cols)
            return
        if key in self._graphs and self._graph_logits.get(key) is not logits:
            del self._graphs[key]
        if key not in self._graphs:
            self._sample_body(cur, logits, cols)     # eager warm-up of this parity, captured below for the next use
            g = torch.cuda.CUDAGraph()
            with capture_graph(g):
                self._sample_body(cur, logits, cols)
            self._graphs[key] = g
            self._graph_logits[key] = logits
            return
        self._graphs[key].replay()

    def _launch_window(self, cur, cols=None):
        """the whole iteration -- K5, forward, K2, K4 -- as ONE hipGraph per (column window, prob-buffer parity): everything it reads
        is in the blob that was uploaded in front of it.  Returns the logits / head partials K2 read (graph-owned, static)."""
        if not self.use_graph:
            logits = self._forward_body(cols)
            self._sample_body(cur, logits, cols)
            return logits
        self._check_graph_buffers()
        key = ("win", cols, cur, self._guidance, self.hook is not None, self._philox, self._k1_regime(), getattr(self, "_greedy", False))
        if key not in self._graphs:
            if self._eager_runs.get(key, 0) < 1:      # one eager run warms up allocations / hipBLASLt before capture
                self._eager_runs[key] = 1
                logits = self._forward_body(cols)
                self._sample_body(cur, logits, cols)
                return logits
            g = torch.cuda.CUDAGraph()
            with capture_graph(g):
                logits = self._forward_body(cols)
                self._sample_body(cur, logits, cols)
            self._graph_logits[key] = logits
            self._graphs[key] = g
        self._graphs[key].replay()
        return self._graph_logits[key]

    def captured_column_windows(self):
        """output-head column windows for which a window graph (one- or two-stage) has been captured"""
        return sorted({k[1] for k in self._graphs if isinstance(k, tuple) and k[0] in ("fwd", "win")}, key=lambda c: (c is None, c))

    def last_head_output(self, cols):
        """the logits / head partials the most recently captured window graph of `cols` hands to K2 (bench.py's scheduler leg)"""
        for k in reversed(list(self._graph_logits)):
            if k[0] in ("fwd", "win") and k[1] == cols:
                return self._graph_logits[k]
        return None

    @torch.no_grad()
    def decode(self, prompt: List[int], spec: WindowSpec, grammar, cfg: SJDConfig, warmup_iters=0, timed_iters=None,
               on_timed_start=None, on_timed_end=None, lead_in_kv=None, continue_after=False, iter_log=None):
        """prompt: accepted ids handed to `_sample` (the context the grammar sees).  Returns (sequence, DecodeStats).
        bench mode: an untimed lead-in runs until kv_len >= lead_in_kv (None: no lead-in), then `warmup_iters` untimed iterations,
        then EXACTLY `timed_iters` iterations bracketed by on_timed_start/on_timed_end (barrier + synchronize live in the
        callbacks); stats.{seconds, tokens, timed_nfe, kv_len, kv_len_start, host_seconds, sync_seconds} cover that region.  The
        decode stops after it unless continue_after (then it runs on to EOS / max_length and stats.{nfe, total_tokens,
        total_seconds} describe the whole decode).  iter_log: list receiving (kv_len before the iteration, rows, accepted,
        host clock after the iteration's sync) per iteration."""
        if cfg.multi_token_init_scheme not in ("random", "repeat_horizon", "sample_horizon"):
            # JL:554-560, 592: anything else (incl. the 'vertical' variants) asserts in the reference
            raise ValueError(f"multi_token_init_scheme should be 'random', 'repeat_horizon' or 'sample_horizon', but got {cfg.multi_token_init_scheme}")
        if cfg.prefix_token_sampler_scheme not in ("speculative_jacobi", "jacobi"):
            raise ValueError(f"prefix_token_sampler_scheme: {cfg.prefix_token_sampler_scheme}")   # JL:1048
        if cfg.max_num_new_tokens > self.Lmax:
            raise ValueError("max_num_new_tokens exceeds the engine's max_window")
        warn_no_grid(cfg, grammar)
        dev, B = self.device, self.B
        scheme = 0 if cfg.prefix_token_sampler_scheme == "speculative_jacobi" else 1
        do_cfg = cfg.do_cfg and (cfg.guidance_scale != 1)                          # JL:1005
        X = [int(t) for t in prompt]
        P = len(X)
        gen = None
        if cfg.seed is not None:                                                   # JL:1021-1023
            set_seed(cfg.seed)
            gen = torch.Generator(cfg.noise_device or dev).manual_seed(cfg.seed)
        host_noise = cfg.noise_device is not None and torch.device(cfg.noise_device).type == "cpu"
        philox = self._philox = not host_noise
        if philox:
            if gen is None:                        # no seed: the device's default generator, which exponential_(generator=None) consumes
                gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
            ph_blocks, ph_seed, ph_off = ops.philox_max_blocks(gen.device), int(gen.initial_seed()), int(gen.get_offset())
            ph_row = ops.philox_step(self.V, ph_blocks)                            # offset step of a [1, V] draw
        l_abs, r_abs = P + cfg.jacobi_loop_interval_l, P + cfg.jacobi_loop_interval_r   # JL:1025
        W = cfg.max_num_new_tokens
        grammar.start(X)
        self.key_start.copy_(spec.key_start.to(device=dev, dtype=torch.int32))
        self.pos_offset.copy_(spec.pos_offset.to(device=dev, dtype=torch.int64))
        self._guidance = float(cfg.guidance_scale)
        self._greedy = greedy = not getattr(cfg, "do_sample", True)
        attn = getattr(self.backbone, "attn", None)
        st = self.state.view
        stats = DecodeStats()
        n, kv_len, first, cur_len, cur, n_prev, m_prev = 1, spec.kv_base, True, P, 0, 1, 1
        carried: List[int] = []
        carried_amax: List[int] = []
        last_amax = None
        t0 = time.perf_counter()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        timed_tok0, timed_nfe0 = 0, 0
        warm_nfe0, timing, timed_done = None, False, False
        t_decode0 = time.perf_counter()
        finished = False
        while not finished:
            if timed_iters is not None and warm_nfe0 is None and (lead_in_kv is None or (not first and kv_len >= lead_in_kv)):
                warm_nfe0 = stats.nfe                                            # the lead-in is over: warm-up iterations start here
            if timed_iters is not None and warm_nfe0 is not None and not timing and not timed_done and stats.nfe == warm_nfe0 + warmup_iters:
                if on_timed_start is not None:
                    on_timed_start()
                timing = True
                timed_tok0, timed_nfe0 = len(X), stats.nfe
                stats.kv_len_start = kv_len
                stats.host_seconds = stats.sync_seconds = 0.0
                t0 = time.perf_counter()
                ev0.record()
            # ---------------- host: integer bookkeeping only ----------------
            t_host0 = time.perf_counter()
            win, resid = None, []
            if first:
                n_rows = 1
                torch.randint(0, cfg.img_vocab_n, (1, 0))
                rules, fresh = grammar.window_rules(1), []
            else:
                n_rows = n
                a = max(0, min(n_prev - m_prev, n - 1))                              # JL:633-639, 657-662
                fr = torch.randint(0, cfg.img_vocab_n, (1, n - 1 - a)).tolist()[0]   # GLOBAL CPU generator (JL:505)
                fresh = [cfg.img_vocab_lo + t for t in fr]                           # img_vocab[rand] (JL:509)
                if cfg.multi_token_init_scheme != "random":                          # spatial init (JL:516-594): copy / re-draw from the left
                    fresh = spatial_fresh_tokens(cfg.multi_token_init_scheme, fresh, len(X) + a, carried[a - 1] if a else X[-1],
                                                 carried_amax[a - 1] if a else last_amax, grammar.grid())
                rules = grammar.window_rules(n_rows)
                if scheme == 0 and n_rows > 1:
                    # the window's ids are all known here (carried ones came with the last read-back): if the grammar can name the
                    # residual rules without a replay, the whole iteration goes out as one graph; otherwise they are computed below,
                    # while the forward runs (K4 is their only reader)
                    win = [X[-1]] + carried[:a] + fresh
                    resid = None if _TWO_STAGE else grammar.fast_residual_rules(win, rules)
            two_stage = resid is None
            use_cfg = do_cfg and not grammar.force_no_cfg()                          # JL:1086-1096
            draws_rs = n_rows > 1 and scheme == 0
            ph = None
            if philox:                                  # offsets of g before the multinomial, the rand and the residual multinomial
                ph_step = ops.philox_step(n_rows * self.V, ph_blocks)
                k2_step = 0 if greedy else ph_step          # greedy: sampling_logits2tokens draws nothing (JL:127-129)
                ph = (ph_blocks, ph_seed, ph_off, ph_off + k2_step, ph_off + k2_step + ph_step)
            self._fill_params(n_rows, kv_len, use_cfg, scheme, fresh, rules, resid or [], head_only=two_stage, philox=ph)
            # ---------------- noise tensors: only for host-drawn noise (the reference's CPU runs) and for observers ----------------
            e1, g_state = None, None
            if host_noise or self.hook is not None:
                self._noise_tensors()
                e1 = self.noise[:n_rows]
                if host_noise:                                                       # parity mode: CPU stream of the reference
                    if greedy:
                        e1.fill_(1.0)
                    else:
                        e1.copy_(torch.empty(n_rows, self.V).exponential_(generator=gen))
                    if draws_rs:
                        self.rs[:n_rows].copy_(torch.rand((1, n_rows, self.V), generator=gen)[0])
                        g_state = gen.get_state()
                        self.noise2.copy_(torch.empty(1, self.V).exponential_(generator=gen))
                else:                           # what the kernels generate, drawn by torch from the same generator state (observers)
                    gen.set_offset(ph_off)
                    if greedy:
                        e1.fill_(1.0)
                    else:
                        e1.exponential_(generator=gen)                               # == torch.multinomial (JL:118)
                    if draws_rs:
                        self.rs[:n_rows].uniform_(0.0, 1.0, generator=gen)           # torch.rand([1,n,V]) (JL:260)
                        self.noise2.exponential_(generator=gen)                      # residual multinomial (JL:237)
            stats.host_seconds += time.perf_counter() - t_host0
            # ---------------- device work ----------------
            cols = None
            if first:
                if attn is not None and hasattr(attn, "params"):
                    attn.params = None                                               # prefill: kv_len passed by value
                tokens, positions = spec.first_tokens.to(dev), spec.first_positions.to(dev)
                self._calibrate_fp8(tokens, positions, self.key_start)
                logits = self.backbone.forward_window(tokens, positions, kv_len, self.key_start)
                lc = logits[0, -1:, :]
                lu = logits[1, -1:, :] if B > 1 else None
                win_len = tokens.shape[1]
                tok_out, amax_out = self._k2_out()
                ops.logits_to_probs_sample(lc, lu, self._guidance, self.params, None if philox else self.noise, self.probs[cur], tok_out,
                                           amax_out_ptr=amax_out)
                self.zero_state[cur].fill_(-1)
                ops.verify_accept(self.params, self.state, self.probs[cur], self.probs[1 - cur], None if philox else self.rs,
                                  None if philox else self.noise2[0], self.scratch, mirror=True)
                if attn is not None and hasattr(attn, "params"):
                    attn.params = self.params                                        # windows: kv_len / n_rows from the blob
            else:
                cols = self.logit_columns(rules)
                if attn is not None and hasattr(attn, "choose_regime"):     # K1: column split while the context is short, key split + combine after
                    ck = self.backbone.cache.k          # [layers, B, H_kv, S, D]
                    attn.choose_regime(kv_len + n_rows, ck.dtype, shape=(B, self.Lmax, getattr(self.backbone, "n_heads", ck.shape[2]), ck.shape[2], ck.shape[4]))
                if two_stage:
                    logits = self._launch_forward(cols)
                    t_host0 = time.perf_counter()
                    resid = grammar.residual_rules(win)
                    self._upload_resid(resid)
                    stats.host_seconds += time.perf_counter() - t_host0
                    self._launch_sample(cur, logits, cols)
                else:
                    logits = self._launch_window(cur, cols)
                win_len = n_rows
                if isinstance(logits, ops.HeadOut):                  # observers get the logits exactly as K2 derived them
                    lc, lu = (self._dbg[0, :n_rows], self._dbg[1, :n_rows] if B > 1 else None) if self.hook is not None else (None, None)
                else:
                    lc = logits[0, :n_rows]
                    lu = logits[1, :n_rows] if B > 1 else None
                if cols is not None and self.hook is not None and not isinstance(logits, ops.HeadOut):       # full-width rows (zeros elsewhere)
                    full = torch.zeros(logits.shape[0], n_rows, self.V, dtype=logits.dtype, device=dev)
                    full[:, :, cols[0]:cols[1]] = logits[:, :n_rows]
                    lc, lu = full[0], (full[1] if B > 1 else None)
            if self.hook is not None:
                self.hook(dict(first=first, n_rows=n_rows, logits_c=lc, logits_u=lu, use_cfg=use_cfg, rules=rules,
                               resid=resid, noise=e1, rs=self.rs[:n_rows], noise2=self.noise2[0], probs=self.probs[cur],
                               prev_probs=self.probs[1 - cur], ctx=list(X), scheme=scheme, two_stage=two_stage))
            # ---------------- the single sync of the iteration ----------------
            t_sync0 = time.perf_counter()
            self.state.wait_mirror(None if _MIRROR_STREAM_WAIT else self._seq)     # K4 wrote the state into pinned host memory and published iter_seq: no D2H copy, no HIP call
            stats.sync_seconds += time.perf_counter() - t_sync0
            m_dev, rejected = int(st.m), bool(st.rejected)
            if int(st.rejected) > 1:
                raise RuntimeError("SJD verify: the residual distribution max(p - q, 0) is empty under the residual grammar rule "
                                   "(the reference's torch.multinomial raises on the NaN probabilities at JL:237)")
            if g_state is not None and not rejected:
                gen.set_state(g_state)                 # host-drawn noise: the residual draw was speculative, rewind it
            if philox:                                 # what torch would have consumed: multinomial [+ rand [+ residual multinomial]]
                ph_off += k2_step + (ph_step if draws_rs else 0) + (ph_row if (draws_rs and rejected) else 0)
                if self.hook is not None:
                    gen.set_offset(ph_off)
            Y = st.tokens[:n_rows]
            A = Y if greedy else st.amax[:n_rows]              # modes of this iteration's target rows (K2 by-product; greedy: the tokens themselves)
            if n_rows <= 1:
                m = win_len                      # is_prefilling_phase short-circuit (JL:344-350)
                emitted, carried, carried_amax, last_amax = [Y[0]], [], [], A[0]
            else:
                m = m_dev
                emitted, carried, carried_amax, last_amax = Y[:m], Y[m:], A[m:], A[m - 1]
            stats.matched.append(m)
            if iter_log is not None:
                iter_log.append((kv_len, n_rows, m, time.perf_counter()))
            # ---------------- next window length, append, rollback ----------------
            n = min(W, r_abs - cur_len) if (l_abs <= cur_len < r_abs) else 1       # JL:1142-1144 (old cur_len)
            X.extend(emitted)
            grammar.push(emitted)
            kv_len += m
            n_prev, m_prev = n_rows, (1 if n_rows <= 1 else m)
            cur = 1 - cur
            first = False
            stats.nfe += 1
            if X[-1] in cfg.eos_token_ids or len(X) >= cfg.max_length:             # JL:1200-1201
                finished = True
            cur_len = len(X)
            if timing and stats.nfe == timed_nfe0 + timed_iters:
                self._close_timed(stats, ev0, ev1, t0, on_timed_end, len(X) - timed_tok0, stats.nfe - timed_nfe0, kv_len)
                timing, timed_done = False, True
                if not continue_after:
                    break
        if not timed_done:          # plain decode, or EOS inside the timed region: the region is what ran since its start
            # a bench-mode decode that ended (EOS / max_length) before its timed region OPENED never called on_timed_start: do not call
            # its partner either (both are barriers in bench.py -- an unmatched one hangs the other ranks), and say so in the stats
            stats.timed_region_reached = timing or timed_iters is None
            self._close_timed(stats, ev0, ev1, t0, on_timed_end if timing else None,
                              len(X) - (timed_tok0 if timing else P), stats.nfe - (timed_nfe0 if timing else 0), kv_len)
        torch.cuda.synchronize()
        check_reduce_timeouts()
        if philox:
            gen.set_offset(ph_off)                     # leave the generator where the reference's draws would have left it
        stats.total_tokens, stats.total_seconds = len(X) - P, time.perf_counter() - t_decode0
        return X, stats

    @staticmethod
    def _close_timed(stats, ev0, ev1, t0, on_timed_end, tokens, nfe, kv_len):
        ev1.record()
        torch.cuda.synchronize()
        if on_timed_end is not None:
            on_timed_end()
        stats.seconds = ev0.elapsed_time(ev1) / 1000.0
        stats.wall_seconds = time.perf_counter() - t0
        stats.tokens, stats.timed_nfe, stats.kv_len = tokens, nfe, kv_len
        stats.timed_host_seconds, stats.timed_sync_seconds = stats.host_seconds, stats.sync_seconds      # frozen: the decode may continue
