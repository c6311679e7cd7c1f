"""Several prompts per GPU in ONE window forward (the reference decodes one prompt per process).

A draft-window forward is a pure weight stream: 13 GB of weights for 32 activation rows.  Two prompts (2 x B_cfg x L = 64 rows)
cost the same stream, so decoding them together almost doubles the accepted tokens per second of a GPU; up to eight prompts (256 rows; the rate is flat beyond four: DESIGN.md 10b;
G1's sub-tiled kernel) fit in one forward: 963 / 1292 tokens/s per MI355X for two / four Lumina 768px prompts against 561 for one.  Every prompt ("slot")
keeps exactly the state machine of `SJDEngine.decode` -- its own window, accept length, KV length, grammar, device generator
and CPU generator for the fresh ids -- so each slot takes the decisions its solo run would take on the same logits; only the
transformer forward is shared:

   host    per slot: n, fresh ids, rules, residual rules, its generator's seed / offsets -> its blob of a contiguous sjd_iter_params
           ARRAY (batch_rows = B_cfg: the kernels K1 / K3 / F2 pick the blob of a batch row, i.e. every prompt has its own kv_len and
           n_rows), one H2D for all
   graph   K5 per slot (window ids of that slot's batch rows), ONE backbone.forward_window over all slots' rows with the output head as
           split-K partials over the union of the slots' column windows, K2 (PART, in-kernel noise) + K4 per slot (own probability
           buffers); ONE hipGraph when every slot's grammar names its residual rules up front (SJDEngine's two-stage launch otherwise)
   host    one stream wait (every K4 wrote its state into pinned host memory), per-slot bookkeeping

A slot that reaches its end token keeps riding along with a one-row dummy window (its tokens are ignored) until every slot is done.
"""
import time
from typing import List

import os
import torch

from . import _lib as L
from . import ops
from .engine import DecodeStats, SJDConfig, WindowSpec, warn_no_grid, capture_graph
from .grammar import spatial_fresh_tokens


class _CacheView:
    """The KV cache rows of one slot (batch rows [lo, hi)) presented as a StaticKVCache to the backbone's prefill."""

    def __init__(self, cache, lo, hi):
        self.k, self.v, self.s_max = cache.k[:, lo:hi], cache.v[:, lo:hi], cache.s_max


class _Slot:
    pass


class SJDBatchEngine:
    def __init__(self, backbone, vocab_size, device, n_prompts, max_window=16, n_batch=2, use_graph=True, narrow_head=True, head_partials=True):
        L.load()                                   # fail loudly if the HIP extension is missing
        if max_window > L.MAX_WINDOW:
            raise ValueError(f"max_window {max_window} > {L.MAX_WINDOW}")
        if n_prompts * n_batch * max_window > 256:
            raise ValueError("the window forward (G1, F1-F3) serves at most 256 rows: n_prompts * n_batch * max_window <= 256")
        rows = n_prompts * n_batch * max_window
        if rows > 128 and getattr(backbone, "_gemm", None) == "sjd" and getattr(backbone, "_packed", None):
            # 129..256 window rows run on kernel G1w (csrc/sjd_gemm_wide.h): the uncompressed packing, 2 / 3 / 4 / 6 / 8 column tiles per
            # workgroup for every projection AND the output head.  Anything else used to fall onto the library-GEMM prefill path silently
            # (slower, not bit-identical to the oracle replays) or to fail at the first forward: say so here instead.
            tiles = getattr(backbone, "G1_WIDE_TILES", (2, 3, 4, 6, 8))
            bad = [k for k, c in backbone.G1_CFG.items() if c[1] not in tiles]
            if getattr(backbone, "_packed_head", None) is not None and backbone.HEAD_CFG[1] not in tiles:
                bad.append("lm_head (HEAD_CFG)")
            if isinstance(backbone._packed[0]["qkv"], ops.PackedZ):
                raise ValueError(f"{rows} window rows per forward need the uncompressed packing: enable_fused(ops, gemm='sjd', compress=False) "
                                 "(the 12-bit stream serves up to 128 rows)")
            if bad:
                raise ValueError(f"{rows} window rows per forward run on kernel G1w with {tiles} column tiles per workgroup; set "
                                 f"model.G1_CFG = dict(model.G1_CFG_256ROW) before enable_fused -- offending launch shapes: {bad}")
        self.backbone, self.V, self.device = backbone, int(vocab_size), torch.device(device)
        self.P, self.nb, self.Lmax, self.B = n_prompts, n_batch, max_window, n_prompts * n_batch
        self.use_graph, self.narrow_head = use_graph, narrow_head
        dev = self.device
        self.params = ops.BlobArray(L.IterParams, self.P, dev)
        self.state = ops.BlobArray(L.State, self.P, dev)
        self.params.view = self.params.blobs[0].view          # what HipWindowAttention's profiling hook looks at
        self.slots = []
        # every slot's buffers at one stride: K5 / K2 / K4 of ALL slots are one launch each (round 6, sjd_slots in include/sjd_hip.h)
        self.probs_all = torch.zeros(self.P, 2, self.Lmax, self.V, dtype=torch.float32, device=dev)
        self.zero_state_all = torch.full((self.P, 2, self.Lmax, 2), -1, dtype=torch.int32, device=dev)
        self.scratch_all = torch.empty(self.P, self.V, dtype=torch.float32, device=dev)
        self.state.mirror_array()
        self.slot_launches = os.environ.get("SJD_SLOT_LAUNCHES", "1") != "0"       # 0: one K5 / K2 / K4 launch per slot (rounds 3-5; A/B aid)
        self._slots_desc = ops.slots_of(self.params, self.state, self.probs_all, self.zero_state_all, self.scratch_all, n_batch)
        for i in range(self.P):
            s = _Slot()
            s.params, s.state = self.params.blobs[i], self.state.blobs[i]
            s.params.view.batch_rows = n_batch
            s.probs = self.probs_all[i]
            s.zero_state = self.zero_state_all[i]        # see SJDEngine.zero_state
            s.noise = s.rs = s.noise2 = None        # only for observers (the parity tests' hook): K2 / K4 generate their noise
            s.scratch = self.scratch_all[i]
            s.tokens_ptr = s.state.field_ptr("tokens")
            s.amax_ptr = s.state.field_ptr("amax")
            self.slots.append(s)
        self.input_ids = torch.zeros(self.B, self.Lmax, dtype=torch.int64, device=dev)
        self.arange = torch.arange(self.Lmax, device=dev)
        self.positions = torch.zeros(self.B, self.Lmax, dtype=torch.int64, device=dev)      # window position ids, written by K5
        self.key_start = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.pos_offset = torch.zeros(self.B, dtype=torch.int64, device=dev)
        nb_params, off = L.ctypes.sizeof(L.IterParams), L.IterParams.kv_len.offset
        assert nb_params % 4 == 0
        # device view of every slot's params->kv_len, repeated for the slot's batch rows
        self.kv_len_dev = self.params.dev.view(torch.int32)[off // 4::nb_params // 4]
        self.row_slot = torch.arange(self.B, device=dev) // n_batch
        self._guidance = 3.0
        # K2 reads the output head's split-K partials (no fp32 logits tensor) when the backbone can produce them
        self.head_partials = bool(head_partials) and getattr(backbone, "supports_head_partials", False)
        self._dbg = None                       # [B, L, V] logits as K2 derived them; allocated only for observers (hook)
        self.hook = None                       # test hook: called per slot and iteration with that slot's device tensors
        self._rule_bytes, self._rule_keep, self._cols_cache = {}, [], {}
        self.reset_graphs()

    def reset_graphs(self):
        """Call after the backbone's cache / weights were re-allocated."""
        self._graphs, self._graph_logits, self._eager_runs = {}, {}, {}
        self.__dict__.pop("_graph_sig", None)

    def _check_graph_buffers(self):
        """see SJDEngine._check_graph_buffers: never replay a graph that holds the address of a re-allocated workspace / cache"""
        from .engine import SJDEngine
        SJDEngine._check_graph_buffers(self)

    # ------------------------------------------------------------------------------------------------
    def _noise_tensors(self, s):
        if s.noise is None:
            dev = self.device
            s.noise = torch.ones(self.Lmax, self.V, dtype=torch.float32, device=dev)
            s.rs = torch.zeros(self.Lmax, self.V, dtype=torch.float32, device=dev)
            s.noise2 = torch.ones(1, self.V, dtype=torch.float32, device=dev)

    def _fill(self, slot, n, kv_len, use_cfg, scheme, fresh, rules, resid):
        """the slot's blob for one iteration, incl. the state of ITS device generator: K2 / K4 generate the multinomial / rand / residual
        noise torch would have drawn from it (SJDEngine._fill_params); slot.ph_step = what one [n, V] draw consumes"""
        p = slot.params.view
        p.n_rows, p.kv_len, p.use_cfg, p.scheme, p.n_fresh = n, kv_len, int(use_cfg), scheme, len(fresh)
        slot.ph_step = ops.philox_step(n * self.V, self._ph_blocks)
        k2_step = 0 if getattr(self, "_greedy", False) else slot.ph_step          # greedy: sampling_logits2tokens draws nothing (JL:127-129)
        p.philox_blocks, p.philox_seed = self._ph_blocks, slot.ph_seed
        p.philox_offset[0], p.philox_offset[1], p.philox_offset[2] = slot.ph_off, slot.ph_off + k2_step, slot.ph_off + k2_step + slot.ph_step
        if fresh:
            p.fresh_tok[:len(fresh)] = fresh
        self._write_rules(slot, L.IterParams.rules.offset, rules)
        if resid:
            self._write_rules(slot, L.IterParams.resid_rules.offset, resid)

    def _write_rules(self, slot, offset, rules):
        """a rule sequence goes into the slot's blob as ONE block copy, its packed bytes cached by the interned structs' identities
        (SJDEngine._write_rules: with four slots the per-element ctypes writes were ~0.1 ms of host time per forward)"""
        key = tuple(map(id, rules))
        blk = self._rule_bytes.get(key)
        if blk is None:
            blk = b"".join(bytes(r) for r in rules)
            if len(self._rule_bytes) < 4096:
                self._rule_bytes[key] = blk
                self._rule_keep.append(list(rules))
        L.ctypes.memmove(slot.params.host.data_ptr() + offset, blk, len(blk))

    def _columns(self, rule_lists):
        """union of the slots' output-head column windows (SJDEngine.logit_columns), None = all columns"""
        if not self.narrow_head:
            return None
        key = tuple(id(r) for rules in rule_lists for r in rules)
        if key in self._cols_cache:
            return self._cols_cache[key]
        cols = self._columns_uncached(rule_lists)
        if len(self._cols_cache) < 4096:
            self._cols_cache[key] = cols
            self._rule_keep.append([r for rules in rule_lists for r in rules])
        return cols

    def _columns_uncached(self, rule_lists):
        lo, hi = self.V, 0
        for rules in rule_lists:
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

    def _per_slot(self, fn):
        """fn(i, slot) for every slot, one after the other on the current stream.  (Round 6 measured the alternative -- the slots' K5 / K2 + K4
        launches as parallel branches of the iteration's hipGraph, forked onto side streams and rejoined: eight prompts 7.2 -> 8.5 ms per step,
        four 4.7 -> 5.5; a fork / join pair costs more than the 5-28 us launches it overlaps.  profiles/r6_slot_fanout_ab.txt)"""
        for i, s in enumerate(self.slots):
            fn(i, s)

    def _forward_body(self, cols):
        def k5(i, s):
            lo, hi = i * self.nb, (i + 1) * self.nb
            ops.reguess(s.params, s.state, self.input_ids[lo:hi], pos_offset=self.pos_offset[lo:hi], positions_out=self.positions[lo:hi])
        if self.slot_launches:
            ops.reguess_slots(self._slots_desc, self.params, self.state, self.input_ids, self.pos_offset, self.positions, self.nb)
        else:
            self._per_slot(k5)
        positions = self.positions
        if self.head_partials:
            return self.backbone.forward_window(self.input_ids, positions, -1, self.key_start, cols=cols, head_partials=True)
        if cols is None:
            return self.backbone.forward_window(self.input_ids, positions, -1, self.key_start)
        return self.backbone.forward_window(self.input_ids, positions, -1, self.key_start, cols=cols)

    def _sample_body(self, cur, logits, cols):
        """K2 + K4 per slot, noise generated in the kernels.  With the output head as split-K partials every slot's K2 reads ITS rows of
        the one G1 launch over the union column window (cond rows of batch row 2i, uncond rows of 2i + 1)."""
        part = isinstance(logits, ops.HeadOut)
        dbg = None
        if part and self.hook is not None:
            if self._dbg is None:
                self._dbg = torch.zeros(self.B, self.Lmax, self.V, dtype=torch.float32, device=self.device)
            dbg = self._dbg
            dbg.zero_()
        def k2_k4(i, s):
            tok_out, amax_out = (s.amax_ptr, s.tokens_ptr) if getattr(self, "_greedy", False) else (s.tokens_ptr, s.amax_ptr)
            if part:
                ops.logits_to_probs_sample_part(logits, self._guidance, s.params, None, s.probs[cur], tok_out, amax_out_ptr=amax_out,
                                                dbg=None if dbg is None else dbg[i * self.nb:(i + 1) * self.nb], row0=i * self.nb * self.Lmax,
                                                urow_off=self.Lmax if self.nb > 1 else 0, zero_state=s.zero_state[cur])
            else:
                lc = logits[i * self.nb]
                lu = logits[i * self.nb + 1] if self.nb > 1 else None
                ops.logits_to_probs_sample(lc, lu, self._guidance, s.params, None, s.probs[cur], tok_out, col0=cols[0] if cols else 0,
                                           amax_out_ptr=amax_out)
                s.zero_state[cur].fill_(-1)
            ops.verify_accept(s.params, s.state, s.probs[cur], s.probs[1 - cur], None, None, s.scratch, mirror=True)
        if self.slot_launches and part and ops.head_slots_ok(logits):
            # one K2 and one K4 launch for all slots (a workgroup per (row, slot) / per slot): eight prompts 8 x (28 + 24) us -> one of each
            greedy = getattr(self, "_greedy", False)
            sl = self._slots_desc
            if dbg is not None:
                sl = ops.slots_of(self.params, self.state, self.probs_all, self.zero_state_all, self.scratch_all, self.nb, dbg=dbg)
            ops.logits_to_probs_sample_part_slots(sl, logits, self._guidance, self.params, self.probs_all, cur, "amax" if greedy else "tokens",
                                                  "tokens" if greedy else "amax", self.state, self.zero_state_all, self.nb, dbg=dbg)
            ops.verify_accept_slots(sl, self.params, self.state, self.probs_all, cur, self.scratch_all)
            return
        self._per_slot(k2_k4)

    def _launch_forward(self, cols):
        if not self.use_graph:
            return self._forward_body(cols)
        self._check_graph_buffers()
        fkey = ("fwd", cols)
        if fkey not in self._graphs:
            if self._eager_runs.get(fkey, 0) < 1:     # one eager run warms up allocations / hipBLASLt before capture
                self._eager_runs[fkey] = 1
                return self._forward_body(cols)
            g = torch.cuda.CUDAGraph()
            with capture_graph(g):
                self._graph_logits[fkey] = self._forward_body(cols)
            self._graphs[fkey] = g
        self._graphs[fkey].replay()
        return self._graph_logits[fkey]

    def _launch_sample(self, cur, logits, cols):
        key = (cur, self._guidance, cols, self.hook is not None, getattr(self, "_greedy", False))
        if not self.use_graph or ("fwd", cols) not in self._graphs:
            self._sample_body(cur, logits, cols)
            return
        if key not in self._graphs:
            self._sample_body(cur, logits, cols)
            g = torch.cuda.CUDAGraph()
            with capture_graph(g):
                self._sample_body(cur, logits, cols)
            self._graphs[key] = g
            return
        self._graphs[key].replay()

    def _launch_window(self, cur, cols):
        """the whole iteration of all slots as ONE hipGraph per (column window, prob-buffer parity) -- SJDEngine._launch_window"""
        if not self.use_graph:
            logits = self._forward_body(cols)
            self._sample_body(cur, logits, cols)
            return logits
        self._check_graph_buffers()
        key = ("win", cols, cur, self._guidance, self.hook is not None, getattr(self, "_greedy", False))
        if key not in self._graphs:
            if self._eager_runs.get(key, 0) < 1:
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
        return sorted({k[1] for k in self._graphs if isinstance(k, tuple) and k[0] in ("fwd", "win")}, key=lambda c: (c is None, c))

    def _upload_resid(self, s, resid):
        """two-stage iterations: a slot's residual rules go up behind the forward, on the same stream (SJDEngine._upload_resid)"""
        off = L.IterParams.resid_rules.offset
        self._write_rules(s, off, resid)
        L.check(L.load().sjd_upload_async(s.params.dev.data_ptr() + off, s.params.host.data_ptr() + off, s.params.nbytes - off, ops._stream()),
                "sjd_upload_async")

    def _observer_noise(self, s, n_rows, scheme):
        """what the slot's kernels generate, drawn by torch from the same generator state -- for observers only (the parity tests' hook)"""
        self._noise_tensors(s)
        s.gen.set_offset(s.ph_off)
        if getattr(self, "_greedy", False):
            s.noise[:n_rows].fill_(1.0)
        else:
            s.noise[:n_rows].exponential_(generator=s.gen)
        if n_rows > 1 and scheme == 0:
            s.rs[:n_rows].uniform_(0.0, 1.0, generator=s.gen)
            s.noise2.exponential_(generator=s.gen)

    # ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decode_many(self, prompts: List[List[int]], specs: List[WindowSpec], grammars, cfg: SJDConfig, seeds=None,
                    warmup_iters=0, timed_iters=None, on_timed_start=None, on_timed_end=None):
        """One SJDConfig for all prompts (seed = cfg.seed + prompt index unless `seeds` is given).  Returns [(sequence, DecodeStats)] in
        prompt order; with timed_iters the stats cover window iterations [warmup_iters, warmup_iters + timed_iters).

        len(prompts) may exceed the number of slots (continuous batching): a slot whose prompt reached its end token is handed the next
        prompt of the list -- state machine, grammar and generators re-created, its KV rows reused from 0, its prompt prefilled eagerly
        over its own batch rows -- while the other slots keep their windows; the captured window graphs are unaffected.  Only when the
        list is exhausted does a finished slot ride along with a one-row dummy window."""
        # GenerationConfig(do_sample=False): K2's MODE of p goes where its draw would (state.tokens; the draw lands in state.amax and is ignored),
        # nothing is consumed from a slot's generator for it, the verify step's draws follow at once (SJDEngine.decode, JL:124-129)
        self._greedy = greedy = not getattr(cfg, "do_sample", True)
        k2_draws = 0 if greedy else 1
        N = len(prompts)
        assert len(specs) == len(grammars) == N and N >= self.P
        if cfg.multi_token_init_scheme not in ("random", "repeat_horizon", "sample_horizon"):
            raise ValueError(f"multi_token_init_scheme should be 'random', 'repeat_horizon' or 'sample_horizon', but got {cfg.multi_token_init_scheme}")
        for g_ in grammars:
            warn_no_grid(cfg, g_)
        if cfg.prefix_token_sampler_scheme not in ("speculative_jacobi", "jacobi"):
            raise ValueError(f"prefix_token_sampler_scheme: {cfg.prefix_token_sampler_scheme}")
        if cfg.max_num_new_tokens > self.Lmax:
            raise ValueError("max_num_new_tokens exceeds the engine's max_window")
        dev, nb = self.device, self.nb
        self._ph_blocks = ops.philox_max_blocks(dev)
        ph_row = ops.philox_step(self.V, self._ph_blocks)                  # offset step of a [1, V] draw
        default_gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
        scheme = 0 if cfg.prefix_token_sampler_scheme == "speculative_jacobi" else 1
        do_cfg = cfg.do_cfg and (cfg.guidance_scale != 1)
        W = cfg.max_num_new_tokens
        self._guidance = float(cfg.guidance_scale)
        attn = getattr(self.backbone, "attn", None)
        full_cache = self.backbone.cache
        results = [None] * N
        next_prompt = 0
        emitted_total = 0                 # tokens appended to any prompt's sequence so far (the timed region counts its increase)
        cur = 0

        def admit(i, buf):
            """prompt `next_prompt` -> slot i: iteration 0 (JL:344-350 short-circuit), a prefill over the slot's own batch rows whose
            distribution lands in probs[buf]"""
            nonlocal next_prompt
            j, s = next_prompt, self.slots[i]
            next_prompt += 1
            s.prompt_index = j
            s.X = [int(t) for t in prompts[j]]
            s.P = len(s.X)
            seed = (seeds[j] if seeds is not None else (None if cfg.seed is None else cfg.seed + j))
            if seed is None:
                # no seed given: every slot needs its OWN Philox stream.  Sharing the device's default generator (round 3) handed every slot
                # the same (seed, offset) pair -- identical prompts in a batch then drew identical noise.  A per-slot seed is
                # drawn FROM the default generator instead: it advances with every admission, and torch.manual_seed still reproduces a run.
                slot_seed = int(torch.randint(0, 2 ** 62, (1,), device=dev, generator=default_gen).item())
                s.gen = torch.Generator(dev).manual_seed(slot_seed)
            else:
                s.gen = torch.Generator(dev).manual_seed(seed)
            s.ph_seed, s.ph_off = int(s.gen.initial_seed()), int(s.gen.get_offset())
            s.cpu_gen = None if seed is None else torch.Generator().manual_seed(seed)     # the prompt's "global CPU generator" (JL:505)
            s.grammar = grammars[j]
            s.grammar.start(s.X)
            s.l_abs, s.r_abs = s.P + cfg.jacobi_loop_interval_l, s.P + cfg.jacobi_loop_interval_r
            s.n, s.kv_len, s.cur_len, s.n_prev, s.m_prev = 1, specs[j].kv_base, s.P, 1, 1
            s.carried, s.carried_amax, s.last_amax, s.finished, s.harvested, s.stats = [], [], None, False, False, DecodeStats()
            s.t_admit = time.perf_counter()
            self.key_start[i * nb:(i + 1) * nb].copy_(specs[j].key_start.to(device=dev, dtype=torch.int32))
            self.pos_offset[i * nb:(i + 1) * nb].copy_(specs[j].pos_offset.to(device=dev, dtype=torch.int64))
            if s.cpu_gen is not None:
                torch.randint(0, cfg.img_vocab_n, (1, 0), generator=s.cpu_gen)
            rules = s.grammar.window_rules(1)
            use_cfg = do_cfg and not s.grammar.force_no_cfg()
            self._fill(s, 1, s.kv_len, use_cfg, scheme, [], rules, [])
            s.params.upload()
            if self.hook is not None:
                self._observer_noise(s, 1, scheme)
            tokens, positions = specs[j].first_tokens.to(dev), specs[j].first_positions.to(dev)
            from .engine import SJDEngine
            SJDEngine._calibrate_fp8(self, tokens, positions, self.key_start[i * nb:(i + 1) * nb])      # (an fp8 cache nobody calibrated: once per backbone)
            self.backbone.cache = _CacheView(full_cache, i * nb, (i + 1) * nb)
            try:
                logits = self.backbone.forward_window(tokens, positions, s.kv_len, self.key_start[i * nb:(i + 1) * nb])
            finally:
                self.backbone.cache = full_cache
            lc = logits[0, -1:, :]
            lu = logits[1, -1:, :] if nb > 1 else None
            tok_out, amax_out = (s.amax_ptr, s.tokens_ptr) if greedy else (s.tokens_ptr, s.amax_ptr)
            ops.logits_to_probs_sample(lc, lu, self._guidance, s.params, None, s.probs[buf], tok_out, amax_out_ptr=amax_out)
            s.zero_state[buf].fill_(-1)
            ops.verify_accept(s.params, s.state, s.probs[buf], s.probs[1 - buf], None, None, s.scratch, mirror=True)
            s.ph_off += s.ph_step * k2_draws                       # the [1, V] multinomial of iteration 0 (greedy: none)
            if self.hook is not None:
                self.hook(j, dict(first=True, n_rows=1, logits_c=lc, logits_u=lu, use_cfg=use_cfg, rules=rules, resid=[],
                                  noise=s.noise[:1], rs=s.rs[:1], noise2=s.noise2[0], probs=s.probs[buf], prev_probs=s.probs[1 - buf],
                                  ctx=list(s.X), scheme=scheme))
                s.gen.set_offset(s.ph_off)
            s.win_len = tokens.shape[1]

        def after_prefill(s):
            """host bookkeeping of iteration 0 (after the state download)"""
            nonlocal emitted_total
            y0 = int(s.state.view.tokens[0])
            s.last_amax = y0 if greedy else int(s.state.view.amax[0])
            s.stats.matched.append(s.win_len)
            s.n = min(W, s.r_abs - s.cur_len) if (s.l_abs <= s.cur_len < s.r_abs) else 1
            s.X.append(y0)
            emitted_total += 1
            s.grammar.push([y0])
            s.kv_len += s.win_len
            s.n_prev, s.m_prev, s.carried, s.carried_amax = 1, 1, [], []
            s.stats.nfe += 1
            s.finished = s.X[-1] in cfg.eos_token_ids or len(s.X) >= cfg.max_length
            s.cur_len = len(s.X)

        def harvest(s):
            s.harvested = True
            s.stats.wall_seconds = time.perf_counter() - s.t_admit
            s.stats.tokens, s.stats.timed_nfe, s.stats.kv_len = len(s.X) - s.P, s.stats.nfe, s.kv_len
            s.gen.set_offset(s.ph_off)                             # leave the generator where the reference's draws would have left it
            results[s.prompt_index] = (s.X, s.stats)

        # ---------------- iteration 0 of the first P prompts ----------------
        if attn is not None and hasattr(attn, "params"):
            attn.params = None
        for i in range(self.P):
            admit(i, cur)
        self.state.wait_mirror()
        for s in self.slots:
            after_prefill(s)
        cur = 1 - cur
        if attn is not None and hasattr(attn, "params"):
            attn.params = self.params                          # windows: kv_len / n_rows of every prompt from its blob

        # ---------------- window iterations, all slots in lock-step ----------------
        it = 0
        timed_started = False
        t0 = time.perf_counter()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        tok0 = emitted_total
        host_s = sync_s = 0.0
        while True:
            # finished prompts leave, queued prompts enter (eager prefill; the window graphs hold none of the buffers it allocates)
            for i, s in enumerate(self.slots):
                if s.finished and not s.harvested:
                    harvest(s)
                    if next_prompt < N:
                        if attn is not None and hasattr(attn, "params"):
                            attn.params = None
                        admit(i, 1 - cur)
                        self.state.wait_mirror()
                        after_prefill(s)
                        if attn is not None and hasattr(attn, "params"):
                            attn.params = self.params
            if all(s.finished for s in self.slots):
                if any(not s.harvested for s in self.slots):
                    continue                                    # (a prompt that ended in its prefill iteration)
                break
            if timed_iters is not None and it == warmup_iters:
                timed_started = True
                if on_timed_start is not None:
                    on_timed_start()
                tok0 = emitted_total
                host_s = sync_s = 0.0
                t0 = time.perf_counter()
                ev0.record()
            t_h = time.perf_counter()
            rule_lists, metas, slow = [], [], []
            for s in self.slots:
                resid, win = [], None
                if s.finished:                                 # dummy one-row window: a forced row (K2 reads no logits), result ignored
                    n_rows, fresh, a = 1, [], 0
                    rules, use_cfg = [ops.make_rule(forced=0)], False
                else:
                    n_rows = s.n
                    a = max(0, min(s.n_prev - s.m_prev, s.n - 1))
                    fr = torch.randint(0, cfg.img_vocab_n, (1, s.n - 1 - a), generator=s.cpu_gen)[0].tolist()
                    fresh = [cfg.img_vocab_lo + t for t in fr]
                    if cfg.multi_token_init_scheme != "random":                      # spatial init (JL:516-594), as in SJDEngine.decode
                        fresh = spatial_fresh_tokens(cfg.multi_token_init_scheme, fresh, len(s.X) + a, s.carried[a - 1] if a else s.X[-1],
                                                     s.carried_amax[a - 1] if a else s.last_amax, s.grammar.grid())
                    rules = s.grammar.window_rules(n_rows)
                    use_cfg = do_cfg and not s.grammar.force_no_cfg()
                    rule_lists.append(rules)
                    if scheme == 0 and n_rows > 1:             # residual rules before the launch when the grammar can name them
                        win = [s.X[-1]] + s.carried[:a] + fresh
                        resid = s.grammar.fast_residual_rules(win, rules)
                        if resid is None:
                            slow.append((s, win, len(metas)))
                self._fill(s, n_rows, s.kv_len, use_cfg, scheme, fresh, rules, resid or [])
                metas.append([n_rows, rules, resid, use_cfg])
            self.params.upload()
            cols = self._columns(rule_lists) if rule_lists else None
            if self.hook is not None:
                for s, mt in zip(self.slots, metas):
                    self._observer_noise(s, mt[0], scheme)
            host_s += time.perf_counter() - t_h
            if slow:                                            # two stages: the missing residual rules are computed under the forward
                logits = self._launch_forward(cols)
                t_h = time.perf_counter()
                for s, win, k in slow:
                    metas[k][2] = s.grammar.residual_rules(win)
                    self._upload_resid(s, metas[k][2])
                host_s += time.perf_counter() - t_h
                self._launch_sample(cur, logits, cols)
            else:
                logits = self._launch_window(cur, cols)
            metas = [tuple(mt) for mt in metas]
            if self.hook is not None:
                part = isinstance(logits, ops.HeadOut)
                for i, (s, (n_rows, rules, resid, use_cfg)) in enumerate(zip(self.slots, metas)):
                    if s.finished:
                        continue
                    if part:                                   # the logits exactly as K2 derived them
                        lg = self._dbg[i * nb:(i + 1) * nb, :n_rows]
                    else:
                        lg = logits[i * nb:(i + 1) * nb, :n_rows]
                        if cols is not None:
                            full = torch.zeros(nb, n_rows, self.V, dtype=logits.dtype, device=dev)
                            full[:, :, cols[0]:cols[1]] = lg
                            lg = full
                    self.hook(s.prompt_index, dict(first=False, n_rows=n_rows, logits_c=lg[0], logits_u=lg[1] if nb > 1 else None,
                                                   use_cfg=use_cfg, rules=rules, resid=resid, noise=s.noise[:n_rows], rs=s.rs[:n_rows],
                                                   noise2=s.noise2[0], probs=s.probs[cur], prev_probs=s.probs[1 - cur], ctx=list(s.X),
                                                   scheme=scheme))
            t_s = time.perf_counter()
            self.state.wait_mirror()                              # the single sync of the iteration
            sync_s += time.perf_counter() - t_s
            for s, (n_rows, _, _, _) in zip(self.slots, metas):
                if s.finished:
                    s.ph_off += s.ph_step * k2_draws           # (the dummy row's draw: keeps blob and generator consistent; never observed)
                    continue
                st = s.state.view
                m_dev, rejected = int(st.m), bool(st.rejected)
                if int(st.rejected) > 1:
                    raise RuntimeError("SJD verify: the residual distribution max(p - q, 0) is empty under the residual grammar rule")
                draws_rs = n_rows > 1 and scheme == 0          # what torch would have consumed: multinomial [+ rand [+ residual multinomial]]
                s.ph_off += s.ph_step * (k2_draws + (1 if draws_rs else 0)) + (ph_row if (draws_rs and rejected) else 0)
                if self.hook is not None:
                    s.gen.set_offset(s.ph_off)
                Y = st.tokens[:n_rows]
                A = Y if greedy else st.amax[:n_rows]          # modes of this iteration's target rows (K2 by-product; greedy: the tokens themselves)
                if n_rows <= 1:
                    m, emitted, s.carried, s.carried_amax, s.last_amax = 1, [Y[0]], [], [], A[0]
                else:
                    m, emitted, s.carried, s.carried_amax, s.last_amax = m_dev, Y[:m_dev], Y[m_dev:], A[m_dev:], A[m_dev - 1]
                s.stats.matched.append(m)
                s.n = min(W, s.r_abs - s.cur_len) if (s.l_abs <= s.cur_len < s.r_abs) else 1      # JL:1142-1144 (old cur_len)
                s.X.extend(emitted)
                emitted_total += len(emitted)
                s.grammar.push(emitted)
                s.kv_len += m
                s.n_prev, s.m_prev = n_rows, (1 if n_rows <= 1 else m)
                s.stats.nfe += 1
                if s.X[-1] in cfg.eos_token_ids or len(s.X) >= cfg.max_length:
                    s.finished = True
                s.cur_len = len(s.X)
            cur = 1 - cur
            it += 1
            if timed_iters is not None and it == warmup_iters + timed_iters:
                break
        ev1.record()
        torch.cuda.synchronize()
        from .engine import check_reduce_timeouts
        check_reduce_timeouts()                    # (a reducing projection that gave up a wait: wrong numbers must not become tokens -- ADVICE r3)
        if on_timed_end is not None and (timed_started or timed_iters is None):      # never an unmatched barrier (SJDEngine.decode)
            on_timed_end()
        seconds = ev0.elapsed_time(ev1) / 1000.0
        for s in self.slots:                                   # prompts still in flight when a timed run stops
            if not s.harvested:
                harvest(s)
        self.run_stats = dict(seconds=seconds, wall_seconds=time.perf_counter() - t0, iterations=it,
                              timed_iterations=(it - warmup_iters) if timed_iters is not None else it,
                              tokens=emitted_total - tok0, host_seconds=host_s, sync_seconds=sync_s, prompts_started=next_prompt,
                              timed_region_reached=bool(timed_started or timed_iters is None))
        out = []
        for r in results:
            if r is None:                                      # never admitted (a timed run that stopped early)
                out.append(None)
                continue
            X, st = r
            st.seconds, st.host_seconds, st.sync_seconds = seconds, host_s, sync_s
            out.append((X, st))
        return out
