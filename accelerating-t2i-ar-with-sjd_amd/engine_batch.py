"""Several prompts per GPU in ONE window forward (the reference decodes one prompt per process).

A draft-window forward is a pure weight stream: 13 GB of weights for 32 activation rows.  Two prompts (2 x B_cfg x L = 64 rows)
cost the same stream, so decoding them together almost doubles the accepted tokens per second of a GPU; up to four prompts (128 rows,
G1's sub-tiled kernel) fit in one forward: 963 / 1292 tokens/s per MI355X for two / four Lumina 768px prompts against 561 for one.  Every prompt ("slot")
keeps exactly the state machine of `SJDEngine.decode` -- its own window, accept length, KV length, grammar, device generator
and CPU generator for the fresh ids -- so each slot takes the decisions its solo run would take on the same logits; only the
transformer forward is shared:

   host    per slot: n, fresh ids, rules -> its blob of a contiguous sjd_iter_params ARRAY (batch_rows = B_cfg: the kernels
           K1 / K3 / F2 pick the blob of a batch row, i.e. every prompt has its own kv_len and n_rows), one H2D for all
   graph 1 K5 per slot (window ids of that slot's batch rows), ONE backbone.forward_window over all slots' rows
   graph 2 K2 + K4 per slot (own noise, own probability buffers)
   host    one D2H of the state array, per-slot bookkeeping

A slot that reaches its end token keeps riding along with a one-row dummy window (its tokens are ignored) until every slot is done.
"""
import time
from typing import List

import torch

from . import _lib as L
from . import ops
from .engine import DecodeStats, SJDConfig, WindowSpec


class _CacheView:
    """The KV cache rows of one slot (batch rows [lo, hi)) presented as a StaticKVCache to the backbone's prefill."""

    def __init__(self, cache, lo, hi):
        self.k, self.v, self.s_max = cache.k[:, lo:hi], cache.v[:, lo:hi], cache.s_max


class _Slot:
    pass


class SJDBatchEngine:
    def __init__(self, backbone, vocab_size, device, n_prompts, max_window=16, n_batch=2, use_graph=True, narrow_head=True):
        L.load()                                   # fail loudly if the HIP extension is missing
        if max_window > L.MAX_WINDOW:
            raise ValueError(f"max_window {max_window} > {L.MAX_WINDOW}")
        if n_prompts * n_batch * max_window > 128:
            raise ValueError("the window forward (G1, F1-F3) serves at most 128 rows: n_prompts * n_batch * max_window <= 128")
        self.backbone, self.V, self.device = backbone, int(vocab_size), torch.device(device)
        self.P, self.nb, self.Lmax, self.B = n_prompts, n_batch, max_window, n_prompts * n_batch
        self.use_graph, self.narrow_head = use_graph, narrow_head
        dev = self.device
        self.params = ops.BlobArray(L.IterParams, self.P, dev)
        self.state = ops.BlobArray(L.State, self.P, dev)
        self.params.view = self.params.blobs[0].view          # what HipWindowAttention's profiling hook looks at
        self.slots = []
        for i in range(self.P):
            s = _Slot()
            s.params, s.state = self.params.blobs[i], self.state.blobs[i]
            s.params.view.batch_rows = n_batch
            s.probs = torch.zeros(2, self.Lmax, self.V, dtype=torch.float32, device=dev)
            s.noise = torch.ones(self.Lmax, self.V, dtype=torch.float32, device=dev)
            s.rs = torch.zeros(self.Lmax, self.V, dtype=torch.float32, device=dev)
            s.noise2 = torch.ones(1, self.V, dtype=torch.float32, device=dev)
            s.scratch = torch.empty(self.V, dtype=torch.float32, device=dev)
            s.tokens_ptr = s.state.field_ptr("tokens")
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
        self.rng_stream = torch.cuda.Stream(device=dev)
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
    def _fill(self, slot, n, kv_len, use_cfg, scheme, fresh, rules, resid):
        p = slot.params.view
        p.n_rows, p.kv_len, p.use_cfg, p.scheme, p.n_fresh = n, kv_len, int(use_cfg), scheme, len(fresh)
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

    def _forward_body(self, cols):
        for i, s in enumerate(self.slots):
            lo, hi = i * self.nb, (i + 1) * self.nb
            ops.reguess(s.params, s.state, self.input_ids[lo:hi], pos_offset=self.pos_offset[lo:hi], positions_out=self.positions[lo:hi])
        positions = self.positions
        if cols is None:
            return self.backbone.forward_window(self.input_ids, positions, -1, self.key_start)
        return self.backbone.forward_window(self.input_ids, positions, -1, self.key_start, cols=cols)

    def _sample_body(self, cur, logits, cols):
        for i, s in enumerate(self.slots):
            lc = logits[i * self.nb]
            lu = logits[i * self.nb + 1] if self.nb > 1 else None
            ops.logits_to_probs_sample(lc, lu, self._guidance, s.params, s.noise, s.probs[cur], s.tokens_ptr, col0=cols[0] if cols else 0)
            ops.verify_accept(s.params, s.state, s.probs[cur], s.probs[1 - cur], s.rs, s.noise2[0], s.scratch, mirror=True)

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
            with torch.cuda.graph(g, capture_error_mode="thread_local"):      # other host threads may use the GPU (servers; the streaming parity tests)
                self._graph_logits[fkey] = self._forward_body(cols)
            self._graphs[fkey] = g
        self._graphs[fkey].replay()
        return self._graph_logits[fkey]

    def _launch_sample(self, cur, logits, noise_ready, cols):
        torch.cuda.current_stream().wait_event(noise_ready)
        key = (cur, self._guidance, cols)
        if not self.use_graph or ("fwd", cols) not in self._graphs:
            self._sample_body(cur, logits, cols)
            return
        if key not in self._graphs:
            self._sample_body(cur, logits, cols)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):      # other host threads may use the GPU (servers; the streaming parity tests)
                self._sample_body(cur, logits, cols)
            self._graphs[key] = g
            return
        self._graphs[key].replay()

    def _draw_noise(self, s, n_rows, scheme):
        """the slot's three noise tensors, in the reference's order and shapes, from the slot's own device generator"""
        s.g_state = None
        s.noise[:n_rows].exponential_(generator=s.gen)
        if n_rows > 1 and scheme == 0:
            s.rs[:n_rows].uniform_(0.0, 1.0, generator=s.gen)
            if s.gen is not None:
                s.g_state = s.gen.get_state()
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
        N = len(prompts)
        assert len(specs) == len(grammars) == N and N >= self.P
        if cfg.multi_token_init_scheme != "random":
            raise NotImplementedError("only multi_token_init_scheme='random' is parity-checkable")
        if cfg.prefix_token_sampler_scheme not in ("speculative_jacobi", "jacobi"):
            raise ValueError(f"prefix_token_sampler_scheme: {cfg.prefix_token_sampler_scheme}")
        if cfg.max_num_new_tokens > self.Lmax:
            raise ValueError("max_num_new_tokens exceeds the engine's max_window")
        dev, nb = self.device, self.nb
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
            s.gen = None if seed is None else torch.Generator(dev).manual_seed(seed)
            s.cpu_gen = None if seed is None else torch.Generator().manual_seed(seed)     # the prompt's "global CPU generator" (JL:505)
            s.grammar = grammars[j]
            s.grammar.start(s.X)
            s.l_abs, s.r_abs = s.P + cfg.jacobi_loop_interval_l, s.P + cfg.jacobi_loop_interval_r
            s.n, s.kv_len, s.cur_len, s.n_prev, s.m_prev = 1, specs[j].kv_base, s.P, 1, 1
            s.carried, s.finished, s.harvested, s.stats = [], False, False, DecodeStats()
            s.t_admit = time.perf_counter()
            self.key_start[i * nb:(i + 1) * nb].copy_(specs[j].key_start.to(device=dev, dtype=torch.int32))
            self.pos_offset[i * nb:(i + 1) * nb].copy_(specs[j].pos_offset.to(device=dev, dtype=torch.int64))
            if s.cpu_gen is not None:
                torch.randint(0, cfg.img_vocab_n, (1, 0), generator=s.cpu_gen)
            rules = s.grammar.window_rules(1)
            use_cfg = do_cfg and not s.grammar.force_no_cfg()
            self._fill(s, 1, s.kv_len, use_cfg, scheme, [], rules, [])
            s.params.upload()
            self._draw_noise(s, 1, scheme)
            tokens, positions = specs[j].first_tokens.to(dev), specs[j].first_positions.to(dev)
            self.backbone.cache = _CacheView(full_cache, i * nb, (i + 1) * nb)
            try:
                logits = self.backbone.forward_window(tokens, positions, s.kv_len, self.key_start[i * nb:(i + 1) * nb])
            finally:
                self.backbone.cache = full_cache
            lc = logits[0, -1:, :]
            lu = logits[1, -1:, :] if nb > 1 else None
            ops.logits_to_probs_sample(lc, lu, self._guidance, s.params, s.noise, s.probs[buf], s.tokens_ptr)
            ops.verify_accept(s.params, s.state, s.probs[buf], s.probs[1 - buf], s.rs, s.noise2[0], s.scratch, mirror=True)
            if self.hook is not None:
                self.hook(j, dict(first=True, n_rows=1, logits_c=lc, logits_u=lu, use_cfg=use_cfg, rules=rules, resid=[],
                                  noise=s.noise[:1], rs=s.rs[:1], noise2=s.noise2[0], probs=s.probs[buf], prev_probs=s.probs[1 - buf],
                                  ctx=list(s.X), scheme=scheme))
            s.win_len = tokens.shape[1]

        def after_prefill(s):
            """host bookkeeping of iteration 0 (after the state download)"""
            nonlocal emitted_total
            y0 = int(s.state.view.tokens[0])
            s.stats.matched.append(s.win_len)
            s.n = min(W, s.r_abs - s.cur_len) if (s.l_abs <= s.cur_len < s.r_abs) else 1
            s.X.append(y0)
            emitted_total += 1
            s.grammar.push([y0])
            s.kv_len += s.win_len
            s.n_prev, s.m_prev, s.carried = 1, 1, []
            s.stats.nfe += 1
            s.finished = s.X[-1] in cfg.eos_token_ids or len(s.X) >= cfg.max_length
            s.cur_len = len(s.X)

        def harvest(s):
            s.harvested = True
            s.stats.wall_seconds = time.perf_counter() - s.t_admit
            s.stats.tokens, s.stats.timed_nfe, s.stats.kv_len = len(s.X) - s.P, s.stats.nfe, s.kv_len
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
                if on_timed_start is not None:
                    on_timed_start()
                tok0 = emitted_total
                host_s = sync_s = 0.0
                t0 = time.perf_counter()
                ev0.record()
            t_h = time.perf_counter()
            rule_lists, metas = [], []
            for s in self.slots:
                if s.finished:                                 # dummy one-row window: a forced row (K2 reads no logits), result ignored
                    n_rows, fresh, a = 1, [], 0
                    rules, use_cfg = [ops.make_rule(forced=0)], False
                else:
                    n_rows = s.n
                    a = max(0, min(s.n_prev - s.m_prev, s.n - 1))
                    fr = torch.randint(0, cfg.img_vocab_n, (1, s.n - 1 - a), generator=s.cpu_gen)[0].tolist()
                    fresh = [cfg.img_vocab_lo + t for t in fr]
                    rules = s.grammar.window_rules(n_rows)
                    use_cfg = do_cfg and not s.grammar.force_no_cfg()
                    rule_lists.append(rules)
                self._fill(s, n_rows, s.kv_len, use_cfg, scheme, fresh, rules, [])
                metas.append([n_rows, rules, [], use_cfg, a, fresh])
            self.params.upload()
            self.rng_stream.wait_stream(torch.cuda.current_stream())       # the previous iteration is done with the noise buffers
            cols = self._columns(rule_lists) if rule_lists else None
            host_s += time.perf_counter() - t_h
            logits = self._launch_forward(cols)                             # everything below overlaps the forward
            off = L.IterParams.resid_rules.offset
            with torch.cuda.stream(self.rng_stream):
                for s, mt in zip(self.slots, metas):
                    n_rows, a, fresh = mt[0], mt[4], mt[5]
                    if not s.finished and scheme == 0 and n_rows > 1:
                        mt[2] = s.grammar.residual_rules([s.X[-1]] + s.carried[:a] + fresh)
                        self._write_rules(s, off, mt[2])
                        s.params.dev[off:].copy_(s.params.host[off:], non_blocking=True)
                    self._draw_noise(s, n_rows, scheme)
                noise_ready = self.rng_stream.record_event()
            self._launch_sample(cur, logits, noise_ready, cols)
            metas = [tuple(mt[:4]) for mt in metas]
            if self.hook is not None:
                for i, (s, (n_rows, rules, resid, use_cfg)) in enumerate(zip(self.slots, metas)):
                    if s.finished:
                        continue
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
                    continue
                st = s.state.view
                m_dev, rejected = int(st.m), bool(st.rejected)
                if int(st.rejected) > 1:
                    raise RuntimeError("SJD verify: the residual distribution max(p - q, 0) is empty under the residual grammar rule")
                if s.g_state is not None and not rejected:
                    s.gen.set_state(s.g_state)
                Y = [int(st.tokens[j]) for j in range(n_rows)]
                if n_rows <= 1:
                    m, emitted, s.carried = 1, [Y[0]], []
                else:
                    m, emitted, s.carried = m_dev, Y[:m_dev], Y[m_dev:]
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
        if on_timed_end is not None:
            on_timed_end()
        seconds = ev0.elapsed_time(ev1) / 1000.0
        for s in self.slots:                                   # prompts still in flight when a timed run stops
            if not s.harvested:
                harvest(s)
        self.run_stats = dict(seconds=seconds, wall_seconds=time.perf_counter() - t0, iterations=it,
                              timed_iterations=(it - warmup_iters) if timed_iters is not None else it,
                              tokens=emitted_total - tok0, host_seconds=host_s, sync_seconds=sync_s, prompts_started=next_prompt)
        out = []
        for r in results:
            if r is None:                                      # never admitted (a timed run that stopped early)
                out.append(None)
                continue
            X, st = r
            st.seconds, st.host_seconds, st.sync_seconds = seconds, host_s, sync_s
            out.append((X, st))
        return out
