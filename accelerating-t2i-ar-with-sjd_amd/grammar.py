"""Host-side integer grammar state -> per-row kernel rules (sjd_row_rule).

The reference evaluates its "3-dim" logits processors with tensor ops and 4+ device->host syncs per call
(SURVEY.md 3.2).  Here the grammar is pure integer bookkeeping that advances incrementally as tokens are
accepted (O(1) per token, no rescans, no syncs); the masks themselves are applied inside kernels K2/K4.

  LuminaGrammar   <- MultiTokensVLLogitsProcessor + MultiTokensInterleavedTopKLogitsWarper
                     (reference scheduler/logit_processor_3dim.py:25-43, 84-155, 190-204)
  TopKTopPGrammar <- TopKLogitsWarper + TopPLogitsWarper3d (reference llamagen/llamagen_solver.py:458-470,
                     scheduler/logit_processor_3dim.py:406-419)
  Emu3Grammar     <- EOLLogitProcessor3d (+ TopK(2048) that HF generate appends)
                     (reference scheduler/jacobi_iteration_emu3.py:44-128, test_emu3.py:81-90)
  AnoleGrammar    <- the 3d processors of image-only mode (reference scheduler/jacobi_iteration_anhole.py:194-232,
                     scheduler/logit_processor_3dim.py:207-353)

Interface: start(ctx) / push(tokens) / window_rules(n) / residual_rules(win) / force_no_cfg().
residual_rules(win)[i-1] is the rule of the residual resample if the first rejection happens at window
position i, i.e. evaluated on ctx + win[1:i] (reference jacobi_iteration_lumina_mgpt.py:297-306).
"""
from . import ops


class _Grammar:
    temperature = 1.0       # HF TemperatureLogitsWarper of the processor list (grammar_from_processors sets it): part of every rule
    top_p = None            # HF TopPLogitsWarper of the processor list (GenerationConfig.top_p < 1): likewise, behind top-k and the temperature

    def start(self, ctx):
        self.reset()
        self.push(ctx)

    def reset(self):
        raise NotImplementedError

    def push(self, tokens):
        for t in tokens:
            self._advance(int(t))

    def _snapshot(self):
        raise NotImplementedError

    def _restore(self, s):
        raise NotImplementedError

    def residual_rules(self, win):
        snap = self._snapshot()
        out = []
        for i in range(1, len(win)):
            out.append(self.window_rules(1)[0])
            self._advance(int(win[i]))
        self._restore(snap)
        return out

    def fast_residual_rules(self, win, rules):
        """residual_rules(win) WITHOUT replaying the grammar, or None when that is not provably the same.  `rules` = window_rules(len(win))
        at the current state.  The reference evaluates the residual processors on ctx + win[1:i] (JL:297-306) and the window processors
        row by row on ctx (LP:84-155): whenever the window tokens move the grammar only by their COUNT, the residual rule of a rejection
        at position i is the window rule of row i - 1.  The engine then has every input of an iteration before it launches anything and
        runs it as ONE hipGraph (K5 -> forward -> K2 -> K4); grammars / states that return None keep the two-stage launch with the
        residual rules computed under the forward."""
        return None

    def force_no_cfg(self):
        return False

    def grid(self):
        """(absolute index of the image's first token, tokens per image row excluding the line token, image-id range lo, hi) while an
        image with line tokens is open, else None.  Drives the spatial draft initialisation (multi_token_init_scheme 'repeat_horizon' / 'sample_horizon',
        reference JL:516-594: `img_width` = logits_processor[0].w_latent_dim, one pad token per row)."""
        return None


def spatial_fresh_tokens(scheme, fresh, n_known, left_tok, left_amax, grid):
    """Draft initialisation from the left neighbour (the paper's spatial-locality-aware token initialisation; reference JL:516-594,
    which is broken at JL:577 in the released tree -- SURVEY.md 8a defect ledger -- so this follows the code's evident intent):
    the fresh draft at absolute index s = n_known + j sits in image column (s - img_start) % (w + 1); if it has a left neighbour in the
    same image row (column >= 1) it takes that neighbour's token ('repeat_horizon') or the mode of the distribution that neighbour was
    drawn from ('sample_horizon': the reference's top-1 re-draw), chaining through the fresh drafts; first-column drafts and drafts whose
    source is not an image token keep the uniformly random id of the 'random' scheme.  The draft distribution stays a one-hot.
      fresh: the random ids already drawn (the global-RNG draw happens for every scheme, JL:519-522);  n_known: tokens before the first
      fresh draft (accepted + last emitted + carried);  left_tok / left_amax: token (and its distribution's mode) just left of it."""
    if scheme not in ("random", "repeat_horizon", "sample_horizon"):
        raise ValueError(f"multi_token_init_scheme should be 'random', 'repeat_horizon' or 'sample_horizon', but got {scheme}")   # JL:560, 592
    if scheme == "random" or grid is None or not fresh:
        return list(fresh)
    img_start, w, img_lo, img_hi = grid
    out = []
    for j, rnd in enumerate(fresh):
        s = n_known + j
        col = (s - img_start) % (w + 1)
        src = left_tok if scheme == "repeat_horizon" else left_amax
        tok = src if (s > img_start and col >= 1 and src is not None and img_lo <= src < img_hi) else rnd
        out.append(int(tok))
        left_tok = left_amax = int(tok)             # a fresh draft's distribution is the one-hot of its token
    return out


class LuminaGrammar(_Grammar):
    def __init__(self, image_top_k=2000, text_top_k=10, image_start_token_id=8197, image_end_token_id=8196,
                 image_next_line_token_id=8803, img_lo=4, img_hi=8196):
        self.image_top_k, self.text_top_k = image_top_k, text_top_k
        self.start_id, self.end_id, self.eol_id = image_start_token_id, image_end_token_id, image_next_line_token_id
        self.img_lo, self.img_hi = img_lo, img_hi
        self._body_rules = {}
        self.reset()

    def reset(self):
        # length, #start, #end, tokens since the last start token, the two grid tokens after it
        self.s = (0, 0, 0, -1, 0, 0)

    def _snapshot(self):
        return self.s

    def _restore(self, s):
        self.s = s

    def _advance(self, t):
        n, ns, ne, since, g1, g2 = self.s
        if since >= 0:
            since += 1
            if since == 1:
                g1 = t
            elif since == 2:
                g2 = t
        if t == self.start_id:
            ns, since, g1, g2 = ns + 1, 0, 0, 0      # "last start token" (LP:96-97)
        if t == self.end_id:
            ne += 1
        self.s = (n + 1, ns, ne, since, g1, g2)

    def force_no_cfg(self):                        # check_is_force_no_cfg (JL:70-80)
        return self.s[1] == self.s[2]

    def fast_residual_rules(self, win, rules):
        _, ns, ne, since, _, _ = self.s
        # inside an image body (grid known) a token only advances the position counter; outside any image it does nothing -- unless it is
        # the start / end token itself.  The two tokens after <start> (since < 2) switch the rule set mid-window: replay those.
        if not ((ns == ne + 1 and since >= 2) or ns == ne):
            return None
        for t in win[1:-1]:
            if t == self.start_id or t == self.end_id:
                return None
        return list(rules[:len(win) - 1])

    def grid(self):
        n, ns, ne, since, g1, g2 = self.s
        if not (ns == ne + 1 and since >= 2):
            return None
        w = (g2 - 8804) * 2
        return (n - (since - 2), w, self.img_lo, self.img_hi) if w > 0 else None

    def window_rules(self, n):
        _, ns, ne, since, g1, g2 = self.s
        k = self.image_top_k if ns == ne + 1 else self.text_top_k      # LP:195-198
        T_, P_ = self.temperature, self.top_p
        if not (ns == ne + 1 and since >= 2):                           # LP:89-102
            return [ops.make_rule((), -1, k, P_, temperature=T_) for _ in range(n)]
        h, w = (g1 - 8804) * 2, (g2 - 8804) * 2                         # LP:107-111
        T = since - 2                                                   # tokens after <start> h w
        l1, l2 = w + 1, (w + 1) * h + 1
        trio = self._body_rules.get((k, T_, P_))
        if trio is None:                                                # the three rules of an image body, made once per top-k
            rng = ((self.img_lo, self.img_hi),)
            trio = self._body_rules[(k, T_, P_)] = (ops.make_rule(rng, -1, k, P_, temperature=T_), ops.make_rule(rng, self.eol_id, k, P_, temperature=T_),
                                                    ops.make_rule(rng, self.end_id, k, P_, temperature=T_))
        rules = [trio[0]] * n
        if l1 > 0:
            for j in range((-(T + 1)) % l1, n, l1):                     # rows with (T + 1 + j) % l1 == 0
                rules[j] = trio[1]                                      # LP:132-137
        if l2 > 0:
            for j in range((-(T + 1)) % l2, n, l2):
                rules[j] = trio[2]                                      # LP:140-145 (evaluated after the line rule: it wins)
        return rules


class TopKTopPGrammar(_Grammar):
    def __init__(self, top_k, top_p=1.0):
        self.top_k, self.top_p = top_k, top_p

    def reset(self):
        pass

    def _advance(self, t):
        pass

    def _snapshot(self):
        return None

    def _restore(self, s):
        pass

    def window_rules(self, n):
        return [ops.make_rule((), -1, self.top_k, self.top_p, temperature=self.temperature) for _ in range(n)]

    def fast_residual_rules(self, win, rules):          # stateless
        return list(rules[:len(win) - 1])


class Emu3Grammar(_Grammar):
    def __init__(self, height, width, visual_lo, visual_n, img_token, eoi_token, eos_token, eol_token, eof_token,
                 pad_token, top_k=2048):
        self.H, self.W = height, width
        self.vis = (visual_lo, visual_lo + visual_n)
        self.img, self.eoi, self.eos, self.eol, self.eof, self.pad = img_token, eoi_token, eos_token, eol_token, eof_token, pad_token
        self.top_k = top_k
        self.reset()

    def reset(self):
        self.since = -1          # tokens after the FIRST img token (offset_cache, JE:50-52)
        self.n = 0

    def _snapshot(self):
        return (self.since, self.n)

    def _restore(self, s):
        self.since, self.n = s

    def _advance(self, t):
        self.n += 1
        if self.since >= 0:
            self.since += 1
        elif t == self.img:
            self.since = 0

    def fast_residual_rules(self, win, rules):
        # tokens after the first image token only count; the padding clause (JE:118-123) keeps python's slice semantics, which differ
        # between a window and a single row once T + n passes the end of the image: replay there
        if self.since < 0 or self.since + len(win) > (self.W + 1) * self.H + 3:
            return None
        return list(rules[:len(win) - 1])

    def grid(self):
        if self.since < 0 or self.since >= (self.W + 1) * self.H:
            return None
        return (self.n - self.since, self.W, self.vis[0], self.vis[1])

    def window_rules(self, n):
        T = self.since
        if T < 0:
            raise ValueError("Emu3 grammar: no image token in the context")
        base = (self.W + 1) * self.H
        forced = [-1] * n
        for j in range(n):
            pos = T + 1 + j
            if pos % (self.W + 1) == 0:
                forced[j] = self.eol
            if pos % (base + 1) == 0:
                forced[j] = self.eof
            if pos % (base + 2) == 0:
                forced[j] = self.eoi
            if pos % (base + 3) == 0:
                forced[j] = self.eos
        if T + n > base + 3:                               # JE:118-123, python slice semantics kept
            for j in range(n)[base + 3 - T:]:
                forced[j] = self.pad
        return [ops.make_rule((self.vis,), f, self.top_k, self.top_p, temperature=self.temperature) for f in forced]


class AnoleGrammar(_Grammar):
    """The restricted multimodal modes of the Anole pipeline (JA:178-260): "image-only" (the SJD image path), "interleaved-text-image"
    (the image-window processors without the global suppression: text outside an image, image ids inside) and "text-only" (image ids and
    the begin / end-of-image tokens suppressed).  Every window row gets the mask of the ACCEPTED prefix (the reference's 3d processors
    use input_ids.shape[1] with no per-row offset)."""

    def __init__(self, vocab_size, prompt_len, max_length, image_seq_length, boi=8197, eoi=8196, eos=2, img_lo=4,
                 img_hi=8196, top_k=2000, mode="image-only"):
        if mode not in ("image-only", "interleaved-text-image", "text-only"):
            raise ValueError(f"AnoleGrammar mode {mode!r}")
        if mode != "image-only" and not (img_hi == eoi and boi == eoi + 1 and vocab_size):
            raise NotImplementedError("text rows need the Chameleon id layout (image ids, <eoi>, <boi> adjacent) and the vocabulary size")
        self.V, self.prompt_len, self.max_length, self.L = vocab_size, prompt_len, max_length, image_seq_length
        self.boi, self.eoi, self.eos, self.img_lo, self.img_hi, self.top_k = boi, eoi, eos, img_lo, img_hi, top_k
        self.mode = mode
        self.reset()

    def reset(self):
        self.ctx = []
        self.boi_at = []            # indices of the <boi> tokens in ctx, ascending: "is there one in the last L tokens" without a scan

    def _snapshot(self):
        return len(self.ctx)

    def _restore(self, s):
        del self.ctx[s:]
        while self.boi_at and self.boi_at[-1] >= s:
            self.boi_at.pop()

    def _advance(self, t):
        if t == self.boi:
            self.boi_at.append(len(self.ctx))
        self.ctx.append(t)

    def fast_residual_rules(self, win, rules):
        # inside an image (a <boi> among the last L tokens, none exactly L + 1 back) every special token is suppressed and only image ids
        # are allowed, whatever the position: as long as that holds for every context length a residual rule is evaluated at
        # (cur .. cur + n - 2) and no draft is a <boi>, all of them are the window's rule
        cur, n, L = len(self.ctx), len(win), self.L
        if self.mode == "text-only":                 # a static mask
            return list(rules[:n - 1]) if n >= 2 else None
        if not self.boi_at or n < 2:
            return None
        b = self.boi_at[-1]
        if b < cur - min(L, cur) or cur + n - 2 > b + L:
            return None
        if len(self.boi_at) > 1 and self.boi_at[-2] + L + 1 >= cur:
            return None
        for t in win[1:-1]:
            if t == self.boi:
                return None
        return list(rules[:n - 1])

    def _allowed(self):
        ctx, cur, L = self.ctx, len(self.ctx), self.L
        offset = L + 1
        at_offset = cur >= offset and ctx[-offset] == self.boi
        window = min(L, cur)
        in_window = bool(self.boi_at) and self.boi_at[-1] >= cur - window and window > 0
        specials = {self.eos, self.boi, self.eoi}
        allowed = set()
        for t in specials:
            ok = True
            if at_offset and t != self.eoi:          # 1. only eoi at the offset
                ok = False
            if (not at_offset) and t == self.eoi:    # 1. eoi nowhere else
                ok = False
            if in_window:                            # 2. inside the image window only image ids
                ok = False
            if t == self.boi and not (self.max_length - L - 1 > cur):   # 3.
                ok = False
            if t == self.eos and self.mode == "image-only" and self.prompt_len <= cur <= self.prompt_len + 1:   # 5. (image-only)
                ok = False
            if ok:
                allowed.add(t)
        img_ok = in_window and not at_offset         # 2. image ids only inside the window; 1. none at the offset
        return img_ok, sorted(allowed)

    def window_rules(self, n):
        if self.mode == "text-only":                 # JA:178-189: everything but image ids, <boi>, <eoi>
            r = ops.make_rule(((0, self.img_lo), (self.boi + 1, self.V)) if self.img_lo > 0 else ((self.boi + 1, self.V),), -1, self.top_k,
                              self.top_p, temperature=self.temperature)
            return [r for _ in range(n)]
        img_ok, specials = self._allowed()
        ranges = []
        pts = [(t, t + 1) for t in specials]
        if img_ok:
            pts.append((self.img_lo, self.img_hi))
        if self.mode == "interleaved-text-image":    # no global suppression: outside an image window every text id is allowed as well
            cur, L = len(self.ctx), self.L
            at_offset = cur >= L + 1 and self.ctx[-(L + 1)] == self.boi
            in_window = bool(self.boi_at) and self.boi_at[-1] >= cur - min(L, cur) and min(L, cur) > 0
            if not at_offset and not in_window:
                pts += [(0, self.img_lo), (self.boi + 1, self.V)] if self.img_lo > 0 else [(self.boi + 1, self.V)]
        pts.sort()
        for lo, hi in pts:                           # merge adjacent intervals
            if ranges and lo <= ranges[-1][1]:
                ranges[-1] = (ranges[-1][0], max(hi, ranges[-1][1]))
            else:
                ranges.append((lo, hi))
        if not ranges:
            raise ValueError("Anole grammar masks every token")
        r = ops.make_rule(ranges, -1, self.top_k, self.top_p, temperature=self.temperature)
        return [r for _ in range(n)]
