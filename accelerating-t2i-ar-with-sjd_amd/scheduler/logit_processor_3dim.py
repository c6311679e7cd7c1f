"""Mirror of reference scheduler/logit_processor_3dim.py: same class names, constructor arguments and grammar semantics.

In the reference these objects mutate a [B, L, V] score tensor with ATen ops (and 4+ host syncs per call).  Here they are
*descriptors*: `_sample` turns the processor list into an integer grammar (sjd_amd/grammar.py) whose per-row rules are
applied inside kernels K2/K4.  Calling a processor directly is therefore not part of the hot path and raises.
"""
import math
from typing import List, Optional

import torch

from .. import grammar as G


def check_eol_in_multitokens(tokenlen, new_pred_tokenlen, line_len):
    """Behaviour of reference logit_processor_3dim.py:25-29: does any of the next `new_pred_tokenlen` positions
    (tokenlen+1 .. tokenlen+new_pred_tokenlen) fall on a multiple of `line_len`?"""
    first_multiple = -(-(tokenlen + 1) // line_len) * line_len
    return first_multiple <= tokenlen + new_pred_tokenlen


def eol_positions_in_multitokens(tokenlen, new_pred_tokenlen, line_len):
    """window rows forced by get_eol_in_multitokens (reference :31-43)."""
    return [j for j in range(new_pred_tokenlen) if (tokenlen + 1 + j) % line_len == 0]


def apply_row_rules(scores, rules, filter_value=-float("inf"), forced_fill=-float("inf")):
    """The row rules of a window (sjd_row_rule per row, what kernels K2 / K4 apply) on a [..., L, V] score tensor, with ATen ops and in the
    kernels' order: forced row -> (-inf, ..., 0 at the forced id, ...) (LP:31-43); allowed id ranges -> everything else `filter_value`; top-k (ties
    kept: scores < k-th largest go, LP:190-204); temperature; top-p (HF's rule, LP:355-419).  This is the processors' direct-call form -- a user
    calling a processor object on a tensor as the reference's objects allow -- not the hot path."""
    out = scores.clone()
    V = out.shape[-1]
    for r, rule in enumerate(rules):
        row = out[..., r, :]
        if rule.forced >= 0:
            row.fill_(forced_fill)
            row[..., rule.forced] = 0
            continue
        if rule.n_ranges > 0:
            keep = torch.zeros(V, dtype=torch.bool, device=out.device)
            for i in range(rule.n_ranges):
                keep[rule.lo[i]:rule.hi[i]] = True
            row.masked_fill_(~keep, filter_value)
        if rule.top_k > 0:
            kth = torch.topk(row, min(int(rule.top_k), V))[0][..., -1, None]
            row.masked_fill_(row < kth, filter_value)
        if rule.temperature != 1.0:
            row.div_(rule.temperature)
        if rule.top_p_thr >= 0:                    # top_p_thr = 1 - top_p: ascending cumulative probabilities <= it are removed, the top token stays
            srt, idx = torch.sort(row, descending=False)
            cum = srt.softmax(dim=-1).cumsum(dim=-1)
            rm = cum <= rule.top_p_thr
            rm[..., -1:] = False
            row.masked_fill_(rm.scatter(-1, idx, rm), filter_value)
    return out


class _Descriptor:
    """A logits processor as a DESCRIPTOR: `_sample` turns the processor list into an integer grammar whose row rules kernels K2 / K4 apply.
    Called on tensors directly (reference usage outside generate()), the processors whose grammar is self-contained -- the Lumina, Emu3 and
    top-k / top-p / temperature ones -- evaluate the same rules with ATen ops (apply_row_rules); the Anole single-purpose processors, which only have a
    grammar as the LIST jacobi_iteration_anhole builds, carry their own tensor form (_drop_ids)."""

    def _solo_rules(self, ctx, n):
        gr = grammar_from_processors([self])
        gr.start(ctx)
        return gr.window_rules(n)

    def __call__(self, input_ids, scores):
        three = scores.dim() >= 3
        sc = scores if three else scores.unsqueeze(-2)
        try:
            rules = self._solo_rules(input_ids[0].tolist(), sc.shape[-2])
        except NotImplementedError as e:
            raise RuntimeError(f"{type(self).__name__} has no grammar of its own (it is one entry of the processor LIST the kernels' grammar is "
                               f"built from, scheduler/jacobi_iteration_anhole.py): {e}") from None
        fv = getattr(self, "filter_value", -float("inf"))
        if fv == "finfo.min":                     # Emu3's processor removes with the dtype's most negative finite value, forced rows included (JE:80-123)
            fv = torch.finfo(sc.dtype).min
            out = apply_row_rules(sc, rules, fv, forced_fill=fv)
        else:
            out = apply_row_rules(sc, rules, fv)
        return out if three else out.squeeze(-2)


class MultiTokensVLLogitsProcessor(_Descriptor):
    def __init__(self, image_start_token_id=None, image_end_token_id=None, image_next_line_token_id=None, patch_size=None,
                 voc_size=None, device="cpu"):
        self.image_start_token_id = image_start_token_id
        self.image_end_token_id = image_end_token_id
        self.image_next_line_token_id = image_next_line_token_id
        self.patch_size, self.voc_size = patch_size, voc_size
        self.h_latent_dim = self.w_latent_dim = None
        self.image_token_range = (4, 8196)           # reference :65


class MultiTokensInterleavedTopKLogitsWarper(_Descriptor):
    def __init__(self, image_top_k: int, text_top_k: int, image_start_token_id=None, image_end_token_id=None,
                 filter_value: float = -float("Inf"), min_tokens_to_keep: int = 1):
        if not isinstance(text_top_k, int) or text_top_k <= 0:
            raise ValueError(f"`text_top_k` has to be a strictly positive integer, but is {text_top_k}")
        if not isinstance(image_top_k, int) or text_top_k <= 0:
            raise ValueError(f"`image_top_k` has to be a strictly positive integer, but is {image_top_k}")
        self.image_top_k = max(image_top_k, min_tokens_to_keep)
        self.text_top_k = max(text_top_k, min_tokens_to_keep)
        self.image_start_token_id, self.image_end_token_id = image_start_token_id, image_end_token_id
        self.filter_value = filter_value

    def _solo_rules(self, ctx, n):
        """LP:190-204: top-k alone, k by whether an image is open (one more <start> than <end> in the context)"""
        from .. import ops
        in_image = ctx.count(self.image_start_token_id) == ctx.count(self.image_end_token_id) + 1
        return [ops.make_rule(top_k=self.image_top_k if in_image else self.text_top_k)] * n


class TopPLogitsWarper3d(_Descriptor):
    def __init__(self, top_p: float, filter_value: float = -float("Inf"), min_tokens_to_keep: int = 1):
        top_p = float(top_p)
        if top_p < 0 or top_p > 1.0:
            raise ValueError(f"`top_p` has to be a float > 0 and < 1, but is {top_p}")
        if not isinstance(min_tokens_to_keep, int) or (min_tokens_to_keep < 1):
            raise ValueError(f"`min_tokens_to_keep` has to be a positive integer, but is {min_tokens_to_keep}")
        self.top_p, self.min_tokens_to_keep = top_p, min_tokens_to_keep


class TopKLogitsWarper(_Descriptor):
    """Stand-in for transformers' TopKLogitsWarper (same attribute); the HF object itself is accepted as well."""

    def __init__(self, top_k: int, filter_value: float = -float("Inf"), min_tokens_to_keep: int = 1):
        self.top_k = max(int(top_k), min_tokens_to_keep)


class TopPLogitsWarper(_Descriptor):
    """Stand-in for transformers' TopPLogitsWarper (same attributes; the HF object is accepted as well): what HF's generate() appends behind
    the user's processors, the temperature and top-k when GenerationConfig.top_p < 1.  Same rule as TopPLogitsWarper3d (LP:207-250 is HF's
    warper with a window axis)."""

    def __init__(self, top_p: float, filter_value: float = -float("Inf"), min_tokens_to_keep: int = 1):
        top_p = float(top_p)
        if top_p < 0 or top_p > 1.0:
            raise ValueError(f"`top_p` has to be a float > 0 and < 1, but is {top_p}")          # HF's own check
        if not isinstance(min_tokens_to_keep, int) or min_tokens_to_keep != 1:
            raise NotImplementedError("min_tokens_to_keep != 1 (the kernels keep the top token, as HF's default does)")
        self.top_p, self.min_tokens_to_keep = top_p, min_tokens_to_keep


class TemperatureLogitsWarper(_Descriptor):
    """Stand-in for transformers' TemperatureLogitsWarper (same attribute; the HF object is accepted as well): scores / temperature, which
    HF's generate() appends behind the user's processors when GenerationConfig.temperature != 1"""

    def __init__(self, temperature: float):
        if not isinstance(temperature, (int, float)) or not (temperature > 0):
            raise ValueError(f"`temperature` (={temperature}) has to be a strictly positive float")        # HF's own check
        self.temperature = float(temperature)

    def _solo_rules(self, ctx, n):
        from .. import ops
        return [ops.make_rule(temperature=self.temperature)] * n


def _drop_ids(scores, ids, keep_only, fired):
    """Anole's single-purpose processors on a [B, (L,) V] tensor (LP:242-256, 280-286, 323-338): per batch row, where `fired` holds (bool [B]),
    either everything EXCEPT `ids` (keep_only) or exactly `ids` is set to the dtype's most negative finite value; the same mask on every window row"""
    sel = torch.zeros(scores.shape[-1], dtype=torch.bool, device=scores.device)
    sel[torch.as_tensor(ids, dtype=torch.long, device=scores.device)] = True
    col = ~sel if keep_only else sel
    hit = fired.to(scores.device).view(-1, *([1] * (scores.dim() - 1))) & col
    return scores.masked_fill(hit, torch.finfo(scores.dtype).min)


class AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d(_Descriptor):
    def __init__(self, trigger_token_id: int, allowed_token_ids: List[int], offset: int, exclusive: bool = False, device="cpu"):
        self.trigger_token_id, self.allowed_token_ids, self.offset, self.exclusive = trigger_token_id, list(allowed_token_ids), offset, exclusive

    def __call__(self, input_ids, scores):
        """called alone (LP:242-256): `offset` tokens behind the trigger only the allowed ids stay; with `exclusive` those ids go everywhere else"""
        B, T = input_ids.shape
        fired = input_ids[:, -self.offset] == self.trigger_token_id if T >= self.offset else torch.zeros(B, dtype=torch.bool, device=input_ids.device)
        out = _drop_ids(scores, self.allowed_token_ids, True, fired)
        return _drop_ids(out, self.allowed_token_ids, False, ~fired) if self.exclusive else out


class AllowOnlyTokensInRelativeWindowLogitsProcessor3d(_Descriptor):
    def __init__(self, trigger_token_id: int, allowed_token_ids: List[int], window_width: int, exclusive: bool = False, device="cpu"):
        self.trigger_token_id, self.allowed_token_ids, self.window_width, self.exclusive = trigger_token_id, list(allowed_token_ids), window_width, exclusive

    def __call__(self, input_ids, scores):
        """called alone (LP:323-338): while the trigger lies among the last `window_width` tokens only the allowed ids stay"""
        w = min(self.window_width, input_ids.shape[1])
        fired = (input_ids[:, -w:] == self.trigger_token_id).any(dim=1)
        out = _drop_ids(scores, self.allowed_token_ids, True, fired)
        return _drop_ids(out, self.allowed_token_ids, False, ~fired) if self.exclusive else out


class SuppressTokensInIndexRangeLogitsProcessor3d(_Descriptor):
    def __init__(self, suppress_tokens: List[int], start_index: int, end_index: Optional[int] = None, device="cpu"):
        self.suppress_tokens, self.start_index = list(suppress_tokens), start_index
        self.end_index = end_index if end_index is not None else math.inf

    def __call__(self, input_ids, scores):
        """called alone (LP:280-286): the listed ids go while start_index <= context length <= end_index"""
        T = input_ids.shape[1]
        on = self.start_index <= T <= self.end_index
        return _drop_ids(scores, self.suppress_tokens, False, torch.full((input_ids.shape[0],), bool(on), dtype=torch.bool, device=input_ids.device))


class SuppressTokensAtBeginLogitsProcessor3d(SuppressTokensInIndexRangeLogitsProcessor3d):
    def __init__(self, begin_suppress_tokens, begin_index, device="cpu"):
        super().__init__(begin_suppress_tokens, begin_index, begin_index + 1, device=device)
        self.begin_index = begin_index


class SuppressTokensLogitsProcessor3d(SuppressTokensInIndexRangeLogitsProcessor3d):
    def __init__(self, suppress_tokens, device="cpu"):
        super().__init__(suppress_tokens, 0, device=device)


def get_double_cfg_input_ids(input_ids, neg_input_ids, pad_category):
    """Behaviour of reference :422-440: [pos; neg] stacked on the batch axis, both right-aligned in a pad-filled frame."""
    width = max(input_ids.shape[1], neg_input_ids.shape[1])
    rows = [torch.nn.functional.pad(t, (width - t.shape[1], 0), value=pad_category) for t in (input_ids, neg_input_ids)]
    return torch.cat(rows, dim=0)


def grammar_from_processors(processors, prompt_len=None, max_length=None, vocab_size=None):
    """LogitsProcessorList -> integer grammar driving kernels K2/K4.  Raises for processors this engine does not know."""
    procs = list(processors)
    temps = [p for p in procs if type(p).__name__ == "TemperatureLogitsWarper"]
    if temps:                     # scale-invariant with respect to every other processor here (masks, top-k): one scalar of the rules
        if len(temps) > 1:
            raise NotImplementedError("more than one TemperatureLogitsWarper in the processor list")
        # K2 / K4 apply mask -> top-k -> temperature -> top-p.  Masks and top-k commute with a positive scale; top-p does NOT: HF evaluates a
        # top-p warper that stands BEFORE the temperature warper at T = 1 (LlamaGen's TopK + TopPLogitsWarper3d with hf_generate's appended
        # temperature warper) -- a different kept set.  Refuse that order instead of silently scaling first.
        ti = procs.index(temps[0])
        if float(temps[0].temperature) != 1.0 and any(type(p).__name__ in ("TopPLogitsWarper", "TopPLogitsWarper3d") and float(getattr(p, "top_p", 1.0)) < 1.0
                                                      for p in procs[:ti]):
            raise NotImplementedError("a top-p warper ahead of TemperatureLogitsWarper(temperature != 1): the kernels apply the temperature before top-p")
        g = grammar_from_processors([p for p in procs if p is not temps[0]], prompt_len=prompt_len, max_length=max_length, vocab_size=vocab_size)
        g.temperature = float(temps[0].temperature)
        return g
    tops = [p for p in procs if type(p).__name__ == "TopPLogitsWarper"]
    if tops:                      # HF's top-p warper behind the family's own processors: one more scalar of every rule (K2 / K4 apply it last)
        if len(tops) > 1 or any(type(p).__name__ == "TopPLogitsWarper3d" for p in procs):
            raise NotImplementedError("more than one top-p warper in the processor list")
        if int(getattr(tops[0], "min_tokens_to_keep", 1)) != 1:
            raise NotImplementedError("TopPLogitsWarper(min_tokens_to_keep != 1)")
        g = grammar_from_processors([p for p in procs if p is not tops[0]], prompt_len=prompt_len, max_length=max_length, vocab_size=vocab_size)
        if float(tops[0].top_p) < 1.0:
            g.top_p = float(tops[0].top_p)
        return g
    names = [type(p).__name__ for p in procs]
    if len(procs) >= 1 and isinstance(procs[0], MultiTokensVLLogitsProcessor):
        vl = procs[0]
        tk = next((p for p in procs[1:] if isinstance(p, MultiTokensInterleavedTopKLogitsWarper)), None)
        extra = [p for p in procs[1:] if p is not tk]
        if extra:
            raise NotImplementedError(f"unsupported logits processors after the Lumina grammar: {[type(p).__name__ for p in extra]}")
        return G.LuminaGrammar(image_top_k=tk.image_top_k if tk else 0, text_top_k=tk.text_top_k if tk else 0,
                               image_start_token_id=vl.image_start_token_id, image_end_token_id=vl.image_end_token_id,
                               image_next_line_token_id=vl.image_next_line_token_id, img_lo=vl.image_token_range[0],
                               img_hi=vl.image_token_range[1])
    if all(n in ("TopKLogitsWarper", "TopPLogitsWarper3d") for n in names) and names:
        k = next((int(p.top_k) for p in procs if type(p).__name__ == "TopKLogitsWarper"), 0)
        p_ = next((float(p.top_p) for p in procs if type(p).__name__ == "TopPLogitsWarper3d"), 1.0)
        return G.TopKTopPGrammar(k, p_)
    if names and names[0] == "EOLLogitProcessor3d":
        h = procs[0]
        k = next((int(p.top_k) for p in procs[1:] if type(p).__name__ == "TopKLogitsWarper"), 0)
        vis = list(h.visual_tokens)
        if vis != list(range(vis[0], vis[0] + len(vis))):
            raise NotImplementedError("Emu3 visual token ids must form one contiguous range")
        return G.Emu3Grammar(h.height, h.width, vis[0], len(vis), h.img_token, h.eoi_token, h.eos_token, h.eol_token,
                             h.eof_token, h.pad_token, top_k=k)
    if names and names[0] == "SuppressTokensLogitsProcessor3d" and all(n == "TopKLogitsWarper" for n in names[1:]):
        # Anole "text-only" (JA:178-189): image ids + <boi> + <eoi> suppressed -- the Chameleon id layout (image ids, <eoi>, <boi> adjacent)
        sup = sorted(procs[0].suppress_tokens)
        if len(sup) < 3 or sup != list(range(sup[0], sup[0] + len(sup))):
            raise NotImplementedError("text-only suppression must be one contiguous id range (image ids, <eoi>, <boi>)")
        if vocab_size is None:
            raise NotImplementedError("the text-only grammar needs the vocabulary size (grammar_from_processors(..., vocab_size=))")
        k = next((int(p.top_k) for p in procs[1:]), 0)
        return G.AnoleGrammar(vocab_size=vocab_size, prompt_len=prompt_len or 0, max_length=max_length or 0, image_seq_length=1,
                              boi=sup[-1], eoi=sup[-2], img_lo=sup[0], img_hi=sup[-2], top_k=k, mode="text-only")
    if names and names[0] == "AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d" and "SuppressTokensAtBeginLogitsProcessor3d" not in names:
        # Anole "interleaved-text-image" (JA:233-260): the image-window processors without the global suppression
        at, win = procs[0], procs[1]
        rng = next(p for p in procs if type(p) is SuppressTokensInIndexRangeLogitsProcessor3d)
        extra = [n for n in names[2:] if n not in ("SuppressTokensInIndexRangeLogitsProcessor3d", "TopKLogitsWarper")]
        if extra:
            raise NotImplementedError(f"unsupported logits processors in the interleaved Anole list: {extra}")
        if vocab_size is None:
            raise NotImplementedError("the interleaved grammar needs the vocabulary size (grammar_from_processors(..., vocab_size=))")
        k = next((int(p.top_k) for p in procs if type(p).__name__ == "TopKLogitsWarper"), 0)
        img = sorted(win.allowed_token_ids)
        L_img = win.window_width
        return G.AnoleGrammar(vocab_size=vocab_size, prompt_len=prompt_len or 0, max_length=rng.start_index + L_img + 1, image_seq_length=L_img,
                              boi=at.trigger_token_id, eoi=at.allowed_token_ids[0], img_lo=img[0], img_hi=img[-1] + 1, top_k=k,
                              mode="interleaved-text-image")
    if names and names[0] == "AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d":
        at, win = procs[0], procs[1]
        begin = next(p for p in procs if isinstance(p, SuppressTokensAtBeginLogitsProcessor3d))
        rng = next(p for p in procs if type(p) is SuppressTokensInIndexRangeLogitsProcessor3d)
        k = next((int(p.top_k) for p in procs if type(p).__name__ == "TopKLogitsWarper"), 0)
        img = sorted(win.allowed_token_ids)
        L_img = win.window_width
        return G.AnoleGrammar(vocab_size=None, prompt_len=begin.begin_index, max_length=rng.start_index + L_img + 1,
                              image_seq_length=L_img, boi=at.trigger_token_id, eoi=at.allowed_token_ids[0],
                              eos=begin.suppress_tokens[0], img_lo=img[0], img_hi=img[-1] + 1, top_k=k)
    raise NotImplementedError(f"no kernel grammar for logits processors {names}")
