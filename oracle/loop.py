"""CPU restatement of the whole SJD decode loop (reference JacobiSampler._sample,
scheduler/jacobi_iteration_lumina_mgpt.py:912-1249), written as the explicit state machine of
SURVEY.md Appendix B on top of the C oracle functions.

TEST INFRASTRUCTURE ONLY.  The transformer forward is a callback so that the same loop checks
(a) the reference's golden token sequences (CPU fp32 backbone + oracle attention) and
(b) the HIP engine step by step ("teacher forcing": the engine's own logits are replayed here).

RNG contract (verified by logging the reference): per iteration
  torch.randint(0,|img_vocab|,(1,n_fresh))  on the GLOBAL CPU generator   (JL:505)
  exponential_[n,V] on generator g  (== torch.multinomial, JL:118)
  rand[1,n,V] on g                  (JL:260; speculative mode, n>1 only)
  exponential_[1,V] on g            (JL:237; only when a rejection happened)
"""
import random
from dataclasses import dataclass, field

import numpy as np
import torch

from . import sjd_oracle as O


@dataclass
class LoopConfig:
    jacobi_loop_interval_l: int = 1
    jacobi_loop_interval_r: int = (768 // 16) ** 2 + 768 // 16
    max_num_new_tokens: int = 16
    guidance_scale: float = 3.0
    seed: int = 42
    do_cfg: bool = True
    prefix_token_sampler_scheme: str = "speculative_jacobi"
    img_vocab_lo: int = 4          # img_vocab = arange(4, 8196) for every model family
    img_vocab_n: int = 8192
    max_length: int = 1 << 30      # MaxLengthCriteria / MaxlenCriteria
    eos_token_ids: tuple = ()      # EosTokenCriteria looks at the LAST appended token only
    multi_token_init_scheme: str = "random"    # 'repeat_horizon' / 'sample_horizon': spatial draft initialisation (JL:516-594)
    do_sample: bool = True         # False: GenerationConfig(do_sample=False) -- the argmax branch of sampling_logits2tokens (JL:127-129)


@dataclass
class Trace:
    windows: list = field(default_factory=list)
    sampled: list = field(default_factory=list)
    matched: list = field(default_factory=list)
    final: list = field(default_factory=list)
    rejected: list = field(default_factory=list)


def set_seed(seed):
    """reference set_seed (JL:36-45)"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def spatial_init(scheme, fresh, n_known, left_tok, left_mode, grid):
    """Restatement of the spatial draft initialisation of JL:516-594 as intended (the released code raises IndexError at JL:577, so
    there is no reference run to pin this against: PARITY UNPINNED).  The fresh draft at absolute index s = n_known + j lies in image
    column (s - img_start) % (w + 1); with a left neighbour in the same row (column >= 1) that is an image token it repeats that token
    ('repeat_horizon', JL:586-590) or takes the top-1 entry of the distribution the neighbour was drawn from ('sample_horizon',
    JL:485-498, 579-585); the draft distribution is a one-hot either way (JL:583-590), so later fresh drafts chain through it."""
    if scheme == "random" or grid is None:
        return list(fresh)
    img_start, w, lo, hi = grid
    out = []
    for j, rnd in enumerate(fresh):
        s = n_known + j
        src = left_tok if scheme == "repeat_horizon" else left_mode
        ok = s > img_start and (s - img_start) % (w + 1) >= 1 and src is not None and lo <= src < hi
        tok = int(src) if ok else int(rnd)
        out.append(tok)
        left_tok = left_mode = tok
    return out


def run(prompt, forward_fn, rules_fn, cfg: LoopConfig, vocab_size, no_cfg_fn=None, resid_rules_fn=None,
        noise_device="cpu", hook=None, grid_fn=None):
    """prompt: list[int] accepted ids handed to _sample.
    forward_fn(window_ids list[int], kv_len int) -> (logits_c [n,V], logits_u [n,V] or None) float32 numpy; the
        callee owns the KV cache: it must write the n window rows at [kv_len, kv_len+n) and attend causally.
    rules_fn(ctx list[int], n) -> n RowRules for the sampling call (JL:106);
    resid_rules_fn(ctx, 1) -> rule of the residual call (JL:297-306); defaults to rules_fn.
    no_cfg_fn(ctx) -> bool (check_is_force_no_cfg, JL:70-80).
    grid_fn(ctx) -> (img_start, w, img_lo, img_hi) or None: image geometry for the spatial init schemes.
    Returns (sequence list[int], Trace)."""
    resid_rules_fn = resid_rules_fn or rules_fn
    X = [int(t) for t in prompt]
    P = len(X)
    V = vocab_size
    do_cfg = cfg.do_cfg and (cfg.guidance_scale != 1)                    # JL:1005
    gen = None
    if cfg.seed is not None:                                             # JL:1021-1023
        set_seed(cfg.seed)
        gen = torch.Generator(noise_device).manual_seed(cfg.seed)
    l_abs, r_abs = P + cfg.jacobi_loop_interval_l, P + cfg.jacobi_loop_interval_r   # JL:1025
    W = cfg.max_num_new_tokens
    speculative = cfg.prefix_token_sampler_scheme == "speculative_jacobi"
    if not speculative and cfg.prefix_token_sampler_scheme != "jacobi":
        raise ValueError(f"prefix_token_sampler_scheme: {cfg.prefix_token_sampler_scheme}")   # JL:1048

    tr = Trace()
    n = 1                       # output_token_num (JL:1017)
    kv_len = 0                  # rows in the cache that belong to this _sample call
    first = True
    carried_tok, carried_rows = [], []     # additional_tokens / additional_scores (JL:419-420)
    p_last = None               # temporary_collected_scores[:, -1] (JL:390-393); one-hot(prompt[-1]) before iter 0
    cur_len = P
    finished = False
    while not finished:
        # ---- window assembly (prepare_inputs_for_generation_jacobi, JL:606-740) ----
        if first:
            win = list(X)                       # prefill: every prompt token, cache_position = arange(P)
            q_rows = None
            n_fresh = 0
            torch.randint(0, cfg.img_vocab_n, (1, 0))
        else:
            a = min(len(carried_tok), n - 1)    # JL:633-639, 657-662
            n_fresh = n - 1 - a
            fresh = torch.randint(0, cfg.img_vocab_n, (1, n_fresh))[0].tolist()   # GLOBAL generator (JL:505)
            fresh = [cfg.img_vocab_lo + t for t in fresh]                          # img_vocab[rand] (JL:509)
            if cfg.multi_token_init_scheme != "random":
                left_tok = carried_tok[a - 1] if a else X[-1]
                left_row = carried_rows[a - 1] if a else p_last
                fresh = spatial_init(cfg.multi_token_init_scheme, fresh, len(X) + a, left_tok, int(np.argmax(left_row)),
                                     grid_fn(list(X)) if grid_fn is not None else None)
            win = [X[-1]] + carried_tok[:a] + fresh
            q_rows = [p_last] + carried_rows[:a] + [None] * n_fresh              # JL:688-701 (None = one-hot)
        ctx = list(X)
        n_rows = 1 if first else n
        # ---- forward (JL:1107) ----
        logits_c, logits_u = forward_fn(win, kv_len)
        logits_c = logits_c[-n_rows:]
        logits_u = None if logits_u is None else logits_u[-n_rows:]
        force_no_cfg = bool(no_cfg_fn(ctx)) if no_cfg_fn is not None else False
        use_u = logits_u if (do_cfg and not force_no_cfg) else None
        # ---- logits -> probs -> tokens (sampling_logits2tokens, JL:82-132) ----
        rules = rules_fn(ctx, n_rows)
        if cfg.do_sample:
            e1 = torch.empty((n_rows, V), dtype=torch.float32, device=noise_device).exponential_(generator=gen)
            Y, Pn = O.logits_to_probs_sample(logits_c, use_u, cfg.guidance_scale, rules, e1.cpu().numpy())
        else:
            # JL:127-129: no draw (the generator is not touched); probabilities = softmax of the processed scores, token = their argmax -- the scores
            # are the CFG-combined logits wherever the rules leave mass
            e1 = torch.ones((n_rows, V), dtype=torch.float32)
            _, Pn = O.logits_to_probs_sample(logits_c, use_u, cfg.guidance_scale, rules, e1.numpy())
            lc = np.asarray(logits_c, dtype=np.float32)
            if use_u is None:
                z = lc
            else:
                lu = np.asarray(use_u, dtype=np.float32)
                z = (np.float32(cfg.guidance_scale) * (lc - lu)).astype(np.float32) + lu
            Y = np.where(Pn > 0, z, -np.inf).argmax(-1)
        if hook is not None:
            hook("sampled", dict(win=win, logits_c=logits_c, logits_u=use_u, rules=rules, noise=e1, Y=Y, P=Pn))
        tr.sampled.append(Y.tolist())
        # ---- prefix matching (prefix_matching_next_tokens, JL:335-376) ----
        rejected = False
        if n_rows <= 1:                          # is_prefilling_phase = (output_token_num <= 1), JL:1136, 344-350
            m = len(win)
            emitted = [int(Y[-1])]
            p_keep = Pn[-1]
            tail_tok, tail_rows = [], []
        else:
            if speculative:
                rs = torch.rand((1, n_rows, V), dtype=torch.float32, device=noise_device, generator=gen)[0]
                resid = [resid_rules_fn(ctx + win[1:i], 1)[0] for i in range(1, n_rows)]
                # the residual multinomial draws from g only when a rejection happens; clone the state so the
                # draw can be offered unconditionally and committed afterwards
                state = gen.get_state()
                e2 = torch.empty((1, V), dtype=torch.float32, device=noise_device).exponential_(generator=gen)
                m, Yc, rejected = O.verify_accept(win, Y, Pn, q_rows, rs.cpu().numpy(), resid, e2[0].cpu().numpy())
                if not rejected:
                    gen.set_state(state)
                if hook is not None:
                    hook("verified", dict(win=win, Y=Y, P=Pn, q_rows=q_rows, rs=rs, resid=resid, noise2=e2[0], m=m,
                                          Yc=Yc, rejected=rejected))
            else:
                m = O.first_mismatch(win, Y)
                Yc = Y
            emitted = [int(t) for t in Yc[:m]]
            p_keep = Pn[m - 1]                   # row m-1 is never overwritten by an accepted draft row (JL:289)
            tail_tok = [int(t) for t in Yc[m:]]
            tail_rows = [Pn[i] for i in range(m, n_rows)]
        tr.windows.append(list(win))
        tr.matched.append(m)
        tr.final.append(list(emitted))
        tr.rejected.append(rejected)
        # ---- next window length uses cur_len BEFORE this iteration's tokens are appended (JL:1142-1144) ----
        n = min(W, r_abs - cur_len) if (l_abs <= cur_len < r_abs) else 1
        # ---- push forward (JL:378-430): append, KV rollback, carry the unverified tail ----
        X.extend(emitted)
        p_last = p_keep
        kv_len += m                              # rows of win[0..m-1] stay; win[m..] are discarded (JL:401-409)
        if len(win) - m > 0:
            carried_tok, carried_rows = tail_tok, tail_rows
        else:
            carried_tok, carried_rows = [], []
        first = False
        # ---- stopping criteria (JL:1200-1201) ----
        if X[-1] in cfg.eos_token_ids or len(X) >= cfg.max_length:
            finished = True
        cur_len = len(X)
    return X, tr
