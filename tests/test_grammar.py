"""CPU: the product's incremental host grammar (sjd_amd/grammar.py) against the oracle's stateless restatement
(oracle/sjd_oracle.py), which is itself pinned to the reference by tests/test_oracle_golden.py."""
import ctypes
import json
import os
import random

import pytest

import numpy as np

from oracle import sjd_oracle as O
from sjd_amd import grammar as G


import collections
FAST_STATS = collections.defaultdict(lambda: [0, 0])     # grammar class -> [declined, taken]


def same(r1, r2):
    a = bytes(ctypes.string_at(ctypes.byref(r1), ctypes.sizeof(r1)))
    b = bytes(ctypes.string_at(ctypes.byref(r2), ctypes.sizeof(r2)))
    if r1.n_ranges != r2.n_ranges or r1.forced != r2.forced or r1.top_k != r2.top_k or r1.top_p_thr != r2.top_p_thr:
        return False
    return all(r1.lo[i] == r2.lo[i] and r1.hi[i] == r2.hi[i] for i in range(r1.n_ranges))


def check(gr, oracle_fn, ctx, n, win=None):
    gr.start(ctx)
    got, want = gr.window_rules(n), oracle_fn(ctx, n)
    assert len(got) == len(want) and all(same(a, b) for a, b in zip(got, want)), (ctx[-6:], n)
    if win is not None:
        res = gr.residual_rules(win)
        for i in range(1, len(win)):
            assert same(res[i - 1], oracle_fn(ctx + win[1:i], 1)[0])
        assert all(same(a, b) for a, b in zip(gr.window_rules(n), want)), "residual_rules must not change the state"
        if len(win) == n:       # the shortcut the single-graph iteration rests on: either it declines, or it IS the replayed result
            fast = gr.fast_residual_rules(win, gr.window_rules(n))
            FAST_STATS[type(gr).__name__][fast is not None] += 1
            if fast is not None:
                assert len(fast) == len(res) and all(same(a, b) for a, b in zip(fast, res)), (ctx[-6:], win)


def test_lumina_grammar_random_contexts():
    rng = random.Random(0)
    for trial in range(300):
        hg, wg = rng.randint(1, 5), rng.randint(1, 5)
        ctx = [rng.randint(8900, 9100) for _ in range(rng.randint(1, 8))]
        if rng.random() < 0.85:
            ctx += [8197]
            body = [8804 + hg, 8804 + wg] + [rng.randint(4, 8195) for _ in range(rng.randint(0, (2 * wg + 1) * 2 * hg + 3))]
            # the two grid tokens are always generated in the single-token phase (jacobi_loop_interval_l >= 3)
            ctx += body[:rng.randint(2, len(body))]
            if rng.random() < 0.1:
                ctx += [8196] + [rng.randint(8900, 9100) for _ in range(rng.randint(0, 3))]
        n = rng.randint(1, 16)
        # drafts are image ids, EOL or end-of-image: a start token can never be drafted inside an open image
        win = [ctx[-1]] + [rng.choice([rng.randint(4, 8195), 8803, 8196]) for _ in range(n - 1)]
        check(G.LuminaGrammar(2000, 10), lambda c, k: O.lumina_rules(c, k, 2000, 10), ctx, n, win)
        g = G.LuminaGrammar(2000, 10)
        g.start(ctx)
        assert g.force_no_cfg() == O.lumina_force_no_cfg(ctx)


def test_lumina_grammar_incremental_push_equals_restart():
    rng = random.Random(1)
    ctx = [9000, 9001, 8197, 8808, 8808]
    g = G.LuminaGrammar(2000, 10)
    g.start(ctx)
    for step in range(90):
        add = [rng.randint(4, 8195) for _ in range(rng.randint(1, 4))]
        g.push(add)
        ctx += add
        want = O.lumina_rules(ctx, 16, 2000, 10)
        assert all(same(a, b) for a, b in zip(g.window_rules(16), want))


def test_topk_topp_grammar():
    for k, p in [(1000, 1.0), (50, 0.9), (0, 0.5)]:
        check(G.TopKTopPGrammar(k, p), lambda c, n: O.llamagen_rules(c, n, k, p), [1, 2, 3], 7, [3, 4, 5, 6])


def test_emu3_grammar():
    rng = random.Random(2)
    tok = dict(img_token=200, eoi_token=201, eos_token=202, eol_token=203, eof_token=204, pad_token=205)
    for trial in range(200):
        H, W = rng.randint(1, 4), rng.randint(1, 6)
        T = rng.randint(0, (W + 1) * H + 8)
        ctx = [rng.randint(300, 2000) for _ in range(rng.randint(1, 5))] + [200] + [rng.randint(3000, 3100) for _ in range(T)]
        n = rng.randint(1, 16)
        win = [ctx[-1]] + [rng.randint(3000, 3100) for _ in range(n - 1)]
        check(G.Emu3Grammar(H, W, 3000, 8192, **tok), lambda c, k: O.emu3_rules(c, k, H, W, 3000, 8192, **tok), ctx, n, win)


def test_anole_grammar():
    rng = random.Random(3)
    V, Lseq, P, maxlen = 9216, 24, 6, 40
    for trial in range(200):
        ctx = [rng.randint(8900, 9100) for _ in range(P)]
        if rng.random() < 0.9:
            ctx += [8197] + [rng.randint(4, 8195) for _ in range(rng.randint(0, Lseq + 2))]
        n = rng.randint(1, 16)
        win = [ctx[-1]] + [rng.randint(4, 8195) for _ in range(n - 1)]
        try:
            want = O.anole_rules(ctx, n, V, P, maxlen, Lseq)
        except ValueError:
            continue
        check(G.AnoleGrammar(V, P, maxlen, Lseq), lambda c, k: O.anole_rules(c, k, V, P, maxlen, Lseq), ctx, n, None)
        # residual rules (replayed, and the shortcut where it applies) against the oracle evaluated on ctx + win[1:i]
        g = G.AnoleGrammar(V, P, maxlen, Lseq)
        g.start(ctx)
        try:
            want_res = [O.anole_rules(ctx + win[1:i], 1, V, P, maxlen, Lseq)[0] for i in range(1, n)]
        except ValueError:
            continue
        res = g.residual_rules(win)
        assert all(same(a, b) for a, b in zip(res, want_res))
        fast = g.fast_residual_rules(win, g.window_rules(n))
        FAST_STATS["AnoleGrammar"][fast is not None] += 1
        if fast is not None:
            assert len(fast) == len(want_res) and all(same(a, b) for a, b in zip(fast, want_res)), (ctx, win)


@pytest.mark.parametrize("mode", ["interleaved-text-image", "text-only"])
def test_anole_grammar_other_modes(mode):
    """AnoleGrammar(mode=) against the oracle restatement of JA:178-189 / 233-260 (pinned to the reference by fn_anole_modes.npz): text
    before, inside and after an image window, the <eoi> slot, a context with no room left for an image; window and residual rules."""
    rng = random.Random(11)
    V, Lseq, P, maxlen = 9216, 24, 6, 70
    for trial in range(300):
        ctx = [rng.randint(8900, 9100) for _ in range(P + rng.randint(0, 30))]
        if rng.random() < 0.7:
            ctx += [8197] + [rng.randint(4, 8195) for _ in range(rng.randint(0, Lseq))]
            if len(ctx) and rng.random() < 0.3 and ctx.count(8197) and len(ctx) - 1 - ctx.index(8197) == Lseq:
                ctx += [8196] + [rng.randint(8900, 9100) for _ in range(rng.randint(0, 3))]
        n = rng.randint(1, 16)
        win = [ctx[-1]] + [rng.choice([rng.randint(4, 8195), rng.randint(8900, 9100)]) for _ in range(n - 1)]
        ofn = lambda c, k: O.anole_rules(c, k, V, P, maxlen, Lseq, mode=mode)
        check(G.AnoleGrammar(V, P, maxlen, Lseq, mode=mode), ofn, ctx, n, None)
        g = G.AnoleGrammar(V, P, maxlen, Lseq, mode=mode)
        g.start(ctx)
        want_res = [ofn(ctx + win[1:i], 1)[0] for i in range(1, n)]
        assert all(same(a, b) for a, b in zip(g.residual_rules(win), want_res))
        fast = g.fast_residual_rules(win, g.window_rules(n))
        if fast is not None:
            assert len(fast) == len(want_res) and all(same(a, b) for a, b in zip(fast, want_res)), (ctx, win)
    with pytest.raises(ValueError):
        G.AnoleGrammar(V, P, maxlen, Lseq, mode="images")


def test_grammars_on_golden_contexts(golden_dir):
    d = np.load(os.path.join(golden_dir, "fn_logits2tokens_lumina.npz"))
    for m in json.loads(str(d["meta"])):
        ctx = d[f"{m['name']}.ctx"][0].tolist()
        check(G.LuminaGrammar(2000, 10), lambda c, k: O.lumina_rules(c, k, 2000, 10), ctx, m["nrows"])


def test_spatial_init_schemes_copy_the_left_neighbour_inside_an_image_row():
    """multi_token_init_scheme 'repeat_horizon' / 'sample_horizon' (reference JL:516-594): product function and oracle
    restatement agree, first-column drafts and control-token sources stay random, fresh drafts chain."""
    from oracle import sjd_oracle as O
    from oracle.loop import spatial_init
    from sjd_amd.grammar import LuminaGrammar, Emu3Grammar, TopKTopPGrammar, spatial_fresh_tokens
    g = LuminaGrammar(2000, 10)
    ctx = [9000] * 5 + [8197, 8808, 8808] + [100, 101, 102]            # 8 x 8 latent grid, three image tokens so far
    g.start(ctx)
    assert g.grid() == O.lumina_grid(ctx) == (8, 8, 4, 8196)
    fresh = [10, 11, 12, 13, 14, 15, 16, 17]
    # absolute indices 11.. -> columns 3,4,5,6,7,8(EOL slot),0,1
    assert spatial_fresh_tokens("repeat_horizon", fresh, len(ctx), 102, 555, g.grid()) == [102] * 6 + [16, 16]
    assert spatial_fresh_tokens("sample_horizon", fresh, len(ctx), 102, 555, g.grid()) == [555] * 6 + [16, 16]
    assert spatial_fresh_tokens("random", fresh, len(ctx), 102, 555, g.grid()) == fresh
    assert spatial_fresh_tokens("repeat_horizon", fresh, len(ctx), 8803, 8803, g.grid())[:2] == [10, 10]     # a line token is never repeated
    for sch in ("repeat_horizon", "sample_horizon"):
        assert spatial_fresh_tokens(sch, fresh, len(ctx) + 2, 102, 555, g.grid()) == spatial_init(sch, fresh, len(ctx) + 2, 102, 555, O.lumina_grid(ctx))
    g.start(ctx[:6])                                                   # grid tokens not yet known -> plain random init
    assert g.grid() is None and O.lumina_grid(ctx[:6]) is None
    assert TopKTopPGrammar(100, 1.0).grid() is None                    # LlamaGen: no img_width -> the reference silently degrades to random
    e = Emu3Grammar(3, 5, 3000, 8192, 200, 201, 202, 203, 204, 205)
    ectx = [300, 301, 200, 3001, 3002]
    e.start(ectx)
    assert e.grid() == O.emu3_grid(ectx, 3, 5, 3000, 8192, 200) == (3, 5, 3000, 11192)
    with pytest.raises(ValueError):
        spatial_fresh_tokens("repeat_vertical", fresh, len(ctx), 102, 555, g.grid())


def test_fast_residual_rules_are_taken_where_it_matters():
    """the shortcut must actually fire for the image-body windows of a decode (drafts = image ids and forced line tokens) -- otherwise
    every iteration silently falls back to the two-stage launch -- and it must equal the replayed rules there"""
    rng = random.Random(5)
    for trial in range(200):
        hg, wg = rng.randint(1, 6), rng.randint(1, 6)
        n_body = rng.randint(0, (2 * wg + 1) * 2 * hg - 2)
        ctx = [rng.randint(8900, 9100) for _ in range(rng.randint(1, 8))] + [8197, 8804 + hg, 8804 + wg] + [rng.randint(4, 8195) for _ in range(n_body)]
        n = rng.randint(2, 16)
        win = [ctx[-1]] + [rng.choice([rng.randint(4, 8195), 8803]) for _ in range(n - 1)]
        g = G.LuminaGrammar(2000, 10)
        g.start(ctx)
        fast = g.fast_residual_rules(win, g.window_rules(n))
        assert fast is not None
        assert all(same(a, b) for a, b in zip(fast, g.residual_rules(win)))
    assert FAST_STATS["Emu3Grammar"][1] > 0 and FAST_STATS["Emu3Grammar"][0] > 0            # (tiny images: most windows cross the image end)
    g = G.Emu3Grammar(90, 90, 151854, 32768, img_token=151851, eoi_token=151853, eos_token=151850, eol_token=151846, eof_token=151847, pad_token=151643)
    g.start([5, 6, 151851] + [151900] * 4000)
    assert g.fast_residual_rules([151900] * 32, g.window_rules(32)) is not None
    assert FAST_STATS["LuminaGrammar"][1] > 0 and FAST_STATS["LuminaGrammar"][0] > 0        # both branches were compared with the replay
    assert FAST_STATS["AnoleGrammar"][1] > 20 and FAST_STATS["AnoleGrammar"][0] > 20
