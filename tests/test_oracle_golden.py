"""Pins the CPU oracle (oracle/) against the golden vectors produced by importing the reference
(tests/golden/make_golden.py).  CPU only.

Tolerances: token ids / accept lengths bit-exact; probabilities |dp| <= 1e-6 + 1e-5*p (the oracle's
canonical softmax keeps 4096 partial sums, torch's vectorised sum over V terms carries ~sqrt(V)*eps
relative error; north_star asks logits within 1e-3).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import sjd_oracle as O
from oracle import loop as OL
from oracle.attention_ref import OracleWindowAttention
from tests.helpers import (llamagen_forward_fn, lumina_forward_fn, make_llamagen, make_chameleon, make_pq)

P_ATOL, P_RTOL = 1e-6, 1e-5


def load(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name))
    return d, json.loads(str(d["meta"]))


def gen_inputs(m, key_logits="logits_seed", key_noise="noise_seed", ctx_gen=None):
    g = torch.Generator().manual_seed(m[key_logits]) if ctx_gen is None else ctx_gen
    logits = torch.randn(2, m["nrows"], m["V"], generator=g) * m["logits_scale"]
    noise = torch.empty(m["nrows"], m["V"]).exponential_(generator=torch.Generator().manual_seed(m[key_noise]))
    return logits.numpy(), noise.numpy()


def check_probs(d, name, toks, probs, cols, exact_support=True):
    assert toks.tolist() == d[f"{name}.tokens"][0].tolist()
    if exact_support:
        assert ((probs > 0).sum(-1) == d[f"{name}.nnz"]).all()
    else:       # sharpened distributions: torch's exp keeps denormals down to e^-103, the canonical exp is 0 below e^-87
        assert ((probs > 0).sum(-1) <= d[f"{name}.nnz"]).all() and ((probs > 0).sum(-1) >= 1).all()
    np.testing.assert_allclose(probs.max(-1), d[f"{name}.pmax"], atol=P_ATOL, rtol=P_RTOL)
    np.testing.assert_allclose(probs[:, cols], d[f"{name}.p_cols"], atol=P_ATOL, rtol=P_RTOL)
    np.testing.assert_allclose(probs.sum(-1), 1.0, atol=1e-5)


def test_canonical_primitives():
    xs = np.linspace(-90, 0, 2001, dtype=np.float32)
    got = np.array([O.expf(x) for x in xs])
    ref = np.exp(xs.astype(np.float64))
    assert np.all(np.abs(got - ref) <= 2e-7 * ref + 1e-45) or np.all(got[xs < -87] == 0)
    ok = xs >= -87
    assert np.max(np.abs(got[ok] - ref[ok]) / ref[ok]) < 3e-7
    assert O.expf(0.0) == 1.0 and O.expf(-1000.0) == 0.0 and O.expf(float("-inf")) == 0.0
    rng = np.random.default_rng(0)
    v = rng.random(65536, dtype=np.float32)
    assert abs(O.canonical_sum(v) - float(v.astype(np.float64).sum())) < 1e-2
    assert O.canonical_sum(np.zeros(5, np.float32)) == 0.0


def test_logits2tokens_lumina(golden_dir):
    d, meta = load(golden_dir, "fn_logits2tokens_lumina.npz")
    cols = d["cols"]
    for m in meta:
        name = m["name"]
        ctx = d[f"{name}.ctx"][0].tolist()
        logits, noise = gen_inputs(m)
        rules = O.lumina_rules(ctx, m["nrows"], m["image_top_k"], m["text_top_k"])
        assert O.lumina_force_no_cfg(ctx) == m["is_force_no_cfg"]
        u = None if m["is_force_no_cfg"] else logits[1]
        toks, probs = O.logits_to_probs_sample(logits[0], u, m["guidance_scale"], rules, noise)
        check_probs(d, name, toks, probs, cols)
        np.testing.assert_allclose(probs[np.arange(len(toks)), toks], d[f"{name}.p_at_tok"], atol=P_ATOL, rtol=P_RTOL)


def test_logits2tokens_greedy(golden_dir):
    """do_sample=False (JL:127-129): warpers off (top-k 0), softmax of the processed scores, token = argmax of the scores -- the oracle's
    distribution with unit noise has its mode there"""
    d, meta = load(golden_dir, "fn_greedy.npz")
    for m in meta:
        name = m["name"]
        ctx = d[f"{name}.ctx"][0].tolist()
        logits = (torch.randn(2, m["nrows"], m["V"], generator=torch.Generator().manual_seed(m["logits_seed"])) * m["logits_scale"]).numpy()
        rules = O.lumina_rules(ctx, m["nrows"], 0, 0)
        assert O.lumina_force_no_cfg(ctx) == m["is_force_no_cfg"]
        u = None if m["is_force_no_cfg"] else logits[1]
        _, probs = O.logits_to_probs_sample(logits[0], u, m["guidance_scale"], rules, np.ones((m["nrows"], m["V"]), np.float32))
        z = logits[0] if u is None else (np.float32(m["guidance_scale"]) * (logits[0] - u) + u)
        toks = np.where(probs > 0, z, -np.inf).argmax(-1)
        check_probs(d, name, toks, probs, d["cols"], exact_support=False)      # (no top-k: torch keeps denormal tails the canonical exp flushes)


def test_greedy_mode_of_p_vs_argmax_of_scores():
    """greedy decoding in the engine emits K2's lowest-index MODE of the softmaxed p; the reference (JL:128) and oracle/loop.py take
    torch.argmax of the processed scores.  This pins where the two can differ: only when the top scores are distinct floats whose exponentials
    round to the same fp32 -- a gap below 2^-25, which exists between floats only below |z| = 0.5 -- and that they agree everywhere else, incl.
    one-ulp near-ties at ordinary logit magnitudes (the documented deviation of SJDEngine._k2_out)."""
    V = 64
    rules = O.lumina_rules([8197, 8808, 8808, 5, 6], 1, 0, 0)                    # inside an image: ids 4..8195 allowed
    ones = np.ones((1, 9216), np.float32)

    def mode_and_argmax(z_top, z_next):
        z = np.full((1, 9216), -30.0, np.float32)
        z[0, 40], z[0, 20] = z_top, z_next                                           # the larger score sits at the HIGHER index
        _, p = O.logits_to_probs_sample(z, None, 1.0, rules, ones)
        return int(p[0].argmax()), int(np.where(p[0] > 0, z[0], -np.inf).argmax())

    for top in (3.0, 17.25, 1.0, 0.75):                                              # |z| >= 0.5: an ulp apart is already distinguishable in p
        nxt = np.nextafter(np.float32(top), np.float32(-np.inf))
        assert mode_and_argmax(top, nxt) == (40, 40)
    top = np.float32(0.01)                                                           # |z| < 0.5: one ulp (9e-10) vanishes in exp(z - zmax): p ties,
    nxt = np.nextafter(top, np.float32(-np.inf))                                     # the mode falls on the LOWER index, the score argmax on the larger score
    assert mode_and_argmax(top, nxt) == (20, 40)
    assert mode_and_argmax(top, np.float32(top - 1e-6)) == (40, 40)                  # ... a gap of 1e-6 is seen again


def test_logits2tokens_llamagen(golden_dir):
    d, meta = load(golden_dir, "fn_logits2tokens_llamagen.npz")
    for m in meta:
        logits, noise = gen_inputs(m)
        rules = O.llamagen_rules([], m["nrows"], m["top_k"], m["top_p"])
        toks, probs = O.logits_to_probs_sample(logits[0], logits[1], m["guidance_scale"], rules, noise)
        check_probs(d, m["name"], toks, probs, d["cols"])


def test_emu3_grammar(golden_dir):
    d, meta = load(golden_dir, "fn_emu3_grammar.npz")
    for m in meta:
        name = m["name"]
        ctx = d[f"{name}.ctx"][0].tolist()
        g = torch.Generator().manual_seed(m["ctx_seed"])
        torch.randint(300, 2000, (1, 9), generator=g)
        torch.randint(m["vis_lo"], m["vis_lo"] + m["vis_n"], (1, m["n_after_img"]), generator=g)
        logits, noise = gen_inputs(m, ctx_gen=g)
        rules = O.emu3_rules(ctx, m["nrows"], m["H"], m["W"], m["vis_lo"], m["vis_n"], m["img_token"], m["eoi_token"],
                             m["eos_token"], m["eol_token"], m["eof_token"], m["pad_token"], m["top_k"])
        toks, probs = O.logits_to_probs_sample(logits[0], logits[1], m["guidance_scale"], rules, noise)
        check_probs(d, name, toks, probs, d["cols"])


def test_anole_grammar(golden_dir):
    d, meta = load(golden_dir, "fn_anole_grammar.npz")
    for m in meta:
        name = m["name"]
        ctx = d[f"{name}.ctx"][0].tolist()
        g = torch.Generator().manual_seed(m["logits_seed"])
        torch.randint(8900, 9200, (1, m["prompt_len"]), generator=g)
        if m["n_after_boi"] >= 0:
            torch.randint(4, 8196, (1, m["n_after_boi"]), generator=g)
        logits, noise = gen_inputs(m, ctx_gen=g)
        rules = O.anole_rules(ctx, m["nrows"], m["V"], m["prompt_len"], m["max_length"], m["image_seq_length"],
                              m["boi"], m["eoi"], m["eos"], top_k=m["top_k"])
        toks, probs = O.logits_to_probs_sample(logits[0], logits[1], m["guidance_scale"], rules, noise)
        check_probs(d, name, toks, probs, d["cols"])


def test_anole_text_only_and_interleaved_modes(golden_dir):
    """multimodal_generation_mode 'text-only' / 'interleaved-text-image' (JA:178-189, 233-260): the processor lists the reference's pipeline
    builds, through its sampling_logits2tokens (tests/golden/make_golden.py::gen_fn_anole_modes) -- tokens bit-exact, probabilities within
    tolerance, in and around an image window."""
    d, meta = load(golden_dir, "fn_anole_modes.npz")
    for m in meta:
        name = m["name"]
        ctx = d[f"{name}.ctx"][0].tolist()
        g = torch.Generator().manual_seed(m["logits_seed"])
        torch.randint(8900, 9200, (1, m["prompt_len"]), generator=g)
        if m["n_after_boi"] >= 0:
            torch.randint(4, 8196, (1, m["n_after_boi"]), generator=g)
        if m["tail"]:
            torch.randint(8900, 9200, (1, m["tail"] - 1), generator=g)
        logits, noise = gen_inputs(m, ctx_gen=g)
        rules = O.anole_rules(ctx, m["nrows"], m["V"], m["prompt_len"], m["max_length"], m["image_seq_length"],
                              m["boi"], m["eoi"], m["eos"], top_k=m["top_k"], mode=m["mode"])
        toks, probs = O.logits_to_probs_sample(logits[0], logits[1], m["guidance_scale"], rules, noise)
        check_probs(d, name, toks, probs, d["cols"])


def test_speculative_sampler(golden_dir):
    d, meta = load(golden_dir, "fn_speculative_sampler.npz")
    for m in meta:
        name, V, L = m["name"], m["V"], m["L"]
        p, q, draft = make_pq(V, L, m["pq_seed"], m["mode"])
        assert draft[0].tolist() == d[f"{name}.draft"][0].tolist()
        adv = d[f"{name}.adv_tokens"][0]
        gen = torch.Generator().manual_seed(m["noise_seed"])
        rs = torch.rand((1, L, V), generator=gen)[0].numpy()
        e2 = torch.empty(1, V).exponential_(generator=gen)[0].numpy()
        ctx = d[f"{name}.ctx"][0].tolist()
        win = draft[0].tolist()
        if m["grammar"] == "lumina":
            rfn = lambda c: O.lumina_rules(c, 1, 2000, 10)[0]
        elif m["grammar"] == "llamagen":
            rfn = lambda c: O.llamagen_rules(c, 1, 100, 1.0)[0]
        else:
            rfn = lambda c: O.rule()
        resid = [rfn(ctx + win[1:i]) for i in range(1, L)]
        q_rows = [q[0, i].numpy() for i in range(L)]
        mm, toks, rej = O.verify_accept(win, adv, p[0].numpy(), q_rows, rs, resid, e2)
        assert mm == int(d[f"{name}.first_misaligned"][0]), name
        assert toks.tolist() == d[f"{name}.tokens"][0].tolist(), name
        assert rej == (mm < L) or mm == L
        # implicit one-hot rows (q_rows[i] is None) must behave exactly like materialised one-hot rows
        onehot = [(q[0, i] == 1).any().item() and float(q[0, i].sum()) == 1.0 for i in range(L)]
        q_imp = [None if (onehot[i] and i > 0) else q_rows[i] for i in range(L)]
        mm2, toks2, _ = O.verify_accept(win, adv, p[0].numpy(), q_imp, rs, resid, e2)
        assert (mm2, toks2.tolist()) == (mm, toks.tolist())


def test_temperature_warper_in_the_processor_list(golden_dir):
    """HF's TemperatureLogitsWarper behind the reference's processors (what transformers' generate() does with
    GenerationConfig.temperature != 1): sampling_logits2tokens and the residual resample of SpeculativeSampler, vectors produced by the
    imported reference (tests/golden/make_golden.py::gen_fn_temperature) -- tokens and accept lengths bit-exact, probabilities within
    the usual tolerance.  Also pins the canonical logarithm the residual path uses."""
    xs = np.concatenate([np.float32(10.0) ** np.linspace(-37, 0, 500, dtype=np.float32), np.linspace(0.5, 2.0, 301, dtype=np.float32)])
    got = np.array([O.lib().sjd_o_logf(float(x)) for x in xs], dtype=np.float64)
    ref = np.log(xs.astype(np.float64))
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-6)) < 4e-7 and O.lib().sjd_o_logf(1.0) == 0.0
    d, meta = load(golden_dir, "fn_temperature.npz")
    cols = d["cols"]
    for m in meta:
        name, T = m["name"], m["temperature"]
        ctx = d[f"{name}.ctx"][0].tolist()
        if m["kind"] == "sample":
            logits, noise = gen_inputs(m)
            rules = O.tempered(lambda c, n: O.lumina_rules(c, n, m["image_top_k"], m["text_top_k"]), T)(ctx, m["nrows"])
            toks, probs = O.logits_to_probs_sample(logits[0], logits[1], m["guidance_scale"], rules, noise)
            check_probs(d, name, toks, probs, cols, exact_support=T >= 1.0)
            np.testing.assert_allclose(probs[np.arange(len(toks)), toks], d[f"{name}.p_at_tok"], atol=P_ATOL, rtol=P_RTOL)
            # and it is not the untempered distribution
            _, p1 = O.logits_to_probs_sample(logits[0], logits[1], m["guidance_scale"], O.lumina_rules(ctx, m["nrows"], 2000, 10), noise)
            live = [i for i, r in enumerate(rules) if r.forced < 0]
            assert np.abs(p1[live] - probs[live]).max() > 1e-3
        else:
            V, L = m["V"], m["L"]
            p, q, draft = make_pq(V, L, m["pq_seed"], m["mode"])
            assert draft[0].tolist() == d[f"{name}.draft"][0].tolist()
            adv = d[f"{name}.adv_tokens"][0]
            gen = torch.Generator().manual_seed(m["noise_seed"])
            rs = torch.rand((1, L, V), generator=gen)[0].numpy()
            e2 = torch.empty(1, V).exponential_(generator=gen)[0].numpy()
            win = draft[0].tolist()
            rfn = O.tempered(lambda c, n: O.lumina_rules(c, n, 2000, 10), T)
            resid = [rfn(ctx + win[1:i], 1)[0] for i in range(1, L)]
            q_rows = [q[0, i].numpy() for i in range(L)]
            mm, toks, rej = O.verify_accept(win, adv, p[0].numpy(), q_rows, rs, resid, e2)
            assert mm == int(d[f"{name}.first_misaligned"][0]) and mm < L and rej, name
            assert toks.tolist() == d[f"{name}.tokens"][0].tolist(), name


def test_reguess(golden_dir):
    d, meta = load(golden_dir, "fn_reguess.npz")
    for m in meta:
        torch.manual_seed(m["global_seed"])
        toks = (4 + torch.randint(0, 8192, (1, m["n"]))).numpy()
        assert toks.tolist() == d[f"{m['name']}.tokens"].tolist()


def unpack(d, name, key):
    flat, offs = d[f"{name}.{key}"], d[f"{name}.{key}_offs"]
    return [flat[offs[i]:offs[i + 1]].tolist() for i in range(len(offs) - 1)]


def check_trace(d, name, tr):
    ref_w, ref_s, ref_f = unpack(d, name, "windows"), unpack(d, name, "sampled"), unpack(d, name, "final")
    ref_m = d[f"{name}.matched"].tolist()
    for it in range(min(len(ref_m), len(tr.matched))):
        assert tr.windows[it] == ref_w[it], f"{name} iter {it} window"
        assert tr.sampled[it] == ref_s[it], f"{name} iter {it} sampled"
        assert tr.matched[it] == ref_m[it], f"{name} iter {it} matched"
        assert tr.final[it] == ref_f[it], f"{name} iter {it} final"
    assert len(tr.matched) == len(ref_m)


def test_loop_llamagen(golden_dir):
    d, meta = load(golden_dir, "loop_llamagen.npz")
    for m in meta:
        name = m["name"]
        model = make_llamagen(m["model_args"], m["weight_seed"], m["embed_token_scale"], OracleWindowAttention())
        latent2 = m["latent"] ** 2
        fwd, first_tok = llamagen_forward_fn(model, m["class_id"], m["cfg"], m["top_k"], m["top_p"], latent2,
                                             m["jacobi"]["seed"])
        jac = m["jacobi"]
        cfg = OL.LoopConfig(jacobi_loop_interval_l=jac["jacobi_loop_interval_l"],
                            jacobi_loop_interval_r=jac["jacobi_loop_interval_r"],
                            max_num_new_tokens=jac["max_num_new_tokens"], guidance_scale=jac["guidance_scale"],
                            seed=jac["seed"], do_cfg=jac["do_cfg"],
                            prefix_token_sampler_scheme=jac["prefix_token_sampler_scheme"], max_length=latent2)
        seq, tr = OL.run([first_tok], fwd, lambda c, n: O.llamagen_rules(c, n, m["top_k"], m["top_p"]), cfg,
                         m["model_args"]["vocab_size"])
        check_trace(d, name, tr)
        assert seq[-latent2:] == d[f"{name}.tokens"][0].tolist(), name
        assert len(tr.matched) == m["nfe"]


@pytest.mark.parametrize("fixture,do_sample", [("loop_lumina.npz", True), ("loop_lumina_greedy.npz", False)])
def test_loop_lumina(golden_dir, fixture, do_sample):
    """whole `_sample` loops of the reference, token for token; the greedy fixture runs GenerationConfig(do_sample=False): no multinomial draw,
    the verify step's draws as before (JL:127-129)"""
    d, meta = load(golden_dir, fixture)
    for m in meta:
        name = m["name"]
        model = make_chameleon(m["config"], m["weight_seed"], m["embed_token_scale"], OracleWindowAttention())
        prompt = d[f"{name}.prompt"][0].tolist()
        fwd = lumina_forward_fn(model, len(prompt), m["max_len"] + 32)
        jac = m["jacobi"]
        cfg = OL.LoopConfig(jacobi_loop_interval_l=jac["jacobi_loop_interval_l"],
                            jacobi_loop_interval_r=jac["jacobi_loop_interval_r"],
                            max_num_new_tokens=jac["max_num_new_tokens"], guidance_scale=jac["guidance_scale"],
                            seed=jac["seed"], do_cfg=jac["do_cfg"],
                            prefix_token_sampler_scheme=jac["prefix_token_sampler_scheme"], max_length=m["max_len"],
                            eos_token_ids=(8196,), do_sample=do_sample)
        seq, tr = OL.run(prompt, fwd, lambda c, n: O.lumina_rules(c, n, 2000, 10), cfg, m["config"]["vocab_size"],
                         no_cfg_fn=O.lumina_force_no_cfg)
        check_trace(d, name, tr)
        assert seq == d[f"{name}.sequence"][0].tolist(), name
        assert len(tr.matched) == m["nfe"]


def test_loop_lumina_autoregressive_baseline(golden_dir):
    """The reference's AR baseline -- HF's sampling loop around FlexARInferenceSolver.create_logits_processor's three non-SJD processors (IS:16-270,
    417-450), one token per forward, the unconditional branch a second forward inside the CFG processor -- is the SJD loop with a ONE-token window
    whose unconditional half sees the prompt from the image-start token on (the SJD sampler keeps only the last prompt token, JL:703-712):
    same grammar (image rows / forced end-of-line / end-of-image / text top-k 10 behind the image), same CFG context, one [1, V] multinomial per
    step and nothing else drawn.  Pinned on three reference runs (tests/golden/loop_lumina_ar.npz), token for token, incl. guidance 1 (no CFG) and
    the four text tokens generated behind the image."""
    d, meta = load(golden_dir, "loop_lumina_ar.npz")
    assert len(meta) == 3
    for m in meta:
        name = m["name"]
        model = make_chameleon(m["config"], m["weight_seed"], m["embed_token_scale"], OracleWindowAttention())
        prompt = d[f"{name}.prompt"][0].tolist()
        max_len = len(prompt) + m["max_new_tokens"]
        fwd = lumina_forward_fn(model, len(prompt), max_len + 32, uncond_start=len(prompt) - 3)      # IS:62-63: the context from <image-start> on
        cfg = OL.LoopConfig(jacobi_loop_interval_l=1, jacobi_loop_interval_r=1 << 20, max_num_new_tokens=1, guidance_scale=m["guidance_scale"],
                            seed=m["seed"], do_cfg=True, max_length=max_len, eos_token_ids=(8710,))
        seq, tr = OL.run(prompt, fwd, lambda c, n: O.lumina_rules(c, n, m["image_top_k"], m["text_top_k"]), cfg, m["config"]["vocab_size"],
                         no_cfg_fn=O.lumina_force_no_cfg)
        want = d[f"{name}.sequence"][0].tolist()
        assert seq == want, (name, [i for i, (a, b) in enumerate(zip(seq, want)) if a != b][:4], len(seq), len(want))
        assert len(tr.matched) == len(want) - len(prompt) and all(a == 1 for a in tr.matched[1:])      # one forward per token: the AR cost


def test_top_p_rule_is_hf_top_p_warper():
    """GenerationConfig.top_p reaches the kernels as the rule scalar TopPLogitsWarper3d uses (reference LP:207-250 = HF's TopPLogitsWarper
    with a window axis).  Pinned here against the third-party warper itself (transformers, the version installed): after a top-k of 50 and a
    temperature, the set of tokens the oracle gives a non-zero probability is the set HF's TopKLogitsWarper -> TopPLogitsWarper leaves
    finite, and the probabilities are the softmax over that set."""
    from transformers.generation.logits_process import TopKLogitsWarper, TopPLogitsWarper, TemperatureLogitsWarper
    g = torch.Generator().manual_seed(7)
    for V, top_k, top_p, T in ((4096, 50, 0.8, 1.0), (8192, 2000, 0.95, 0.7), (1000, 0, 0.5, 1.3), (512, 5, 0.999, 1.0)):
        logits = (torch.randn(6, V, generator=g) * 3.0).float()
        s = TemperatureLogitsWarper(T)(None, logits.clone()) if T != 1.0 else logits.clone()
        if top_k:
            s = TopKLogitsWarper(top_k)(None, s)
        s = TopPLogitsWarper(top_p)(None, s)
        want = torch.softmax(s, dim=-1).numpy()
        rules = [O.rule((), -1, top_k, top_p, temperature=T) for _ in range(6)]
        noise = np.ones((6, V), dtype=np.float32)
        toks, probs = O.logits_to_probs_sample(logits.numpy(), None, 1.0, rules, noise)
        assert ((probs > 0) == (want > 0)).all(), (V, top_k, top_p, T)
        np.testing.assert_allclose(probs, want, atol=P_ATOL, rtol=P_RTOL)
