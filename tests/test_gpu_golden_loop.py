"""GPU: the HIP engine itself (fp32 backbone on the GPU, exact-fp32 K1 variant, kernels K2-K5) reproduces the token
sequences the REFERENCE produced on CPU (tests/golden/loop_*.npz) bit-exactly.  The noise is drawn from a CPU generator with
the reference's seed (SJDConfig.noise_device="cpu") because the golden runs were CPU runs; everything else is the product
path.  This is BASELINE.json config 0 ("LlamaGen class-conditional ... bit-exact token check") plus the Lumina flavour."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _load(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name))
    return d, json.loads(str(d["meta"]))


def test_llamagen_golden_tokens_on_gpu(dev, golden_dir):
    import sjd_amd.ops as ops
    from sjd_amd.engine import SJDEngine, SJDConfig
    from sjd_amd.frontends import llamagen_window_spec
    from sjd_amd.grammar import TopKTopPGrammar
    from tests.helpers import make_llamagen, llamagen_prefill_sample
    d, meta = _load(golden_dir, "loop_llamagen.npz")
    for m in meta:
        name, jac, N = m["name"], m["jacobi"], m["latent"] ** 2
        model = make_llamagen(m["model_args"], m["weight_seed"], m["embed_token_scale"], ops.HipWindowAttention(n_split=1),
                              dtype=torch.float32, device=dev)
        T = 1
        model.setup_cache(batch=2, s_max=((T + N + 64 + 31) // 32) * 32)
        cond = torch.tensor([m["class_id"], model.num_classes], device=dev)
        zeros = torch.zeros(2, dtype=torch.int32, device=dev)
        logits = model.forward_embeds(model.embed_condition(cond), torch.zeros(2, 1, dtype=torch.long, device=dev), 0, zeros)
        torch.manual_seed(jac["seed"])
        first = int(llamagen_prefill_sample(logits.float().cpu(), m["cfg"], 1.0, m["top_k"], m["top_p"])[0, 0])
        cfg = SJDConfig(jacobi_loop_interval_l=jac["jacobi_loop_interval_l"], jacobi_loop_interval_r=jac["jacobi_loop_interval_r"],
                        max_num_new_tokens=jac["max_num_new_tokens"], guidance_scale=jac["guidance_scale"], seed=jac["seed"],
                        prefix_token_sampler_scheme=jac["prefix_token_sampler_scheme"], max_length=N, noise_device="cpu")
        eng = SJDEngine(model, m["model_args"]["vocab_size"], dev, max_window=jac["max_num_new_tokens"], use_graph=False)
        seq, stats = eng.decode([first], llamagen_window_spec(first, T, dev), TopKTopPGrammar(m["top_k"], m["top_p"]), cfg)
        assert seq[-N:] == d[f"{name}.tokens"][0].tolist(), name
        assert stats.matched == d[f"{name}.matched"].tolist(), name
        assert stats.nfe == m["nfe"]


@pytest.mark.parametrize("fixture,do_sample", [("loop_lumina.npz", True), ("loop_lumina_greedy.npz", False)])
def test_lumina_golden_tokens_on_gpu(dev, golden_dir, fixture, do_sample):
    """the reference's whole-loop traces replayed through the product engine; the greedy fixture = GenerationConfig(do_sample=False): K2's mode
    instead of its draw, no multinomial consumed from the generator (JL:127-129)"""
    import sjd_amd.ops as ops
    from sjd_amd.engine import SJDEngine, SJDConfig
    from sjd_amd.frontends import lumina_window_spec
    from sjd_amd.grammar import LuminaGrammar
    from tests.helpers import make_chameleon
    d, meta = _load(golden_dir, fixture)
    for m in meta:
        name, jac = m["name"], m["jacobi"]
        model = make_chameleon(m["config"], m["weight_seed"], m["embed_token_scale"], ops.HipWindowAttention(n_split=1),
                               dtype=torch.float32, device=dev)
        prompt = d[f"{name}.prompt"][0].tolist()
        model.setup_cache(batch=2, s_max=((m["max_len"] + 64 + 31) // 32) * 32)
        cfg = SJDConfig(jacobi_loop_interval_l=jac["jacobi_loop_interval_l"], jacobi_loop_interval_r=jac["jacobi_loop_interval_r"],
                        max_num_new_tokens=jac["max_num_new_tokens"], guidance_scale=jac["guidance_scale"], seed=jac["seed"],
                        prefix_token_sampler_scheme=jac["prefix_token_sampler_scheme"], max_length=m["max_len"],
                        eos_token_ids=(8196,), noise_device="cpu", do_sample=do_sample)
        eng = SJDEngine(model, m["config"]["vocab_size"], dev, max_window=jac["max_num_new_tokens"], use_graph=False)
        seq, stats = eng.decode(prompt, lumina_window_spec(prompt, dev), LuminaGrammar(2000, 10), cfg)
        assert seq == d[f"{name}.sequence"][0].tolist(), name
        assert stats.matched == d[f"{name}.matched"].tolist(), name


def test_llamagen_solver_generate_reproduces_reference_golden(dev, golden_dir):
    """A16 + A15 + A1 through the PRODUCT entry point: LlamaGenSolver.generate (its own prefill / sample / top_k_top_p_filtering,
    reference LS:34-104, 371-456) on the golden models reproduces the reference's whole token sequences -- first token included --
    for all five golden runs (plain Jacobi and top-p 0.95 among them).  noise_device="cpu": the golden runs were CPU runs."""
    import sjd_amd.ops as ops
    from llamagen.llamagen_solver import LlamaGenSolver, renew_llamagen
    from scheduler.jacobi_iteration_lumina_mgpt import renew_sampler
    from tests.helpers import make_llamagen
    d, meta = _load(golden_dir, "loop_llamagen.npz")
    assert len(meta) == 5
    for m in meta:
        name, jac, N = m["name"], m["jacobi"], m["latent"] ** 2
        model = make_llamagen(m["model_args"], m["weight_seed"], m["embed_token_scale"], ops.HipWindowAttention(n_split=1),
                              dtype=torch.float32, device=dev)
        model.__class__ = renew_llamagen(model.__class__)                      # test_llamagen.py:85-88
        model._init_new_params(**jac)
        model.__class__ = renew_sampler(model.__class__)
        model._init_new_params(**jac)
        model.sjd_use_graph = False
        solver = LlamaGenSolver(model=model, image_top_k=m["top_k"], image_top_p=m["top_p"], noise_device="cpu")
        torch.manual_seed(jac["seed"])                                         # the golden driver seeds the global generator first
        toks = solver.generate(torch.tensor([m["class_id"]], device=dev), N, None, cfg_scale=m["cfg"], temperature=1.0,
                               top_k=m["top_k"], top_p=m["top_p"], sample_logits=True)
        assert toks.shape == (1, N)
        assert toks[0].tolist() == d[f"{name}.tokens"][0].tolist(), name      # incl. the first token (product sampler, LS:75-84)
        assert model.last_sjd_stats.matched == d[f"{name}.matched"].tolist(), name
        assert model.last_sjd_stats.nfe == m["nfe"]


def _max_logit_gap(rec_gpu, rec_cpu):
    assert len(rec_gpu) == len(rec_cpu)
    worst = 0.0
    for g, c in zip(rec_gpu, rec_cpu):
        for a, b in ((g[0], c[0]), (g[1], c[1])):
            if a is None:
                continue
            n = a.shape[0]
            worst = max(worst, float(np.abs(a - b[-n:]).max()))
    return worst


def test_golden_loop_logits_within_1e3_of_cpu_oracle_forward(dev, golden_dir):
    """north_star: 'bit-exact token IDs; logits within 1e-3'.  The token half is asserted above; here EVERY iteration's fp32 logits of
    the GPU engine (hipBLASLt fp32 GEMMs, exact-fp32 K1) are compared with the CPU oracle forward (torch CPU fp32 + fp64 window
    attention) of the same weights on the same windows: |delta| <= 1e-3 for every row the sampler reads."""
    import sjd_amd.ops as ops
    from oracle import loop as OL
    from oracle import sjd_oracle as O
    from oracle.attention_ref import OracleWindowAttention
    from sjd_amd.engine import SJDEngine, SJDConfig
    from sjd_amd.frontends import lumina_window_spec, llamagen_window_spec
    from sjd_amd.grammar import LuminaGrammar, TopKTopPGrammar
    from tests.helpers import make_chameleon, make_llamagen, lumina_forward_fn, llamagen_forward_fn
    TOL = 1e-3
    report = {}

    def rec_hook(store):
        def hook(dd):
            store.append((dd["logits_c"].float().cpu().numpy().copy(), None if dd["logits_u"] is None else dd["logits_u"].float().cpu().numpy().copy()))
        return hook

    def rec_fwd(fwd, store):
        def f(win, kv_len):
            lc, lu = fwd(win, kv_len)
            store.append((lc.copy(), None if lu is None else lu.copy()))
            return lc, lu
        return f

    d, meta = _load(golden_dir, "loop_lumina.npz")
    for m in meta[:2]:
        name, jac = m["name"], m["jacobi"]
        prompt = d[f"{name}.prompt"][0].tolist()
        cpu_rec, gpu_rec = [], []
        cpu_model = make_chameleon(m["config"], m["weight_seed"], m["embed_token_scale"], OracleWindowAttention())
        lcfg = OL.LoopConfig(jacobi_loop_interval_l=jac["jacobi_loop_interval_l"], jacobi_loop_interval_r=jac["jacobi_loop_interval_r"],
                             max_num_new_tokens=jac["max_num_new_tokens"], guidance_scale=jac["guidance_scale"], seed=jac["seed"],
                             do_cfg=jac["do_cfg"], prefix_token_sampler_scheme=jac["prefix_token_sampler_scheme"], max_length=m["max_len"],
                             eos_token_ids=(8196,))
        seq_cpu, _ = OL.run(prompt, rec_fwd(lumina_forward_fn(cpu_model, len(prompt), m["max_len"] + 32), cpu_rec),
                            lambda c, n: O.lumina_rules(c, n, 2000, 10), lcfg, m["config"]["vocab_size"], no_cfg_fn=O.lumina_force_no_cfg)
        model = make_chameleon(m["config"], m["weight_seed"], m["embed_token_scale"], ops.HipWindowAttention(n_split=1), dtype=torch.float32, device=dev)
        model.setup_cache(batch=2, s_max=((m["max_len"] + 64 + 31) // 32) * 32)
        cfg = SJDConfig(jacobi_loop_interval_l=jac["jacobi_loop_interval_l"], jacobi_loop_interval_r=jac["jacobi_loop_interval_r"],
                        max_num_new_tokens=jac["max_num_new_tokens"], guidance_scale=jac["guidance_scale"], seed=jac["seed"],
                        prefix_token_sampler_scheme=jac["prefix_token_sampler_scheme"], max_length=m["max_len"], eos_token_ids=(8196,),
                        noise_device="cpu")
        eng = SJDEngine(model, m["config"]["vocab_size"], dev, max_window=jac["max_num_new_tokens"], use_graph=False, narrow_head=False)
        eng.hook = rec_hook(gpu_rec)
        seq_gpu, _ = eng.decode(prompt, lumina_window_spec(prompt, dev), LuminaGrammar(2000, 10), cfg)
        assert seq_gpu == seq_cpu
        report["lumina/" + name] = _max_logit_gap(gpu_rec, cpu_rec)
    d, meta = _load(golden_dir, "loop_llamagen.npz")
    for m in meta[:2]:
        name, jac, N = m["name"], m["jacobi"], m["latent"] ** 2
        cpu_rec, gpu_rec = [], []
        cpu_model = make_llamagen(m["model_args"], m["weight_seed"], m["embed_token_scale"], OracleWindowAttention())
        fwd, first = llamagen_forward_fn(cpu_model, m["class_id"], m["cfg"], m["top_k"], m["top_p"], N, jac["seed"])
        lcfg = OL.LoopConfig(jacobi_loop_interval_l=jac["jacobi_loop_interval_l"], jacobi_loop_interval_r=jac["jacobi_loop_interval_r"],
                             max_num_new_tokens=jac["max_num_new_tokens"], guidance_scale=jac["guidance_scale"], seed=jac["seed"],
                             do_cfg=jac["do_cfg"], prefix_token_sampler_scheme=jac["prefix_token_sampler_scheme"], max_length=N)
        seq_cpu, _ = OL.run([first], rec_fwd(fwd, cpu_rec), lambda c, n: O.llamagen_rules(c, n, m["top_k"], m["top_p"]), lcfg,
                            m["model_args"]["vocab_size"])
        model = make_llamagen(m["model_args"], m["weight_seed"], m["embed_token_scale"], ops.HipWindowAttention(n_split=1), dtype=torch.float32, device=dev)
        model.setup_cache(batch=2, s_max=((1 + N + 64 + 31) // 32) * 32)
        cond = torch.tensor([m["class_id"], model.num_classes], device=dev)
        zeros = torch.zeros(2, dtype=torch.int32, device=dev)
        model.forward_embeds(model.embed_condition(cond), torch.zeros(2, 1, dtype=torch.long, device=dev), 0, zeros)   # cond rows -> cache
        cfg = SJDConfig(jacobi_loop_interval_l=jac["jacobi_loop_interval_l"], jacobi_loop_interval_r=jac["jacobi_loop_interval_r"],
                        max_num_new_tokens=jac["max_num_new_tokens"], guidance_scale=jac["guidance_scale"], seed=jac["seed"],
                        prefix_token_sampler_scheme=jac["prefix_token_sampler_scheme"], max_length=N, noise_device="cpu")
        eng = SJDEngine(model, m["model_args"]["vocab_size"], dev, max_window=jac["max_num_new_tokens"], use_graph=False)
        eng.hook = rec_hook(gpu_rec)
        seq_gpu, _ = eng.decode([first], llamagen_window_spec(first, 1, dev), TopKTopPGrammar(m["top_k"], m["top_p"]), cfg)
        assert seq_gpu == seq_cpu
        report["llamagen/" + name] = _max_logit_gap(gpu_rec, cpu_rec)
    print("max |logit_gpu - logit_cpu_oracle| per golden run:", report)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "golden_logit_gap.json"), "w") as f:
            json.dump(dict(tolerance=TOL, max_abs_gap_per_golden_run=report), f)
    assert max(report.values()) <= TOL, report
