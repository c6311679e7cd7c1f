"""CPU: the drop-in boundary -- import paths, names, argument defaults and error behaviour of the mirror."""
import inspect

import pytest
import torch


def test_reference_import_paths_resolve():
    from scheduler.jacobi_iteration_lumina_mgpt import (renew_pipeline_sampler, renew_sampler, renew_backbone, renew_pipeline,  # noqa
                                                        SpeculativeSampler, sampling_logits2tokens, prefix_matching_next_tokens,
                                                        find_first_misaligned_token_inds, check_is_force_no_cfg, set_seed)
    from scheduler.logit_processor_3dim import (MultiTokensVLLogitsProcessor, MultiTokensInterleavedTopKLogitsWarper,  # noqa
                                                TopPLogitsWarper3d, get_double_cfg_input_ids, check_eol_in_multitokens,
                                                AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d,
                                                AllowOnlyTokensInRelativeWindowLogitsProcessor3d,
                                                SuppressTokensInIndexRangeLogitsProcessor3d, SuppressTokensAtBeginLogitsProcessor3d,
                                                SuppressTokensLogitsProcessor3d)
    from scheduler.jacobi_iteration_emu3 import renew_solver  # noqa
    from scheduler.jacobi_iteration_anhole import renew_pipeline_sampler as anole_rps  # noqa
    from llamagen.llamagen_solver import LlamaGenSolver, renew_llamagen  # noqa
    from llamagen.llamagen import GPT_models  # noqa
    from lumina_mgpt.inference_solver import FlexARInferenceSolver  # noqa  (test_lumina_mgpt.py:94)
    assert set(GPT_models) == {'GPT-B', 'GPT-L', 'GPT-XL', 'GPT-XXL', 'GPT-XXXL', 'GPT-1B', 'GPT-3B', 'GPT-7B'}


def test_init_new_params_defaults_match_reference():
    """reference jacobi_iteration_lumina_mgpt.py:865-878"""
    from scheduler.jacobi_iteration_lumina_mgpt import renew_sampler
    from sjd_amd.backbones import LlamaGenBackbone, LlamaGenArgs
    cls = renew_sampler(LlamaGenBackbone)
    sig = inspect.signature(cls._init_new_params)
    want = dict(jacobi_loop_interval_l=1, jacobi_loop_interval_r=(768 // 16) ** 2 + 768 // 16, max_num_new_tokens=16,
                guidance_scale=3.0, seed=42, multi_token_init_scheme='random', do_cfg=True,
                prefix_token_sampler_scheme='speculative_jacobi', use_chameleon_tokenizer=True, _init_doubled_attn_mask_cfg=False)
    for k, v in want.items():
        assert sig.parameters[k].default == v, k
    ss = inspect.signature(cls._sample)
    assert list(ss.parameters)[:8] == ["self", "input_ids", "logits_processor", "stopping_criteria", "generation_config",
                                       "synced_gpus", "streamer", "logits_warper"]
    m = LlamaGenBackbone(LlamaGenArgs(dim=64, n_layer=1, n_head=2, vocab_size=9000, block_size=16))
    m.__class__ = cls
    m._init_new_params(prefix_token_sampler_scheme="bogus", image_top_k=5, text_top_k=3)     # extra kwargs are swallowed
    assert m.prefix_token_sampler_scheme == "bogus"


def test_engine_rejects_unknown_scheme_and_broken_init_schemes():
    """ValueError at reference JL:1048; unknown init schemes assert at JL:560 / 592."""
    from sjd_amd.engine import SJDConfig, SJDEngine
    eng = SJDEngine.__new__(SJDEngine)
    eng.Lmax = 16
    with pytest.raises(ValueError):
        SJDEngine.decode.__wrapped__(eng, [1], None, None, SJDConfig(prefix_token_sampler_scheme="bogus"))
    with pytest.raises(ValueError):                  # only the horizon variants exist (the 'vertical' ones assert at JL:554-560)
        SJDEngine.decode.__wrapped__(eng, [1], None, None, SJDConfig(multi_token_init_scheme="repeat_vertical"))


def test_processor_descriptors_and_grammar_mapping():
    from scheduler.logit_processor_3dim import (MultiTokensVLLogitsProcessor, MultiTokensInterleavedTopKLogitsWarper,
                                                TopPLogitsWarper3d, TopKLogitsWarper, grammar_from_processors)
    import sjd_amd.grammar as G
    with pytest.raises(ValueError):
        MultiTokensInterleavedTopKLogitsWarper(image_top_k=10, text_top_k=0)
    with pytest.raises(ValueError):
        TopPLogitsWarper3d(top_p=1.5)
    vl = MultiTokensVLLogitsProcessor(8197, 8196, 8803, 32, 65536)
    g = grammar_from_processors([vl, MultiTokensInterleavedTopKLogitsWarper(2000, 10, 8197, 8196)])
    assert isinstance(g, G.LuminaGrammar) and g.image_top_k == 2000 and g.text_top_k == 10
    g = grammar_from_processors([TopKLogitsWarper(1000), TopPLogitsWarper3d(0.9)])
    assert isinstance(g, G.TopKTopPGrammar) and (g.top_k, g.top_p) == (1000, 0.9)
    from transformers.generation.logits_process import TopKLogitsWarper as HFTopK
    assert grammar_from_processors([HFTopK(top_k=50)]).top_k == 50
    with pytest.raises(NotImplementedError):
        grammar_from_processors([object()])
    # called on tensors directly the self-contained processors evaluate their rules with ATen ops (tests/test_processor_calls.py pins that to the
    # reference's objects); in text mode the Lumina processor leaves the scores alone (LP:89-92)
    sc = torch.randn(1, 3, 9216)
    assert torch.equal(vl(torch.tensor([[9000, 9001, 9002]]), sc), sc)
    # Anole's other restricted modes (JA:178-189, 233-260): their processor lists map to AnoleGrammar(mode=)
    from scheduler.logit_processor_3dim import (AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d as AtOff,
                                                AllowOnlyTokensInRelativeWindowLogitsProcessor3d as InWin,
                                                SuppressTokensInIndexRangeLogitsProcessor3d as InRange, SuppressTokensLogitsProcessor3d as Supp)
    img = list(range(4, 8196))
    g = grammar_from_processors([Supp(img + [8197, 8196]), TopKLogitsWarper(10)], prompt_len=5, max_length=50, vocab_size=9216)
    assert isinstance(g, G.AnoleGrammar) and g.mode == "text-only" and (g.img_lo, g.eoi, g.boi, g.top_k) == (4, 8196, 8197, 10)
    g.start([9000] * 5)
    r = g.window_rules(2)[0]
    assert r.n_ranges == 2 and (r.lo[0], r.hi[0], r.lo[1], r.hi[1]) == (0, 4, 8198, 9216)
    g = grammar_from_processors([AtOff(8197, [8196], 25, True), InWin(8197, img, 24, True), InRange([8197], 50 - 24 - 1)],
                                prompt_len=5, max_length=50, vocab_size=9216)
    assert isinstance(g, G.AnoleGrammar) and g.mode == "interleaved-text-image" and (g.L, g.max_length) == (24, 50)
    with pytest.raises(NotImplementedError):
        grammar_from_processors([Supp(img + [8197, 8196])], prompt_len=5, max_length=50)          # no vocabulary size
    with pytest.raises(NotImplementedError):
        grammar_from_processors([Supp([5, 9, 100])], vocab_size=9216)
    # GenerationConfig.top_p: HF's TopPLogitsWarper (the HF object or the stand-in) behind any family's processors = a scalar of every rule
    from transformers.generation.logits_process import TopPLogitsWarper as HFTopP
    from scheduler.logit_processor_3dim import TopPLogitsWarper, TemperatureLogitsWarper
    for tp in (HFTopP(top_p=0.8), TopPLogitsWarper(0.8)):
        g = grammar_from_processors([vl, MultiTokensInterleavedTopKLogitsWarper(2000, 10, 8197, 8196), TemperatureLogitsWarper(0.9), tp])
        assert isinstance(g, G.LuminaGrammar) and g.top_p == 0.8 and g.temperature == 0.9
        g.start([5, 6, 7])
        r = g.window_rules(3)[0]
        assert abs(r.top_p_thr - (1.0 - 0.8)) < 1e-6 and abs(r.temperature - 0.9) < 1e-6 and r.top_k == 10
    assert grammar_from_processors([vl, HFTopP(top_p=1.0)]).top_p is None
    with pytest.raises(NotImplementedError):
        grammar_from_processors([TopKLogitsWarper(10), TopPLogitsWarper3d(0.9), HFTopP(top_p=0.8)])
    with pytest.raises(NotImplementedError):
        TopPLogitsWarper(0.8, min_tokens_to_keep=2)
    # a top-p warper AHEAD of the temperature warper is evaluated by HF at T = 1; the kernels scale first -> refused, not silently reordered
    with pytest.raises(NotImplementedError):
        grammar_from_processors([vl, MultiTokensInterleavedTopKLogitsWarper(2000, 10, 8197, 8196), HFTopP(top_p=0.8), TemperatureLogitsWarper(0.9)])
    with pytest.raises(NotImplementedError):
        grammar_from_processors([TopKLogitsWarper(10), TopPLogitsWarper3d(0.9), TemperatureLogitsWarper(0.7)])
    assert grammar_from_processors([TopKLogitsWarper(10), TopPLogitsWarper3d(0.9), TemperatureLogitsWarper(1.0)]).temperature == 1.0


def test_get_double_cfg_input_ids_and_emu3_inputs():
    """reference logit_processor_3dim.py:422-440, jacobi_iteration_emu3.py:234-278"""
    from scheduler.logit_processor_3dim import get_double_cfg_input_ids
    from sjd_amd.frontends import emu3_window_spec
    pos, neg = torch.tensor([[5, 6, 7, 8, 9]]), torch.tensor([[3, 4]])
    both = get_double_cfg_input_ids(pos, neg, pad_category=0)
    assert both.tolist() == [[5, 6, 7, 8, 9], [0, 0, 0, 3, 4]]
    spec = emu3_window_spec([5, 6, 7, 8, 9], [3, 4], 0, "cpu")
    assert spec.first_tokens.tolist() == both.tolist()
    assert spec.key_start.tolist() == [0, 3] and spec.pos_offset.tolist() == [0, -3]
    assert spec.first_positions.tolist() == [[0, 1, 2, 3, 4], [1, 1, 1, 0, 1]]


def test_integer_helpers():
    from scheduler.jacobi_iteration_lumina_mgpt import find_first_misaligned_token_inds, check_is_force_no_cfg
    from scheduler.logit_processor_3dim import check_eol_in_multitokens
    assert find_first_misaligned_token_inds(torch.tensor([[1, 2, 3, 4]]), torch.tensor([[2, 3, 9, 9]])) == [3]
    assert find_first_misaligned_token_inds(torch.tensor([[1, 2, 3]]), torch.tensor([[2, 3, 7]])) == [3]
    assert check_is_force_no_cfg(torch.tensor([[9, 8197, 5]]), 8197, 8196) is False
    assert check_is_force_no_cfg(torch.tensor([[9, 8197, 5, 8196]]), 8197, 8196) is True
    assert check_is_force_no_cfg(torch.tensor([[9]]), None, None) is False
    assert check_eol_in_multitokens(7, 16, 9) and not check_eol_in_multitokens(0, 3, 9)


def test_reference_driver_import_statements_resolve():
    """The import lines of the reference's three drivers and of its eval wrapper, written out (names only -- nothing of the
    reference is read at run time): test_lumina_mgpt.py:10,101; test_emu3.py:16,145; test_llamagen.py:17-21; model_loader.py:19-22,
    126, 225-229, 502."""
    from lumina_mgpt.inference_solver import FlexARInferenceSolver  # noqa: F401
    from scheduler.jacobi_iteration_lumina_mgpt import renew_pipeline_sampler  # noqa: F401
    from emu3.mllm.processing_emu3 import Emu3Processor  # noqa: F401
    from scheduler.jacobi_iteration_emu3 import renew_solver  # noqa: F401
    from llamagen.tokenizer.tokenizer_image.vq_model import VQ_models  # noqa: F401
    from llamagen.language.t5 import T5Embedder  # noqa: F401
    from llamagen.llamagen import GPT_models  # noqa: F401
    from llamagen.llamagen_solver import LlamaGenSolver, renew_llamagen, generate  # noqa: F401
    from scheduler.jacobi_iteration_lumina_mgpt import renew_sampler  # noqa: F401
    from scheduler.jacobi_iteration_anhole import renew_pipeline_sampler as renew_pipeline_sampler_anhole  # noqa: F401
    from scheduler.jacobi_iteration_emu3 import renew_solver as renew_solver_emu3  # noqa: F401
    from model_wrappers.model_loader import load_pretrained_model, get_forward_func  # noqa: F401
    from llamagen.llamagen_solver import generate as llamagen_original_generate
    sig = inspect.signature(llamagen_original_generate)                            # LS:145
    assert list(sig.parameters)[:6] == ["model", "cond", "max_new_tokens", "emb_masks", "cfg_scale", "cfg_interval"]
    assert set(VQ_models) == {"VQ-16", "VQ-8"}


def test_model_loader_dispatch_and_defaults():
    """reference ML:347-359, 564-574: name-substring dispatch, NotImplementedError otherwise; loader keyword defaults (ML:25-35, 62-77,
    112-127, 194-224)."""
    import model_wrappers.model_loader as ML
    with pytest.raises(NotImplementedError):
        ML.load_pretrained_model("some-other-model")
    with pytest.raises(NotImplementedError):
        ML.get_forward_func("some-other-model", None)
    d = {k: v.default for k, v in inspect.signature(ML.load_lumina_mgpt).parameters.items()}
    assert (d["model_name"], d["target_size"], d["max_num_new_tokens"], d["guidance_scale"], d["multi_token_init_scheme"]) == \
        ("Alpha-VLLM/Lumina-mGPT-7B-768", 768, 16, 7.0, "random")
    d = {k: v.default for k, v in inspect.signature(ML.load_emu3).parameters.items()}
    assert (d["model_name"], d["target_size"], d["image_top_k"], d["prefix_token_sampler_scheme"]) == ("BAAI/Emu3-Gen", 720, 2048, "speculative_jacobi")
    d = {k: v.default for k, v in inspect.signature(ML.load_llamagen).parameters.items()}
    assert d["guidance_scale"] == 7.5 and d["image_top_k"] == 1000 and d["backbone_params"]["gpt_model"] == "GPT-XL"
    d = {k: v.default for k, v in inspect.signature(ML.load_anole).parameters.items()}
    assert d["model_name"] == "leloy/Anole-7b-v0.1-hf" and d["target_size"] == 512
    with pytest.raises(FileNotFoundError):           # no hub access: a missing local checkpoint is an error, not a download
        ML.load_pretrained_model("Alpha-VLLM/Lumina-mGPT-7B-768", cache_dir="/nonexistent")


def test_emu3_processor_grammar_helper_and_generation_prompt():
    """reference emu3/mllm/processing_emu3.py:246-290 and the 'G' branch of __call__ (:150-165)"""
    from types import SimpleNamespace as NS
    from emu3.mllm.processing_emu3 import Emu3Processor
    from tests.helpers import Emu3StubTokenizer
    proc = Emu3Processor(None, NS(config=NS(codebook_size=8192), spatial_scale_factor=8), Emu3StubTokenizer())
    fn = proc.build_prefix_constrained_fn(3, 5)
    assert (fn.height, fn.width, fn.img_token, fn.eol_token, fn.eof_token, fn.eoi_token, fn.eos_token, fn.pad_token) == (3, 5, 200, 203, 204, 201, 202, 205)
    assert fn.visual_tokens == list(range(3000, 3000 + 8192))
    assert proc.calculate_generate_size("1:1", 720 * 720, 8) == (90, 90) and proc.calculate_generate_size("4:3", 518400, 8) == (78, 104)
    out = proc(text="ab", mode="G", ratio="1:1", image_area=24 * 40, return_tensors="pt")
    ids = out.input_ids[0].tolist()
    assert ids[0] == 1 and ids[-1] == 200 and 2 in ids and out.image_size == [[4, 4]]
    with pytest.raises(ValueError):
        proc(text=["a", "b"], mode="G")


def test_hf_generate_builds_criteria_and_topk_then_calls_sample():
    """HF generate's contract as the drivers rely on it (test_emu3.py:81-90, 163-169): max_new_tokens -> max_length, EOS criterion,
    TopKLogitsWarper(generation_config.top_k) appended AFTER the user's processors, attention_mask / neg_input_ids passed through."""
    from transformers import GenerationConfig
    from scheduler.jacobi_iteration_lumina_mgpt import hf_generate
    seen = {}

    class M:
        args = type("A", (), {"max_position_embeddings": 64})()

        def _sample(self, ids, procs, crit, gc, synced, streamer, **kw):
            seen.update(ids=ids, procs=[type(p).__name__ for p in procs], eos=[c.eos_token_id for c in crit if hasattr(c, "eos_token_id")],
                        max_len=[c.max_length for c in crit if hasattr(c, "max_length")], kw=kw, gc=gc)
            return ids

    ids = torch.zeros(1, 10, dtype=torch.long)
    gc = GenerationConfig(use_cache=True, eos_token_id=202, pad_token_id=205, max_new_tokens=40960, do_sample=True, top_k=2048)

    class Proc:
        pass

    hf_generate(M(), ids, gc, logits_processor=[Proc()], attention_mask=torch.ones(2, 10), neg_input_ids=torch.ones(1, 4, dtype=torch.long))
    assert seen["procs"] == ["Proc", "TopKLogitsWarper"] and seen["eos"] == [[202]]
    assert seen["max_len"] == [64]                                   # 10 + 40960, bounded by the model's context
    assert set(seen["kw"]) == {"attention_mask", "neg_input_ids"} and gc.max_length != 64       # the caller's config is not mutated
    # temperature: a TemperatureLogitsWarper behind the user's processors and in front of the top-k warper, as HF orders its warpers
    hf_generate(M(), ids, GenerationConfig(do_sample=True, temperature=0.7, top_k=50, max_new_tokens=4), logits_processor=[Proc()])
    assert seen["procs"] == ["Proc", "TemperatureLogitsWarper", "TopKLogitsWarper"]
    # top_p: HF's TopPLogitsWarper behind temperature and top-k
    hf_generate(M(), ids, GenerationConfig(do_sample=True, temperature=0.7, top_k=50, top_p=0.9, max_new_tokens=4), logits_processor=[Proc()])
    assert seen["procs"] == ["Proc", "TemperatureLogitsWarper", "TopKLogitsWarper", "TopPLogitsWarper"]
    # greedy (round 5): HF builds no warpers when do_sample is False; `_sample` then takes the argmax of the processed scores (JL:127-129)
    hf_generate(M(), ids, GenerationConfig(do_sample=False, temperature=0.7, top_k=50, max_new_tokens=4), logits_processor=[Proc()])
    assert seen["procs"] == ["Proc"]
    with pytest.raises(NotImplementedError):
        hf_generate(M(), ids, GenerationConfig(do_sample=True, num_beams=2, max_new_tokens=4))


def test_flexar_solver_method_surface():
    """FlexARInferenceSolver keeps the reference's method names (IS:273-450) -- generate, create_logits_processor, decode_ids, decode_image,
    create_image_grid, get_streamer; the two that need the reference's tokenizer / VQ-GAN assets say so instead of failing somewhere inside"""
    import pytest
    from PIL import Image
    from lumina_mgpt.inference_solver import FlexARInferenceSolver
    for name in ("generate", "generate_ids", "create_logits_processor", "decode_ids", "decode_image", "create_image_grid", "get_streamer"):
        assert callable(getattr(FlexARInferenceSolver, name)), name
    imgs = [Image.new("RGB", (4, 3), (i * 40, 0, 0)) for i in range(6)]
    grid = FlexARInferenceSolver.create_image_grid(imgs, 2, 3)                       # IS:405-415
    assert grid.size == (12, 6) and grid.getpixel((5, 4)) == (160, 0, 0) and grid.getpixel((11, 0)) == (80, 0, 0)
    s = FlexARInferenceSolver.__new__(FlexARInferenceSolver)
    s.item_processor = None
    with pytest.raises(NotImplementedError):
        s.get_streamer()
    with pytest.raises(NotImplementedError):
        s.decode_image([8197, 8196])

    class _IP:
        tokenizer = None

        def decode_image(self, toks):
            return ("image", len(toks))
    s.item_processor = _IP()
    assert s.decode_image([8197, 5, 8196]) == ("image", 3)
