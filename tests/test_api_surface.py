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
    """ValueError at reference JL:1048; the horizon init schemes crash in the released reference."""
    from sjd_amd.engine import SJDConfig, SJDEngine
    eng = SJDEngine.__new__(SJDEngine)
    eng.Lmax = 16
    with pytest.raises(ValueError):
        SJDEngine.decode.__wrapped__(eng, [1], None, None, SJDConfig(prefix_token_sampler_scheme="bogus"))
    with pytest.raises(NotImplementedError):
        SJDEngine.decode.__wrapped__(eng, [1], None, None, SJDConfig(multi_token_init_scheme="repeat_horizon"))


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
    with pytest.raises(RuntimeError):
        vl(None, None)


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
