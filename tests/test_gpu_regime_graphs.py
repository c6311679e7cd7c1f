"""the two K1 forms are different kernels and every regime's forward graph owns its own static logits / head partials, so a
captured {K2, K4} graph must never be replayed on the other regime's buffer.  Behavioural check: a decode whose context crosses the
column-split / key-split threshold in the middle of the image, with hipGraphs, in the one-graph and in the two-stage form, must emit exactly
the tokens of the eager decode (the kernels are the same, the draws are Philox streams keyed by the seed: any stale-buffer read shows up as a
different token)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _decode(use_graph, two_stage, threshold, seed=3):
    import sjd_amd.engine as E
    import sjd_amd.ops as ops
    import sjd_amd.synthetic as synthetic
    from sjd_amd.engine import SJDEngine, SJDConfig, WindowSpec
    from sjd_amd.grammar import LuminaGrammar
    from tests.helpers import make_chameleon
    device, V, P, hg, wg, window = "cuda:0", 9216, 12, 4, 4, 16
    conf = dict(vocab_size=V, hidden_size=512, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=4, max_position_embeddings=512, rms_norm_eps=1e-5, rope_theta=10000.0)
    attn = ops.HipWindowAttention(n_split=2)
    attn.COLSPLIT_MAX_KEYS = {"16bit": threshold, "fp8": threshold}
    model = make_chameleon(conf, 23, 0.25, attn, dtype=torch.bfloat16, device=device)
    model.enable_fused(ops, gemm="sjd")
    prompt = torch.cat([synthetic.synthetic_prompt(P - 3, seed, lo=8900, hi=9200), torch.tensor([[8197, 8804 + hg, 8804 + wg]])], dim=1)
    n_img = (2 * wg + 1) * 2 * hg
    max_len = P + n_img + 1 + 4
    model.setup_cache(batch=2, s_max=((max_len + 64 + 31) // 32) * 32)
    cfg = SJDConfig(jacobi_loop_interval_l=3, jacobi_loop_interval_r=n_img - 10, max_num_new_tokens=window, guidance_scale=3.0, seed=seed,
                    prefix_token_sampler_scheme="speculative_jacobi", max_length=max_len, eos_token_ids=(8196,))
    ids = prompt.to(device)
    spec = WindowSpec(first_tokens=ids.repeat(2, 1), first_positions=torch.stack([torch.arange(P), torch.tensor([1] * (P - 1) + [0])]).to(device),
                      key_start=torch.tensor([0, P - 1], dtype=torch.int32), pos_offset=torch.tensor([0, -(P - 1)], dtype=torch.long), kv_base=0)
    eng = SJDEngine(model, V, device, max_window=window, use_graph=use_graph)
    old = E._TWO_STAGE
    E._TWO_STAGE = two_stage
    try:
        regimes, seqs = set(), []
        for image in range(2):                       # the second image replays the graphs of BOTH regimes captured during the first
            log = []
            seq, st = eng.decode(prompt[0].tolist(), spec, LuminaGrammar(2000, 10), cfg, iter_log=log)
            seqs.append(seq)
            regimes.add(model.attn.regime)
        n_graphs = len(eng._graphs)
    finally:
        E._TWO_STAGE = old
    return seqs, n_graphs


@pytest.mark.parametrize("two_stage", [False, True])
def test_graph_decode_across_the_k1_regime_threshold_matches_eager(two_stage):
    threshold = 12 + 40            # crossed after ~40 accepted tokens of a 72-token image
    eager, _ = _decode(False, two_stage, threshold)
    graph, n_graphs = _decode(True, two_stage, threshold)
    assert n_graphs >= 2, "both regimes must have been captured"
    assert graph[0] == eager[0] and graph[1] == eager[1], "a graph replayed on the other regime's buffer"
    assert eager[0] == eager[1], "same seed, same image"
    # and the threshold really is crossed: pinned to one side the decode is the same sequence (the two K1 forms agree to the last bit on these
    # shapes only by luck, so compare lengths, not tokens)
    pinned, _ = _decode(True, two_stage, 10 ** 6)
    assert len(pinned[0]) == len(graph[0])
