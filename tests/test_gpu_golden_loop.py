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


def test_lumina_golden_tokens_on_gpu(dev, golden_dir):
    import sjd_amd.ops as ops
    from sjd_amd.engine import SJDEngine, SJDConfig
    from sjd_amd.frontends import lumina_window_spec
    from sjd_amd.grammar import LuminaGrammar
    from tests.helpers import make_chameleon
    d, meta = _load(golden_dir, "loop_lumina.npz")
    for m in meta:
        name, jac = m["name"], m["jacobi"]
        model = make_chameleon(m["config"], m["weight_seed"], m["embed_token_scale"], ops.HipWindowAttention(n_split=1),
                               dtype=torch.float32, device=dev)
        prompt = d[f"{name}.prompt"][0].tolist()
        model.setup_cache(batch=2, s_max=((m["max_len"] + 64 + 31) // 32) * 32)
        cfg = SJDConfig(jacobi_loop_interval_l=jac["jacobi_loop_interval_l"], jacobi_loop_interval_r=jac["jacobi_loop_interval_r"],
                        max_num_new_tokens=jac["max_num_new_tokens"], guidance_scale=jac["guidance_scale"], seed=jac["seed"],
                        prefix_token_sampler_scheme=jac["prefix_token_sampler_scheme"], max_length=m["max_len"],
                        eos_token_ids=(8196,), noise_device="cpu")
        eng = SJDEngine(model, m["config"]["vocab_size"], dev, max_window=jac["max_num_new_tokens"], use_graph=False)
        seq, stats = eng.decode(prompt, lumina_window_spec(prompt, dev), LuminaGrammar(2000, 10), cfg)
        assert seq == d[f"{name}.sequence"][0].tolist(), name
        assert stats.matched == d[f"{name}.matched"].tolist(), name
