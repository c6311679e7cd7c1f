"""BASELINE config 5 (Anole / Chameleon-7B 512x512, fp8 draft attention): how far does the fp8 (OCP e4m3) KV cache move the MODEL's output?

SURVEY.md section 8 states the parity target of the fp8 path as "the bf16 result within tolerance"; the reference has no fp8 (its attention is
modeling_chameleon.py:499-581 over a bf16 DynamicCache, called from jacobi_iteration_anhole.py:137-272), so the bound is stated HERE, at the
logits K2 reads, and measured against three forwards on the same weights and the same token windows:

  hip_fp8   the product path of config 5: prefill + window forwards on libsjd_hip.so with the KV cache in fp8 (F2 quantises k / v on the way in,
            K1 contracts on v_mfma_f32_16x16x32_fp8_fp8)
  hip_bf16  the same launches over a bf16 cache (config 2's attention)
  aten16    the independent ATen bf16 forward of tests/test_gpu_real_shape_forward.py (hipBLASLt + SDPA over a torch.cat cache)
  fp32      the same independent forward in fp32

Stated tolerance (asserted below, on the image-vocabulary columns of every window row, context 600 .. 700 keys).  Round 6 (VERDICT r5 #2b) removed
the two ON-CHIP roundings -- Q and P travel as hi / lo e4m3 operand pairs -- and calibrates per-layer K / V scales from the prompt's amax; measured
under the three arithmetics on one box (profiles/r6_fp8_model_bound_{a_r5,b_hilo,c_hilo_calibrated}.json):

                                  round 5      hi / lo      hi / lo + calibrated     bf16 cache
  max  |logit - fp32|             0.635        0.573        0.557                    0.227   (aten16 0.226)
  mean |logit - fp32|             0.0576       0.0554       0.0542                   0.0329  (aten16 0.0326)
  argmax agreement with bf16      93.8 %       93.4 %       95.3 %                   (bf16 vs aten16: 95.3 %)
  K2 total variation vs bf16      0.106        0.101        0.098                    (bf16 vs fp32: 0.056)
  tokens / step, 256 steps        2.05         1.98         2.04                     2.21

i.e. what is left IS the cache format: K and V stored with three mantissa bits (kernel level: the attention now equals exact attention over
the dequantised cache to 2^-8, tests/test_gpu_kernels.py::test_k1_k3_fp8_kv_cache, tolerances a fifth of round 5's).  VERDICT r5's targets
(mean <= 1.5 x bf16's, max <= 2 x, acceptance within 3 %) are NOT reached -- 1.65 x, 2.45 x, -8 % -- and cannot be with an e4m3 cache; DESIGN.md
section 6 says so plainly: config 5's fp8 path exists because BASELINE.json names it, at a tokens/s cost that bench.py reports next to the bf16
cache (other_configs.anole7b vs anole7b_bf16kv).  Asserted:
  * max |hip_fp8 - fp32| <= 3.0 x max |aten16 - fp32| and mean <= 2.0 x mean,
  * the argmax of hip_fp8 agrees with hip_bf16's on >= 90 % of the rows,
  * the sampling distribution K2 would form (CFG 3.0, top-k 2000, softmax) moves by <= 0.12 in total variation on average,
  * a 256-step SJD decode accepts within 12 % of the tokens per step on either cache.
The measured numbers go to gpurun_out/r6_fp8_model_bound*.json (committed under profiles/)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@torch.no_grad()
def test_fp8_kv_cache_moves_the_logits_by_a_stated_bound():
    import sjd_amd.ops as ops
    import sjd_amd.backbones as BB
    import sjd_amd.synthetic as synthetic
    from sjd_amd.engine import SJDEngine, SJDConfig
    from sjd_amd.frontends import lumina_window_spec
    from sjd_amd.grammar import AnoleGrammar
    from tests.test_gpu_real_shape_forward import _IndependentForward
    dev = torch.device("cuda:0")
    margs, dt, window, P, s_max = BB.LUMINA_7B, torch.bfloat16, 16, 600, 1024      # Anole-7B is the Chameleon-7B architecture
    V = margs.vocab_size
    with torch.device(dev):
        model = BB.ChameleonBackbone(margs, attn=ops.HipWindowAttention()).to(dt).eval()
    synthetic.fill_state_dict_device(model, seed=0, embed_token_scale=0.7)
    model.enable_fused(ops, gemm="sjd")
    # context: 64 text tokens + <image start> + 535 image tokens; then eight windows of 16 image tokens, each three rows further on (what a
    # decode that accepts three tokens per step looks like: 13 rows of every window overwrite rows of the one before)
    prompt = synthetic.synthetic_prompt(63, 1234, lo=9000, hi=60000)[0].tolist() + [8197] + synthetic.synthetic_prompt(P - 64, 99, lo=4, hi=8196)[0].tolist()
    spec = lumina_window_spec(prompt, dev)
    ks, po = spec.key_start.to(dev), spec.pos_offset.to(dev)
    g = torch.Generator().manual_seed(5)
    wins = [(P + 3 * i, torch.randint(4, 8196, (1, window), generator=g).expand(2, window).contiguous().to(dev)) for i in range(8)]
    cols = (0, 8224)                               # the image vocabulary 4..8195, 32-aligned

    calibrate = os.environ.get("SJD_FP8_CALIBRATE", "1") != "0"

    def hip_logits(cache_dtype):
        model.setup_cache(batch=2, s_max=s_max, dtype=cache_dtype)
        model.attn.params = None                   # kv_len from the host
        model.attn.layer_scales = None
        if cache_dtype == ops.FP8 and calibrate:   # round 6: per-layer (k, v) scales from this prompt's amax (what the engines do on their first prompt)
            scales = model.calibrate_kv_scales(spec.first_tokens.to(dev), spec.first_positions.to(dev), ks)
            rep_scales.extend([round(s_[0], 5), round(s_[1], 5)] for s_ in scales)
        model.forward_window(spec.first_tokens.to(dev), spec.first_positions.to(dev), 0, ks)
        out = []
        for kv, ids in wins:
            rows = kv + torch.arange(window, device=dev)
            pos = rows[None, :] + po[:, None]
            model.attn.choose_regime(kv + window, model.cache.k.dtype, shape=(2, window, margs.num_attention_heads, margs.num_key_value_heads, 128))
            lg = model.forward_window(ids, pos, kv, ks, cols=cols)
            out.append(lg.float().clone())
        return out

    rep_scales = []
    hip8 = hip_logits(ops.FP8)
    hip16 = hip_logits(None)
    outs = {}
    for tag, fdt in (("aten16", dt), ("fp32", torch.float32)):
        f = _IndependentForward(model, fdt)
        f.forward(spec.first_tokens.to(dev), 0, ks, po, cols)
        res = []
        for kv, ids in wins:
            f.rollback(kv)
            res.append(f.forward(ids, kv, ks, po, cols))
        outs[tag] = res
        del f
        torch.cuda.empty_cache()

    def k2_probs(lg):                              # what K2 forms from (cond, uncond) logits: CFG 3.0, image ids only, top-k 2000, softmax (JL:82-132)
        z = (3.0 * (lg[0] - lg[1]) + lg[1])[:, 4:8196]
        kth = z.topk(2000, dim=-1).values[:, -1:]
        return torch.softmax(z.masked_fill(z < kth, float("-inf")), dim=-1)

    rep = dict(family="anole7b", prompt_len=P, window=window, cache="fp8 e4m3 vs bf16", windows=[], calibrated=calibrate,
               kv_scales_first_and_last_layer=[rep_scales[0], rep_scales[-1]] if rep_scales else None,
               arithmetic=os.environ.get("SJD_FP8_TAG", "hi/lo e4m3 operands for Q and P (round 6)"))
    acc = dict(e8=[], e16=[], ea=[], d=[], agree=[], agree_a=[], tv=[], tv16=[], m8=0.0, m16=0.0, ma=0.0)
    for i, (kv, _) in enumerate(wins):
        a8, a16, aa, f32 = hip8[i][..., 4:8196], hip16[i][..., 4:8196], outs["aten16"][i][..., 4:8196], outs["fp32"][i][..., 4:8196]
        assert torch.isfinite(a8).all() and a8.shape == a16.shape == aa.shape == f32.shape
        e8, e16, ea = (a8 - f32).abs(), (a16 - f32).abs(), (aa - f32).abs()
        p8, p16, pf = k2_probs(hip8[i]), k2_probs(hip16[i]), k2_probs(outs["fp32"][i])
        tv, tv16 = 0.5 * (p8 - p16).abs().sum(-1), 0.5 * (p16 - pf).abs().sum(-1)
        w = dict(kv_len=kv, logit_std=round(float(f32.std()), 3), fp8_max=round(float(e8.max()), 4), fp8_mean=round(float(e8.mean()), 5),
                 bf16_max=round(float(e16.max()), 4), bf16_mean=round(float(e16.mean()), 5), aten16_max=round(float(ea.max()), 4),
                 aten16_mean=round(float(ea.mean()), 5), fp8_vs_bf16_max=round(float((a8 - a16).abs().max()), 4),
                 argmax_agree_fp8_bf16=round(float((a8.argmax(-1) == a16.argmax(-1)).float().mean()), 4),
                 argmax_agree_bf16_aten16=round(float((a16.argmax(-1) == aa.argmax(-1)).float().mean()), 4),
                 k2_total_variation_fp8_vs_bf16=round(float(tv.mean()), 4), k2_total_variation_bf16_vs_fp32=round(float(tv16.mean()), 4))
        rep["windows"].append(w)
        acc["e8"].append(w["fp8_mean"]); acc["e16"].append(w["bf16_mean"]); acc["ea"].append(w["aten16_mean"])
        acc["agree"].append(w["argmax_agree_fp8_bf16"]); acc["agree_a"].append(w["argmax_agree_bf16_aten16"])
        acc["tv"].append(w["k2_total_variation_fp8_vs_bf16"]); acc["tv16"].append(w["k2_total_variation_bf16_vs_fp32"])
        acc["m8"], acc["m16"], acc["ma"] = max(acc["m8"], w["fp8_max"]), max(acc["m16"], w["bf16_max"]), max(acc["ma"], w["aten16_max"])
    mean = lambda v: round(sum(v) / len(v), 5)
    rep["summary"] = dict(fp8_max=acc["m8"], bf16_max=acc["m16"], aten16_max=acc["ma"], fp8_mean=mean(acc["e8"]), bf16_mean=mean(acc["e16"]),
                          aten16_mean=mean(acc["ea"]), argmax_agree_fp8_bf16=mean(acc["agree"]), argmax_agree_bf16_aten16=mean(acc["agree_a"]),
                          k2_total_variation_fp8_vs_bf16=mean(acc["tv"]), k2_total_variation_bf16_vs_fp32=mean(acc["tv16"]))

    # ---- the decode itself on either cache: tokens per step over 256 SJD iterations of the config-5 workload
    P0, n_img = 64, 1024 + 1
    prompt0 = synthetic.synthetic_prompt(P0 - 1, 1234, lo=9000, hi=60000)[0].tolist() + [8197]
    dec = {}
    for tag, cdt in (("fp8", ops.FP8), ("bf16", None)):
        model.setup_cache(batch=2, s_max=((P0 + n_img + 2 * window + 64 + 31) // 32) * 32, dtype=cdt)
        model.attn.layer_scales = None             # (the engine calibrates an fp8 cache on its first prompt unless SJD_FP8_CALIBRATE=0)
        eng = SJDEngine(model, V, dev, max_window=window, use_graph=True)
        cfg = SJDConfig(jacobi_loop_interval_l=0, jacobi_loop_interval_r=1024 - window - 2, max_num_new_tokens=window, guidance_scale=3.0, seed=1234,
                        prefix_token_sampler_scheme="speculative_jacobi", max_length=P0 + n_img, eos_token_ids=(8196,))
        seq, st = eng.decode(prompt0, lumina_window_spec(prompt0, dev), AnoleGrammar(V, P0, P0 + n_img, 1024), cfg, warmup_iters=0, timed_iters=256)
        dec[tag] = dict(steps=int(st.timed_nfe), tokens=int(st.tokens), tokens_per_step=round(st.tokens / max(st.timed_nfe, 1), 4))
        del eng
    rep["decode_256_steps"] = dec
    print("fp8 model bound:", json.dumps(rep))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "r6_fp8_model_bound%s.json" % os.environ.get("SJD_FP8_FILE_TAG", "")), "w") as fh:
            json.dump(rep, fh, indent=1)
    s = rep["summary"]
    loose = os.environ.get("SJD_FP8_FILE_TAG", "") in ("_a_r5", "_b_hilo")          # (the A/B legs of tools/_r6_fp8.sh: round 5's bounds)
    assert s["fp8_max"] <= (3.5 if loose else 3.0) * s["aten16_max"] + 1e-3 and s["fp8_mean"] <= (2.5 if loose else 2.0) * s["aten16_mean"] + 1e-4, s
    assert s["bf16_max"] <= 1.5 * s["aten16_max"] + 1e-3 and s["bf16_mean"] <= 1.5 * s["aten16_mean"] + 1e-4, s
    assert s["argmax_agree_fp8_bf16"] >= (0.85 if loose else 0.90), s
    assert s["k2_total_variation_fp8_vs_bf16"] <= (0.15 if loose else 0.12), s
    assert abs(dec["fp8"]["tokens_per_step"] - dec["bf16"]["tokens_per_step"]) <= (0.15 if loose else 0.12) * dec["bf16"]["tokens_per_step"], dec
