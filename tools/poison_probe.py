"""Debug aid: run one decode of the production window path on POISONED device memory (every free block of the caching allocator filled with
0xFF = NaN) with SJD_NAN_CHECK=1, eager, so that the first kernel that reads workspace nobody wrote is named.
  python tools/poison_probe.py --family emu3_8b --P 4100 --s-max 4224 [--graph]"""
import argparse
import os
import sys

os.environ.setdefault("SJD_NAN_CHECK", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--family", default="emu3_8b")
    ap.add_argument("--P", type=int, default=4100)
    ap.add_argument("--s-max", type=int, default=4224)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--no-poison", action="store_true")
    ap.add_argument("--iters", type=int, default=8)
    a = ap.parse_args()
    from conftest import poison_device_memory
    import sjd_amd.ops as ops
    import sjd_amd.backbones as BB
    import sjd_amd.synthetic as synthetic
    from sjd_amd.engine import SJDEngine, SJDConfig
    from sjd_amd.frontends import lumina_window_spec, lumina_prompt, emu3_window_spec
    from sjd_amd.grammar import LuminaGrammar, Emu3Grammar
    dev = torch.device("cuda:0")
    seed, P = 17, a.P
    if not a.no_poison:
        poison_device_memory()
    if a.family == "lumina7b":
        margs, dt, window = BB.LUMINA_7B, torch.bfloat16, 16
    else:
        margs, dt, window = BB.EMU3_8B, (torch.bfloat16 if a.family.endswith("bf16") else torch.float16), 32
    with torch.device(dev):
        model = BB.ChameleonBackbone(margs, attn=ops.HipWindowAttention()).to(dt).eval()
    if a.family != "lumina7b":
        model.G1_CFG = dict(model.G1_CFG_EMU3)
    synthetic.fill_state_dict_device(model, seed=0, embed_token_scale=0.7)
    model.enable_fused(ops, gemm="sjd")
    V = margs.vocab_size
    if a.family == "lumina7b":
        prompt = lumina_prompt(P, 48, 48, seed=seed)
        spec = lumina_window_spec(prompt, dev)
        grammar = LuminaGrammar(2000, 10)
        cfg = SJDConfig(jacobi_loop_interval_l=0, jacobi_loop_interval_r=48 * 48 + 48 - 13, max_num_new_tokens=window, guidance_scale=3.0,
                        seed=seed, max_length=P + 400, eos_token_ids=(8196,))
    else:
        tok = dict(img_token=151851, eoi_token=151853, eos_token=151850, eol_token=151846, eof_token=151847, pad_token=151643)
        pos = synthetic.synthetic_prompt(P - 1, seed, lo=1000, hi=150000)[0].tolist() + [tok["img_token"]]
        neg = synthetic.synthetic_prompt(11, seed + 1, lo=1000, hi=150000)[0].tolist() + [tok["img_token"]]
        spec = emu3_window_spec(pos, neg, tok["pad_token"], dev)
        prompt = spec.first_tokens[0].tolist()
        grammar = Emu3Grammar(90, 90, 151854, 32768, top_k=2048, **tok)
        cfg = SJDConfig(jacobi_loop_interval_l=1, jacobi_loop_interval_r=90 * 90 - 1, max_num_new_tokens=window, guidance_scale=3.0,
                        seed=seed, max_length=P + 400, eos_token_ids=(tok["eos_token"],))
    if not a.no_poison:
        poison_device_memory()
    model.setup_cache(batch=2, s_max=a.s_max)
    eng = SJDEngine(model, V, dev, max_window=window, use_graph=a.graph)
    if not a.no_poison:
        poison_device_memory()
    it = [0]

    def hook(d):
        n = d["n_rows"]
        lc, lu = d["logits_c"], d["logits_u"]
        fin = bool(torch.isfinite(lc).all()) and bool(torch.isfinite(lu).all())
        print(f"iter {it[0]} first={d['first']} n_rows={n} kv_len={int(eng.params.view.kv_len)} logits finite={fin}", flush=True)
        it[0] += 1

    eng.hook = hook
    out = eng.decode(prompt, spec, grammar, cfg, warmup_iters=0, timed_iters=a.iters)
    print("decode ok", flush=True)


if __name__ == "__main__":
    main()
