#!/usr/bin/env python3
"""What the iteration boundary costs: the SAME captured iteration graph (K5, forward, K2, K4 at Lumina-7B shapes)
  (a) replayed back to back, one sync at the end        -> device time of an iteration (kernels + the gaps inside the graph)
  (b) blob upload + replay + host spin on K4's mirror   -> the engine's loop without its host arithmetic
  (c) the engine's own ms per step over the same region
(b) - (a) is what the hand-over host -> device -> host costs per iteration; (c) - (b) the host's arithmetic.
  python tools/boundary_probe.py [--model lumina7b|anole7b] [--iters 200]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--kv", type=int, default=1216)
    a = ap.parse_args()
    import sjd_amd.ops as ops
    import sjd_amd.backbones as BB
    import sjd_amd.synthetic as synthetic
    from sjd_amd.engine import SJDEngine, SJDConfig
    from sjd_amd.frontends import lumina_window_spec, lumina_prompt
    from sjd_amd.grammar import LuminaGrammar
    dev = torch.device("cuda:0")
    margs, window, grid = BB.LUMINA_7B, 16, 48
    with torch.device(dev):
        model = BB.ChameleonBackbone(margs, attn=ops.HipWindowAttention()).to(torch.bfloat16).eval()
    synthetic.fill_state_dict_device(model, seed=0, embed_token_scale=0.7)
    model.enable_fused(ops, gemm="sjd")
    prompt = lumina_prompt(a.kv - 700, grid, grid, seed=5)
    spec = lumina_window_spec(prompt, dev)
    cfg = SJDConfig(jacobi_loop_interval_l=0, jacobi_loop_interval_r=grid * grid + grid - 13, max_num_new_tokens=window, guidance_scale=3.0,
                    seed=5, max_length=len(prompt) + 700, eos_token_ids=(8196,))
    model.setup_cache(batch=2, s_max=((len(prompt) + 700 + 2 * window + 64 + 31) // 32) * 32)
    eng = SJDEngine(model, margs.vocab_size, dev, max_window=window, use_graph=True)
    log = []
    seq, stats = eng.decode(prompt, spec, LuminaGrammar(2000, 10), cfg, iter_log=log)
    torch.cuda.synchronize()
    steps = [(log[i + 1][3] - log[i][3]) * 1e3 for i in range(len(log) - 81, len(log) - 1)]          # the last 80 iterations: every graph captured long ago
    engine_ms = sum(steps) / len(steps)
    keys = [k for k in eng._graphs if isinstance(k, tuple) and k[0] == "win"]
    g = eng._graphs[keys[-1]]
    # (a) back to back
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        g.replay()
    torch.cuda.synchronize()
    back_to_back = (time.perf_counter() - t0) * 1e3 / a.iters
    # (b) upload + replay + spin on the mirror
    t0 = time.perf_counter()
    for _ in range(a.iters):
        eng._seq = (eng._seq % 0x7FFFFFF0) + 1
        eng.params.view.iter_seq = eng._seq
        eng.params.upload()
        g.replay()
        eng.state.wait_mirror(eng._seq)
    loop = (time.perf_counter() - t0) * 1e3 / a.iters
    if os.environ.get("SJD_PROBE_PROFILE") == "1":      # where the host's share goes: cProfile over a second decode (graphs already captured)
        import cProfile
        import pstats
        eng2 = eng
        pr = cProfile.Profile()
        pr.enable()
        seq2, stats2 = eng2.decode(prompt, spec, LuminaGrammar(2000, 10), cfg)
        pr.disable()
        print("iterations", stats2.nfe)
        pstats.Stats(pr).sort_stats("tottime").print_stats(28)
    print(json.dumps(dict(model="lumina7b", kv_len=int(stats.kv_len), graph_key=str(keys[-1])[:120], iters=a.iters,
                          a_back_to_back_ms=round(back_to_back, 4), b_upload_replay_spin_ms=round(loop, 4), c_engine_ms_per_step=round(engine_ms, 4),
                          boundary_us=round((loop - back_to_back) * 1e3, 1), host_arithmetic_us=round((engine_ms - loop) * 1e3, 1))))


if __name__ == "__main__":
    main()
