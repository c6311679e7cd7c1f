#!/usr/bin/env python3
"""Randomised teacher-forced whole-loop parity (engine vs CPU oracle) over seeds / window sizes / intervals / schemes / flavours.
Not part of the test suite (minutes of GPU time): `python tools/fuzz_loops.py --n 40`.  Exits non-zero on the first mismatch."""
import argparse
import os
import random
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=24)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--poison", action="store_true", help="fill the allocator's free pool with 0xFF (NaN / -1) before every case: "
                    "a kernel that reads workspace nobody wrote then fails in every order (tests/conftest.py::poison_device_memory)")
    a = ap.parse_args()
    from tests import gpu_loop_check as G
    from tests.conftest import poison_device_memory
    rnd = random.Random(a.seed)
    ok = 0
    for i in range(a.n):
        kind = rnd.choice(["llamagen", "lumina", "lumina", "emu3", "anole", "anole_api", "batch", "batch"])
        seed = rnd.randrange(1, 10000)
        if a.poison:
            poison_device_memory(total_gib=12)
        try:
            if kind == "llamagen":
                kw = dict(seed=seed, window=rnd.choice([4, 8, 16, 32, 64]), scheme=rnd.choice(["speculative_jacobi", "jacobi"]),
                          embed_token_scale=rnd.choice([0.1, 0.25, 0.5, 1.0]), top_k=rnd.choice([0, 10, 1000]),
                          top_p=rnd.choice([1.0, 0.95, 0.7]), use_graph=rnd.random() < 0.5, latent=rnd.choice([8, 12, 16]))
                r = G.teacher_forced_llamagen_check(**kw)
            elif kind == "lumina":
                kw = dict(seed=seed, window=rnd.choice([2, 8, 16, 32, 64]), hg=rnd.choice([2, 3, 4]), wg=rnd.choice([2, 4, 5]),
                          kv_heads=rnd.choice([4, 2, 1]), l=rnd.choice([0, 1, 3]), embed_token_scale=rnd.choice([0.1, 0.25, 0.6]),
                          scheme=rnd.choice(["speculative_jacobi", "speculative_jacobi", "jacobi"]), use_graph=rnd.random() < 0.6,
                          fused=True, gemm=rnd.choice(["torch", "sjd"]), fp8_kv=rnd.random() < 0.3,
                          dtype=rnd.choice([torch.bfloat16, torch.float16]),
                          init_scheme=rnd.choice(["random", "random", "repeat_horizon", "sample_horizon"]),
                          temperature=rnd.choice([1.0, 1.0, 0.6, 0.85, 1.4, 2.2]), top_p=rnd.choice([None, None, 0.95, 0.8, 0.5]))
                r = G.teacher_forced_lumina_check(**kw)
                r.pop("windows", None)
            elif kind == "emu3":
                kw = dict(seed=seed, H=rnd.choice([2, 3, 4]), W=rnd.choice([3, 5, 6]), window=rnd.choice([8, 16, 32, 64]),
                          pos_len=rnd.choice([5, 9, 12]), neg_len=rnd.choice([3, 5, 12]), gemm=rnd.choice(["torch", "sjd"]),
                          use_graph=rnd.random() < 0.6, init_scheme=rnd.choice(["random", "repeat_horizon", "sample_horizon"]))
                r = G.teacher_forced_emu3_check(**kw)
                r = {k: v for k, v in r.items() if k != "gen"}
            elif kind == "anole_api":
                kw = dict(seed=seed, img_len=rnd.choice([24, 36, 49]), window=rnd.choice([4, 16]), P=rnd.choice([6, 10, 13]),
                          mode=rnd.choice(["image-only", "interleaved-text-image", "text-only"]), extra_new_tokens=rnd.choice([0, 9, 20]),
                          gemm=rnd.choice(["torch", "sjd"]))
                r = G.teacher_forced_anole_api_check(**kw)
                r = {k: v for k, v in r.items() if k != "gen"}
            elif kind == "anole":
                kw = dict(seed=seed, img_len=rnd.choice([24, 40, 57]), window=rnd.choice([4, 16]), fp8_kv=rnd.random() < 0.5,
                          gemm=rnd.choice(["torch", "sjd"]), use_graph=rnd.random() < 0.6)
                r = G.teacher_forced_anole_check(**kw)
            else:
                n_prompts = rnd.choice([2, 2, 3, 4])
                kw = dict(seed=seed, n_prompts=n_prompts, window=rnd.choice([8, 16] if n_prompts < 4 else [4, 8, 16]),
                          P=rnd.choice([(12, 9, 14, 7), (7, 7, 7, 7), (10, 15, 5, 11)]), gemm=rnd.choice(["torch", "sjd"]),
                          use_graph=rnd.random() < 0.6, fp8_kv=rnd.random() < 0.3,
                          init_scheme=rnd.choice(["random", "random", "repeat_horizon", "sample_horizon"]))
                r = G.teacher_forced_batch_check(**kw)
            ok += 1
            print(f"[{i}] ok {kind} {kw} -> {r if not isinstance(r, list) else r[:2]}", flush=True)
        except Exception:
            print(f"[{i}] MISMATCH {kind} {kw}", flush=True)
            traceback.print_exc()
            sys.exit(1)
    print(f"{ok}/{a.n} runs identical to the oracle")


if __name__ == "__main__":
    main()
