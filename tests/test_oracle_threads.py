"""CPU: the oracle's K2 restatement runs one window row per OpenMP thread (bench.py's cpu_baseline uses all host cores); its result must
not depend on the thread count (round-2 regression: a static accumulator array made the parallel run racy)."""
import numpy as np
import torch

from oracle import sjd_oracle as O


def test_k2_restatement_is_thread_count_independent():
    g = torch.Generator().manual_seed(0)
    V, L = 16384, 16
    lg = (torch.randn(2, L, V, generator=g) * 3).numpy()
    noise = torch.empty(L, V).exponential_(generator=g).numpy()
    rules = O.lumina_rules([9000] * 5 + [8197, 8808, 8808] + [100] * 7, L, 2000, 10)
    out = []
    for n in (1, 4, 8):
        assert O.set_threads(n) == n
        toks, probs = O.logits_to_probs_sample(lg[0], lg[1], 3.0, rules, noise)
        out.append((toks.copy(), probs.view(np.uint32).copy()))
    for toks, bits in out[1:]:
        assert (toks == out[0][0]).all() and np.array_equal(bits, out[0][1])
