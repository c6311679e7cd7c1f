"""CPU: host-side decisions added in round 4 -- which K1 form a window launch takes, which shapes the one-launch MLP serves."""
import os

import pytest
import torch


def _ops():
    import sjd_amd.ops as ops
    return ops


def test_colsplit_serves_single_prompt_mha_windows_only():
    ops = _ops()
    assert ops.colsplit_ok(2, 16, 32, 32, 128, torch.bfloat16) and ops.colsplit_ok(2, 16, 32, 32, 128, ops.FP8)
    assert ops.colsplit_ok(1, 5, 32, 32, 128, torch.float16)
    assert not ops.colsplit_ok(2, 16, 32, 8, 128, torch.bfloat16)          # grouped-query heads: the ring kernel's shapes
    assert not ops.colsplit_ok(2, 32, 32, 32, 128, torch.bfloat16)         # two row chunks
    assert not ops.colsplit_ok(2, 16, 12, 12, 64, torch.bfloat16)          # head size 64
    assert not ops.colsplit_ok(8, 16, 32, 32, 128, torch.bfloat16)         # four prompts per forward fill the chip without any split
    assert not ops.colsplit_ok(2, 16, 32, 32, 128, torch.float32)


def test_k1_regime_follows_the_context_length(monkeypatch):
    ops = _ops()
    monkeypatch.delenv("SJD_K1_REGIME", raising=False)
    a = ops.HipWindowAttention()
    assert a.regime == "keysplit"
    lim16, lim8 = a.COLSPLIT_MAX_KEYS["16bit"], a.COLSPLIT_MAX_KEYS["fp8"]
    assert lim16 < lim8                                                    # fp8 rows are half as long: the column split stays ahead twice as far
    assert a.choose_regime(64 + 16, torch.bfloat16) == "colsplit" and a.regime == "colsplit"
    assert a.choose_regime(lim16, torch.float16) == "colsplit"
    assert a.choose_regime(lim16 + 1, torch.bfloat16) == "keysplit" and a.regime == "keysplit"
    assert a.choose_regime(lim16 + 1, ops.FP8) == "colsplit" and a.choose_regime(lim8 + 1, ops.FP8) == "keysplit"
    monkeypatch.setenv("SJD_K1_REGIME", "keysplit")
    b = ops.HipWindowAttention()
    assert b.choose_regime(80, torch.bfloat16) == "keysplit"              # pinned (A/B aid)


def test_mlp_pair_shapes():
    ops = _ops()
    dev = torch.device("cpu")
    # not two 12-bit packed weights -> never
    assert not ops.mlp_pair_ok(32, 11008, 4096, object(), object(), 768, 8, dev)
    z = lambda kc: ops.PackedZ(torch.zeros(1, dtype=torch.uint8), torch.zeros(1, 1, 32, 2, dtype=torch.int32), 1, 1, kc, False, 0)
    gu, dn = z(2048), z(768)
    # shape rules are checked before the device is asked how many workgroups it holds
    assert not ops.mlp_pair_ok(64, 11008, 4096, gu, dn, 768, 8, dev)      # a 64-row window
    assert not ops.mlp_pair_ok(32, 11008, 2048, gu, dn, 768, 8, dev)      # hidden != 4096
    assert not ops.mlp_pair_ok(32, 11008, 4096, gu, dn, 800, 8, dev)      # KC of down not a multiple of 64
    assert not ops.mlp_pair_ok(32, 11008, 4096, gu, dn, 768, 6, dev)      # six column tiles per down workgroup
    assert not ops.mlp_pair_ok(32, 11008, 4096, gu, z(896), 768, 8, dev)  # packed with another K chunk


def test_engine_graph_keys_carry_the_k1_regime():
    """the two K1 forms are different kernels: a graph captured under one must never be replayed under the other"""
    import inspect
    import sjd_amd.engine as E
    src = inspect.getsource(E.SJDEngine)
    assert src.count("self._k1_regime()") >= 3 and "choose_regime(kv_len + n_rows" in src
