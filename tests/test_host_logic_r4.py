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


def test_choose_regime_follows_the_launch_shape():
    """a shape the column split does not serve is "keysplit" whatever the context length or the pin (the behavioural graph-vs-eager
    check across the threshold is tests/test_gpu_regime_graphs.py)"""
    ops = _ops()
    attn = ops.HipWindowAttention.__new__(ops.HipWindowAttention)
    attn._pin_regime, attn.regime = None, "keysplit"
    assert attn.choose_regime(100, torch.bfloat16, shape=(2, 16, 32, 32, 128)) == "colsplit"
    assert attn.choose_regime(4000, torch.bfloat16, shape=(2, 16, 32, 32, 128)) == "keysplit"
    assert attn.choose_regime(100, torch.float16, shape=(2, 32, 32, 8, 128)) == "keysplit"        # Emu3: GQA, 32-row window
    assert attn.choose_regime(100, torch.bfloat16, shape=(8, 16, 32, 32, 128)) == "keysplit"       # four prompts per forward
    attn._pin_regime = "colsplit"
    assert attn.choose_regime(100, torch.float16, shape=(2, 32, 32, 8, 128)) == "keysplit"
    assert attn.choose_regime(4000, torch.bfloat16, shape=(2, 16, 32, 32, 128)) == "colsplit"


def test_head_combine_serves_wide_aligned_windows_only():
    """K2a (sjd_head_combine) is put in front of K2 for wide head windows only, and only where its 16-byte accesses are legal"""
    ops = _ops()
    import sjd_amd._lib as L

    def hp(n_cols, row_stride=None, chunk_stride=None, part=1 << 20):
        h = L.HeadPartials()
        h.part, h.n_chunks, h.n_cols = part, 2, n_cols
        h.row_stride = n_cols if row_stride is None else row_stride
        h.chunk_stride = 64 * h.row_stride if chunk_stride is None else chunk_stride
        return h

    assert ops.head_combine_ok(hp(32800))                       # Emu3's visual window (+ padding to whole 32-column tiles)
    assert not ops.head_combine_ok(hp(8224))                    # Lumina's image window: the extra launch costs more than it saves
    assert not ops.head_combine_ok(hp(32800, part=(1 << 20) + 4))     # planes not 16-byte aligned
    assert not ops.head_combine_ok(hp(32802))                   # not whole float4 groups
    assert not ops.head_combine_ok(hp(32800, row_stride=32802))


def test_emu3_on_the_12bit_stream_has_its_own_launch_shapes():
    import sjd_amd.backbones as BB
    z, raw = BB.ChameleonBackbone.G1_CFG_EMU3_Z, BB.ChameleonBackbone.G1_CFG_EMU3
    assert set(z) == set(raw) == {"qkv", "o", "gate_up", "down"}
    for name in z:                                              # <= 8 waves at 64 rows on the 12-bit stream; the uncompressed set runs on kernel G1w since late
        assert z[name][1] <= 8                                  # round 6 (its own chunks; the second number = column tiles per workgroup)
        assert raw[name][1] in BB.ChameleonBackbone.G1_WIDE_TILES and 4096 % 16 == 0 and raw[name][0] % 64 == 0
    assert (z["qkv"][0], z["o"][0], z["gate_up"][0], z["down"][0]) == (512, 512, 2048, 896)      # round 4's sweep (profiles/r4_g1z_sweep_emu3_64rows.jsonl)
    # 256 workgroups for q|k|v (6144 columns) and o (4096 columns) at K = 4096
    assert (6144 // 32 // z["qkv"][1]) * (4096 // z["qkv"][0]) == 256 and (4096 // 32 // z["o"][1]) * (4096 // z["o"][0]) == 256
