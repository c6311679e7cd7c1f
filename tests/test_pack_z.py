"""The 12-bit lossless weight stream of kernels G1z / G1sz (sjd_amd.ops.pack_weight_z, include/sjd_hip.h): a NumPy restatement of what the
kernel does with a record -- decode (low byte + 4-bit code + unit base), then patch the unit's exceptions -- must give back the uncompressed
packed stream of ops.pack_weight bit for bit.  CPU only (the GPU side is tests/test_gpu_glue.py::test_g1z_*)."""
import numpy as np
import pytest
import torch

import sjd_amd.ops as ops


def decode_z(pz):
    """PackedZ -> the uint16 stream pack_weight(weight, KC, step_major) holds (the kernel's g1z_decode + g1z_patch, restated)."""
    N, K, KC = pz.N, pz.K, pz.KC
    T = N // 32
    data = pz.data.cpu().numpy().view(np.uint32)
    exc = pz.exc.cpu().numpy().view(np.uint32)                   # [n_chunks, T, cap, 2], cap = 32 / 64 / 128
    out, off = [], 0
    for c, k0 in enumerate(range(0, K, KC)):
        S = min(KC, K - k0) // 16
        P = (S + 1) // 2                                         # record pairs (k-steps 2p, 2p + 1) of 384 words
        rec = data[off:off + T * P * 384]
        off += T * P * 384
        rec = rec.reshape(P, T, 384).transpose(1, 0, 2) if pz.step_major else rec.reshape(T, P, 384)
        lo = rec[..., :256].reshape(T, P, 64, 2, 2).transpose(0, 1, 3, 2, 4).reshape(T, 2 * P, 64, 2)[:, :S]      # [t, s, lane, {0..3, 4..7}]
        cw = rec[..., 256:].reshape(T, P, 64, 2).transpose(0, 1, 3, 2).reshape(T, 2 * P, 64)[:, :S]
        base4 = (exc[c, :, 0, 0].astype(np.uint32) * np.uint32(0x01010101))[:, None, None]
        hA = ((((cw & 0x08080808) << 4) | (cw & 0x07070707)) + base4).astype(np.uint32)
        c2 = cw >> 4
        hB = ((((c2 & 0x08080808) << 4) | (c2 & 0x07070707)) + base4).astype(np.uint32)
        w16 = np.zeros((T, S, 64, 8), dtype=np.uint16)
        for j in range(4):
            w16[..., j] = ((lo[..., 0] >> (8 * j)) & 0xFF) | (((hA >> (8 * j)) & 0xFF) << 8)
            w16[..., 4 + j] = ((lo[..., 1] >> (8 * j)) & 0xFF) | (((hB >> (8 * j)) & 0xFF) << 8)
        for t in range(T):                                       # exceptions: entries 1 .. cap - 1 of the unit's header
            if exc[c, t, 0, 1] == 0xFFFFFFFF:                     # a raw unit: entry 1 = where its records are (spliced in below)
                assert exc[c, t, 0, 0] == 0 and np.all(exc[c, t, 2:, 0] == 0xFFFFFFFF)
                continue
            n = int(exc[c, t, 0, 1])
            for i in range(1, 1 + n):
                pos, val = int(exc[c, t, i, 0]), int(exc[c, t, i, 1])
                w16[t, pos >> 9, (pos >> 3) & 63, pos & 7] = val & 0xFFFF
            assert np.all(exc[c, t, 1 + n:, 0] == 0xFFFFFFFF)
        if pz.n_raw:                                             # raw units (round 6): zero-filled in the stream, verbatim in raw_data
            idx = pz.raw_index.cpu().numpy()
            rd = pz.raw_data.cpu().view(torch.int16).numpy().view(np.uint16)       # [n_raw, KC / 16, 512] = [s, lane, j]
            for u, (cc, t) in enumerate(idx):
                if cc == c:
                    assert np.all(w16[t] == 0)                                     # the stream holds zeros there
                    roff = int(exc[c, t, 1, 0])                                    # byte offset of the unit's records from the start of `exc`
                    assert roff == exc.size * 4 + u * (KC // 16) * 1024 and roff % 16 == 0
                    assert pz.raw_data.data_ptr() + u * (KC // 16) * 1024 == pz.exc.data_ptr() + roff     # ... ONE allocation
                    w16[t] = rd[u, :S].reshape(S, 64, 8)
        if pz.step_major:
            w16 = w16.transpose(1, 0, 2, 3)
        out.append(w16.reshape(-1))
    return np.concatenate(out)


def raw_stream(w, KC, step_major):
    return ops.pack_weight(w, KC, step_major).view(torch.int16).cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("N,K,KC,step_major", [(64, 256, 128, False), (96, 512, 512, True), (128, 1040, 512, False), (64, 896, 896, True),
                                               (32, 4096, 2048, True)])
def test_z_stream_decodes_to_the_uncompressed_stream(N, K, KC, step_major):
    g = torch.Generator().manual_seed(N + K)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    pz = ops.pack_weight_z(w, KC, step_major)
    pairs = sum((min(KC, K - k0) // 16 + 1) // 2 for k0 in range(0, K, KC))      # an odd last k-step of a chunk is padded to a pair
    assert pz is not None and pz.data.numel() == (N // 32) * pairs * 1536
    assert np.array_equal(decode_z(pz), raw_stream(w, KC, step_major))


def test_z_exceptions_carry_outliers_zeros_and_specials():
    g = torch.Generator().manual_seed(7)
    N, K, KC = 64, 512, 256
    w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16)
    # a handful of weights far outside any 16-binade window, in both units' first and last k-steps / lanes / elements
    w[0, 0], w[31, 255], w[32, 256], w[63, 511] = 0.0, 3.0e4, -1.0e-30, float("inf")
    w[5, 17], w[40, 300] = -0.0, float("nan")
    w[7, 8:12] = torch.tensor([1e-20, -1e-20, 1e20, -1e20]).to(torch.bfloat16)
    for sm in (False, True):
        pz = ops.pack_weight_z(w, KC, sm)
        assert pz is not None and pz.n_exceptions >= 10
        assert np.array_equal(decode_z(pz), raw_stream(w, KC, sm))


def test_z_packer_declines_only_what_the_format_cannot_hold():
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(64, 256, generator=g) * 0.02).to(torch.bfloat16)
    assert ops.pack_weight_z(w.to(torch.float16), 128) is None                    # fp16: the high byte does not concentrate
    assert ops.pack_weight_z(torch.zeros(32, 8192, dtype=torch.bfloat16), 8192) is None      # KC > 4096: the header's position field
    w3 = w.clone()
    w3[0, :64] = torch.logspace(-30, 30, 64).to(torch.bfloat16)                    # 64 weights spread over 200 binades in one unit: a header of 64
    pz3 = ops.pack_weight_z(w3, 128)
    assert pz3 is not None and pz3.cap == 64 and pz3.n_raw == 0 and np.array_equal(decode_z(pz3), raw_stream(w3, 128, False))


@pytest.mark.parametrize("step_major", [False, True])
def test_z_raw_units_carry_what_no_header_holds(step_major):
    """round 6: a unit with more than 127 out-of-window weights used to make the packer decline the WHOLE matrix; now that unit alone travels
    verbatim (PackedZ.raw_*), the rest of the matrix stays in the 12-bit stream, and the pack statistics say so."""
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(96, 400, generator=g) * 0.02).to(torch.bfloat16)             # chunks of 128, 128, 128, 16 columns
    w[:16, :128] = 0.0                                                             # half a unit of zeros (pruned / padded rows): unit (0, 0)
    w[40, 128:256] = 0.0                                                           # one zero row inside unit (1, 1): 128 zeros
    w[64:72, 256:288] = torch.logspace(-30, 30, 256).reshape(8, 32).to(torch.bfloat16)      # 256 weights over 200 binades: unit (2, 2)
    w[32:64, :128] = 0.0                                                           # a unit of NOTHING but zeros packs (its window sits at exponent 0)
    pz = ops.pack_weight_z(w, 128, step_major)
    assert pz is not None and pz.n_raw == 3 and pz.raw_index.tolist() == [[0, 0], [1, 1], [2, 2]]
    assert pz.stats["units"] == 12 and pz.stats["raw_units"] == 3 and pz.stats["max_exceptions"] <= 127
    assert pz.raw_data.shape == (3, 8, 512)
    assert np.array_equal(decode_z(pz), raw_stream(w, 128, step_major))
    w[:20, 384:] = 0.0                                                             # a raw unit in the ragged last chunk (one k-step): padded to KC / 16 records
    pz = ops.pack_weight_z(w, 128, step_major)
    assert pz.n_raw == 4 and pz.raw_index.tolist()[-1] == [3, 0] and np.array_equal(decode_z(pz), raw_stream(w, 128, step_major))


def test_z_gateup_packing_lists_whole_pairs():
    """[Wg; Wu] packed for sjd_gateup_silu_z (KC = K / 2): a raw unit anywhere in (gate tile t | up tile t) x (K half 0 | 1) lists the pair t, its
    four units in the order [K half][gate | up]"""
    g = torch.Generator().manual_seed(9)
    I, K = 128, 512
    w = (torch.randn(2 * I, K, generator=g) * 0.02).to(torch.bfloat16)
    w[I + 32:I + 40, 256:512] = 0.0                                                # up tile 1, K half 1: eight zero rows
    w[96:100, :256] = 0.0                                                          # gate tile 3, K half 0: four zero rows
    pz = ops.pack_weight_z(w, K // 2, True, gateup=True)
    assert pz.raw_tiles.tolist() == [1, 3] and pz.n_raw_pairs == 2 and pz.n_raw == 8
    assert pz.raw_index.tolist() == [[0, 1], [0, 5], [1, 1], [1, 5], [0, 3], [0, 7], [1, 3], [1, 7]]
    assert np.array_equal(decode_z(pz), raw_stream(w, K // 2, True))
    assert ops.pack_weight_z(w, K // 2, True).n_raw == 2                           # without the closure: the two raw units alone


def test_z_header_capacity_follows_the_weights():
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(64, 2048, generator=g) * 0.02).to(torch.bfloat16)
    assert ops.pack_weight_z(w, 1024).cap == 32
    torch.manual_seed(5)
    heavy = (torch.distributions.StudentT(2.0).sample((64, 2048)) * 0.015).to(torch.bfloat16)      # very heavy tails: 50-90 exceptions per unit
    for KC, cap in ((1024, 64), (2048, 128)):
        pz = ops.pack_weight_z(heavy, KC)
        assert pz is not None and pz.cap == cap and np.array_equal(decode_z(pz), raw_stream(heavy, KC, False))


def test_z_window_follows_the_bulk_not_the_outlier():
    g = torch.Generator().manual_seed(11)
    w = (torch.randn(32, 2048, generator=g) * 0.02).to(torch.bfloat16)
    w[3, 100] = 50.0                                                               # one outlier 11 binades above the bulk
    pz = ops.pack_weight_z(w, 2048)
    assert pz is not None and pz.n_exceptions < 16
    assert np.array_equal(decode_z(pz), raw_stream(w, 2048, False))
