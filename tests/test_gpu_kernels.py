"""GPU parity: HIP kernels (through the C-ABI) vs the CPU oracle on the same seeded inputs.

Bar: K2 / K4 / K5 bit-exact (token ids, accept length, AND fp32 probabilities -- the oracle and the kernels share a
canonical summation order / exp); K1+K3 within bf16 tolerance of the fp64 attention reference.
"""
import ctypes
import zlib

import numpy as np
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sjd_oracle as O
from oracle.attention_ref import OracleWindowAttention
from tests.helpers import make_pq


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _ops():
    import sjd_amd.ops as ops
    import sjd_amd._lib as L
    L.load()
    return ops, L


def to_dev_rule(L, ops, r):
    rr = L.RowRule()
    ctypes.memmove(ctypes.byref(rr), ctypes.byref(r), ctypes.sizeof(rr))
    return rr


def run_k2(dev, logits, use_cfg, guidance, rules, noise, max_rows=None):
    ops, L = _ops()
    n, V = noise.shape
    max_rows = max_rows or n
    params = ops.DeviceBlob(L.IterParams, dev)
    params.view.n_rows, params.view.use_cfg = n, int(use_cfg)
    for j, r in enumerate(rules):
        params.view.rules[j] = to_dev_rule(L, ops, r)
    params.upload()
    lg = torch.from_numpy(logits).to(dev)
    nz = torch.zeros(max_rows, V, device=dev)
    nz[:n] = torch.from_numpy(noise).to(dev)
    probs = torch.full((max_rows, V), -7.0, device=dev)
    toks = torch.full((max_rows,), -5, dtype=torch.int64, device=dev)
    ops.logits_to_probs_sample(lg[0], lg[1], guidance, params, nz, probs, ctypes.c_void_p(toks.data_ptr()))
    torch.cuda.synchronize()
    return toks[:n].cpu().numpy(), probs[:n].cpu().numpy()


K2_CASES = [
    # name, V, n, rules builder
    ("lumina_image_65536", 65536, 16, lambda n: O.lumina_rules([9000] * 5 + [8197, 8828, 8828] + [100] * 40, n, 2000, 10)),
    ("lumina_eol_rows", 9216, 16, lambda n: O.lumina_rules([9000] * 5 + [8197, 8808, 8808] + [100] * 3, n, 2000, 10)),
    ("lumina_text_mode", 65536, 3, lambda n: O.lumina_rules([9000] * 7, n, 2000, 10)),
    ("llamagen_16384", 16384, 16, lambda n: O.llamagen_rules([], n, 1000, 1.0)),
    ("llamagen_no_topk", 16384, 5, lambda n: O.llamagen_rules([], n, 0, 1.0)),
    ("emu3_odd_vocab", 184622, 8, lambda n: O.emu3_rules([5, 6, 151851] + [151854 + 7] * 88, n, 90, 90, 151854, 32768,
                                                         151851, 151853, 151850, 151846, 151847, 151643, 2048)),
    ("tiny_vocab", 1000, 2, lambda n: O.llamagen_rules([], n, 10, 1.0)),
    ("llamagen_top_p_0.9", 16384, 16, lambda n: O.llamagen_rules([], n, 1000, 0.9)),
    ("llamagen_top_p_only", 16384, 4, lambda n: O.llamagen_rules([], n, 0, 0.5)),
    ("lumina_range_top_p", 65536, 4, lambda n: [O.rule(((4, 8196),), -1, 2000, 0.95) for _ in range(n)]),
    # HF's TemperatureLogitsWarper behind the processors (GenerationConfig.temperature != 1): sharpened, flattened, with top-p, with EOL rows
    ("lumina_image_T0.7", 65536, 16, lambda n: O.tempered(lambda c, k: O.lumina_rules(c, k, 2000, 10), 0.7)([9000] * 5 + [8197, 8828, 8828] + [100] * 40, n)),
    ("lumina_eol_rows_T1.5", 9216, 16, lambda n: O.tempered(lambda c, k: O.lumina_rules(c, k, 2000, 10), 1.5)([9000] * 5 + [8197, 8808, 8808] + [100] * 3, n)),
    ("llamagen_top_p_T0.8", 16384, 8, lambda n: O.tempered(lambda c, k: O.llamagen_rules(c, k, 1000, 0.9), 0.8)([], n)),
    ("emu3_T2.0", 184622, 4, lambda n: O.tempered(lambda c, k: O.emu3_rules(c, k, 90, 90, 151854, 32768, 151851, 151853, 151850, 151846, 151847,
                                                                            151643, 2048), 2.0)([5, 6, 151851] + [151854 + 7] * 88, n)),
]


@pytest.mark.parametrize("name,V,n,builder", K2_CASES, ids=[c[0] for c in K2_CASES])
@pytest.mark.parametrize("use_cfg", [True, False])
def test_k2_bit_exact(dev, name, V, n, builder, use_cfg):
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    logits = (torch.randn(2, n, V, generator=g) * 3.0).numpy()
    noise = torch.empty(n, V).exponential_(generator=g).numpy()
    rules = builder(n)
    toks_ref, probs_ref = O.logits_to_probs_sample(logits[0], logits[1] if use_cfg else None, 3.0, rules, noise)
    toks, probs = run_k2(dev, logits, use_cfg, 3.0, rules, noise, max_rows=16)
    assert toks.tolist() == toks_ref.tolist()
    assert np.array_equal(probs.view(np.uint32), probs_ref.view(np.uint32)), \
        f"max |dp| = {np.abs(probs - probs_ref).max()}"


@pytest.mark.parametrize("name,V,n,builder,cols", [
    ("lumina_image_65536", 65536, 16, K2_CASES[0][3], (0, 8224)),
    ("lumina_eol_rows", 9216, 16, K2_CASES[1][3], (0, 8224)),
    ("emu3_odd_vocab", 184622, 8, K2_CASES[5][3], (151840, 184622)),
    ("lumina_range_top_p", 65536, 4, K2_CASES[9][3], (0, 8224)),
], ids=lambda v: v if isinstance(v, str) else None)
def test_k2_column_window_of_the_output_head(dev, name, V, n, builder, cols):
    """The engine evaluates the output head only for the vocabulary columns the grammar allows (engine.logit_columns) and hands
    K2 the compact logits with the column offset: bit-identical to the oracle on the full-width logits."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    logits = (torch.randn(2, n, V, generator=g) * 3.0).numpy()
    noise = torch.empty(n, V).exponential_(generator=g).numpy()
    rules = builder(n)
    toks_ref, probs_ref = O.logits_to_probs_sample(logits[0], logits[1], 3.0, rules, noise)
    params = ops.DeviceBlob(L.IterParams, dev)
    params.view.n_rows, params.view.use_cfg = n, 1
    for j, r in enumerate(rules):
        params.view.rules[j] = to_dev_rule(L, ops, r)
    params.upload()
    compact = torch.from_numpy(np.ascontiguousarray(logits[:, :, cols[0]:cols[1]])).to(dev)
    nz = torch.from_numpy(noise).to(dev)
    probs = torch.full((n, V), -7.0, device=dev)
    toks = torch.full((n,), -5, dtype=torch.int64, device=dev)
    ops.logits_to_probs_sample(compact[0], compact[1], 3.0, params, nz, probs, ctypes.c_void_p(toks.data_ptr()), col0=cols[0])
    torch.cuda.synchronize()
    assert toks.cpu().tolist() == toks_ref.tolist()
    assert np.array_equal(probs.cpu().numpy().view(np.uint32), probs_ref.view(np.uint32))

    class _E:            # the host-side choice of that window
        V, narrow_head = None, True
        _cols_cache, _rule_keep = {}, []
    from sjd_amd.engine import SJDEngine
    e = _E()
    e.V = V
    want = cols if 2 * (cols[1] - cols[0]) <= V else None      # only worth a separate graph when it halves the head
    assert SJDEngine.logit_columns(e, [to_dev_rule(L, ops, r) for r in rules]) == want


def test_k2_matches_torch_softmax(dev):
    """Independent fp32 torch reference of the same op (tolerance, not bit-exact)."""
    V, n = 65536, 16
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(2, n, V, generator=g) * 3.0
    noise = torch.empty(n, V).exponential_(generator=g)
    rules = O.llamagen_rules([], n, 2000, 1.0)
    toks, probs = run_k2(dev, logits.numpy(), True, 3.0, rules, noise.numpy())
    z = 3.0 * (logits[0] - logits[1]) + logits[1]
    kth = torch.topk(z, 2000)[0][..., -1, None]
    ref = torch.softmax(z.masked_fill(z < kth, -float("inf")), dim=-1)
    np.testing.assert_allclose(probs, ref.numpy(), atol=1e-6, rtol=1e-5)
    assert toks.tolist() == torch.argmax(ref / noise, dim=-1).tolist()


def run_k4(dev, win, Y, p, q_rows, rs, resid, e2, scheme=0, max_rows=16):
    ops, L = _ops()
    n, V = p.shape
    params = ops.DeviceBlob(L.IterParams, dev)
    state = ops.DeviceBlob(L.State, dev)
    params.view.n_rows, params.view.scheme = n, scheme
    params.view.iter_seq = 1000 + n + scheme
    for j, r in enumerate(resid):
        params.view.resid_rules[j] = to_dev_rule(L, ops, r)
    prev = torch.zeros(max_rows, V, device=dev)
    for i in range(n):
        state.view.win_tok[i] = int(win[i])
        state.view.tokens[i] = int(Y[i])
        if q_rows[i] is None:
            state.view.q_src[i] = -1
        else:
            state.view.q_src[i] = i
            prev[i] = torch.from_numpy(np.asarray(q_rows[i])).to(dev)
    params.upload()
    state.dev.copy_(state.host)
    probs = torch.zeros(max_rows, V, device=dev)
    probs[:n] = torch.from_numpy(p).to(dev)
    rsd = torch.zeros(max_rows, V, device=dev)
    rsd[:n] = torch.from_numpy(rs).to(dev)
    # K4 writes the state into the blob's pinned host copy itself (host_mirror): what the host reads after ONE stream wait must be
    # byte for byte what a D2H copy of the device state gives
    state.host.fill_(0xA5)
    ops.verify_accept(params, state, probs, prev, rsd, torch.from_numpy(e2).to(dev), torch.empty(V, device=dev), mirror=True)
    state.wait_mirror(seq=params.view.iter_seq)          # host spin on the sequence word K4 publishes behind the state (no HIP call)
    mirrored = bytes(state.host.numpy().tobytes())
    st = state.download()
    assert mirrored == state.host.numpy().tobytes(), "K4's host mirror differs from the device state"
    return st.m, [st.tokens[i] for i in range(n)], bool(st.rejected)


@pytest.mark.parametrize("mode,L,grammar", [("carried", 16, None), ("mixed", 16, None), ("fresh", 16, None),
                                            ("far", 16, None), ("equal", 8, None), ("mixed", 16, "lumina"),
                                            ("fresh", 2, None), ("mixed", 16, "llamagen"), ("mixed", 32, None),
                                            ("mixed", 16, "llamagen_topp"), ("far", 16, "llamagen_topp"),
                                            ("mixed", 16, "lumina_T0.7"), ("far", 16, "lumina_T1.5"), ("fresh", 16, "llamagen_topp_T0.6"),
                                            ("mixed", 32, "plain_T2.0")])
@pytest.mark.parametrize("V", [9216, 65536])
def test_k4_bit_exact(dev, mode, L, grammar, V):
    seed = 9000 + L + len(mode)
    p, q, draft = make_pq(V, L, seed, mode)
    g = torch.Generator().manual_seed(seed + 1)
    adv = torch.multinomial(p[0], 1, generator=g)[:, 0].numpy()
    rs = torch.rand((L, V), generator=g).numpy()
    e2 = torch.empty(V).exponential_(generator=g).numpy()
    win = draft[0].tolist()
    ctx = [9000] * 5 + [8197, 8808, 8808] + [100] * 3
    if grammar == "lumina":
        rfn = lambda c: O.lumina_rules(c, 1, 2000, 10)[0]
    elif grammar == "llamagen":
        rfn = lambda c: O.llamagen_rules(c, 1, 100, 1.0)[0]
    elif grammar == "llamagen_topp":
        rfn = lambda c: O.llamagen_rules(c, 1, 200, 0.8)[0]
    elif grammar == "lumina_T0.7":          # residual resample under a TemperatureLogitsWarper: softmax(log(max(p - q, 0)) / T)
        rfn = lambda c: O.tempered(lambda cc, k: O.lumina_rules(cc, k, 2000, 10), 0.7)(c, 1)[0]
    elif grammar == "lumina_T1.5":
        rfn = lambda c: O.tempered(lambda cc, k: O.lumina_rules(cc, k, 2000, 10), 1.5)(c, 1)[0]
    elif grammar == "llamagen_topp_T0.6":
        rfn = lambda c: O.tempered(lambda cc, k: O.llamagen_rules(cc, k, 200, 0.8), 0.6)(c, 1)[0]
    elif grammar == "plain_T2.0":
        rfn = lambda c: O.rule(temperature=2.0)
    else:
        rfn = lambda c: O.rule()
    resid = [rfn(ctx + win[1:i]) for i in range(1, L)]
    onehot = [bool((q[0, i] == 1).any()) and float(q[0, i].sum()) == 1.0 and i > 0 for i in range(L)]
    q_rows = [None if onehot[i] else q[0, i].numpy() for i in range(L)]
    m_ref, tok_ref, rej_ref = O.verify_accept(win, adv, p[0].numpy(), q_rows, rs, resid, e2)
    m, tok, rej = run_k4(dev, win, adv, p[0].numpy(), q_rows, rs, resid, e2, max_rows=32)
    assert (m, rej) == (m_ref, rej_ref)
    assert tok[:m] == tok_ref[:m].tolist()
    assert tok[m:] == adv[m:].tolist()       # the unverified tail is carried unchanged


def test_k4_jacobi_scheme(dev):
    V, L = 9216, 16
    g = torch.Generator().manual_seed(3)
    win = torch.randint(4, 8196, (L,), generator=g).tolist()
    Y = list(win[1:]) + [77]
    Y[6] = (Y[6] + 1) % 8000
    p = torch.softmax(torch.randn(L, V, generator=g), -1).numpy()
    m, tok, rej = run_k4(dev, win, Y, p, [None] * L, np.zeros((L, V), np.float32), [O.rule()] * L, np.ones(V, np.float32),
                         scheme=1)
    assert m == O.first_mismatch(win, Y) == 7 and not rej and tok == Y


def test_k5_window_assembly(dev):
    ops, L = _ops()
    params = ops.DeviceBlob(L.IterParams, dev)
    state = ops.DeviceBlob(L.State, dev)
    ids = torch.zeros(2, 16, dtype=torch.int64, device=dev)
    for n_prev, m, n in [(16, 3, 16), (16, 16, 16), (16, 1, 4), (1, 1, 16), (8, 5, 1), (16, 9, 12)]:
        Y = list(range(1000, 1000 + n_prev))
        a = max(0, min(n_prev - m, n - 1))
        fresh = list(range(5000, 5000 + n - 1 - a))
        state.view.m, state.view.n_prev = m, n_prev
        for i, t in enumerate(Y):
            state.view.tokens[i] = t
        params.view.n_rows, params.view.n_fresh = n, len(fresh)
        for i, t in enumerate(fresh):
            params.view.fresh_tok[i] = t
        params.upload()
        state.dev.copy_(state.host)
        ops.reguess(params, state, ids)
        torch.cuda.synchronize()
        st = state.download()
        want = [Y[m - 1]] + Y[m:m + a] + fresh
        assert [st.win_tok[i] for i in range(n)] == want
        assert [st.q_src[i] for i in range(n)] == [m - 1 + i for i in range(a + 1)] + [-1] * (n - 1 - a)
        assert ids[0, :n].tolist() == want and ids[1, :n].tolist() == want
        # the same launch with the position ids: kv_len + i + pos_offset[b]
        params.view.kv_len = 700 + n
        params.upload()
        off, pos = torch.tensor([0, -37], dtype=torch.int64, device=dev), torch.full((2, 16), -1, dtype=torch.int64, device=dev)
        ids.zero_()
        ops.reguess(params, state, ids, pos_offset=off, positions_out=pos)
        torch.cuda.synchronize()
        assert ids[0, :n].tolist() == want and ids[1, :n].tolist() == want
        assert pos[0].tolist() == [700 + n + i for i in range(16)] and pos[1].tolist() == [700 + n - 37 + i for i in range(16)]


class _Cache:
    def __init__(self, k, v):
        self.k, self.v = k, v


ATTN_CASES = [
    # name, B, H, Hkv, D, S_max, kv_len, n, key_start, dtype
    ("mha_d128_mid", 2, 4, 4, 128, 1280, 1216, 16, [0, 59], torch.bfloat16),
    ("mha_d128_empty_cache", 2, 4, 4, 128, 128, 0, 16, [0, 0], torch.bfloat16),
    ("mha_d128_unaligned", 2, 2, 2, 128, 256, 77, 16, [0, 33], torch.bfloat16),
    ("mha_d128_short_window", 2, 2, 2, 128, 256, 100, 5, [0, 10], torch.bfloat16),
    ("gqa4_d128", 2, 8, 2, 128, 4224, 4100, 16, [3, 0], torch.float16),
    ("gqa2_d128", 1, 4, 2, 128, 512, 300, 16, [0], torch.bfloat16),
    ("mha_d64_llamagen", 2, 12, 12, 64, 288, 200, 16, [0, 0], torch.bfloat16),
    ("prefill_60_rows", 2, 4, 4, 128, 128, 0, 60, [0, 59], torch.bfloat16),
    ("prefill_17_rows_offset", 1, 2, 2, 128, 128, 9, 17, [2], torch.bfloat16),
    # shared-tile kernel: (q head of the group, row chunk) pairs of one workgroup read each K/V tile once
    ("gqa4_window32_emu3", 2, 8, 2, 128, 2112, 2000, 32, [5, 0], torch.float16),
    ("gqa4_window20_ragged_chunk", 2, 8, 2, 128, 512, 333, 20, [0, 40], torch.bfloat16),
    ("gqa4_window32_short_cache", 1, 4, 1, 128, 128, 7, 32, [0], torch.bfloat16),
    ("gqa2_window32", 2, 4, 2, 128, 512, 300, 32, [0, 11], torch.bfloat16),
]


@pytest.mark.parametrize("case", ATTN_CASES, ids=[c[0] for c in ATTN_CASES])
@pytest.mark.parametrize("n_split", [1, 8, "colsplit"])
def test_k1_k3_attention(dev, case, n_split):
    ops, L = _ops()
    name, B, H, Hkv, D, S_max, kv_len, n, key_start, dtype = case
    regime = "keysplit"
    if n_split == "colsplit":        # round 4: no key splits, four workgroups per (batch, head) split the output columns (k1_dsplit)
        if not ops.colsplit_ok(B, n, H, Hkv, D, dtype):
            pytest.skip("the column split serves multi-head 16-row windows of head size 128")
        n_split, regime = 4, "colsplit"
    g = torch.Generator().manual_seed(len(name))
    kc = torch.randn(1, B, Hkv, S_max, D, generator=g).to(dtype)
    vc = torch.randn(1, B, Hkv, S_max, D, generator=g).to(dtype)
    q = (torch.randn(B, n, H, D, generator=g) * 1.5).to(dtype)
    k = torch.randn(B, n, Hkv, D, generator=g).to(dtype)
    v = torch.randn(B, n, Hkv, D, generator=g).to(dtype)
    ref_cache = _Cache(kc.clone(), vc.clone())
    ref = OracleWindowAttention()(0, q, k, v, ref_cache, kv_len, key_start).float()
    dcache = _Cache(kc.clone().to(dev), vc.clone().to(dev))
    attn = ops.HipWindowAttention(n_split=n_split)
    attn.regime = regime
    out = attn(0, q.to(dev), k.to(dev), v.to(dev), dcache, kv_len, key_start)
    torch.cuda.synchronize()
    # K3: the cache rows were appended exactly
    assert torch.equal(dcache.k.cpu()[0, :, :, :kv_len + n], ref_cache.k[0, :, :, :kv_len + n])
    assert torch.equal(dcache.v.cpu()[0, :, :, :kv_len + n], ref_cache.v[0, :, :, :kv_len + n])
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    # rows with at least one visible key
    vis = torch.tensor([[kv_len + i >= key_start[b] for i in range(n)] for b in range(B)])
    err = (got - ref).abs()
    assert err[vis].max() < 3e-2, f"max err {err[vis].max()}"
    assert err[vis].mean() < 3e-3, f"mean err {err[vis].mean()}"
    assert (got[~vis] == 0).all()
    # The kernel's rounding points restated (the fp8 test does the same for its P): scores and the softmax sum are
    # fp32, each probability is rounded ONCE to the 16-bit MFMA operand type before P.V (relative error <= u = 2^-8 bf16 / 2^-11 fp16: half an ulp,
    # whatever running maximum it was taken against), the output is rounded once more.  So element by element
    #     |out - exact| <= u * (sum_j p_j |v_jd|  +  |exact_d|)            (+ fp16: N * 2^-25 max|v| for subnormal probabilities)
    # with exact = fp64 attention over the same 16-bit operands -- about two output ulps; a wrong tile edge, mask bit or split merge
    # is orders of magnitude above it, where the 3e-2 above would hide it.
    u = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    G = H // Hkv
    Kd, Vd = ref_cache.k[0].double(), ref_cache.v[0].double()
    for b in range(B):
        for i in range(n):
            lo, hi = key_start[b], kv_len + i + 1
            if hi <= lo:
                continue
            for h in range(H):
                kk, vv = Kd[b, h // G, lo:hi], Vd[b, h // G, lo:hi]
                p = torch.softmax((kk @ q[b, i, h].double()) / D ** 0.5, dim=0)
                exact = (p[:, None] * vv).sum(0)
                bound = u * 1.05 * ((p[:, None] * vv.abs()).sum(0) + exact.abs()) + 2e-6
                if dtype == torch.float16:
                    bound = bound + (hi - lo) * 2.0 ** -25 * float(vv.abs().max())
                e = (got[b, i, h].double() - exact).abs()
                assert (e <= bound).all(), f"{name} b{b} row{i} head{h}: err {float(e.max()):.3e} over the rounding bound {float(bound[e.argmax()]):.3e}"


_DIRECT_SCRIPT = r"""
import hashlib, sys, torch
sys.path.insert(0, ".")
import sjd_amd.ops as ops
from tests.test_gpu_kernels import _Cache
dev = torch.device("cuda:0")
h = hashlib.sha256()
for fp8 in (False, True):
    for (B, H, n, kv_len, ks, n_valid) in ((8, 32, 16, 700, [0, 3, 0, 650, 0, 0, 9, 0], 16), (2, 4, 16, 37, [0, 30], 5), (1, 2, 7, 0, [0], 7)):
        g = torch.Generator().manual_seed(B * 1000 + n)
        S = 1024
        kc, vc = torch.randn(1, B, H, S, 128, generator=g).bfloat16().to(dev), torch.randn(1, B, H, S, 128, generator=g).bfloat16().to(dev)
        q = (torch.randn(B, n, H, 128, generator=g) * 1.5).bfloat16().to(dev)
        k, v = torch.randn(B, n, H, 128, generator=g).bfloat16().to(dev), torch.randn(B, n, H, 128, generator=g).bfloat16().to(dev)
        attn = ops.HipWindowAttention(n_split=1)
        if fp8:
            kc, vc = kc.float().to(ops.FP8), vc.float().to(ops.FP8)
        out = attn(0, q, k, v, _Cache(kc, vc), kv_len, ks)
        torch.cuda.synchronize()
        h.update(out.cpu().view(torch.int16).numpy().tobytes())
print(h.hexdigest())
"""


def test_k1_single_split_direct_output_is_what_combine_would_write(dev):
    """n_split == 1: k1_partial normalises and writes the 16-bit output itself (no workspace round trip, no k1_combine launch); the bytes
    are those of partial + combine (SJD_K1_NO_DIRECT=1, read once per process -> two subprocesses), 16-bit and fp8 caches."""
    import subprocess
    import sys
    outs = []
    for env_extra in ({}, {"SJD_K1_NO_DIRECT": "1"}):
        env = dict(os.environ, **env_extra)
        env.pop("SJD_K1_NO_DIRECT", None) if not env_extra else None
        r = subprocess.run([sys.executable, "-c", _DIRECT_SCRIPT], capture_output=True, text=True, env=env,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert len(outs[0]) == 64 and outs[0] == outs[1]


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("B,H,Hkv,D,n,kv_len,ks,n_split,dtype", [
    (2, 32, 32, 128, 16, 1216, [0, 59], 4, torch.bfloat16),       # Lumina-7B mid-image: the production launch
    (2, 32, 32, 128, 16, 2368, [0, 63], 4, torch.bfloat16),
    (2, 32, 32, 128, 16, 40, [0, 39], 4, torch.bfloat16),         # one effective split: the direct output, no exchange
    (2, 32, 32, 128, 16, 200, [0, 63], 4, torch.bfloat16),        # two effective splits of four
    (2, 4, 4, 128, 5, 700, [0, 10], 8, torch.float16),            # ragged window, eight splits
    (2, 4, 2, 128, 16, 600, [3, 0], 4, torch.bfloat16),           # two q heads per kv head in one workgroup
    (2, 12, 12, 64, 16, 250, [0, 0], 2, torch.bfloat16),          # LlamaGen head size
    (1, 4, 4, 128, 60, 500, [7], 4, torch.bfloat16)])             # prefill-like: four row chunks
def test_k1_splits_merged_in_the_kernel_equal_partial_plus_combine(dev, fp8, B, H, Hkv, D, n, kv_len, ks, n_split, dtype):
    """K1 in one launch (round 3): the key splits are merged by the last of their workgroups to finish -- device-coherent exchange, then
    k1_combine's arithmetic in split order.  The output BYTES are those of k1_partial + k1_combine, for 30 launches in a row (the
    tickets re-arm themselves, no stale partial is ever merged) and when replayed from a hipGraph."""
    ops, L = _ops()
    if fp8 and D != 128:
        pytest.skip("fp8 K1 cases use head size 128")
    g = torch.Generator().manual_seed(kv_len + n + H)
    S = ((kv_len + n + 63) // 32) * 32
    ksd = torch.tensor(ks, dtype=torch.int32, device=dev)
    kcs = [torch.randn(B, Hkv, S, D, generator=g).to(dtype).to(dev) for _ in range(3)]
    vcs = [torch.randn(B, Hkv, S, D, generator=g).to(dtype).to(dev) for _ in range(3)]
    if fp8:
        kcs, vcs = [t.float().to(ops.FP8) for t in kcs], [t.float().to(ops.FP8) for t in vcs]
    qs = [(torch.randn(B, n, H, D, generator=g) * 1.5).to(dtype).to(dev) for _ in range(3)]
    ws = ops.attention_workspace(B, H, n, D, n_split, dev)

    def run(i, out, merged):
        if fp8:
            ops.draft_window_attention_fp8(qs[i % 3], kcs[i % 3], vcs[i % 3], out, 1.0, 1.0, ksd, None, kv_len, n_split, ws, merged=merged)
        else:
            ops.draft_window_attention(qs[i % 3], kcs[i % 3], vcs[i % 3], out, ksd, None, kv_len, n_split, ws, merged=merged)

    ref, got = [torch.empty_like(qs[0]) for _ in range(3)], [torch.full_like(qs[0], 7.0) for _ in range(30)]
    for i in range(3):
        run(i, ref[i], False)
    for i in range(30):
        run(i, got[i], True)
    torch.cuda.synchronize()
    for i in range(30):
        assert torch.equal(got[i].view(torch.int16), ref[i % 3].view(torch.int16)), (i, (got[i].float() - ref[i % 3].float()).abs().max())
    outs = [torch.full_like(qs[0], 7.0) for _ in range(3)]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        run(0, outs[0], True)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(3):
            run(i, outs[i], True)
    for _ in range(10):
        graph.replay()
    torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(outs[i].view(torch.int16), ref[i].view(torch.int16))
    assert int(ops.k1_tickets(B, Hkv, n, dev).abs().sum()) == 0          # every ticket re-armed


def test_k1_device_side_kv_len(dev):
    """kv_len / n_rows read from the device-resident sjd_iter_params blob (shape-static launch)."""
    ops, L = _ops()
    B, H, D, S_max, n_max = 2, 4, 128, 512, 16
    g = torch.Generator().manual_seed(11)
    kc = torch.randn(1, B, H, S_max, D, generator=g).to(torch.bfloat16)
    vc = torch.randn(1, B, H, S_max, D, generator=g).to(torch.bfloat16)
    attn = ops.HipWindowAttention(n_split=4)
    attn.params = ops.DeviceBlob(L.IterParams, dev)
    dcache = _Cache(kc.clone().to(dev), vc.clone().to(dev))
    for kv_len, n in [(37, 16), (200, 7), (201, 1)]:
        q = torch.randn(B, n_max, H, D, generator=g).to(torch.bfloat16)
        k = torch.randn(B, n_max, H, D, generator=g).to(torch.bfloat16)
        v = torch.randn(B, n_max, H, D, generator=g).to(torch.bfloat16)
        ref_cache = _Cache(dcache.k.cpu().clone(), dcache.v.cpu().clone())
        ref = OracleWindowAttention()(0, q[:, :n], k[:, :n], v[:, :n], ref_cache, kv_len, [0, 5]).float()
        attn.params.view.kv_len, attn.params.view.n_rows = kv_len, n
        attn.params.upload()
        out = attn(0, q.to(dev), k.to(dev), v.to(dev), dcache, -1, [0, 5])
        torch.cuda.synchronize()
        err = (out[:, :n].float().cpu() - ref).abs()
        assert err.max() < 3e-2 and err.mean() < 3e-3


def test_k1_ignores_nan_beyond_the_valid_cache_length(dev):
    """Cache rows >= kv_len + n (padding rows of a shape-static window, stale data) may hold anything, incl. NaN."""
    ops, L = _ops()
    B, H, D, S_max, kv_len, n = 2, 4, 128, 256, 70, 9
    g = torch.Generator().manual_seed(2)
    kc = torch.randn(1, B, H, S_max, D, generator=g).to(torch.bfloat16)
    vc = torch.randn(1, B, H, S_max, D, generator=g).to(torch.bfloat16)
    q = torch.randn(B, 16, H, D, generator=g).to(torch.bfloat16)
    k = torch.randn(B, 16, H, D, generator=g).to(torch.bfloat16)
    v = torch.randn(B, 16, H, D, generator=g).to(torch.bfloat16)
    k[:, n:], v[:, n:] = float("nan"), float("nan")           # what the padding rows of the window may append
    kc[..., kv_len + 16:, :], vc[..., kv_len + 16:, :] = float("nan"), float("inf")
    ref_cache = _Cache(kc.clone(), vc.clone())
    ref = OracleWindowAttention()(0, q[:, :n], k[:, :n], v[:, :n], ref_cache, kv_len, [0, 3]).float()
    attn = ops.HipWindowAttention(n_split=4)
    attn.params = ops.DeviceBlob(L.IterParams, dev)
    attn.params.view.kv_len, attn.params.view.n_rows = kv_len, n
    attn.params.upload()
    dcache = _Cache(kc.clone().to(dev), vc.clone().to(dev))
    out = attn(0, q.to(dev), k.to(dev), v.to(dev), dcache, -1, [0, 3]).float().cpu()
    assert torch.isfinite(out).all()
    assert (out[:, n:] == 0).all()
    assert (out[:, :n] - ref).abs().max() < 3e-2


@pytest.mark.parametrize("H,Hkv,n,kv_len,n_split,ks", [
    (8, 2, 32, 4101, 16, [0, 4089]),     # Emu3's window late in an image: 130 tiles over 16 splits = 9 each, the sixteenth gets none
    (8, 2, 1, 4100, 32, [0, 4088]),      # its one-row iteration: 5 tiles each, splits 26..31 none
    (4, 4, 16, 1217, 8, [0, 59]),        # multi-head window (k1_partial), the last split empty
    (4, 4, 48, 0, 4, [0, 40]),           # a prompt whose first chunks lie wholly in front of key_start: no visible key at all -> zeros
    (8, 2, 48, 0, 4, [0, 40]),           # the same through the shared-tile kernels
])
def test_k1_empty_splits_on_a_poisoned_workspace(dev, H, Hkv, n, kv_len, n_split, ks):
    """A split below the effective count that holds no tile (tiles-per-split rounds up) and a chunk that sees no key must still merge as
    (m = -inf, l = 0, O = 0).  Round 4: k1_partial_shared / k1_partial_ring returned from such a split without writing its partial and
    k1_combine merged whatever the workspace held -- zero pages in a fresh process, NaN after an unlucky predecessor.  The workspace is
    filled with NaN here, so the result depends on written partials only."""
    ops, L = _ops()
    B, D = 2, 128
    S_max = ((kv_len + n + 127) // 128) * 128
    g = torch.Generator().manual_seed(5)
    kc = torch.randn(1, B, Hkv, S_max, D, generator=g).to(torch.float16)
    vc = torch.randn(1, B, Hkv, S_max, D, generator=g).to(torch.float16)
    q = torch.randn(B, n, H, D, generator=g).to(torch.float16)
    k = torch.randn(B, n, Hkv, D, generator=g).to(torch.float16)
    v = torch.randn(B, n, Hkv, D, generator=g).to(torch.float16)
    ref = OracleWindowAttention()(0, q, k, v, _Cache(kc.clone(), vc.clone()), kv_len, ks).float()
    attn = ops.HipWindowAttention(n_split=n_split)
    dcache = _Cache(kc.clone().to(dev), vc.clone().to(dev))
    attn._workspace(B, H, n, D, dev).fill_(float("nan"))
    out = attn(0, q.to(dev), k.to(dev), v.to(dev), dcache, kv_len, ks).float().cpu()
    assert torch.isfinite(out).all()
    hidden = torch.tensor([[kv_len + i < ks[b] for i in range(n)] for b in range(B)])
    assert (out[hidden] == 0).all()                       # rows with no visible key: defined (zero) output
    assert (out[~hidden] - ref[~hidden]).abs().max() < 3e-2


FP8_CASES = [
    # name, B, H, Hkv, D, S_max, kv_len, n, key_start, dtype, (k_scale, v_scale)
    ("fp8_mha_d128_mid", 2, 4, 4, 128, 1280, 1216, 16, [0, 59], torch.bfloat16, (1.0, 1.0)),
    ("fp8_mha_d128_empty_cache", 2, 4, 4, 128, 128, 0, 16, [0, 0], torch.bfloat16, (1.0, 1.0)),
    ("fp8_mha_d128_unaligned_scaled", 2, 2, 2, 128, 256, 77, 16, [0, 33], torch.bfloat16, (0.5, 2.0)),
    ("fp8_short_window", 2, 2, 2, 128, 256, 100, 5, [0, 10], torch.float16, (1.0, 0.25)),
    ("fp8_gqa4_d128", 2, 8, 2, 128, 1056, 1000, 16, [3, 0], torch.float16, (1.0, 1.0)),
    ("fp8_gqa2_window32", 1, 4, 2, 128, 512, 300, 32, [0], torch.bfloat16, (1.0, 1.0)),
    ("fp8_mha_d64", 2, 12, 12, 64, 288, 200, 16, [0, 0], torch.bfloat16, (1.0, 1.0)),
]


@pytest.mark.parametrize("case", FP8_CASES, ids=[c[0] for c in FP8_CASES])
@pytest.mark.parametrize("n_split", [1, 8, "colsplit"])
def test_k1_k3_fp8_kv_cache(dev, case, n_split):
    """BASELINE config 5: K3 quantises the new rows to OCP e4m3 exactly like torch's cast; K1 over the fp8 cache (fp8 MFMA for both
    contractions, q and P rounded to fp8 in the kernel) stays within fp8 tolerance of exact attention over the SAME dequantised
    cache -- and therefore of the bf16 result (the reference has no fp8; that is the stated parity target)."""
    ops, L = _ops()
    name, B, H, Hkv, D, S_max, kv_len, n, key_start, dtype, (sk, sv) = case
    g = torch.Generator().manual_seed(len(name))
    k_all = torch.randn(1, B, Hkv, S_max, D, generator=g) * sk * 1.2
    v_all = torch.randn(1, B, Hkv, S_max, D, generator=g) * sv * 1.2
    q = (torch.randn(B, n, H, D, generator=g) * 1.5).to(dtype)
    k = (torch.randn(B, n, Hkv, D, generator=g) * sk * 1.2).to(dtype)
    v = (torch.randn(B, n, Hkv, D, generator=g) * sv * 1.2).to(dtype)
    kc8 = (k_all / sk).to(ops.FP8)
    vc8 = (v_all / sv).to(ops.FP8)
    dcache = _Cache(kc8.clone().to(dev), vc8.clone().to(dev))
    regime = "keysplit"
    if n_split == "colsplit":        # round 4: k1_dsplit_fp8 (column split, no key splits)
        if not ops.colsplit_ok(B, n, H, Hkv, D, ops.FP8):
            pytest.skip("the column split serves multi-head 16-row windows of head size 128")
        n_split, regime = 4, "colsplit"
    attn = ops.HipWindowAttention(n_split=n_split)
    attn.regime = regime
    attn.kv_scale = (sk, sv)
    out = attn(0, q.to(dev), k.to(dev), v.to(dev), dcache, kv_len, key_start)
    torch.cuda.synchronize()
    # K3: appended rows are torch's round-to-nearest-even e4m3 cast of x / scale, bit for bit; older rows untouched
    want_k = (k.float() / sk).to(ops.FP8).permute(0, 2, 1, 3).contiguous().view(torch.uint8)
    want_v = (v.float() / sv).to(ops.FP8).permute(0, 2, 1, 3).contiguous().view(torch.uint8)
    got_k, got_v = dcache.k.cpu().view(torch.uint8)[0], dcache.v.cpu().view(torch.uint8)[0]
    assert torch.equal(got_k[:, :, kv_len:kv_len + n], want_k) and torch.equal(got_v[:, :, kv_len:kv_len + n], want_v)
    assert torch.equal(got_k[:, :, :kv_len], kc8.view(torch.uint8)[0, :, :, :kv_len])
    # K1: exact attention over the dequantised cache (fp64 oracle on bf16-free operands)
    ref_cache = _Cache(dcache.k.cpu().float() * sk, dcache.v.cpu().float() * sv)
    kd, vd = ref_cache.k[0].double(), ref_cache.v[0].double()
    G = H // Hkv
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    def hilo(x):                                           # the kernel's on-chip operands since round 6: hi = fp8(x), lo = fp8(16 (x - hi)); value hi + lo / 16
        x = x.float()
        hi = x.to(ops.FP8).float()
        return (hi + ((x - hi) * 16.0).to(ops.FP8).float() / 16.0).double()
    q8 = hilo(q)                                          # the kernel's q operand (two e4m3 operands, one more MFMA)
    for b in range(B):
        for i in range(n):
            lo, hi = key_start[b], kv_len + i + 1
            if hi <= lo:
                assert (got[b, i] == 0).all()
                continue
            for h in range(H):
                kk, vv = kd[b, h // G, lo:hi], vd[b, h // G, lo:hi]
                # (1) the kernel arithmetic restated (hi / lo q; P = hi / lo of 256 e, e = exp(s - max); l from the unrounded e) -- the kernel
                #     rounds e against its running per-tile max, so the two P roundings are independent; with the residual operand what is left
                #     of a rounding is 2^-8 relative: the tolerance is a FIFTH of round 5's single-operand one (0.5 / 0.1 sv conc)
                sc = (kk @ q8[b, i, h]) / D ** 0.5
                e = torch.exp(sc - sc.max())
                p8 = hilo(e * 256) / 256
                emu = ((p8[:, None] * vv).sum(0) / e.sum()).float()
                err = (got[b, i, h] - emu).abs()
                conc = float((e / e.sum()).pow(2).sum().sqrt())          # rounding noise on P scales with sqrt(sum p^2) |v|
                assert err.max() < 0.1 * sv * conc + 5e-3 and err.mean() < 0.02 * sv * conc + 1e-3, \
                    f"{name} b{b} row{i} head{h}: vs emulation max {err.max():.4f} mean {err.mean():.5f} conc {conc:.3f}"
                # (2) exact attention over the same cache (what a 16-bit K1 would return up to bf16 rounding): with hi / lo operands the kernel
                #     IS that up to 2^-8 on q and P plus the output's rounding -- a fifth of round 5's fp8 tolerance (1.0 / 0.25 sv conc)
                p = torch.softmax((kk @ q[b, i, h].double()) / D ** 0.5, dim=0)
                want = (p[:, None] * vv).sum(0).float()
                err = (got[b, i, h] - want).abs()
                assert err.max() < 0.2 * sv * conc + 1e-2 and err.mean() < 0.05 * sv * conc + 2e-3, \
                    f"{name} b{b} row{i} head{h}: vs exact max {err.max():.4f} mean {err.mean():.5f} conc {conc:.3f}"


K1F_CASES = [
    # kv_len, key_start, valid rows (None: all 16, host kv_len; int: through a device params blob), S_max
    (0, (0, 0), None, 128), (5, (0, 3), None, 128), (37, (0, 36), 9, 128), (64, (0, 63), None, 160), (100, (0, 63), None, 192),
    (448, (0, 17), 16, 512), (1216, (0, 63), None, 1280), (2368, (0, 63), 5, 2432), (31, (0, 0), 1, 128),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("qk_norm,fold", [(True, True), (False, False), (True, False)])
@pytest.mark.parametrize("kv_len,key_start,n_valid,S_max", K1F_CASES)
def test_k1f_fused_attention_matches_f2_then_k1(dev, dtype, qk_norm, fold, kv_len, key_start, n_valid, S_max):
    """K1F (F2 + K1 + combine in one launch) against the three-launch path on the same G1 partials: identical K/V cache rows (same
    arithmetic and rounding points as F2), attention output within the K1 tolerance of an fp32 softmax over the appended cache."""
    ops, L = _ops()
    B, n, H, D, hid = 2, 16, 6, 128, 512
    g = torch.Generator().manual_seed(kv_len + 7 * H)
    x = torch.randn(B * n, hid, generator=g).to(dtype).to(dev)
    w = (torch.randn(3 * H * D, hid, generator=g) / hid ** 0.5 * 2.0).to(dtype).to(dev)
    part = ops.skinny_gemm(x, ops.pack_weight(w, 128), 3 * H * D, hid, 128)
    assert part.n_chunks == 4
    rn = (ops.residual_sumsq(x.clone(), None), hid, 1e-5) if fold else None
    mk = lambda a, b_: (a + b_ * torch.randn(1, D, generator=g)).to(dtype).to(dev)
    qn = (mk(1.0, 0.2), mk(0.0, 0.1), mk(1.0, 0.2), mk(0.0, 0.1)) if qk_norm else (None,) * 4
    inv = (1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))).to(dev)
    ks = torch.tensor(key_start, dtype=torch.int32, device=dev)
    pos = (kv_len + torch.arange(n)[None] - torch.tensor(key_start)[:, None]).reshape(-1).to(dev)
    base_k = torch.randn(1, B, H, S_max, D, generator=g).to(dtype).to(dev)
    base_v = torch.randn(1, B, H, S_max, D, generator=g).to(dtype).to(dev)
    params = None
    if n_valid is not None:
        params = ops.DeviceBlob(L.IterParams, dev)
        params.view.n_rows, params.view.kv_len = n_valid, kv_len
        params.upload()
    nv = n if n_valid is None else n_valid
    c1, c2 = _Cache(base_k.clone(), base_v.clone()), _Cache(base_k.clone(), base_v.clone())
    q = ops.qknorm_rope_append(part, c1.k[0], c1.v[0], *qn, inv, pos, B, n, H, H, D, params, 0 if params else kv_len, dtype=dtype, row_norm=rn)
    attn = ops.HipWindowAttention(n_split=4)
    attn.params = params
    ref3 = attn.attend(0, q, c1, kv_len, ks)
    out = ops.qkv_attention_fused(part, c2.k[0], c2.v[0], *qn, inv, pos, B, n, H, D, params, 0 if params else kv_len, ks, row_norm=rn, dtype=dtype)
    # K1Fs: the same with the key tiles split over four workgroups per (batch, head) + the split combine (identical cache rows, the
    # output of the one-workgroup form up to the order in which the key parts are merged)
    c3 = _Cache(base_k.clone(), base_v.clone())
    ws = ops.attention_workspace(B, H, n, D, 4, dev)
    out_s = ops.qkv_attention_fused(part, c3.k[0], c3.v[0], *qn, inv, pos, B, n, H, D, params, 0 if params else kv_len, ks, row_norm=rn, dtype=dtype,
                                    n_split=4, workspace=ws)
    torch.cuda.synchronize()
    assert torch.equal(c1.k, c3.k) and torch.equal(c1.v, c3.v)
    assert torch.isfinite(out_s.float()).all()
    ds = (out_s.float() - out.float()).abs()
    assert ds.max() < 4e-2 and ds.mean() < 2e-3, (float(ds.max()), float(ds.mean()))
    assert torch.equal(c1.k, c2.k) and torch.equal(c1.v, c2.v)                      # appended rows bit-identical, nothing else touched
    assert not torch.equal(c2.k[0, :, :, kv_len:kv_len + n], base_k[0, :, :, kv_len:kv_len + n])
    # fp32 softmax over the appended cache with the 16-bit q of F2
    K, V = c1.k[0].float(), c1.v[0].float()
    S = torch.einsum("bihd,bhjd->bhij", q.float(), K) / D ** 0.5
    j = torch.arange(S_max, device=dev)[None, None, None, :]
    i = torch.arange(n, device=dev)[None, None, :, None]
    vis = (j >= ks.view(B, 1, 1, 1)) & (j <= kv_len + i) & (j < kv_len + nv)
    P = torch.softmax(S.masked_fill(~vis, float("-inf")), dim=-1)
    ref = torch.einsum("bhij,bhjd->bihd", torch.nan_to_num(P), V)
    got = out.float()
    assert torch.isfinite(got).all()
    live = torch.zeros(B, n, dtype=torch.bool, device=dev)
    live[:, :nv] = True
    for b in range(B):
        live[b] &= (kv_len + torch.arange(n, device=dev)) >= int(ks[b])
    err = (got - ref).abs()[live]
    assert err.max() < 3e-2 and err.mean() < 3e-3, (float(err.max()), float(err.mean()))
    assert (got[:, nv:] == 0).all()                                                 # padding rows of a shape-static window
    d3 = (got - ref3.float()).abs()[live]
    assert d3.max() < 4e-2 and d3.mean() < 2e-3, (float(d3.max()), float(d3.mean()))


def test_backbone_window_forward_fused_attention_equals_unfused(dev):
    """the G1 window forward with K1F against the same forward with F2 + K1 + combine (model.k1_fused = False): same cache, close logits"""
    ops, L = _ops()
    from tests.helpers import make_chameleon
    conf = dict(vocab_size=9216, hidden_size=1024, intermediate_size=1024, num_hidden_layers=3, num_attention_heads=8,
                num_key_value_heads=8, max_position_embeddings=512, rms_norm_eps=1e-5, rope_theta=10000.0)
    outs, caches = [], []
    for fused in (False, True):
        m = make_chameleon(conf, 23, 0.5, ops.HipWindowAttention(n_split=2), dtype=torch.bfloat16, device=dev)
        m.G1_CFG = dict(qkv=(256, 8, True), o=(256, 8, False), gate_up=(512, 8, True), down=(256, 8, False))
        m.enable_fused(ops, gemm="sjd")
        m.k1_fused = fused
        m.setup_cache(batch=2, s_max=128)
        toks = torch.randint(4, 9000, (2, 40), generator=torch.Generator().manual_seed(1)).to(dev)
        ks = torch.tensor([0, 7], dtype=torch.int32, device=dev)
        m.forward_window(toks, torch.arange(40)[None].repeat(2, 1).to(dev), 0, ks)
        toks2 = torch.randint(4, 9000, (2, 16), generator=torch.Generator().manual_seed(2)).to(dev)
        outs.append(m.forward_window(toks2, (40 + torch.arange(16))[None].repeat(2, 1).to(dev), 40, ks))
        caches.append((m.cache.k.clone(), m.cache.v.clone()))
    assert torch.equal(caches[0][0][0], caches[1][0][0]) and torch.equal(caches[0][1][0], caches[1][1][0])     # layer 0: identical inputs
    assert (outs[0] - outs[1]).abs().mean() < 0.03 and (outs[0] - outs[1]).abs().max() < 0.5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N_packed,col0,n_cols,K,KC", [(9216, 0, 8224, 512, 128), (12320, 3008, 8224, 1024, 512), (65536, 0, 8224, 4096, 512),
                                                      (2048, 1024, 1024, 256, 256)])
def test_g1_column_window_of_a_packed_weight(dev, dtype, N_packed, col0, n_cols, K, KC):
    """sjd_skinny_gemm_cols: the launch over a tile window of ONE packed weight equals the same columns of the full launch, bit for bit"""
    ops, L = _ops()
    g = torch.Generator().manual_seed(N_packed + col0)
    x = torch.randn(32, K, generator=g).to(dtype).to(dev)
    w = (torch.randn(N_packed, K, generator=g) / K ** 0.5).to(dtype).to(dev)
    for sm in (True, False):
        wp = ops.pack_weight(w, KC, sm)
        full = ops.skinny_gemm(x, wp, N_packed, K, KC, waves=8, step_major=sm)
        win = ops.skinny_gemm_cols(x, wp, N_packed, K, KC, col0, n_cols, waves=8, step_major=sm)
        assert win.n_chunks == full.n_chunks and win.data.shape == (full.n_chunks, 32, n_cols)
        assert torch.equal(win.data, full.data[:, :, col0:col0 + n_cols])


@pytest.mark.parametrize("dtype,fold", [(torch.bfloat16, True), (torch.float16, True), (torch.bfloat16, False)])
@pytest.mark.parametrize("name,V,n,builder,cols", [
    ("lumina_image_window", 65536, 16, lambda O: O.lumina_rules([9000] * 61 + [8197, 8828, 8828] + [100] * 40, 16, 2000, 10), (0, 8224)),
    ("emu3_visual_window_odd_vocab", 184622, 8, lambda O: O.emu3_rules([1000] * 20 + [151851] + [151860] * 7, 8, 90, 90, 151854, 32768, top_k=2048,
                                                                         img_token=151851, eoi_token=151853, eos_token=151850, eol_token=151846,
                                                                         eof_token=151847, pad_token=151643), (151840, 184640)),
    ("text_rows_full_vocab", 9216, 4, lambda O: O.lumina_rules([9000] * 12, 4, 2000, 10), (0, 9216)),
])
def test_k2_on_the_unmaterialised_output_head(dev, dtype, fold, name, V, n, builder, cols):
    """K2 reading the lm_head split-K partials (chunk sum, folded-norm row scale, 16-bit rounding inside the kernel) gives
    bit-identical probabilities and tokens to the dense-logits K2 fed with the logits it derived, and those logits equal the
    torch restatement dtype(r * sum_c part_c) (within one 16-bit rounding where rsqrt differs in the last bit)."""
    ops, L = _ops()
    from oracle import sjd_oracle as O
    rules = builder(O)
    g = torch.Generator().manual_seed(V + n)
    Lmax, n_chunks, hidden = 16, 4, 2048
    n_cols = cols[1] - cols[0]
    part = ops.Partials((torch.randn(n_chunks, 32, n_cols, generator=g) * 1.5).to(dev), n_chunks, n_cols)
    sumsq = (hidden / 4 * (0.5 + torch.rand(4, 32, generator=g))).to(dev)
    head = ops.HeadOut(part, cols[0], Lmax, dtype, row_norm=(sumsq, hidden, 1e-5) if fold else None)
    params = ops.DeviceBlob(L.IterParams, dev)
    params.view.n_rows, params.view.use_cfg = n, 1
    for j, r in enumerate(rules):
        params.view.rules[j] = to_dev_rule(L, ops, r)
    params.upload()
    noise = torch.empty(Lmax, V).exponential_(generator=g).to(dev)
    probs_a, probs_b = torch.zeros(Lmax, V, device=dev), torch.zeros(Lmax, V, device=dev)
    toks_a, toks_b = torch.zeros(Lmax, dtype=torch.int64, device=dev), torch.zeros(Lmax, dtype=torch.int64, device=dev)
    dbg = torch.zeros(2, Lmax, V, device=dev)
    ops.logits_to_probs_sample_part(head, 3.0, params, noise, probs_a, ctypes_ptr(toks_a), dbg=dbg)
    # the same K2 on materialised logits = what the partial path derived
    ops.logits_to_probs_sample(dbg[0], dbg[1], 3.0, params, noise, probs_b, ctypes_ptr(toks_b))
    torch.cuda.synchronize()
    assert torch.equal(toks_a[:n], toks_b[:n])
    assert torch.equal(probs_a[:n].view(torch.int32), probs_b[:n].view(torch.int32))
    # K2a (round 4: wide windows are combined on the whole chip first, sjd_head_combine) against K2 reading the planes itself: the same bits
    # either way -- scores (the observers' copy), probabilities, tokens.  The Emu3 case takes the K2a route by default, the others are forced.
    was = ops._HEAD_COMBINE_MIN_COLS
    try:
        took_k2a = n_cols >= was
        ops._HEAD_COMBINE_MIN_COLS = (1 << 30) if took_k2a else 4
        probs_c, toks_c, dbg_c = torch.zeros(Lmax, V, device=dev), torch.zeros(Lmax, dtype=torch.int64, device=dev), torch.zeros(2, Lmax, V, device=dev)
        ops.logits_to_probs_sample_part(head, 3.0, params, noise, probs_c, ctypes_ptr(toks_c), dbg=dbg_c)
        torch.cuda.synchronize()
    finally:
        ops._HEAD_COMBINE_MIN_COLS = was
    assert name != "emu3_visual_window_odd_vocab" or took_k2a
    assert torch.equal(toks_a[:n], toks_c[:n]) and torch.equal(probs_a[:n].view(torch.int32), probs_c[:n].view(torch.int32))
    for row in range(n):
        r = rules[row]
        if r.forced < 0:
            lo = min(r.lo[i] for i in range(r.n_ranges)) if r.n_ranges else 0
            hi = max(r.hi[i] for i in range(r.n_ranges)) if r.n_ranges else V
            assert torch.equal(dbg[:, row, lo:hi].view(torch.int32), dbg_c[:, row, lo:hi].view(torch.int32))
    # torch restatement of the derived logits
    s = part.data[0].clone()
    for c in range(1, n_chunks):
        s = s + part.data[c]
    if fold:
        s = s * torch.rsqrt(sumsq.sum(0) / hidden + 1e-5)[:, None]
    ref = s.to(dtype).float()
    for b in range(2):
        for row in range(n):
            r = rules[row]
            if r.forced >= 0:
                continue
            lo = min(r.lo[i] for i in range(r.n_ranges)) if r.n_ranges else 0
            hi = max(r.hi[i] for i in range(r.n_ranges)) if r.n_ranges else V
            got = dbg[b, row, lo:hi]
            want = ref[b * Lmax + row, lo - cols[0]:hi - cols[0]]
            ulp = want.abs() * (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10) + 1e-6
            assert ((got - want).abs() <= ulp).all(), (name, b, row)
            assert (got == want).float().mean() > 0.99


def ctypes_ptr(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


def test_k2_zero_state_skips_only_what_is_zero(dev):
    """sjd_head_partials::zero_state (round 4): K2 does not rewrite the zeros outside a row's window when the window it recorded for the row
    lies inside the current one.  A probs buffer driven through a SEQUENCE of rule sets -- image rows, rows that become forced EOL rows
    and back, a window that shrinks, one that moves, text rows over the whole vocabulary -- must hold, after every launch, exactly what a
    poisoned-then-fresh buffer without a state holds."""
    ops, L = _ops()
    V, Lmax, n, n_chunks = 9216, 16, 12, 3
    g = torch.Generator().manual_seed(5)
    img = lambda: O.rule(((4, 8196),), -1, 2000, None)
    seqs = [
        [img() for _ in range(n)],                                                         # image rows
        [img() if i % 5 else O.rule((), 8803, 0, None) for i in range(n)],                 # some rows forced (EOL)
        [img() for _ in range(n)],                                                         # ... and image rows again
        [O.rule(((1000, 3000),), -1, 50, None) for _ in range(n)],                         # a narrower window inside the old one
        [O.rule(((2000, 9000),), -1, 0, None) for _ in range(n)],                          # a window that sticks out of the recorded one
        [O.rule((), -1, 10, None) for _ in range(n)],                                      # text rows: the whole vocabulary
        [O.rule(((4, 8196),), -1, 2000, None) if i < 7 else O.rule((), 8196, 0, None) for i in range(n)],
    ]
    cols = (0, V)
    probs_s = torch.full((Lmax, V), -7.0, device=dev)
    zst = torch.full((Lmax, 2), -1, dtype=torch.int32, device=dev)
    toks_s, toks_f = torch.zeros(Lmax, dtype=torch.int64, device=dev), torch.zeros(Lmax, dtype=torch.int64, device=dev)
    for it, rules in enumerate(seqs):
        part = ops.Partials((torch.randn(n_chunks, 32, V, generator=g) * 1.5).to(dev), n_chunks, V)
        head = ops.HeadOut(part, cols[0], Lmax, torch.bfloat16)
        params = ops.DeviceBlob(L.IterParams, dev)
        params.view.n_rows, params.view.use_cfg = n, 1
        for j, r in enumerate(rules):
            params.view.rules[j] = to_dev_rule(L, ops, r)
        params.upload()
        noise = torch.empty(Lmax, V).exponential_(generator=g).to(dev)
        probs_f = torch.full((Lmax, V), float("nan"), device=dev)                          # fresh, poisoned, no state: everything is rewritten
        ops.logits_to_probs_sample_part(head, 3.0, params, noise, probs_f, ctypes_ptr(toks_f))
        ops.logits_to_probs_sample_part(head, 3.0, params, noise, probs_s, ctypes_ptr(toks_s), zero_state=zst)
        torch.cuda.synchronize()
        assert torch.equal(toks_s[:n], toks_f[:n]), it
        assert torch.equal(probs_s[:n].view(torch.int32), probs_f[:n].view(torch.int32)), (it, int((probs_s[:n] != probs_f[:n]).sum()))
        st = zst.cpu()
        for i, r in enumerate(rules):
            want = (r.forced, r.forced + 1) if r.forced >= 0 else ((min(r.lo[a] for a in range(r.n_ranges)), max(r.hi[a] for a in range(r.n_ranges))) if r.n_ranges else (0, V))
            assert tuple(st[i].tolist()) == want, (it, i, st[i].tolist(), want)
    assert (zst[n:] == -1).all() and (probs_s[n:] == -7.0).all()                          # rows beyond n_rows: untouched


@pytest.mark.parametrize("n_slots,n_batch,fold", [(3, 2, True), (8, 2, False), (2, 1, True)])
def test_slot_launches_equal_one_launch_per_slot(dev, n_slots, n_batch, fold):
    """round 6, at the C-ABI: sjd_reguess_slots / sjd_logits_to_probs_sample_part_slots / sjd_verify_accept_slots over P slots with DIFFERENT row counts, rules,
    KV lengths, generator seeds and states against the one-slot entry points called once per slot on a second, identical set of buffers: window ids and
    position ids, probabilities (bit patterns), tokens, argmax rows, zero states and the whole state blobs after K4 -- device copy and pinned host mirror --
    must be the same; two iterations, so that the second one verifies against the first one's rows."""
    import copy
    ops, L = _ops()
    from oracle import sjd_oracle as O
    V, Lmax, n_chunks, hidden = 9216, 16, 3, 1024
    g = torch.Generator().manual_seed(100 * n_slots + n_batch)
    prows = ops._prows(n_slots * n_batch * Lmax)
    n_cols = 8224
    part = ops.Partials((torch.randn(n_chunks, prows, n_cols, generator=g) * 1.5).to(dev), n_chunks, n_cols)
    sumsq = (hidden / 4 * (0.5 + torch.rand(4, prows, generator=g))).to(dev)
    head = ops.HeadOut(part, 0, Lmax if n_batch > 1 else 0, torch.bfloat16, row_norm=(sumsq, hidden, 1e-5) if fold else None)
    ctx = [9000] * 9 + [8197, 8808, 8808]
    sets = []
    for which in range(2):          # 0: one launch per slot, 1: slot launches
        params, state = ops.BlobArray(L.IterParams, n_slots, dev), ops.BlobArray(L.State, n_slots, dev)
        probs = torch.zeros(n_slots, 2, Lmax, V, device=dev)
        zst = torch.full((n_slots, 2, Lmax, 2), -1, dtype=torch.int32, device=dev)
        scratch = torch.empty(n_slots, V, device=dev)
        ids = torch.zeros(n_slots * n_batch, Lmax, dtype=torch.int64, device=dev)
        pos = torch.zeros(n_slots * n_batch, Lmax, dtype=torch.int64, device=dev)
        poff = (-torch.arange(n_slots * n_batch, dtype=torch.int64) * 3).to(dev)
        state.mirror_array()
        sets.append(dict(params=params, state=state, probs=probs, zst=zst, scratch=scratch, ids=ids, pos=pos, poff=poff))
    blocks = ops.philox_max_blocks(torch.device(dev))
    for it in range(2):
        cur = it & 1
        for s_ in sets:
            for i in range(n_slots):
                n = [16, 9, 1, 12, 16, 5, 2, 13][i % 8] if it else [1, 1, 1, 1, 1, 1, 1, 1][i % 8]
                p = s_["params"].blobs[i].view
                rules = O.lumina_rules(ctx + [100 + i] * (3 * i + it), n, 2000, 10)
                p.n_rows, p.kv_len, p.use_cfg, p.scheme, p.batch_rows, p.iter_seq = n, 12 + 5 * i + it, int(n_batch > 1), 0, n_batch, 7 + it
                a = max(0, min(n - 1, (s_["state"].blobs[i].view.n_prev - s_["state"].blobs[i].view.m) if it else 0))
                p.n_fresh = n - 1 - a
                for j in range(p.n_fresh):
                    p.fresh_tok[j] = 4 + (37 * i + 11 * j) % 8192
                for j, r in enumerate(rules):
                    p.rules[j] = to_dev_rule(L, ops, r)
                    p.resid_rules[j] = to_dev_rule(L, ops, r)
                step = ops.philox_step(n * V, blocks)
                p.philox_blocks, p.philox_seed = blocks, 1000 + i
                p.philox_offset[0], p.philox_offset[1], p.philox_offset[2] = 40 * it, 40 * it + step, 40 * it + 2 * step
                if it == 0:
                    st = s_["state"].blobs[i].view
                    st.m, st.rejected, st.n_prev = 1, 0, 1
                    st.tokens[0] = ctx[-1]
            s_["params"].upload()
            if it == 0:
                s_["state"].upload()
        a_, b_ = sets
        # K5
        for i in range(n_slots):
            lo, hi = i * n_batch, (i + 1) * n_batch
            ops.reguess(a_["params"].blobs[i], a_["state"].blobs[i], a_["ids"][lo:hi], pos_offset=a_["poff"][lo:hi], positions_out=a_["pos"][lo:hi])
        sl = ops.slots_of(b_["params"], b_["state"], b_["probs"], b_["zst"], b_["scratch"], n_batch)
        ops.reguess_slots(sl, b_["params"], b_["state"], b_["ids"], b_["poff"], b_["pos"], n_batch)
        torch.cuda.synchronize()
        assert torch.equal(a_["ids"], b_["ids"]) and torch.equal(a_["pos"], b_["pos"]), f"K5, iteration {it}"
        # K2
        for i in range(n_slots):
            s0 = a_["state"].blobs[i]
            ops.logits_to_probs_sample_part(head, 3.0, a_["params"].blobs[i], None, a_["probs"][i, cur], s0.field_ptr("tokens"), amax_out_ptr=s0.field_ptr("amax"),
                                            row0=i * n_batch * Lmax, urow_off=Lmax if n_batch > 1 else 0, zero_state=a_["zst"][i, cur])
        ops.logits_to_probs_sample_part_slots(sl, head, 3.0, b_["params"], b_["probs"], cur, "tokens", "amax", b_["state"], b_["zst"], n_batch)
        torch.cuda.synchronize()
        assert torch.equal(a_["probs"].view(torch.int32), b_["probs"].view(torch.int32)) and torch.equal(a_["zst"], b_["zst"]), f"K2, iteration {it}"
        assert float(b_["probs"][:, cur].sum()) > n_slots - 0.5
        # K4
        for i in range(n_slots):
            ops.verify_accept(a_["params"].blobs[i], a_["state"].blobs[i], a_["probs"][i, cur], a_["probs"][i, 1 - cur], None, None, a_["scratch"][i], mirror=True)
        ops.verify_accept_slots(sl, b_["params"], b_["state"], b_["probs"], cur, b_["scratch"])
        a_["state"].wait_mirror()
        b_["state"].wait_mirror()
        mir_a = bytes(a_["state"].host.numpy().tobytes())
        mir_b = bytes(b_["state"].host.numpy().tobytes())
        assert mir_a == mir_b, f"K4 (host mirrors), iteration {it}"
        assert torch.equal(a_["state"].dev, b_["state"].dev), f"K4 (device states), iteration {it}"
        for i in range(n_slots):
            st = b_["state"].blobs[i].view
            assert 1 <= st.m <= max(1, b_["params"].blobs[i].view.n_rows)
