"""GPU: fused glue kernels F1-F3 against the plain PyTorch ops they replace (fp32/bf16 references of the same op),
and the fused backbone forward against the unfused one."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,hidden", [(32, 4096), (2, 512), (7, 8192)])
def test_f1_add_rmsnorm(dev, dtype, rows, hidden):
    import sjd_amd.ops as ops
    from sjd_amd.backbones import _CRMSNorm
    g = torch.Generator().manual_seed(rows + hidden)
    h = torch.randn(rows, hidden, generator=g).to(dtype).to(dev)
    d = torch.randn(rows, hidden, generator=g).to(dtype).to(dev)
    norm = _CRMSNorm(hidden, 1e-5).to(dev).to(dtype)
    norm.weight.data = (1 + 0.1 * torch.randn(hidden, generator=g)).to(dtype).to(dev)
    for delta in (None, d):
        h1 = h.clone()
        y = ops.add_rmsnorm(h1, delta, norm.weight, 1e-5)
        href = h if delta is None else h + delta
        yref = norm(href)
        assert torch.equal(h1, href)
        torch.testing.assert_close(y.float(), yref.float(), atol=2e-2, rtol=2e-2)
        assert (y.float() - yref.float()).abs().mean() < 2e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H,Hkv,D,qk_norm", [(32, 32, 128, True), (8, 2, 128, False), (12, 12, 64, True)])
def test_f2_qknorm_rope_append(dev, dtype, H, Hkv, D, qk_norm):
    import sjd_amd.ops as ops
    from sjd_amd.backbones import _HeadLayerNorm, _rotate_half
    B, n, S, kv_len = 2, 16, 96, 37
    g = torch.Generator().manual_seed(H + D)
    qkv = torch.randn(B * n, (H + 2 * Hkv) * D, generator=g).to(dtype).to(dev)
    kc = torch.zeros(B, Hkv, S, D, dtype=dtype, device=dev)
    vc = torch.zeros_like(kc)
    qn, kn = _HeadLayerNorm(D, H).to(dev).to(dtype), _HeadLayerNorm(D, Hkv).to(dev).to(dtype)
    for m in (qn, kn):
        m.weight.data = (1 + 0.2 * torch.randn(1, D, generator=g)).to(dtype).to(dev)
        m.bias.data = (0.1 * torch.randn(1, D, generator=g)).to(dtype).to(dev)
    inv_freq = (1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))).to(dev)
    positions = (torch.tensor([[1000], [941]]) + torch.arange(n)[None]).to(dev)
    args = (qn.weight, qn.bias, kn.weight, kn.bias) if qk_norm else (None,) * 4
    q = ops.qknorm_rope_append(qkv, kc, vc, *args, inv_freq, positions.reshape(-1).contiguous(), B, n, H, Hkv, D, None, kv_len)
    # reference: the unfused ATen sequence of ChameleonBackbone.forward_window
    x = qkv.view(B, n, H + 2 * Hkv, D)
    qr, kr, vr = x[:, :, :H], x[:, :, H:H + Hkv], x[:, :, H + Hkv:]
    if qk_norm:
        qr, kr = qn(qr), kn(kr)
    freqs = positions[:, :, None].float() * inv_freq[None, None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = emb.cos().to(dtype)[:, :, None, :], emb.sin().to(dtype)[:, :, None, :]
    qr = qr * cos + _rotate_half(qr) * sin
    kr = kr * cos + _rotate_half(kr) * sin
    tol = dict(atol=4e-2, rtol=4e-2)
    torch.testing.assert_close(q.float(), qr.float(), **tol)
    torch.testing.assert_close(kc[:, :, kv_len:kv_len + n].float(), kr.transpose(1, 2).float(), **tol)
    assert torch.equal(vc[:, :, kv_len:kv_len + n], vr.transpose(1, 2))
    assert (q.float() - qr.float()).abs().mean() < 4e-3
    assert kc[:, :, :kv_len].abs().sum() == 0 and kc[:, :, kv_len + n:].abs().sum() == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_f3_silu_mul(dev, dtype):
    import sjd_amd.ops as ops
    g = torch.Generator().manual_seed(3)
    gu = (torch.randn(32, 2 * 11008, generator=g) * 2).to(dtype).to(dev)
    y = ops.silu_mul(gu)
    ref = F.silu(gu[:, :11008]) * gu[:, 11008:]
    torch.testing.assert_close(y.float(), ref.float(), atol=2e-2, rtol=2e-2)


def test_fused_forward_matches_unfused(dev):
    import sjd_amd.ops as ops
    from tests.helpers import make_chameleon
    conf = dict(vocab_size=9216, hidden_size=512, intermediate_size=1024, num_hidden_layers=3, num_attention_heads=4,
                num_key_value_heads=2, max_position_embeddings=512, rms_norm_eps=1e-5, rope_theta=10000.0)
    outs = []
    for fused in (False, True):
        m = make_chameleon(conf, 23, 0.5, ops.HipWindowAttention(n_split=2), dtype=torch.bfloat16, device=dev)
        if fused:
            m.enable_fused(ops)
        m.setup_cache(batch=2, s_max=128)
        toks = torch.randint(4, 9000, (2, 40), generator=torch.Generator().manual_seed(1)).to(dev)
        pos = torch.arange(40)[None].repeat(2, 1).to(dev)
        ks = torch.tensor([0, 7], dtype=torch.int32, device=dev)
        l1 = m.forward_window(toks, pos, 0, ks)
        toks2 = torch.randint(4, 9000, (2, 16), generator=torch.Generator().manual_seed(2)).to(dev)
        l2 = m.forward_window(toks2, (40 + torch.arange(16))[None].repeat(2, 1).to(dev), 40, ks)
        outs.append((l1, l2))
    for a, b in zip(outs[0], outs[1]):
        # bf16 end-to-end: compare on the logit scale (std ~3)
        assert (a - b).abs().mean() < 0.05 and (a - b).abs().max() < 0.6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,KC", [(32, 12288, 4096, 1024), (32, 4096, 4096, 512), (32, 22016, 4096, 2048),
                                        (32, 4096, 11008, 1024), (7, 256, 176, 64), (32, 64, 2048, 2048),
                                        (64, 6144, 4096, 1024), (47, 4096, 14336, 1280), (64, 64, 96, 32)])
@pytest.mark.parametrize("waves,step_major", [(4, False), (8, True), (11, True), (3, False)])
def test_g1_skinny_gemm(dev, dtype, M, N, K, KC, waves, step_major):
    """G1 weight-streaming projection (split-K partials) against an fp32 matmul of the same bf16/fp16 operands."""
    import sjd_amd.ops as ops
    g = torch.Generator().manual_seed(N + K)
    x = torch.randn(M, K, generator=g).to(dtype).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).to(dev)
    part = ops.skinny_gemm(x, ops.pack_weight(w, KC, step_major), N, K, KC, waves=waves, step_major=step_major)
    assert part.n_chunks == (K + KC - 1) // KC
    got = part.data.sum(0)[:M]
    ref = x.float() @ w.float().t()
    torch.testing.assert_close(got, ref, atol=2e-3, rtol=2e-3)
    assert part.data.shape[1] == ((M + 31) // 32) * 32
    if M < part.data.shape[1]:
        assert part.data[:, M:].abs().max() == 0        # missing rows are zero, not garbage


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,KC", [(128, 4096, 11008, 896), (96, 12288, 4096, 896), (128, 22016, 4096, 2048), (100, 512, 1376, 256),
                                        (128, 64, 96, 32), (128, 4096, 4096, 1040), (65, 256, 176, 64), (128, 6144, 4096, 512), (64, 6144, 4096, 2048)])
@pytest.mark.parametrize("waves,step_major", [(8, True), (8, False), (6, False), (3, True), (4, True), (4, False)])
def test_g1_skinny_gemm_three_and_four_row_tiles(dev, dtype, M, N, K, KC, waves, step_major):
    """65..128-row windows (three / four prompts per forward), and 64-row windows whose chunk does not fit in LDS (KC > 1280): the
    sub-tiled G1 (activation double-buffered through LDS 256 columns at a time) against an fp32 matmul; chunk lengths that are not whole
    sub-tiles / whole 8-step groups included."""
    import sjd_amd.ops as ops
    g = torch.Generator().manual_seed(N + K + M)
    x = torch.randn(M, K, generator=g).to(dtype).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).to(dev)
    part = ops.skinny_gemm(x, ops.pack_weight(w, KC, step_major), N, K, KC, waves=waves, step_major=step_major)
    assert part.n_chunks == (K + KC - 1) // KC and part.data.shape[1] == ((M + 31) // 32) * 32
    torch.testing.assert_close(part.data.sum(0)[:M], x.float() @ w.float().t(), atol=2e-3, rtol=2e-3)
    if M < part.data.shape[1]:
        assert part.data[:, M:].abs().max() == 0
    # the same rows through the 32-row kernel, 32 at a time: identical partial planes (same chunking, same accumulation order)
    for r0 in range(0, M, 32):
        p32 = ops.skinny_gemm(x[r0:r0 + 32].contiguous(), ops.pack_weight(w, KC, step_major), N, K, KC, waves=waves, step_major=step_major)
        rows = min(32, M - r0)
        assert torch.equal(p32.data[:, :rows], part.data[:, r0:r0 + rows])
    with pytest.raises(RuntimeError):
        ops.skinny_gemm(x, ops.pack_weight(w, KC, step_major), N, K, KC, waves=11, step_major=step_major)      # > 8 waves: 256 VGPRs needed


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,KC", [(64, 6144, 4096, 1024), (64, 4096, 14336, 1792), (64, 28672, 4096, 2048), (33, 512, 1376, 256), (48, 64, 96, 32),
                                        (40, 256, 176, 64), (64, 4096, 4096, 512), (57, 352, 4096, 4096)])
@pytest.mark.parametrize("tiles,step_major", [(4, True), (8, True), (8, False), (2, True), (3, False), (6, True)])
def test_g1w_two_row_tiles(dev, dtype, M, N, K, KC, tiles, step_major):
    """late round 6: 33..64-row windows on the uncompressed stream (Emu3's window of 32 with CFG in fp16; two prompts per forward) run on kernel G1w with
    two row tiles, bf16 AND fp16 -- against an fp32 matmul and plane for plane against the 32-row kernel fed 32 rows at a time (bit-identical); Emu3's
    shapes, ragged chunks, column counts that leave waves without a tile, one K chunk, rows that are not whole tiles; poisoned planes."""
    import sjd_amd.ops as ops
    g = torch.Generator().manual_seed(N + K + M)
    x = torch.randn(M, K, generator=g).to(dtype).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).to(dev)
    wp = ops.pack_weight(w, KC, step_major)
    torch.full((2 * ((K + KC - 1) // KC) * 64 * N,), float("nan"), device=dev)      # (freed at once: the allocator hands the block to the planes)
    part = ops.skinny_gemm(x, wp, N, K, KC, waves=tiles, step_major=step_major)
    assert part.n_chunks == (K + KC - 1) // KC and part.data.shape[1] == 64
    torch.testing.assert_close(part.data.sum(0)[:M], x.float() @ w.float().t(), atol=2e-3, rtol=2e-3)
    if M < 64:
        assert part.data[:, M:].abs().max() == 0
    if KC <= 2560:
        for r0 in range(0, M, 32):
            p32 = ops.skinny_gemm(x[r0:r0 + 32].contiguous(), wp, N, K, KC, waves=4, step_major=step_major)
            rows = min(32, M - r0)
            assert torch.equal(p32.data[:, :rows], part.data[:, r0:r0 + rows])


@pytest.mark.parametrize("M,N,K,KC", [(256, 4096, 11008, 1376), (192, 12288, 4096, 2048), (160, 22016, 4096, 2048), (224, 4096, 4096, 896), (130, 512, 1376, 256),
                                        (256, 64, 96, 32), (255, 256, 176, 64), (256, 4096, 11008, 1408), (256, 12288, 4096, 832), (256, 352, 4096, 512),
                                        (256, 4096, 11008, 688), (200, 2048, 4096, 128), (256, 1056, 1024, 256)])      # (16 / 32 / 4 chunks: every branch of the XCD-aware block map)
@pytest.mark.parametrize("tiles,step_major", [(4, True), (4, False), (8, True), (8, False), (2, True), (3, False), (6, True)])
def test_g1_skinny_gemm_five_to_eight_row_tiles(dev, M, N, K, KC, tiles, step_major):
    """129..256-row windows (five to eight prompts per forward): kernel G1w (round 6: activation stages by LDS-DMA, hand-counted vmcnt, one or two
    column tiles per wave) -- against an fp32 matmul, and plane for plane against the 32-row kernel fed 32 rows at a time (same chunking, same
    accumulation order: bit-identical); every column-tile count the launcher serves, ragged chunks, column counts that leave waves without a tile,
    rows that are not whole tiles.  The planes are POISONED first: every element must be written."""
    import sjd_amd.ops as ops
    g = torch.Generator().manual_seed(N + K + M)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    wp = ops.pack_weight(w, KC, step_major)
    torch.full((2 * ((K + KC - 1) // KC) * 256 * N,), float("nan"), device=dev)      # (freed at once: the allocator hands the block to the planes)
    part = ops.skinny_gemm(x, wp, N, K, KC, waves=tiles, step_major=step_major)
    assert part.n_chunks == (K + KC - 1) // KC and part.data.shape[1] == ((M + 31) // 32) * 32
    torch.testing.assert_close(part.data.sum(0)[:M], x.float() @ w.float().t(), atol=2e-3, rtol=2e-3)
    if M < part.data.shape[1]:
        assert part.data[:, M:].abs().max() == 0
    if KC <= 2560:
        for r0 in range(0, M, 32):
            p32 = ops.skinny_gemm(x[r0:r0 + 32].contiguous(), wp, N, K, KC, waves=4, step_major=step_major)
            rows = min(32, M - r0)
            assert torch.equal(p32.data[:, :rows], part.data[:, r0:r0 + rows])
    with pytest.raises(RuntimeError):
        ops.skinny_gemm(x, wp, N, K, KC, waves=5, step_major=step_major)          # more than 128 rows: 2, 3, 4, 6 or 8 column tiles per workgroup
    with pytest.raises(RuntimeError):
        ops.skinny_gemm(x.to(torch.float16), wp, N, K, KC, waves=4, step_major=step_major)      # ... and bf16 only


def test_g1_wide_chunk_neighbours_do_not_leak(dev):
    """G1w fetches the activation columns past a ragged chunk's end out of range (zeros): NaN / Inf sitting right behind a chunk's last k-step -- in
    the next chunk, or in the next row behind the matrix's last column -- must not reach the planes of the chunk in front of them."""
    import sjd_amd.ops as ops
    M, N, K, KC = 200, 256, 176, 80            # chunks of 5, 5 and 1 k-steps: every chunk ends inside a stage of four
    g = torch.Generator().manual_seed(11)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    wp = ops.pack_weight(w, KC, True)
    clean = ops.skinny_gemm(x, wp, N, K, KC, waves=8, step_major=True).data.clone()
    assert torch.isfinite(clean).all()
    xa = x.clone()
    xa[:, 80:96] = float("nan")                # the first k-step of chunk 1: right behind chunk 0's end, inside chunk 0's last stage
    got = ops.skinny_gemm(xa, wp, N, K, KC, waves=8, step_major=True).data
    assert torch.equal(got[0], clean[0]) and torch.equal(got[2], clean[2]) and torch.isnan(got[1][:M]).all()
    xb = x.clone()
    xb[:, 160:] = float("inf")                 # chunk 2: behind chunk 1's end
    got = ops.skinny_gemm(xb, wp, N, K, KC, waves=8, step_major=True).data
    assert torch.equal(got[0], clean[0]) and torch.equal(got[1], clean[1])
    xc = x.clone()
    xc[:, :48] = float("nan")                  # the head of every row: what lies behind the LAST chunk's end (the next row's first columns)
    got = ops.skinny_gemm(xc, wp, N, K, KC, waves=8, step_major=True).data
    assert torch.equal(got[1], clean[1]) and torch.equal(got[2], clean[2])


def test_partials_feed_glue_kernels(dev):
    import sjd_amd.ops as ops
    g = torch.Generator().manual_seed(5)
    dt = torch.bfloat16
    x = torch.randn(32, 512, generator=g).to(dt).to(dev)
    w = (torch.randn(1024, 512, generator=g) / 512 ** 0.5).to(dt).to(dev)
    part = ops.skinny_gemm(x, ops.pack_weight(w, 128), 1024, 512, 128)
    dense = F.linear(x, w)
    # F1
    h = torch.randn(32, 1024, generator=g).to(dt).to(dev)
    nw = (1 + 0.1 * torch.randn(1024, generator=g)).to(dt).to(dev)
    h1, h2 = h.clone(), h.clone()
    y1 = ops.add_rmsnorm(h1, part, nw, 1e-5)
    y2 = ops.add_rmsnorm(h2, dense, nw, 1e-5)
    assert (h1.float() - h2.float()).abs().max() < 0.07 and (y1.float() - y2.float()).abs().mean() < 5e-3
    # F3
    a1 = ops.silu_mul(part, rows=32, dtype=dt)
    a2 = ops.silu_mul(dense)
    assert (a1.float() - a2.float()).abs().mean() < 5e-3
    # F2 (D=128: H=4, Hkv=2 -> 8 head rows of 128 = 1024 columns)
    kc1, vc1 = torch.zeros(2, 2, 64, 128, dtype=dt, device=dev), torch.zeros(2, 2, 64, 128, dtype=dt, device=dev)
    kc2, vc2 = torch.zeros_like(kc1), torch.zeros_like(vc1)
    inv = (1.0 / (10000.0 ** (torch.arange(0, 128, 2).float() / 128))).to(dev)
    pos = torch.arange(32, device=dev)
    q1 = ops.qknorm_rope_append(part, kc1, vc1, None, None, None, None, inv, pos, 2, 16, 4, 2, 128, None, 5)
    q2 = ops.qknorm_rope_append(dense, kc2, vc2, None, None, None, None, inv, pos, 2, 16, 4, 2, 128, None, 5)
    assert (q1.float() - q2.float()).abs().mean() < 5e-3 and (kc1.float() - kc2.float()).abs().mean() < 5e-3
    assert (vc1.float() - vc2.float()).abs().max() < 0.07


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("B,n,H,Hkv,n_chunks,qk_norm,folded", [(16, 16, 32, 32, 2, True, True), (8, 16, 32, 8, 4, False, True), (6, 16, 8, 4, 5, True, False),
                                                         (5, 16, 8, 8, 9, True, True), (16, 13, 4, 4, 1, False, False)])
def test_f2_many_row_kernel_is_the_one_head_kernel_bit_for_bit(dev, monkeypatch, fp8, B, n, H, Hkv, n_chunks, qk_norm, folded):
    """round 6: windows of more than 64 rows run F2 with four heads of a token per wave (f2_qknorm_rope_append_rows: the planes of four heads in
    flight together, the rotary angle computed once instead of once per head) -- q and the cache rows must be those of the one-head kernel
    (SJD_F2_ROWS=0), bit for bit: 256 rows of Lumina's 96 heads, Emu3's GQA shape, ragged chunk counts, with and without the folded RMSNorm"""
    import sjd_amd.ops as ops
    D, S, kv_len, T = 128, 64, 21, B * n
    g = torch.Generator().manual_seed(B * 100 + n_chunks)
    ncol = (H + 2 * Hkv) * D
    planes = torch.zeros(n_chunks, ops._prows(T), ncol)
    planes[:, :T] = torch.randn(n_chunks, T, ncol, generator=g)
    part = ops.Partials(planes.to(dev), n_chunks, ncol)
    w = lambda: (1 + 0.2 * torch.randn(1, D, generator=g)).to(torch.bfloat16).to(dev)
    args = (w(), w(), w(), w()) if qk_norm else (None,) * 4
    inv = (1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))).to(dev)
    pos = (torch.randint(0, 3000, (B, 1), generator=g) + torch.arange(n)[None]).reshape(-1).contiguous().to(dev)
    rn = None
    if folded:
        rn = (torch.rand(3, ops._prows(T), generator=g).add(0.5).mul(1000.0).to(dev), 4096, 1e-5)
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("SJD_F2_ROWS", flag)
        if fp8:
            kc, vc = torch.zeros(B, Hkv, S, D, dtype=torch.uint8, device=dev).view(ops.FP8), torch.zeros(B, Hkv, S, D, dtype=torch.uint8, device=dev).view(ops.FP8)
        else:
            kc, vc = torch.zeros(B, Hkv, S, D, dtype=torch.bfloat16, device=dev), torch.zeros(B, Hkv, S, D, dtype=torch.bfloat16, device=dev)
        q = ops.qknorm_rope_append(part, kc, vc, *args, inv, pos, B, n, H, Hkv, D, None, kv_len, kv_scale=(0.05, 0.03), dtype=torch.bfloat16, row_norm=rn)
        torch.cuda.synchronize()
        outs.append((q, kc.view(torch.uint8) if fp8 else kc, vc.view(torch.uint8) if fp8 else vc))
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)
    assert outs[0][0].float().abs().sum() > 0 and outs[0][1][:, :, kv_len:kv_len + n].float().abs().sum() > 0


def test_g1_forward_matches_library_gemm_forward(dev):
    import sjd_amd.ops as ops
    from tests.helpers import make_chameleon
    conf = dict(vocab_size=9216, hidden_size=512, intermediate_size=1024, num_hidden_layers=3, num_attention_heads=4,
                num_key_value_heads=2, max_position_embeddings=512, rms_norm_eps=1e-5, rope_theta=10000.0)
    outs = []
    for gemm, fold in (("torch", False), ("sjd", False), ("sjd", True)):
        m = make_chameleon(conf, 23, 0.5, ops.HipWindowAttention(n_split=2), dtype=torch.bfloat16, device=dev)
        m.G1_CFG = dict(qkv=(256, 8, True), o=(128, 4, False), gate_up=(512, 8, True), down=(256, 4, False))
        for layer in m.model.layers:          # non-trivial norm gains, so that folding them into the packed weights is exercised
            for nm in (layer.input_layernorm, layer.post_attention_layernorm):
                nm.weight.data = (1 + 0.2 * torch.randn(nm.weight.shape, generator=torch.Generator().manual_seed(9))).to(nm.weight)
        m.enable_fused(ops, gemm=gemm, fold_norm=fold)
        m.setup_cache(batch=2, s_max=128)
        toks = torch.randint(4, 9000, (2, 40), generator=torch.Generator().manual_seed(1)).to(dev)
        ks = torch.tensor([0, 7], dtype=torch.int32, device=dev)
        m.forward_window(toks, torch.arange(40)[None].repeat(2, 1).to(dev), 0, ks)
        toks2 = torch.randint(4, 9000, (2, 16), generator=torch.Generator().manual_seed(2)).to(dev)
        outs.append(m.forward_window(toks2, (40 + torch.arange(16))[None].repeat(2, 1).to(dev), 40, ks))
    for o in outs[1:]:
        assert (outs[0] - o).abs().mean() < 0.05 and (outs[0] - o).abs().max() < 0.6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("step_major", [False, True])
@pytest.mark.parametrize("M,I,K", [(32, 11008, 4096), (17, 11008, 4096), (32, 14336, 4096), (32, 1408, 512), (5, 128, 1024), (32, 2752, 2048),
                                   (64, 11008, 4096), (40, 14336, 4096), (64, 1408, 1024), (33, 2752, 2048),       # 64-row windows: four staging phases
                                   (128, 11008, 4096), (96, 11008, 4096), (65, 2752, 4096), (100, 14336, 4096)])   # 65..128 rows: four row tiles
@pytest.mark.parametrize("with_norm", [True, False])
def test_g1_gateup_silu_matches_g1_then_f3(dev, dtype, step_major, M, I, K, with_norm):
    """G1s (the gate|up projection with SiLU * up as its epilogue, one launch) is BIT-IDENTICAL to G1 (two K halves) followed by F3 on the
    two partial planes -- same MFMA sequence per (tile, K half), same summation order and rounding points -- from the same packed weight."""
    import sjd_amd.ops as ops
    g = torch.Generator().manual_seed(I + K + M)
    x = torch.randn(M, K, generator=g).to(dtype).to(dev)
    w = (torch.randn(2 * I, K, generator=g) / K ** 0.5).to(dtype).to(dev)
    wp = ops.pack_weight(w, K // 2, step_major)
    assert ops.gateup_silu_ok(M, I, K, K // 2)
    rn = (ops.residual_sumsq(x.clone(), None), K, 1e-5) if with_norm else None
    ref = ops.silu_mul(ops.skinny_gemm(x, wp, 2 * I, K, K // 2, waves=8, step_major=step_major), rows=M, dtype=dtype, row_norm=rn)
    got = ops.gateup_silu(x, wp, I, K, step_major, row_norm=rn)
    torch.cuda.synchronize()
    assert got.shape == (M, I) and torch.equal(got.view(torch.int16), ref.view(torch.int16)), (got.float() - ref.float()).abs().max()
    # and it is the MLP's first half: silu(x Wg^T) * (x Wu^T) against fp32 math on the same operands
    if not with_norm:
        gf, uf = x.float() @ w[:I].float().t(), x.float() @ w[I:].float().t()
        want = torch.nn.functional.silu(gf) * uf
        d = (got.float() - want).abs()
        assert d.mean() < 6e-3 and d.max() < 0.2, (d.mean(), d.max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,KC,waves,step_major", [
    (32, 4096, 4096, 512, 8, False),       # Lumina-7B o projection: 16 column groups x 8 chunks
    (32, 4096, 11008, 896, 8, False),      # ... down projection: 16 x 13, ragged last chunk
    (17, 4096, 11008, 896, 8, False),      # ragged rows (rows >= M are never touched)
    (32, 4096, 14336, 896, 8, False),      # Emu3's intermediate size: 16 chunks
    (5, 512, 1024, 256, 4, True), (32, 1024, 512, 128, 2, False), (32, 2048, 2752, 512, 8, True)])
def test_g1_reduce_epilogue_matches_g1_then_f1r(dev, dtype, M, N, K, KC, waves, step_major):
    """G1 with F1r as its tail (one launch; the workgroups of a 512-column slice exchange their split-K planes device-coherently and reduce
    them in the producer's tail) is BIT-IDENTICAL to G1 followed by F1r: the residual stream h and the per-slice sums of squares -- also
    when the launch is replayed from a hipGraph (the ticket re-arms itself) and for 50 launches in a row (no stale plane is ever read)."""
    import sjd_amd.ops as ops
    g = torch.Generator().manual_seed(N + K + M)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).to(dev)
    wp = ops.pack_weight(w, KC, step_major)
    assert ops.skinny_gemm_reduce_ok(M, N, K, KC, waves, dev)
    # few-chunk shapes (a workgroup would have to reduce more rows than it has wave pairs) are refused, by the host check and by the library
    assert not ops.skinny_gemm_reduce_ok(24, 512, 512, 512, 8, dev) and not ops.skinny_gemm_reduce_ok(4, 512, 256, 896, 8, dev)
    with pytest.raises(RuntimeError):
        ops.skinny_gemm_reduce(torch.zeros(24, 512, dtype=dtype, device=dev), torch.zeros(512 * 512, dtype=dtype, device=dev), 512, 512, 512,
                               torch.zeros(24, 512, dtype=dtype, device=dev), waves=8)
    for it in range(3):
        x = torch.randn(M, K, generator=g).to(dtype).to(dev)
        h0 = torch.randn(M, N, generator=g).to(dtype).to(dev)
        h_ref = h0.clone()
        ss_ref = ops.residual_sumsq(h_ref, ops.skinny_gemm(x, wp, N, K, KC, waves=waves, step_major=step_major))
        h_got = h0.clone()
        ss_got = ops.skinny_gemm_reduce(x, wp, N, K, KC, h_got, waves=waves, step_major=step_major)
        torch.cuda.synchronize()
        assert torch.equal(h_got.view(torch.int16), h_ref.view(torch.int16)), (h_got.float() - h_ref.float()).abs().max()
        assert torch.equal(ss_got[:, :M].view(torch.int32), ss_ref[:, :M].view(torch.int32))
    # a chain of dependent launches on one residual stream, eager and as a hipGraph replay
    xs = [torch.randn(M, K, generator=g).to(dtype).to(dev) * 0.1 for _ in range(4)]
    h_ref = h0.clone()
    for i in range(48):
        ss_ref = ops.residual_sumsq(h_ref, ops.skinny_gemm(xs[i % 4], wp, N, K, KC, waves=waves, step_major=step_major))
    h_got = h0.clone()
    for i in range(48):
        ss_got = ops.skinny_gemm_reduce(xs[i % 4], wp, N, K, KC, h_got, waves=waves, step_major=step_major)
    torch.cuda.synchronize()
    assert torch.equal(h_got.view(torch.int16), h_ref.view(torch.int16)) and torch.equal(ss_got[:, :M], ss_ref[:, :M])
    h_g = h0.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ops.skinny_gemm_reduce(xs[0], wp, N, K, KC, h0.clone(), waves=waves, step_major=step_major)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(4):
            ss_g = ops.skinny_gemm_reduce(xs[i], wp, N, K, KC, h_g, waves=waves, step_major=step_major)
    h_g.copy_(h0)
    for _ in range(12):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(h_g.view(torch.int16), h_ref.view(torch.int16)) and torch.equal(ss_g[:, :M], ss_ref[:, :M])
    assert ops.reduce_timeouts() == 0


def test_g1_gateup_silu_refuses_what_it_does_not_serve(dev):
    import sjd_amd.ops as ops
    import sjd_amd._lib as L
    assert ops.gateup_silu_ok(65, 11008, 4096, 2048) and not ops.gateup_silu_ok(65, 11008, 4096, 2048, packed_z=True) and not ops.gateup_silu_ok(129, 11008, 4096, 2048) and not ops.gateup_silu_ok(32, 11008, 4096, 1024) and not ops.gateup_silu_ok(32, 100, 4096, 2048)
    assert ops.gateup_silu_ok(64, 11008, 4096, 2048) and not ops.gateup_silu_ok(40, 128, 512, 256)
    x = torch.zeros(130, 4096, dtype=torch.bfloat16, device=dev)
    wp = torch.zeros(2 * 128 * 4096, dtype=torch.bfloat16, device=dev)
    with pytest.raises(L.SjdLibraryError):
        ops.gateup_silu(x, wp, 128, 4096)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [32, 17, 64, 96, 128])
@pytest.mark.parametrize("N,K,KC", [(4096, 4096, 256), (512, 1376, 256), (1536, 2752, 1024)])
def test_f1r_residual_sumsq_and_row_norm_consumers(dev, dtype, M, N, K, KC):
    """F1r: the residual half of F1 (h bit-identical), per-slice sums of h^2; F2 / F3 with `row_norm` on a projection of the RAW
    residual stream through a gain-folded weight match F1 -> projection -> F2 / F3 (the folded-norm forward)."""
    import sjd_amd.ops as ops
    g = torch.Generator().manual_seed(N + K + M)
    x = torch.randn(M, K, generator=g).to(dtype).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).to(dev)
    h = torch.randn(M, N, generator=g).to(dtype).to(dev)
    gamma = (1 + 0.2 * torch.randn(N, generator=g)).to(dtype).to(dev)
    part = ops.skinny_gemm(x, ops.pack_weight(w, KC), N, K, KC)
    h_a, h_b, h_c = h.clone(), h.clone(), h.clone()
    y = ops.add_rmsnorm(h_a, part, gamma, 1e-5)
    sumsq = ops.residual_sumsq(h_b, part)
    assert torch.equal(h_a, h_b) and sumsq.shape == ((N + 511) // 512, ((M + 31) // 32) * 32)
    torch.testing.assert_close(sumsq.sum(0)[:M], h_b.float().pow(2).sum(-1), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(ops.residual_sumsq(h_c, None).sum(0)[:M], h.float().pow(2).sum(-1), rtol=1e-4, atol=1e-3)
    assert torch.equal(h_c, h)
    # second projection [2I, N] consumed by F3, and a qkv-shaped one consumed by F2
    I = 512
    w2 = (torch.randn(2 * I, N, generator=g) / N ** 0.5).to(dtype).to(dev)
    w2f = (w2.float() * gamma.float()[None, :]).to(dtype)
    kc2 = 256 if N % 256 == 0 else N
    ref = ops.silu_mul(ops.skinny_gemm(y, ops.pack_weight(w2, kc2), 2 * I, N, kc2), rows=M, dtype=dtype)
    got = ops.silu_mul(ops.skinny_gemm(h_b, ops.pack_weight(w2f, kc2), 2 * I, N, kc2), rows=M, dtype=dtype, row_norm=(sumsq, N, 1e-5))
    d = (ref.float() - got.float()).abs()
    assert d.mean() < 4e-3 and d.max() < 0.15, (d.mean(), d.max())
    B, n, H, Hkv, D = 1, M, 2, 1, 128
    wq = (torch.randn((H + 2 * Hkv) * D, N, generator=g) / N ** 0.5).to(dtype).to(dev)
    wqf = (wq.float() * gamma.float()[None, :]).to(dtype)
    inv = (1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))).to(dev)
    pos = torch.arange(M, device=dev)
    caches = [torch.zeros(B, Hkv, 192, D, dtype=dtype, device=dev) for _ in range(4)]
    q_ref = ops.qknorm_rope_append(ops.skinny_gemm(y, ops.pack_weight(wq, kc2), wq.shape[0], N, kc2), caches[0], caches[1],
                                   None, None, None, None, inv, pos, B, n, H, Hkv, D, None, 3, dtype=dtype)
    q_got = ops.qknorm_rope_append(ops.skinny_gemm(h_b, ops.pack_weight(wqf, kc2), wq.shape[0], N, kc2), caches[2], caches[3],
                                   None, None, None, None, inv, pos, B, n, H, Hkv, D, None, 3, dtype=dtype, row_norm=(sumsq, N, 1e-5))
    for a_, b_ in ((q_ref, q_got), (caches[0], caches[2]), (caches[1], caches[3])):
        d = (a_.float() - b_.float()).abs()
        assert d.mean() < 4e-3 and d.max() < 0.15, (d.mean(), d.max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H,Hkv,D,qk_norm", [(8, 8, 128, True), (8, 2, 128, False), (12, 12, 64, True)])
def test_f2_into_fp8_cache(dev, dtype, H, Hkv, D, qk_norm):
    """F2 with an fp8 (e4m3) cache = F2 into a 16-bit cache followed by torch's cast of x / scale, bit for bit; q unchanged."""
    import sjd_amd.ops as ops
    B, n, S, kv_len, sk, sv = 2, 16, 96, 37, 0.5, 2.0
    g = torch.Generator().manual_seed(H + D)
    qkv = torch.randn(B * n, (H + 2 * Hkv) * D, generator=g).to(dtype).to(dev)
    qn = [(1 + 0.2 * torch.randn(1, D, generator=g)).to(dtype).to(dev), (0.1 * torch.randn(1, D, generator=g)).to(dtype).to(dev),
          (1 + 0.2 * torch.randn(1, D, generator=g)).to(dtype).to(dev), (0.1 * torch.randn(1, D, generator=g)).to(dtype).to(dev)]
    if not qk_norm:
        qn = [None] * 4
    inv = (1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))).to(dev)
    pos = (500 + torch.arange(B * n)).to(dev)
    kc, vc = torch.zeros(B, Hkv, S, D, dtype=dtype, device=dev), torch.zeros(B, Hkv, S, D, dtype=dtype, device=dev)
    kc8, vc8 = torch.zeros(B, Hkv, S, D, device=dev).to(ops.FP8), torch.zeros(B, Hkv, S, D, device=dev).to(ops.FP8)
    q_a = ops.qknorm_rope_append(qkv, kc, vc, *qn, inv, pos, B, n, H, Hkv, D, None, kv_len)
    q_b = ops.qknorm_rope_append(qkv, kc8, vc8, *qn, inv, pos, B, n, H, Hkv, D, None, kv_len, kv_scale=(sk, sv))
    assert torch.equal(q_a, q_b)
    assert torch.equal((kc.float() / sk).to(ops.FP8).view(torch.uint8), kc8.view(torch.uint8))
    assert torch.equal((vc.float() / sv).to(ops.FP8).view(torch.uint8), vc8.view(torch.uint8))
    assert kc8.view(torch.uint8)[:, :, kv_len:kv_len + n].float().abs().sum() > 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_16bit_window_forward_error_against_fp32_forward(dev, dtype):
    """Reported and bounded: max |delta logit| of the hand-written 16-bit window forward (G1 + F1r/F2/F3 + K1, folded norm) against an
    fp32 forward of the SAME weights (hipBLASLt fp32 GEMMs + the exact-fp32 K1), next to the same figure for the plain PyTorch-ROCm
    16-bit forward (ATen GEMMs / norms / RoPE; attention = K1).  The hand-written path must not be further from fp32 than the
    library path is (x1.5 + 1e-3 slack): its rounding points are the reference's."""
    import json
    import os
    import sjd_amd.ops as ops
    from tests.helpers import make_chameleon
    conf = dict(vocab_size=9216, hidden_size=1024, intermediate_size=2048, num_hidden_layers=4, num_attention_heads=8,
                num_key_value_heads=8, max_position_embeddings=512, rms_norm_eps=1e-5, rope_theta=10000.0)
    toks = torch.randint(4, 9000, (2, 48), generator=torch.Generator().manual_seed(1)).to(dev)
    toks2 = torch.randint(4, 9000, (2, 16), generator=torch.Generator().manual_seed(2)).to(dev)
    ks = torch.tensor([0, 7], dtype=torch.int32, device=dev)
    outs = {}
    for tag, dt, fused in (("fp32", torch.float32, None), ("aten16", dtype, None), ("hip16", dtype, "sjd")):
        m = make_chameleon(conf, 23, 0.5, ops.HipWindowAttention(n_split=2), dtype=torch.float32, device=dev)
        # the 16-bit models hold the 16-bit ROUNDED weights; the fp32 reference uses exactly those values
        m = m.to(dtype).to(dt) if dt == torch.float32 else m.to(dt)
        if fused:
            m.G1_CFG = dict(qkv=(512, 8, True), o=(256, 8, False), gate_up=(512, 8, True), down=(512, 8, False))
            m.enable_fused(ops, gemm=fused)
        m.setup_cache(batch=2, s_max=128)
        m.forward_window(toks, torch.arange(48)[None].repeat(2, 1).to(dev), 0, ks)
        outs[tag] = m.forward_window(toks2, (48 + torch.arange(16))[None].repeat(2, 1).to(dev), 48, ks).float()
    e_aten = (outs["aten16"] - outs["fp32"]).abs()
    e_hip = (outs["hip16"] - outs["fp32"]).abs()
    rep = dict(dtype=str(dtype), logit_std=round(float(outs["fp32"].std()), 3), aten16_max=round(float(e_aten.max()), 5),
               aten16_mean=round(float(e_aten.mean()), 6), hip16_max=round(float(e_hip.max()), 5), hip16_mean=round(float(e_hip.mean()), 6))
    print("16-bit forward vs fp32:", rep)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"logit_error_{str(dtype).split('.')[-1]}.json"), "w") as f:
            json.dump(rep, f)
    assert e_hip.max() <= 1.5 * e_aten.max() + 1e-3 and e_hip.mean() <= 1.5 * e_aten.mean() + 1e-4, rep


def _z_weight(N, K, gen, dev, outliers):
    """bf16 weight with a few values far outside the 16-binade window of their unit (exceptions: patched into the operand registers)"""
    w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(torch.bfloat16)
    outliers = min(outliers, N * K // 4096)            # (keep toy shapes inside the 32-entry header)
    if outliers:
        idx_n = torch.randint(0, N, (outliers,), generator=gen)
        idx_k = torch.randint(0, K, (outliers,), generator=gen)
        vals = torch.tensor([0.0, 37.0, -1e-12, 2e-9, -512.0, 1e-30])[torch.randint(0, 6, (outliers,), generator=gen)]
        w[idx_n, idx_k] = vals.to(torch.bfloat16)
        w[0, 0], w[N - 1, K - 1], w[31, 7], w[N - 32, K - 16] = 0.0, 96.0, -3e-15, 1e-8       # first / last lane, element and k-step of a unit
    return w.to(dev)


@pytest.mark.parametrize("M,N,K,KC,waves,step_major", [
    (32, 12288, 4096, 896, 8, True),       # Lumina-7B q|k|v: ragged last chunk, step-major
    (32, 4096, 4096, 512, 6, False),       # o
    (32, 4096, 11008, 896, 8, False),      # down
    (32, 22016, 4096, 2048, 8, True),      # gate|up as plain G1z (128 k-steps per unit)
    (17, 4096, 11008, 896, 8, False), (64, 4096, 4096, 512, 8, False), (40, 12288, 4096, 896, 8, True),
    (32, 8224, 4096, 1024, 4, True), (5, 512, 1024, 256, 4, True), (32, 1024, 528, 128, 2, False), (64, 2048, 2752, 512, 11, True),
    (32, 96, 48, 16, 3, False),
    # the sub-tiled kernel: 65..128 rows (three / four prompts per forward), and a 64-row window whose K chunk does not fit in LDS
    (128, 4096, 11008, 896, 8, False), (96, 12288, 4096, 896, 8, True), (128, 22016, 4096, 2048, 8, True), (100, 512, 1376, 256, 3, True),
    (64, 4096, 4096, 2048, 8, True), (128, 8224, 4096, 1024, 4, True), (70, 1024, 528, 128, 6, False),
    # Emu3-8B in bf16 (what the reference's test_emu3.py:27 runs): the 64-row launch shapes of G1_CFG_EMU3 -- q|k|v 6144 columns, o, down with
    # K 14336, gate|up 28672 columns as plain G1z
    (64, 6144, 4096, 512, 8, False), (64, 4096, 14336, 896, 8, False), (64, 28672, 4096, 2048, 8, True),
    # ... and of G1_CFG_EMU3_Z (round 4: the launch shapes the compressed stream runs at)
    (64, 6144, 4096, 512, 6, True), (64, 4096, 4096, 512, 4, True), (64, 4096, 14336, 896, 8, True)])
@pytest.mark.parametrize("outliers", [0, 300])
def test_g1z_matches_g1_bit_for_bit(dev, M, N, K, KC, waves, step_major, outliers):
    """G1z (the projection over the 12-bit lossless weight stream) writes the SAME split-K planes as G1 over the uncompressed packing of the
    same weight -- every MFMA operand is reconstructed bit for bit, exceptions included -- also from a hipGraph and over a column window."""
    import sjd_amd.ops as ops
    g = torch.Generator().manual_seed(N + K + M + outliers)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = _z_weight(N, K, g, dev, outliers)
    wp, wz = ops.pack_weight(w, KC, step_major), ops.pack_weight_z(w, KC, step_major)
    assert wz is not None and wz.data.numel() <= wp.numel() * 2 * 0.76 + 1536 * (N // 32) * -(-K // KC)
    if outliers:
        assert wz.n_exceptions >= 4
    ref = ops.skinny_gemm(x, wp, N, K, KC, waves, step_major).data
    got = ops.skinny_gemm(x, wz, N, K, KC, waves, step_major).data
    torch.cuda.synchronize()
    assert got.shape == ref.shape and torch.equal(got.view(torch.int32), ref.view(torch.int32)), (got - ref).abs().max()
    if N >= 256:                           # a column window of the packed weight (the output head on the grammar's columns)
        c0, nc = 64, N - 160
        ref_c = ops.skinny_gemm_cols(x, wp, N, K, KC, c0, nc, waves, step_major).data
        got_c = ops.skinny_gemm_cols(x, wz, N, K, KC, c0, nc, waves, step_major).data
        assert torch.equal(got_c.view(torch.int32), ref_c.view(torch.int32)) and torch.equal(got_c, ref[:, :, c0:c0 + nc])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ops.skinny_gemm(x, wz, N, K, KC, waves, step_major)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        cap = ops.skinny_gemm(x, wz, N, K, KC, waves, step_major).data
    cap.zero_()
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(cap.view(torch.int32), ref.view(torch.int32))


@pytest.mark.parametrize("M,N,K,KC,n_wg", [
    (32, 12288, 4096, 512, 256),        # q|k|v at the engine's shape: 8 K chunks x 32 workgroups, 12 column tiles (4 per consumer) each
    (32, 4096, 11008, 688, 256),        # down: 16 chunks of 43 k-steps (an odd count: the last pair is half empty), 8 tiles per workgroup (3 / 3 / 2)
    (32, 4096, 4096, 512, 256),         # o: 4 tiles per workgroup (2 / 1 / 1)
    (17, 1024, 1536, 512, 24),          # fewer rows, a small launch: 3 chunks x 8 workgroups
    (32, 4096, 4096, 1024, 128),        # KC 1024: rings of two slots
    (5, 256, 640, 512, 2),              # a ragged last chunk (128 columns = 4 pairs: one slot), one workgroup per chunk walking 8 tiles
])
@pytest.mark.parametrize("outliers", [0, 300, 3000, 3001])
def test_g1_engine_matches_g1z_bit_for_bit(dev, M, N, K, KC, n_wg, outliers):
    """round 5 stage A: G1z in the loader / consumer form (LDS-DMA weight rings, csrc/sjd_gemm_engine.h) writes the SAME split-K planes as
    g1z_skinny_gemm -- the records take another road into the MFMA (HBM -> LDS -> registers), the decode, the operands and their order do not --
    eagerly, from poisoned output, forty launches in a row, and from a hipGraph; no bounded poll gives up."""
    import sjd_amd.ops as ops
    g = torch.Generator().manual_seed(N + K + M + outliers)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = _z_weight(N, K, g, dev, min(outliers, 300))
    if outliers >= 3000:                   # one unit with 50 / 100 out-of-window weights: headers of 64 / 128 entries (two / four header DMAs per slot)
        n_bad = 50 if outliers == 3000 else 100
        rr = torch.randint(32, 64, (n_bad,), generator=g)
        cc = torch.randperm(min(KC, K), generator=g)[:n_bad]
        w[rr.to(dev), cc.to(dev)] = torch.tensor(37.0, dtype=torch.bfloat16, device=dev)
    wz = ops.pack_weight_z(w, KC, False)
    assert wz is not None
    if outliers >= 3000:
        assert wz.cap == (64 if outliers == 3000 else 128)
    ref = ops.skinny_gemm(x, wz, N, K, KC, 8, False).data
    t0 = ops.engine_timeouts()
    for _ in range(3):
        got = ops.skinny_gemm_engine(x, wz, n_wg=n_wg).data
    torch.cuda.synchronize()
    assert got.shape == ref.shape and torch.equal(got.view(torch.int32), ref.view(torch.int32)), (got - ref).abs().max()
    if N >= 1024:                          # a column window (tile0 > 0)
        c0, nc = 64, N - 192
        got_c = ops.skinny_gemm_engine(x, wz, n_wg=max(-(-K // KC), (min(n_wg, 64) // -(-K // KC)) * -(-K // KC)), col0=c0, n_cols=nc).data
        assert torch.equal(got_c, ref[:, :, c0:c0 + nc])
    outs = [ops.skinny_gemm_engine(x, wz, n_wg=n_wg).data for _ in range(40)]
    torch.cuda.synchronize()
    assert all(torch.equal(o.view(torch.int32), ref.view(torch.int32)) for o in outs)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ops.skinny_gemm_engine(x, wz, n_wg=n_wg)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        cap = ops.skinny_gemm_engine(x, wz, n_wg=n_wg).data
    cap.fill_(float("nan"))
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(cap.view(torch.int32), ref.view(torch.int32))
    assert ops.engine_timeouts() == t0


@pytest.mark.parametrize("step_major", [False, True])
@pytest.mark.parametrize("M,I,K", [(32, 11008, 4096), (17, 11008, 4096), (32, 1408, 512), (5, 128, 1024), (32, 2752, 2048),
                                   (64, 11008, 4096), (40, 1408, 1024), (64, 2752, 2048), (64, 14336, 4096)])       # (the last: Emu3-8B bf16)
@pytest.mark.parametrize("with_norm", [True, False])
@pytest.mark.parametrize("outliers", [0, 200])
def test_g1sz_matches_g1s_bit_for_bit(dev, step_major, M, I, K, with_norm, outliers):
    """G1sz (gate|up + SiLU * up over the 12-bit stream) is BIT-IDENTICAL to G1s over the uncompressed packing (and hence to G1 + F3)."""
    import sjd_amd.ops as ops
    g = torch.Generator().manual_seed(I + K + M + outliers)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = _z_weight(2 * I, K, g, dev, outliers)
    wp, wz = ops.pack_weight(w, K // 2, step_major), ops.pack_weight_z(w, K // 2, step_major)
    assert wz is not None
    rn = (ops.residual_sumsq(x.clone(), None), K, 1e-5) if with_norm else None
    ref = ops.gateup_silu(x, wp, I, K, step_major, row_norm=rn)
    got = ops.gateup_silu(x, wz, I, K, step_major, row_norm=rn)
    torch.cuda.synchronize()
    assert got.shape == (M, I) and torch.equal(got.view(torch.int16), ref.view(torch.int16)), (got.float() - ref.float()).abs().max()


def _hostile_weights(kind, N, K, g):
    """weight statistics a real checkpoint can have and synthetic Gaussians never do (reference IS:287-289, ML:83-140)"""
    w = torch.randn(N, K, generator=g) / K ** 0.5
    if kind == "zero_rows":                    # pruned output channels
        w[[5, 37, 38]] = 0.0
        w[100:110] = 0.0
    elif kind == "zero_blocks":                # pruned input blocks + a half-zero tile
        w[:, 64:192] = 0.0
        w[48:64, 300:] = 0.0
    elif kind == "student_t3":
        torch.manual_seed(int(g.initial_seed()))
        w = torch.distributions.StudentT(3.0).sample((N, K)) * 0.01
    elif kind == "scale_spread":               # a norm gain folded into the columns with a 100 x per-channel spread, plus dead channels
        gain = torch.logspace(-2, 0, K)[torch.randperm(K, generator=g)]
        gain[::37] = 0.0
        w = w * gain[None, :]
    return w.to(torch.bfloat16)


@pytest.mark.parametrize("kind", ["zero_rows", "zero_blocks", "student_t3", "scale_spread"])
@pytest.mark.parametrize("M,N,K,KC,waves,step_major", [(32, 512, 1024, 512, 6, True), (17, 256, 880, 256, 4, False), (64, 512, 2048, 1024, 8, True),
                                                       (128, 256, 1024, 512, 8, True), (64, 512, 2048, 512, 4, True), (40, 256, 880, 256, 4, False)])
def test_g1z_raw_units_match_g1_bit_for_bit(dev, kind, M, N, K, KC, waves, step_major):
    """round 6: the matrix ALWAYS packs -- units the 12-bit format cannot hold travel verbatim and the fix-up launch (csrc/sjd_gemm_raw.h) recomputes
    their tiles: the planes are those of G1 on the uncompressed stream, bit for bit, for every weight statistic above; also through a column
    window (the output head's launch) that holds only some of the raw units."""
    import sjd_amd.ops as ops
    g = torch.Generator().manual_seed(N + K + M)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = _hostile_weights(kind, N, K, g).to(dev)
    wp, wz = ops.pack_weight(w, KC, step_major), ops.pack_weight_z(w, KC, step_major)
    assert wz is not None and wz.stats["max_exceptions"] <= 127
    if kind in ("zero_rows", "zero_blocks"):
        assert wz.n_raw > 0
    a = ops.skinny_gemm(x, wp, N, K, KC, waves=waves, step_major=step_major)
    z = ops.skinny_gemm(x, wz, N, K, KC, waves=waves, step_major=step_major)
    assert torch.equal(a.data, z.data)
    lo, n = 64, N - 128
    a = ops.skinny_gemm_cols(x, wp, N, K, KC, lo, n, waves=waves, step_major=step_major)
    z = ops.skinny_gemm_cols(x, wz, N, K, KC, lo, n, waves=waves, step_major=step_major)
    assert torch.equal(a.data, z.data)


@pytest.mark.parametrize("kind", ["zero_rows", "zero_blocks", "student_t3", "scale_spread"])
@pytest.mark.parametrize("T,inter,hidden,step_major", [(32, 512, 1024, True), (9, 256, 512, False), (64, 512, 2048, True), (32, 11008, 4096, True)])
def test_g1sz_raw_pairs_match_g1s_bit_for_bit(dev, kind, T, inter, hidden, step_major):
    """the same for the fused gate|up kernel: a raw unit lists its (gate tile, up tile) pair, the fix-up redoes both accumulations and the SiLU
    epilogue of the pair -- the activations are G1s's on the uncompressed stream, bit for bit, with and without the folded RMSNorm's row scale"""
    import sjd_amd.ops as ops
    g = torch.Generator().manual_seed(inter + hidden + T)
    x = torch.randn(T, hidden, generator=g).to(torch.bfloat16).to(dev)
    w = _hostile_weights(kind, 2 * inter, hidden, g).to(dev)
    wp, wz = ops.pack_weight(w, hidden // 2, step_major), ops.pack_weight_z(w, hidden // 2, step_major, gateup=True)
    assert wz is not None
    if kind in ("zero_rows", "zero_blocks"):
        assert wz.n_raw_pairs > 0 and wz.n_raw == 4 * wz.n_raw_pairs
    ss = ops.residual_sumsq(x.clone())
    for rn in (None, (ss, hidden, 1e-5)):
        ya = ops.gateup_silu(x, wp, inter, hidden, step_major, row_norm=rn)
        yz = ops.gateup_silu(x, wz, inter, hidden, step_major, row_norm=rn)
        assert torch.equal(ya.view(torch.int16), yz.view(torch.int16))
    # the SAME packed weight through the plane kernel (what a 65..128-row window runs): its raw units in plane order
    pa = ops.skinny_gemm(x, wp, 2 * inter, hidden, hidden // 2, waves=8, step_major=step_major)
    pz = ops.skinny_gemm(x, wz, 2 * inter, hidden, hidden // 2, waves=8, step_major=step_major)
    assert torch.equal(pa.data, pz.data)


@pytest.mark.parametrize("M,N,K,KC,tiles,step_major", [
    (256, 12288, 4096, 2048, 4, True), (160, 4096, 11008, 1408, 4, True), (256, 22016, 4096, 2048, 8, True), (200, 4096, 4096, 1024, 2, False),
    (130, 512, 1376, 272, 3, False), (64, 4096, 11008, 688, 6, True), (48, 256, 176, 48, 2, True), (256, 1056, 4096, 4096, 3, True), (96, 6144, 4096, 1024, 4, True),
    (64, 6144, 4096, 512, 3, True), (128, 8224, 4096, 1024, 8, True)])
@pytest.mark.parametrize("kind", ["gauss", "outliers", "wide_headers", "raw"])
def test_g1w_over_the_12bit_stream_experimental(dev, M, N, K, KC, tiles, step_major, kind):
    """late round 6 experiment (libsjd_hip_exp.so: sjd_skinny_gemm_z_wide; measured slower than the product's kernels): kernel G1w decoding the
    12-bit record pairs in front of its MFMAs -- the same planes as G1 over the uncompressed packing, bit for bit: 33..256 rows, every tile count, odd k-step
    counts per chunk (a half-empty last pair), ragged chunks, one K chunk, both record orders, headers of 128 exceptions, raw units (through the fix-up
    launch), a column window."""
    import sjd_amd.ops as ops
    g = torch.Generator().manual_seed(N + K + M)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    if kind == "raw":
        w = _hostile_weights("zero_rows", N, K, g).to(dev)
    else:
        w = _z_weight(N, K, g, dev, {"gauss": 0, "outliers": 300, "wide_headers": 3000}[kind])
    wp, wz = ops.pack_weight(w, KC, step_major), ops.pack_weight_z(w, KC, step_major)
    assert wz is not None and (kind != "raw" or wz.n_raw > 0)
    ref = ops.skinny_gemm(x, wp, N, K, KC, tiles, step_major).data
    got = ops.skinny_gemm_z_wide(x, wz, tiles).data
    torch.cuda.synchronize()
    assert got.shape == ref.shape and torch.equal(got.view(torch.int32), ref.view(torch.int32)), (got - ref).abs().max()
    if N >= 256:
        c0, nc = 64, N - 160
        got_c = ops.skinny_gemm_z_wide(x, wz, tiles, col0=c0, n_cols=nc).data
        assert torch.equal(got_c, ref[:, :, c0:c0 + nc])


def test_g1z_refuses_what_it_does_not_serve(dev):
    import sjd_amd._lib as L
    import sjd_amd.ops as ops
    w = (torch.randn(64, 256) * 0.02).to(torch.bfloat16).to(dev)
    wz = ops.pack_weight_z(w, 128)
    x16 = torch.randn(8, 256, device=dev).to(torch.float16)
    out = torch.empty(2, 32, 64, device=dev)
    lib = L.load()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.sjd_skinny_gemm_z(p(x16), p(wz.data), p(wz.exc), 32, p(out), 8, 64, 256, 128, 2, 0, 1, 64, 0, s) != 0      # fp16
    assert lib.sjd_skinny_gemm_z(p(x16), p(wz.data), p(wz.exc), 32, p(out), 129, 64, 256, 128, 2, 0, 0, 64, 0, s) != 0    # > 128 rows
    assert lib.sjd_skinny_gemm_z(p(x16), p(wz.data), p(wz.exc), 32, p(out), 96, 64, 256, 128, 12, 0, 0, 64, 0, s) != 0    # sub-tiled kernel: <= 8 waves
    assert lib.sjd_skinny_gemm_z(p(x16), p(wz.data), None, 32, p(out), 8, 64, 256, 128, 2, 0, 0, 64, 0, s) != 0           # no header table
    assert lib.sjd_skinny_gemm_z(p(x16), p(wz.data), p(wz.exc), 48, p(out), 8, 64, 256, 128, 2, 0, 0, 64, 0, s) != 0      # header capacity
    assert lib.sjd_gateup_silu_z(p(x16), p(wz.data), p(wz.exc), 32, p(out), 8, 64, 512, 0, 1, None, s) != 0               # fp16


@pytest.mark.parametrize("M,N,K,KC,waves,step_major", [(32, 4096, 4096, 1024, 6, True), (64, 2048, 2752, 1024, 8, False), (32, 1024, 4096, 2048, 8, True)])
def test_g1z_wide_exception_headers(dev, M, N, K, KC, waves, step_major):
    """heavy-tailed weights (Student-t, 2 degrees of freedom: 50-100 out-of-window weights per unit) make the packer choose headers of 64 or
    128 entries -- a lane then patches from two register sets -- and the planes still equal G1's bit for bit; the same for G1sz"""
    import sjd_amd.ops as ops
    g = torch.Generator().manual_seed(N + K)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    torch.manual_seed(N + K)
    w = (torch.distributions.StudentT(2.0).sample((N, K)) * 0.015).to(torch.bfloat16).to(dev)
    wp, wz = ops.pack_weight(w, KC, step_major), ops.pack_weight_z(w, KC, step_major)
    assert wz is not None and wz.cap in (64, 128), wz and wz.cap
    ref = ops.skinny_gemm(x, wp, N, K, KC, waves, step_major).data
    got = ops.skinny_gemm(x, wz, N, K, KC, waves, step_major).data
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int32), ref.view(torch.int32))
    if K == 4096 and N % 128 == 0:
        wp2, wz2 = ops.pack_weight(w, K // 2, step_major), ops.pack_weight_z(w, K // 2, step_major)
        assert wz2 is not None and wz2.cap in (64, 128)
        a = ops.gateup_silu(x, wp2, N // 2, K, step_major)
        b = ops.gateup_silu(x, wz2, N // 2, K, step_major)
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


@pytest.mark.parametrize("M,I,KC_dn,step_major_dn", [(32, 11008, 768, False), (17, 11008, 768, False), (32, 11008, 896, False), (32, 2816, 704, True)])
@pytest.mark.parametrize("with_norm", [True, False])
def test_mlp_pair_matches_g1sz_then_g1z(dev, M, I, KC_dn, step_major_dn, with_norm):
    """sjd_mlp_pair_z (round-4 experiment: gate|up + SiLU * up and the down projection in ONE launch, the down workgroups' weight stream
    started ahead of the dependency edge) writes the activation and the split-K planes of G1sz followed by G1z, bit for bit -- eagerly, forty
    launches in a row (the arrival counters re-arm themselves) and replayed from a hipGraph; no wait is ever abandoned."""
    import sjd_amd.ops as ops
    hid = 4096
    g = torch.Generator().manual_seed(I + M + KC_dn)
    x = torch.randn(M, hid, generator=g).to(torch.bfloat16).to(dev)
    wgu = _z_weight(2 * I, hid, g, dev, 100)
    wdn = _z_weight(hid, I, g, dev, 100)
    gu = ops.pack_weight_z(wgu, hid // 2, True)
    dn = ops.pack_weight_z(wdn, KC_dn, step_major_dn)
    assert gu is not None and dn is not None
    if not ops.mlp_pair_ok(M, I, hid, gu, dn, KC_dn, 8, dev):
        pytest.skip("launch larger than the device holds at once")
    rn = (ops.residual_sumsq(x.clone(), None), hid, 1e-5) if with_norm else None
    y_ref = ops.gateup_silu(x, gu, I, hid, True, row_norm=rn)
    p_ref = ops.skinny_gemm(y_ref, dn, hid, I, KC_dn, 8, step_major_dn).data
    torch.cuda.synchronize()
    t0 = ops.mlp_pair_timeouts()
    for it in range(40):
        y, part = ops.mlp_pair(x, gu, dn, I, hid, KC_dn, row_norm=rn)
        if it in (0, 39):
            torch.cuda.synchronize()
            assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16)), (it, (y.float() - y_ref.float()).abs().max())
            assert part.data.shape == p_ref.shape and torch.equal(part.data.view(torch.int32), p_ref.view(torch.int32)), (it, (part.data - p_ref).abs().max())
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ops.mlp_pair(x, gu, dn, I, hid, KC_dn, row_norm=rn)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        outs = [ops.mlp_pair(x, gu, dn, I, hid, KC_dn, row_norm=rn) for _ in range(3)]
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    for y, part in outs:
        assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16)) and torch.equal(part.data.view(torch.int32), p_ref.view(torch.int32))
    assert ops.mlp_pair_timeouts() == t0
