#!/usr/bin/env python3
"""bench.py -- accepted image-tokens/s of the MI355X-native SJD engine on BASELINE.json's headline workload.

Workload (config.workload): Lumina-mGPT-7B architecture (Chameleon-7B dims, bf16, random-init synthetic weights --
no checkpoints reach the GPU box), one 768x768 prompt per GPU (P=64 incl. <start> h w, 48x(48+1) image tokens),
draft window 16, CFG 3.0 (cond||uncond batch of 2), image top-k 2000, speculative_jacobi, seed 1234+rank.
A "step" is ONE SJD iteration (= one transformer forward over the draft window + the whole hand-written hot path).

Where the K timed steps sit: the cost of a step grows with the KV length, so the decode first runs an UNTIMED lead-in (real SJD
iterations from the prompt) until the KV length is such that the W warm-up + K timed steps are centred on the mean KV length of
a whole image (P + image/2 = 1240 for 768x768; `--kv-center 0` times from the prompt as round 1 did).  At N=1 the same decode
then continues to the end of the image, so the line also carries the MEASURED whole-image NFE / tokens/s (`whole_image`) and
ms/step at KV lengths {64, 1216, 2368} (`per_kv`), plus bounded side legs: `floor` (1 / ms_per_step), `acceptance_probe_unscaled_embeddings` (plain random embeddings),
`torch_baseline` (the reference's data flow in PyTorch-ROCm ops on the same weights at the same KV length: `vs_baseline`), and
`cpu_baseline` (the oracle's scheduler step on the host cores).

python bench.py [--gpus N] [--steps K] [--warmup W]
   N > 1 without a torch.distributed environment: re-executes itself under `python -m torch.distributed.run --nproc-per-node N`
   (one rank per GPU, RCCL); under the driver's own torchrun launch it reads RANK / LOCAL_RANK / WORLD_SIZE.  Prompt i -> rank i
   (weak scaling), no collective inside the decode, ONE all_gather of [tokens, steps, seconds] at the end.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# kernel arguments in device memory instead of host-coherent memory (the HIP runtime reads this when it is loaded, i.e. before torch is
# imported): 3.507 against 3.522 ms/step on one box; a deployment sets it the same way (conventions)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--model", default="lumina7b", choices=["lumina7b", "lumina_tiny", "emu3_8b", "anole7b"],
                    help="lumina7b = BASELINE.json's metric config; emu3_8b = config 3 (720x720, GQA 32/8, V=184622, fp16)")
    ap.add_argument("--dtype", default=None, choices=[None, "bf16", "fp16"])
    ap.add_argument("--kv", default="auto", choices=["auto", "16bit", "fp8"],
                    help="KV-cache element type: the activation dtype, or OCP fp8 e4m3 with the fp8-MFMA K1 (auto: fp8 for anole7b = BASELINE config 5)")
    ap.add_argument("--prompts-per-gpu", type=int, default=1,
                    help="decode this many independent prompts per GPU in ONE window forward (SJDBatchEngine; 1 = the reference's "
                         "operating point and BASELINE.json's configuration; prompts x B_cfg x window must stay <= 128 rows)")
    ap.add_argument("--queue-prompts", type=int, default=0,
                    help="with --prompts-per-gpu > 1: total prompts per GPU (>= --prompts-per-gpu); a slot whose image is complete takes the "
                         "next prompt of the queue (continuous batching).  0 = one prompt per slot")
    ap.add_argument("--embed-token-scale", type=float, default=0.7)
    ap.add_argument("--window", type=int, default=16)
    ap.add_argument("--n-split", type=int, default=0, help="K1 key splits (0 = auto from batch x kv heads)")
    ap.add_argument("--kv-center", type=int, default=-1,
                    help="KV length the timed region is centred on (-1: the mean over a whole image, P + image/2; 0: no lead-in, time from the prompt)")
    ap.add_argument("--no-whole-image", action="store_true", help="stop after the timed region (default at N=1: decode on to the end of the image)")
    ap.add_argument("--no-floor", action="store_true")
    ap.add_argument("--floor-steps", type=int, default=96)
    ap.add_argument("--no-torch-baseline", action="store_true")
    ap.add_argument("--no-ar-baseline", action="store_true")
    ap.add_argument("--ar-steps", type=int, default=96)
    ap.add_argument("--torch-baseline-steps", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-iters", type=int, default=0, help="scheduler steps of the cpu_baseline sample (0: sized for ~15 s)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--gemm", default="sjd", choices=["sjd", "torch"], help="window projections: G1 weight-streaming kernel or hipBLASLt")
    ap.add_argument("--tunableop", action="store_true", help="enable PyTorch TunableOp GEMM selection")
    ap.add_argument("--no-fold-norm", action="store_true", help="keep F1 (RMSNorm before the projection) instead of the folded-norm forward")
    ap.add_argument("--compress", action="store_true", help="keep the 12-bit weight stream for three / four prompts per forward too")
    ap.add_argument("--no-compress", action="store_true", help="stream the packed bf16 weights uncompressed (G1 / G1s) instead of the lossless "
                    "12-bit stream (G1z / G1sz); results are bit-identical either way")
    ap.add_argument("--no-fused", action="store_true", help="plain ATen element-wise glue instead of the fused F1-F3 kernels")
    ap.add_argument("--k1-launches", type=int, default=320, help="launches of the K1 micro-measurement")
    ap.add_argument("--total-prompts", type=int, default=0,
                    help="config 4's shape: decode a queue of this many prompts (whole images), split contiguously over the --gpus ranks "
                         "(prompt i keeps seed 1234 + i whatever N is; one all_gather at the end).  0 = the default per-rank K-step bench")
    ap.add_argument("--no-pin", action="store_true", help="multi-rank runs: do not pin each rank to its GPU's NUMA node")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the compact emu3_8b / anole7b records (BASELINE configs 3 and 5)")
    return ap.parse_args(argv)


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` with no torch.distributed environment: start N ranks ourselves (the reference's fan-out is one
    process per GPU, dataset_tools/multi_gpu_infer_with_prompt.py:146-172) and hand our stdout through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    return subprocess.call(cmd, env=env)


def build_model(args, device):
    import torch
    import sjd_amd.backbones as BB
    import sjd_amd.ops as ops
    import sjd_amd.synthetic as synthetic
    if args.model in ("lumina7b", "anole7b"):      # Anole-7B is the Chameleon-7B architecture
        margs = BB.LUMINA_7B
    elif args.model == "emu3_8b":
        margs = BB.EMU3_8B
    else:
        margs = BB.ChameleonArgs(vocab_size=65536, hidden_size=1024, intermediate_size=2048, num_hidden_layers=4,
                                 num_attention_heads=8, num_key_value_heads=8)
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype or ("fp16" if args.model == "emu3_8b" else "bf16")]
    attn = ops.HipWindowAttention(n_split=args.n_split or None)
    with torch.device(device):
        model = BB.ChameleonBackbone(margs, attn=attn).to(dt).eval()
    if args.prompts_per_gpu > 1 and args.model != "emu3_8b":      # 64-row windows: the staged activation chunk (64 x KC) must fit in LDS
        rows_ = 2 * args.prompts_per_gpu * args.window
        model.G1_CFG = dict(model.G1_CFG_64ROW if rows_ <= 64 else model.G1_CFG_128ROW if rows_ <= 128 else model.G1_CFG_256ROW)
    if args.model == "emu3_8b":      # 64-row windows (draft window 32) need activation chunks <= 1280 columns
        model.G1_CFG = dict(model.G1_CFG_EMU3)
    if os.environ.get("SJD_HEAD_CFG"):     # tuning aid: JSON [KC, waves, step_major] of the output head's G1 launch
        h = json.loads(os.environ["SJD_HEAD_CFG"])
        model.HEAD_CFG = (int(h[0]), int(h[1]), bool(h[2]))
    if os.environ.get("SJD_G1_CFG"):       # tuning aid: JSON {"o": [KC, waves, step_major], ...} overriding the per-projection launch shapes
        over = json.loads(os.environ["SJD_G1_CFG"])
        model.G1_CFG = dict(model.G1_CFG, **{k: (int(v[0]), int(v[1]), bool(v[2])) for k, v in over.items()})
    synthetic.fill_state_dict_device(model, seed=0, embed_token_scale=args.embed_token_scale)
    if not args.no_fused:
        # (three / four prompts per forward run on the sub-tiled kernel, which is bound by its staging barriers, not by HBM: the 12-bit stream
        # serves it -- g1z_skinny_gemm_tiled, bit-identical -- but measured 1-2 % slower there, profiles/r3_g1z_microbench.txt; --compress forces it)
        model.enable_fused(ops, gemm=args.gemm, fold_norm=not args.no_fold_norm,
                           compress=False if (args.no_compress or (args.prompts_per_gpu > 2 and not args.compress)) else None)
    return model, margs, attn


def measure_k1(args, model, attn, device, kv_len):
    """The attention kernel the engine runs, measured live with HIP events on the stream it runs on, at KV length `kv_len` (same
    B/H/D/window as the decode), cycling over all layers' caches like a real iteration (~40 MB/layer x 32 streams from HBM; a single layer
    would sit in the 256 MB Infinity Cache).  MHA 16-row window: K1F (QK-norm + RoPE + append + attention + merge, one launch, fed with
    q|k|v split-K partials); otherwise k1_partial (the split combine is a separate launch and not included)."""
    import ctypes
    import torch
    import sjd_amd._lib as L
    import sjd_amd.ops as ops
    lib = L.load()
    B, n, H, D = 2, args.window, model.n_heads, model.head_dim
    nl = model.cache.k.shape[0]
    kc, vc = model.cache.k, model.cache.v
    kv_len = min(int(kv_len), model.cache.s_max - n)
    adt = model.lm_head.weight.dtype
    ks = torch.tensor([0, 0], dtype=torch.int32, device=device) if model.n_kv_heads != model.n_heads else torch.tensor([0, 63], dtype=torch.int32, device=device)
    esz, Hkv = kc.element_size(), kc.shape[2]
    rows0, rows1 = kv_len + n, kv_len + n - int(ks[1])   # visible key rows of the cond / uncond batch row
    kv_bytes = 2 * Hkv * (rows0 + rows1) * D * esz
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import sjd_amd.backbones as BB_
    if getattr(model, "k1_fused", BB_._K1_FUSED_DEFAULT) and getattr(model, "_gemm", None) == "sjd" and ops.fused_attention_ok(B, n, H, Hkv, D, kc.dtype):
        n_chunks = (model.args.hidden_size + model.G1_CFG["qkv"][0] - 1) // model.G1_CFG["qkv"][0]
        parts = [ops.Partials(torch.randn(n_chunks, 32, 3 * H * D, device=device), n_chunks, 3 * H * D) for _ in range(4)]
        a0 = model.model.layers[0].self_attn
        qn = (a0.q_norm.weight, a0.q_norm.bias, a0.k_norm.weight, a0.k_norm.bias) if model.args.qk_norm else (None,) * 4
        pos = (kv_len + torch.arange(n, device=device)[None] - ks[:, None].long()).reshape(-1).contiguous()
        sumsq = torch.full((model.args.hidden_size // 512, 32), 512.0, device=device)
        rn = (sumsq, model.args.hidden_size, model.args.rms_norm_eps) if getattr(model, "_fold_norm", False) else None
        run = lambda i: ops.qkv_attention_fused(parts[i % 4], kc[i % nl], vc[i % nl], *qn, model._inv_freq32, pos, B, n, H, D, None, kv_len, ks,
                                                row_norm=rn, dtype=adt)
        for i in range(nl):
            run(i)
        torch.cuda.synchronize()
        e0.record()
        for i in range(args.k1_launches):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        alg = kv_bytes + n_chunks * B * n * 3 * H * D * 4 + 2 * B * n * Hkv * D * esz + B * n * H * D * esz     # + partials in, K/V rows + out written
        avg_ms = e0.elapsed_time(e1) / args.k1_launches
        return dict(kernel="k1f_qkv_attention (QK-norm + RoPE + KV append + draft-window attention, one launch)", launches=args.k1_launches,
                    avg_ms=avg_ms, avg_bytes=alg, avg_kv_rows=kv_len + n, gbps=alg / 1e9 / (avg_ms / 1e3))
    n_split = attn.n_split or 8
    q = torch.randn(B, n, H, D, device=device).to(adt)
    out = torch.empty_like(q)
    ws = ops.attention_workspace(B, H, n, D, n_split, device)
    if ops.colsplit_ok(B, n, H, Hkv, D, kc.dtype) and attn.choose_regime(kv_len + n, kc.dtype) == "colsplit":
        # the short-context form the engine launches at this KV length: k1_dsplit(_fp8), one launch, back-to-back between one event pair
        for i in range(nl):
            ops.draft_window_attention_colsplit(q, kc[i], vc[i], out, ks, None, kv_len, attn.kv_scale)
        torch.cuda.synchronize()
        e0.record()
        for i in range(args.k1_launches):
            ops.draft_window_attention_colsplit(q, kc[i % nl], vc[i % nl], out, ks, None, kv_len, attn.kv_scale)
        e1.record()
        torch.cuda.synchronize()
        alg = kv_bytes + 2 * B * n * H * D * 2
        avg_ms = e0.elapsed_time(e1) / args.k1_launches
        return dict(kernel="k1_dsplit" + ("_fp8" if kc.dtype == ops.FP8 else "") + " (column split: one launch, no combine)", launches=args.k1_launches,
                    avg_ms=avg_ms, avg_bytes=alg, avg_kv_rows=kv_len + n, gbps=alg / 1e9 / (avg_ms / 1e3))
    if kc.dtype == ops.FP8:        # fp8 cache: k1_partial_fp8 + k1_combine, back-to-back launches between one event pair
        sk, sv = attn.kv_scale
        for i in range(nl):
            ops.draft_window_attention_fp8(q, kc[i], vc[i], out, sk, sv, ks, None, kv_len, n_split, ws)
        torch.cuda.synchronize()
        e0.record()
        for i in range(args.k1_launches):
            ops.draft_window_attention_fp8(q, kc[i % nl], vc[i % nl], out, sk, sv, ks, None, kv_len, n_split, ws)
        e1.record()
        torch.cuda.synchronize()
        alg = kv_bytes + B * n * H * D * 2
        avg_ms = e0.elapsed_time(e1) / args.k1_launches
        return dict(kernel="k1_partial_fp8 + k1_combine", launches=args.k1_launches, avg_ms=avg_ms, avg_bytes=alg, avg_kv_rows=kv_len + n,
                    gbps=alg / 1e9 / (avg_ms / 1e3))
    evs = [(ctypes.c_void_p(lib.sjd_event_create()), ctypes.c_void_p(lib.sjd_event_create())) for _ in range(args.k1_launches)]
    for i in range(nl):
        ops.draft_window_attention(q, kc[i], vc[i], out, ks, None, kv_len, n_split, ws)
    torch.cuda.synchronize()
    for i, (ev0, ev1) in enumerate(evs):
        ops.draft_window_attention(q, kc[i % nl], vc[i % nl], out, ks, None, kv_len, n_split, ws, ev0, ev1)
    torch.cuda.synchronize()
    ms = [lib.sjd_event_elapsed_ms(ev0, ev1) for ev0, ev1 in evs]
    for ev0, ev1 in evs:
        lib.sjd_event_destroy(ev0)
        lib.sjd_event_destroy(ev1)
    alg = kv_bytes + B * n * H * D * esz     # K,V rows once per kv head + q
    avg_ms = sum(ms) / len(ms)
    return dict(kernel="k1_partial (draft-window attention)", launches=len(ms), avg_ms=avg_ms, avg_bytes=alg, avg_kv_rows=kv_len + n,
                gbps=alg / 1e9 / (avg_ms / 1e3))


def measure_k1_pair(args, model, attn, device, kv_len, reps=4):
    """What one layer's draft-window attention costs INSIDE the engine's hipGraph: the launches K1 consists of for this shape (k1 partial
    + k1_combine, or the single-launch form) for all layers' caches captured in one graph -- the way an iteration runs them -- and `reps`
    replays timed with HIP events on the replay stream.  The eager per-launch figure of measure_k1 carries ~5 us of launch latency per
    launch and times the partial kernel only; this one is the pair a step pays for."""
    import ctypes
    import torch
    import sjd_amd._lib as L
    import sjd_amd.ops as ops
    lib = L.load()
    hip = ctypes.CDLL("libamdhip64.so")
    B, n, H, D = 2, args.window, model.n_heads, model.head_dim
    kc, vc = model.cache.k, model.cache.v
    nl, Hkv, esz = kc.shape[0], kc.shape[2], kc.element_size()
    kv_len = min(int(kv_len), model.cache.s_max - n)
    adt = model.lm_head.weight.dtype
    ks = torch.tensor([0, 0], dtype=torch.int32, device=device) if model.n_kv_heads != model.n_heads else torch.tensor([0, 63], dtype=torch.int32, device=device)
    n_split = attn.n_split or 8
    q = torch.randn(B, n, H, D, device=device).to(adt)
    out = torch.empty_like(q)
    ws = ops.attention_workspace(B, H, n, D, n_split, device)
    fp8 = kc.dtype == ops.FP8

    # the launch form the engine uses at this context length (column split while it is short, key splits + combine after)
    regime = attn.choose_regime(kv_len + n, kc.dtype) if ops.colsplit_ok(B, n, H, Hkv, D, kc.dtype) else "keysplit"

    def one(i):
        if regime == "colsplit":
            ops.draft_window_attention_colsplit(q, kc[i], vc[i], out, ks, None, kv_len, attn.kv_scale)
        elif fp8:
            ops.draft_window_attention_fp8(q, kc[i], vc[i], out, attn.kv_scale[0], attn.kv_scale[1], ks, None, kv_len, n_split, ws)
        else:
            ops.draft_window_attention(q, kc[i], vc[i], out, ks, None, kv_len, n_split, ws)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        one(0)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(nl):
            one(i)
    graph.replay()
    torch.cuda.synchronize()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    e0, e1 = ctypes.c_void_p(lib.sjd_event_create()), ctypes.c_void_p(lib.sjd_event_create())
    hip.hipEventRecord(e0, stream)
    for _ in range(reps):
        graph.replay()
    hip.hipEventRecord(e1, stream)
    torch.cuda.synchronize()
    avg_ms = lib.sjd_event_elapsed_ms(e0, e1) / (reps * nl)
    lib.sjd_event_destroy(e0)
    lib.sjd_event_destroy(e1)
    rows0, rows1 = kv_len + n, kv_len + n - int(ks[1])
    alg = 2 * Hkv * (rows0 + rows1) * D * esz + B * n * H * D * 2
    return dict(avg_us=round(avg_ms * 1e3, 2), launches=reps * nl, avg_bytes=int(alg), achieved=round(alg / 1e9 / (avg_ms / 1e3), 1),
                frac=round(alg / 1e9 / (avg_ms / 1e3) / 8000.0, 4), n_split=int(n_split), regime=regime, avg_kv_rows=kv_len + n,
                what="all K1 launches of a layer (partial + combine where the shape needs it), 32 layers' caches in ONE hipGraph replay, HIP events on the replay stream")


def measure_g1(args, model, device, rounds=6):
    """Dominant hand-written kernel by time: G1 (weight-streaming projections).  One full pass over the model's own packed
    weights (32 layers x {qkv, o, gate|up, down} = 13.0 GB, so every launch streams from HBM) is captured in a hipGraph -- the
    way the engine launches it -- and `rounds` replays are timed with HIP events on the replay stream.  Algorithmic bytes per
    launch = the weight matrix (N*K*2) + the activations."""
    import ctypes
    import torch
    import sjd_amd._lib as L
    import sjd_amd.ops as ops
    lib = L.load()
    hip = ctypes.CDLL("libamdhip64.so")
    H, Hkv, D, hid, inter = model.n_heads, model.n_kv_heads, model.head_dim, model.args.hidden_size, model.args.intermediate_size
    shapes = dict(qkv=((H + 2 * Hkv) * D, hid), o=(hid, H * D), gate_up=(2 * inter, hid), down=(hid, inter))
    # the activation rows of the decode's own launches: B_cfg (= 2: cond || uncond) x prompts per forward x draft window -- 32 for the headline,
    # 64 for Emu3's window of 32 (round 3 staged 32 rows for every model and so priced Emu3's projections on a launch shape it does not run)
    rows = min(256, 2 * max(1, args.prompts_per_gpu) * args.window)
    xs = {k: torch.randn(rows, K, device=device).to(model.lm_head.weight.dtype) for k, (N, K) in shapes.items()}
    cfg = model.G1_CFG

    import sjd_amd.backbones as BB
    # the product runs gate|up as kernel G1s (the projection with SiLU * up as its epilogue) when the shape allows: measure what it runs
    want_fused = getattr(model, "gateup_fused", BB._GATEUP_FUSED_DEFAULT)
    fused_mlp = want_fused and (rows <= 64 or want_fused == "tall") and ops.gateup_silu_ok(rows, inter, hid, cfg["gate_up"][0],
                                                                                          isinstance(model._packed[0]["gate_up"], ops.PackedZ))
    rn = (ops.residual_sumsq(xs["gate_up"].clone(), None), hid, 1e-5)

    def one_pass():
        for li in range(len(model._packed)):
            for name, (N, K) in shapes.items():
                if name == "gate_up" and fused_mlp:
                    ops.gateup_silu(xs[name], model._packed[li][name], inter, hid, cfg[name][2], row_norm=rn)
                else:
                    ops.skinny_gemm(xs[name], model._packed[li][name], N, K, cfg[name][0], cfg[name][1], cfg[name][2])

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        one_pass()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        one_pass()
    graph.replay()
    torch.cuda.synchronize()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    e0, e1 = ctypes.c_void_p(lib.sjd_event_create()), ctypes.c_void_p(lib.sjd_event_create())
    hip.hipEventRecord(e0, stream)
    for r in range(rounds):
        graph.replay()
    hip.hipEventRecord(e1, stream)
    torch.cuda.synchronize()
    n = rounds * len(model._packed) * len(shapes)
    tot_ms = lib.sjd_event_elapsed_ms(e0, e1)
    # algorithmic bytes of a launch = the weight stream AS STORED (the lossless 12-bit form is what the algorithm has to move; the bf16 size is
    # reported next to it) + the activation rows
    wbytes = lambda p, N, K: p.nbytes() if isinstance(p, ops.PackedZ) else N * K * 2
    tot_b = rounds * sum(wbytes(pk[name], N, K) + rows * K * 2 for pk in model._packed for name, (N, K) in shapes.items())
    tot_b16 = rounds * len(model._packed) * sum(N * K * 2 + rows * K * 2 for N, K in shapes.values())
    nz = sum(isinstance(pk[name], ops.PackedZ) for pk in model._packed for name in shapes)
    cs = getattr(model, "compress_stats", None) or {}
    pack = {k: cs[k] for k in ("matrices", "compressed", "declined", "units", "raw_units", "exceptions", "max_exceptions_per_unit") if k in cs}
    if pack.get("units"):
        pack["exceptions_per_unit"] = round(pack.get("exceptions", 0) / pack["units"], 2)
    return dict(launches=n, avg_ms=tot_ms / n, avg_bytes=tot_b / n, gbps=tot_b / 1e9 / (tot_ms / 1e3), fused_mlp=bool(fused_mlp), rows=rows,
                compressed_launches=rounds * nz, avg_bytes_bf16=tot_b16 / n, gbps_bf16_equivalent=tot_b16 / 1e9 / (tot_ms / 1e3), pack=pack)


def cpu_baseline(args, gpu_sched_ms=None):
    """The reference's scheduler step (logits->probs->sample + verify/accept) as restated by the CPU oracle, timed on this host on a
    bounded sample of the same workload: V=65536, L=16, CFG, image top-k 2000.  The window rows of the K2 restatement are independent
    and run one per OpenMP thread on all host cores (min(nproc, L) threads do work); the accept scan is sequential, as in the reference."""
    import torch
    from oracle import sjd_oracle as O
    V, L = 65536, args.window
    nproc = os.cpu_count() or 1
    threads = O.set_threads(min(nproc, L))          # one window row per thread: more threads than rows only spin
    g = torch.Generator().manual_seed(0)
    ctx = [9000] * 61 + [8197, 8828, 8828] + [100] * 40
    rules = O.lumina_rules(ctx, L, 2000, 10)
    resid = [O.lumina_rules(ctx, 1, 2000, 10)[0] for _ in range(L - 1)]
    ring = []                                   # eight pre-drawn input sets, cycled: the sample times the scheduler step, not the RNG
    for _ in range(8):
        ring.append(((torch.randn(2, L, V, generator=g) * 3.0).numpy(), torch.empty(L, V).exponential_(generator=g).numpy(),
                     torch.rand(L, V, generator=g).numpy(), torch.empty(V).exponential_(generator=g).numpy()))

    def step(it, prev):
        logits, noise, rs, e2 = ring[it % len(ring)]
        toks, probs = O.logits_to_probs_sample(logits[0], logits[1], 3.0, rules, noise)
        q_rows = [None] * L if prev is None else [prev[i] for i in range(L)]
        win = [100] + toks[:-1].tolist()
        O.verify_accept(win, toks, probs, q_rows, rs, resid, e2)
        return probs

    prev = None
    t0 = time.perf_counter()
    for it in range(3):                          # probe: size the sample for ~15 s of CPU work
        prev = step(it, prev)
    probe = (time.perf_counter() - t0) / 3
    n_it = args.cpu_baseline_iters or int(min(20000, max(50, 15.0 / max(probe, 1e-4))))
    t_total = 0.0
    for it in range(n_it):
        t0 = time.perf_counter()
        prev = step(it, prev)
        t_total += time.perf_counter() - t0
    sec = t_total / n_it
    out = {"value": round(1.0 / sec, 2), "unit": "SJD scheduler steps/s (logits->tokens + verify/accept only; no transformer forward)",
           "cores": min(threads, L), "host_cores": nproc, "kind": "port",
           "kind_detail": ("C/OpenMP restatement of the reference's scheduler step (oracle/sjd_oracle.c), NOT the reference's Python/ATen CPU "
                           "path: that one measured 46.7 + 14.1 = 60.8 ms per step at 8 cores in the build container (SURVEY.md 8d)"),
           "ms_per_step": round(sec * 1e3, 3),
           "sample": f"{n_it} SJD scheduler steps, V=65536, L={L}, CFG, top-k 2000 ({t_total:.1f} s of CPU work)"}
    if gpu_sched_ms:
        out["gpu_same_work_ms_per_step"] = round(gpu_sched_ms, 4)
    return out


def measure_scheduler_gpu(eng, grammar0, prompt, window, reps=200):
    """K2 + K4 (the GPU side of what cpu_baseline times) on a full image-body window: HIP-event time per step.  The iteration blob is
    set to `window` image rows (rules of a context just inside the image); logits are those of the engine's last body forward."""
    import copy
    import torch
    g = copy.deepcopy(grammar0)
    ctx = list(prompt) + [100] * 5
    g.start(ctx)
    rules = g.window_rules(window)
    cols = eng.logit_columns(rules)
    logits = eng.last_head_output(cols)
    if logits is None:
        return None
    resid = g.residual_rules([100] * window)
    import sjd_amd.ops as ops
    step = ops.philox_step(window * eng.V, ops.philox_max_blocks(eng.device))
    eng._fill_params(window, 100, True, 0, [], rules, resid, philox=(ops.philox_max_blocks(eng.device), 1234, 0, step, 2 * step))      # in-kernel noise, as the decode runs it
    ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        eng._sample_body(0, logits, cols)
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        eng._sample_body(0, logits, cols)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


def per_kv_table(iter_log, points, half=48):
    """median ms/step of the iterations whose KV length lies within +-half of each point (host clock after each iteration's sync)"""
    out = {}
    for S in points:
        d = sorted((iter_log[i][3] - iter_log[i - 1][3]) * 1e3 for i in range(1, len(iter_log))
                   if abs(iter_log[i][0] - S) <= half and iter_log[i][1] > 1)
        if d:
            out[str(S)] = {"ms_per_step": round(d[len(d) // 2], 4), "iterations": len(d)}
    return out


def workload_of(args, margs, rank, device, prompt_index=None):
    """-> dict(prompt, spec, grammar, cfg, P, n_img, grid, workload, tau_est) of BASELINE.json's configs 2 / 3 / 5 for this rank (or for
    prompt `prompt_index` of a queue: the seed then depends on the prompt, not on the rank that happens to decode it)"""
    from sjd_amd.engine import SJDConfig
    from sjd_amd.grammar import LuminaGrammar
    from sjd_amd.frontends import lumina_window_spec, lumina_prompt
    import sjd_amd.synthetic as synthetic
    sd = rank if prompt_index is None else prompt_index
    grid = 48
    if args.model == "emu3_8b":
        from sjd_amd.frontends import emu3_window_spec
        from sjd_amd.grammar import Emu3Grammar
        tok = dict(img_token=151851, eoi_token=151853, eos_token=151850, eol_token=151846, eof_token=151847, pad_token=151643)
        Hh = Ww = 90                                        # 720x720 / 8 (reference test_emu3.py:121-122)
        pos = synthetic.synthetic_prompt(63, 1234 + sd, lo=1000, hi=150000)[0].tolist() + [tok["img_token"]]
        neg = synthetic.synthetic_prompt(11, 4321 + sd, lo=1000, hi=150000)[0].tolist() + [tok["img_token"]]
        spec = emu3_window_spec(pos, neg, tok["pad_token"], device)
        prompt = spec.first_tokens[0].tolist()
        P, n_img = len(prompt), (Ww + 1) * Hh + 2
        grammar = Emu3Grammar(Hh, Ww, 151854, 32768, top_k=2048, **tok)
        cfg = SJDConfig(jacobi_loop_interval_l=1, jacobi_loop_interval_r=Hh * Ww - 1, max_num_new_tokens=args.window,
                        guidance_scale=3.0, seed=1234 + sd, prefix_token_sampler_scheme="speculative_jacobi",
                        max_length=P + n_img + 1, eos_token_ids=(tok["eos_token"],))
        workload = f"Emu3-Gen 8B architecture 720x720 (90x91 visual tokens), pos/neg prompt CFG 3.0, top-k 2048, draft window {args.window}, fp16"
    elif args.model == "anole7b":
        from sjd_amd.grammar import AnoleGrammar
        P, n_img = 64, 1024 + 1                              # 512x512 -> 32x32 VQ tokens + <eoi>; no line tokens (config 5)
        prompt = synthetic.synthetic_prompt(P - 1, 1234 + sd, lo=9000, hi=60000)[0].tolist() + [8197]
        spec = lumina_window_spec(prompt, device)
        grammar = AnoleGrammar(margs.vocab_size, P, P + n_img, 1024)
        cfg = SJDConfig(jacobi_loop_interval_l=0, jacobi_loop_interval_r=1024 - args.window - 2, max_num_new_tokens=args.window,
                        guidance_scale=3.0, seed=1234 + sd, prefix_token_sampler_scheme="speculative_jacobi",
                        max_length=P + n_img, eos_token_ids=(8196,))
        workload = (f"Anole/Chameleon-7B architecture 512x512 (1024 image tokens, image-only grammar), draft window {args.window}, CFG 3.0, "
                    f"top-k 2000, bf16")
    else:
        P = 64
        n_img = grid * (grid + 1)
        prompt = lumina_prompt(P, grid, grid, seed=1234 + sd)
        spec = lumina_window_spec(prompt, device)
        grammar = LuminaGrammar(2000, 10)
        cfg = SJDConfig(jacobi_loop_interval_l=0, jacobi_loop_interval_r=grid * grid + grid - 10 - 3,
                        max_num_new_tokens=args.window, guidance_scale=3.0, seed=1234 + sd,
                        prefix_token_sampler_scheme="speculative_jacobi", max_length=P + n_img + 1, eos_token_ids=(8196,))
        workload = (f"{'Lumina-mGPT-7B' if args.model == 'lumina7b' else args.model} 768x768, 1 prompt/GPU, draft window {args.window}, "
                    f"CFG 3.0 (batch 2), top-k 2000, bf16")
    return dict(prompt=prompt, spec=spec, grammar=grammar, cfg=cfg, P=P, n_img=n_img, grid=grid, workload=workload, tau_est=2.3)


def roofline_blocks(args, prof, prof_g1, pair=None):
    """-> (roofline, roofline_k1 or None) from the live HIP-event measurements.  `traffic` is NOT measured in this run: it is the PMC figure
    of the committed rocprofv3 --pmc pass at the same shapes (profiles/*_traffic.json), labelled as such, or null."""
    peak = 8000.0

    def traffic_of(fname):
        if args.model != "lumina7b":
            return None, None                   # the committed PMC traffic files were collected at the Lumina-7B shapes
        tpath = os.path.join(ROOT, "profiles", fname)
        try:
            t = json.load(open(tpath))
            return t.get("hbm_bytes_per_launch"), f"profiles/{fname}: {t.get('source', 'rocprofv3 --pmc pass')} -- the builder's pass at these shapes, not this run"
        except Exception:
            return None, None

    def rocprof_avg():
        """average G1 launch duration of the committed rocprofv3 --kernel-trace --stats pass over the bench decode (tools/profile_round.sh ->
        tools/make_traffic_json.py -> profiles/rocprof_g1.json): must agree with this run's graph-pass `avg_us`"""
        if args.model != "lumina7b":
            return None, None
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", "rocprof_g1.json")))
            return t.get("avg_us_per_g1_launch"), t.get("source")
        except Exception:
            return None, None

    k1_block = g1_block = None
    if prof is not None:
        tr, src = traffic_of("k1_traffic.json")
        k1_block = {"kernel": prof.get("kernel", "k1_partial (draft-window attention)"), "bound": "hbm", "achieved": round(prof["gbps"], 1),
                    "peak": peak, "unit": "GB/s", "frac": round(prof["gbps"] / peak, 4), "traffic": tr, "traffic_source": src,
                    "avg_us": round(prof["avg_ms"] * 1e3, 2), "avg_bytes": int(prof["avg_bytes"]),
                    "avg_kv_rows": round(prof["avg_kv_rows"], 1), "launches": prof["launches"],
                    "kernel_time_source": "eager launches, one HIP event pair per launch (or per batch of back-to-back launches) on the launch "
                                          "stream: includes the ~5 us launch latency of an eager launch; `pair_in_graph` is what a step pays"}
        if pair is not None:
            k1_block["pair_in_graph"] = pair
    if prof_g1 is not None:       # the dominant kernel by time (~70 % of an iteration)
        g1_name = ("g1_skinny_gemm x3 + g1_gateup_silu (weight-streaming window projections, gate|up with SiLU*up as its epilogue; 128 launches / iteration)"
                   if prof_g1.get("fused_mlp") else "g1_skinny_gemm (weight-streaming window projections, 128 launches / iteration)")
        z = prof_g1.get("compressed_launches", 0) > 0
        if z:
            g1_name = g1_name.replace("g1_skinny_gemm", "g1z_skinny_gemm").replace("g1_gateup_silu", "g1z_gateup_silu")
        tr, src = traffic_of("g1z_traffic.json" if z else "g1_traffic.json")
        rows_g1 = prof_g1.get("rows", 32)
        if not z and rows_g1 > 32:
            # windows of more than 32 rows on the uncompressed stream run on kernel G1w (round 6): its OWN PMC passes at the product launch shapes of
            # 128 and 256 rows (profiles/g1w_traffic.json, tools/_r6_g1w_traffic.sh); other row counts have no committed pass: null, not another kernel's
            g1_name = g1_name.replace("g1_skinny_gemm", "g1_wide (kernel G1w)")
            tr = src = None
            if args.model == "lumina7b" and rows_g1 in (128, 256):
                try:
                    t = json.load(open(os.path.join(ROOT, "profiles", "g1w_traffic.json")))
                    tr = t[f"rows_{rows_g1}"]["hbm_bytes_per_launch"]
                    src = f"profiles/g1w_traffic.json (rows_{rows_g1}): {t.get('source')} -- the builder's pass at these shapes, not this run"
                except Exception:
                    tr = src = None
        # `achieved` = ALGORITHMIC bytes per launch (the bf16 weight matrix N*K*2 + the activation rows) / the measured launch time.
        # With the lossless 12-bit stream the kernel MOVES fewer bytes than that (`traffic`, `stored_bytes`): the rate on the bytes actually
        # moved -- what the HBM pipe sees -- is reported next to it (`hbm_GBps_on_stored_bytes`, `frac_on_stored_bytes`).
        alg_b = prof_g1.get("avg_bytes_bf16", prof_g1["avg_bytes"])
        alg_gbps = prof_g1.get("gbps_bf16_equivalent", prof_g1["gbps"])
        g1_block = {"kernel": g1_name, "bound": "hbm", "achieved": round(alg_gbps, 1), "peak": peak, "unit": "GB/s",
                    "frac": round(alg_gbps / peak, 4), "traffic": tr, "traffic_source": src, "avg_us": round(prof_g1["avg_ms"] * 1e3, 2),
                    "avg_bytes": int(alg_b), "launches": prof_g1["launches"], "rows": prof_g1.get("rows", 32),
                    "kernel_time_source": "one pass over all layers' packed weights captured in a hipGraph (the way the engine launches them), "
                                          "replays timed with HIP events on the replay stream; avg over the four projection shapes"}
        ra, rsrc = rocprof_avg()
        if ra is not None and prof_g1.get("rows", 32) == 32:
            g1_block.update({"rocprof_avg_us": ra, "rocprof_source": rsrc})
        if z:
            g1_block.update({"weight_stream": f"lossless 12-bit (G1z / G1sz; {prof_g1['compressed_launches']} of {prof_g1['launches']} launches; "
                                              "results bit-identical to the bf16 stream): `achieved` prices the bf16 bytes of SURVEY.md 8(d), "
                                              "the kernel moves `stored_bytes`",
                             "pack": prof_g1.get("pack"),       # (round 6: units that travel verbatim -- `raw_units` -- and exceptions per unit; synthetic Gaussians: 0 raw)
                             "stored_bytes": int(prof_g1["avg_bytes"]),
                             "hbm_GBps_on_stored_bytes": round(prof_g1["gbps"], 1),
                             "frac_on_stored_bytes": round(prof_g1["gbps"] / peak, 4)})
    return (g1_block, k1_block) if g1_block is not None else (k1_block, None)


def other_config(base_args, model_name, window, device, steps=64, warmup=8, dtype=None, kv="auto"):
    """Compact record of another BASELINE.json configuration measured in THIS run (configs 3 and 5 next to the headline's config 2):
    the model is built, decoded through a real lead-in to its mean KV length, `steps` SJD iterations are timed, G1 and K1 are measured
    with HIP events exactly as for the headline; then everything is freed."""
    import copy
    import gc
    import torch
    a = copy.copy(base_args)
    a.model, a.window, a.dtype, a.kv, a.prompts_per_gpu, a.n_split = model_name, window, dtype, kv, 1, 0
    t0 = time.perf_counter()
    from sjd_amd.engine import SJDEngine
    import sjd_amd.ops as ops_
    model, margs, attn = build_model(a, device)
    w = workload_of(a, margs, 0, device)
    P, n_img = w["P"], w["n_img"]
    fp8_kv = kv == "fp8" or (kv == "auto" and model_name == "anole7b")
    model.setup_cache(batch=2, s_max=((P + n_img + 2 * window + 64 + 31) // 32) * 32, dtype=ops_.FP8 if fp8_kv else None)
    eng = SJDEngine(model, margs.vocab_size, device, max_window=window, use_graph=not a.no_graph)
    lead = int(P + n_img // 2 - w["tau_est"] * (steps / 2.0 + warmup))
    sync = torch.cuda.synchronize
    if not a.no_graph:
        # untimed, as for the headline: the hipGraphs of both K1 regimes x both probability-buffer parities are captured before the clock starts.  (Until
        # late round 6 a key first met inside the 64 timed steps was captured there: ~13 ms, +5 % on a 0.26 s window -- the legs read 3.83 or 3.98 ms
        # for Emu3 in bf16 depending on where the window happened to start.)
        pin0 = getattr(attn, "_pin_regime", None)
        for pin in ("keysplit", "colsplit"):
            if hasattr(attn, "_pin_regime"):
                attn._pin_regime = pin
            eng.decode(w["prompt"], w["spec"], copy.deepcopy(w["grammar"]), w["cfg"], warmup_iters=0, timed_iters=warmup + 8)
        if hasattr(attn, "_pin_regime"):
            attn._pin_regime = pin0
        sync()
    seq, st = eng.decode(w["prompt"], w["spec"], w["grammar"], w["cfg"], warmup_iters=warmup, timed_iters=steps, on_timed_start=sync,
                         on_timed_end=sync, lead_in_kv=lead if lead > P + window else None)
    prof = measure_k1(a, model, attn, device, kv_len=(st.kv_len_start + st.kv_len) // 2)
    pair = measure_k1_pair(a, model, attn, device, kv_len=(st.kv_len_start + st.kv_len) // 2)
    prof_g1 = measure_g1(a, model, device)
    r, rk1 = roofline_blocks(a, prof, prof_g1, pair)
    keep = ("kernel", "achieved", "frac", "avg_us", "avg_bytes", "launches", "avg_kv_rows", "stored_bytes", "frac_on_stored_bytes", "rows", "pair_in_graph")
    wl = w["workload"]
    if dtype == "bf16" and wl.endswith(", fp16"):
        wl = wl[:-6] + ", bf16 (the dtype the reference's test_emu3.py:27 loads the model in)"
    out = {"workload": wl + (", fp8 (e4m3) KV cache + fp8-MFMA draft attention" if fp8_kv else ""),
           "dtype": "fp16" if model.lm_head.weight.dtype == torch.float16 else "bf16", "steps": st.timed_nfe,
           "ms_per_step": round(st.seconds / max(st.timed_nfe, 1) * 1e3, 4), "tokens_per_step": round(st.tokens / max(st.timed_nfe, 1), 4),
           "tokens_per_s": round(st.tokens / max(st.seconds, 1e-9), 2), "kv_len": [st.kv_len_start, st.kv_len],
           "timed_region_reached": bool(st.timed_region_reached),
           "roofline": {k: r[k] for k in keep if r and k in r}, "roofline_k1": {k: rk1[k] for k in keep if rk1 and k in rk1}}
    del eng, model, attn, seq
    gc.collect()
    torch.cuda.empty_cache()
    out["wall_s"] = round(time.perf_counter() - t0, 1)
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))
    if args.tunableop:
        os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
        os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")
        os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/sjd_tunableop_%d.csv")
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from sjd_amd.parallel import gather_report, pin_to_gpu_numa_node, run_prompt_queue
    pinned = None
    if world > 1 and not args.no_pin:           # eight host loops, one sync every ~3 ms each: keep every rank on the cores next to its GPU
        pinned = pin_to_gpu_numa_node(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), local_rank)
    if world > 1 or os.environ.get("SJD_FORCE_DIST") == "1":
        dist.init_process_group(backend="nccl", device_id=device)          # RCCL on ROCm

    from sjd_amd.engine import SJDEngine
    import sjd_amd.synthetic as synthetic

    model, margs, attn = build_model(args, device)
    w = workload_of(args, margs, rank, device)
    prompt, spec, grammar, cfg, P, n_img, grid, workload = (w[k] for k in ("prompt", "spec", "grammar", "cfg", "P", "n_img", "grid", "workload"))
    tau_est = w["tau_est"]                         # accepted tokens / step used only to place the lead-in
    s_max = ((P + n_img + 2 * args.window + 64 + 31) // 32) * 32
    fp8_kv = args.kv == "fp8" or (args.kv == "auto" and args.model == "anole7b")
    import sjd_amd.ops as ops_
    if fp8_kv:
        workload += ", fp8 (e4m3) KV cache + fp8-MFMA draft attention"
    PP = args.prompts_per_gpu
    model.setup_cache(batch=2 * PP, s_max=s_max, dtype=ops_.FP8 if fp8_kv else None)
    if PP > 1:
        if args.model != "lumina7b" and args.model != "lumina_tiny":
            raise SystemExit("--prompts-per-gpu > 1 is wired for the Lumina workload")
        from sjd_amd.engine_batch import SJDBatchEngine
        from sjd_amd.frontends import lumina_window_spec, lumina_prompt
        eng = SJDBatchEngine(model, margs.vocab_size, device, PP, max_window=args.window, use_graph=not args.no_graph)
        NQ = max(PP, args.queue_prompts)
        prompts = [lumina_prompt(P, grid, grid, seed=1234 + rank * NQ + i) for i in range(NQ)]
        specs = [lumina_window_spec(p_, device) for p_ in prompts]
        workload += f", {PP} prompts per GPU sharing one window forward"
        if NQ > PP:
            workload += f" (continuous batching over a queue of {NQ} prompts)"
    else:
        eng = SJDEngine(model, margs.vocab_size, device, max_window=args.window, use_graph=not args.no_graph)

    def sync_all():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    import copy
    grammar0 = copy.deepcopy(grammar)              # pristine grammar for the side legs

    # ---- config 4's shape: a queue of M prompts over the N ranks (the reference's fan-out), whole images, ONE all_gather
    if args.total_prompts > 0:
        if PP > 1:
            raise SystemExit("--total-prompts decodes one prompt at a time per GPU (use --queue-prompts with --prompts-per-gpu)")
        eng.decode(prompt, spec, copy.deepcopy(grammar0), cfg, warmup_iters=0, timed_iters=args.warmup + 8)      # untimed: graph captures, allocations
        per_prompt = []

        def decode_one(i):
            wi = workload_of(args, margs, rank, device, prompt_index=i)
            seq_i, st_i = eng.decode(wi["prompt"], wi["spec"], wi["grammar"], wi["cfg"])
            per_prompt.append((i, st_i.total_tokens, st_i.nfe, bool(seq_i[-1] in wi["cfg"].eos_token_ids)))
            return st_i.total_tokens, st_i.nfe

        q = run_prompt_queue(args.total_prompts, decode_one, sync=torch.cuda.synchronize, device=device)
        if rank == 0:
            out = {"metric": "accepted image-tokens/s (SJD, whole images, prompt queue over the ranks; BASELINE.json config 4's shape)",
                   "value": round(q["tokens_per_s"], 2), "unit": "image-tokens/s", "n_gpus": world, "steps": int(q["steps"]), "warmup": args.warmup,
                   "ms_per_step": round(q["seconds"] / max(max(r[1] for r in q["per_rank"]), 1) * 1e3, 4), "higher_is_better": True,
                   "scaling": "strong", "vs_baseline": None, "dtype": "fp16" if model.lm_head.weight.dtype == torch.float16 else "bf16",
                   "data": "synthetic", "tokens_per_step": round(q["tokens_per_step"], 4),
                   "config": {"workload": workload + f", queue of {args.total_prompts} prompts split contiguously over {world} rank(s), whole images, "
                                          f"random-init synthetic weights (embed_token_scale={args.embed_token_scale})",
                              "prompts": args.total_prompts, "parallelism": f"prompt-parallel x{world}", "prompt_len": P, "image_tokens": n_img},
                   "per_rank": [{"tokens": int(r[0]), "steps": int(r[1]), "seconds": round(r[2], 4)} for r in q["per_rank"]],
                   "rank0_prompts": [{"prompt": i, "tokens": t, "nfe": n, "finished": f} for i, t, n, f in per_prompt],
                   "cpu_affinity_rank0": (f"{len(pinned)} cores of the GPU's NUMA node" if pinned else None)}
            _print_line(out)
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    # ---- where the timed region sits
    kv_center = (P + n_img // 2) if args.kv_center < 0 else args.kv_center
    lead_in_kv = None
    if kv_center > 0 and PP == 1:
        lead_in_kv = int(kv_center - tau_est * (args.steps / 2.0 + args.warmup))
        if lead_in_kv <= P + args.window:
            lead_in_kv = None
    whole_image = (PP == 1 and not args.no_whole_image)      # every rank decodes on to the end of ITS image: `value` is the whole-image rate
    iter_log = []
    capture_s = 0.0
    if whole_image and not args.no_graph:
        # untimed, as in the queue leg above: the image's hipGraphs -- both K1 regimes x both probability-buffer parities -- are captured before
        # the clock starts (an engine serves many images with the graphs of its first; each capture is an eager iteration + a recording, ~13 ms)
        pin0 = getattr(attn, "_pin_regime", None)
        torch.cuda.synchronize()
        t_cap0 = time.perf_counter()
        for pin in ("keysplit", "colsplit"):
            if hasattr(attn, "_pin_regime"):
                attn._pin_regime = pin
            eng.decode(prompt, spec, copy.deepcopy(grammar0), cfg, warmup_iters=0, timed_iters=args.warmup + 8)
        if hasattr(attn, "_pin_regime"):
            attn._pin_regime = pin0
        torch.cuda.synchronize()
        capture_s = time.perf_counter() - t_cap0
    t_wall0 = time.perf_counter()
    if PP > 1:
        res = eng.decode_many(prompts, specs, [copy.deepcopy(grammar) for _ in range(len(prompts))], cfg, warmup_iters=args.warmup,
                              timed_iters=args.steps, on_timed_start=sync_all, on_timed_end=sync_all)
        seq, stats = res[0]
        rs = eng.run_stats                                       # all slots of this GPU; steps = shared window forwards
        stats.tokens, stats.timed_nfe, stats.seconds = rs["tokens"], rs["timed_iterations"], rs["seconds"]
        stats.host_seconds, stats.sync_seconds = rs["host_seconds"], rs["sync_seconds"]
        stats.timed_host_seconds, stats.timed_sync_seconds = rs["host_seconds"], rs["sync_seconds"]
        stats.kv_len_start = P
    else:
        seq, stats = eng.decode(prompt, spec, grammar, cfg, warmup_iters=args.warmup, timed_iters=args.steps, on_timed_start=sync_all,
                                on_timed_end=sync_all, lead_in_kv=lead_in_kv, continue_after=whole_image, iter_log=iter_log)
        if not stats.timed_region_reached:          # the image ended before the region opened: keep the other ranks' two barriers matched
            sync_all()
            sync_all()
    decode_wall = time.perf_counter() - t_wall0
    kv_mid = (stats.kv_len_start + stats.kv_len) // 2
    prof = measure_k1(args, model, attn, device, kv_len=kv_mid)
    pair = measure_k1_pair(args, model, attn, device, kv_len=kv_mid) if args.prompts_per_gpu == 1 else None
    prof_g1 = measure_g1(args, model, device) if (args.gemm == "sjd" and not args.no_fused) else None
    # whole image of this rank: from the end of the prefill iteration to the last iteration (host clock after each iteration's sync)
    img_tok = img_nfe = 0
    img_s = 0.0
    if whole_image and len(iter_log) > 1:
        img_tok, img_nfe, img_s = stats.total_tokens - 1, stats.nfe - 1, iter_log[-1][3] - iter_log[0][3]
    rep = gather_report(stats.tokens, stats.timed_nfe, stats.seconds, device, extra=(img_tok, img_nfe, img_s))   # ONE RCCL all_gather (48 B/rank)
    if rank != 0:
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    tot_tokens = sum(r[0] for r in rep)
    tot_steps = sum(r[1] for r in rep)
    t_max = max(r[2] for r in rep)
    tps_window = tot_tokens / t_max
    tok_per_step = tot_tokens / max(tot_steps, 1)
    dt_name = "fp16" if model.lm_head.weight.dtype == torch.float16 else "bf16"
    have_img = whole_image and all(r[5] > 0 for r in rep)
    tps_image = sum(r[3] for r in rep) / max(r[5] for r in rep) if have_img else None
    out = {
        "metric": ("accepted image-tokens/s (SJD, Emu3 720px, BASELINE.json config 3)" if args.model == "emu3_8b"
                   else "accepted image-tokens/s (SJD, Anole/Chameleon-7B 512px, fp8 draft attention, BASELINE.json config 5)" if args.model == "anole7b"
                   else "accepted image-tokens/s (SJD, Lumina-mGPT-7B 768px); tokens_per_step = 1/steps-to-converge rate"),
        # value: every rank's WHOLE image (all its accepted tokens over the slowest rank's decode time) -- the K timed steps give
        # ms_per_step, which does not depend on the acceptance luck of a short window; their tokens/s is value_window
        "value": round(tps_image if have_img else tps_window, 2), "unit": "image-tokens/s",
        "value_basis": ("whole image(s): sum over ranks of accepted tokens / slowest rank's decode time (prefill iteration excluded; hipGraphs captured in an untimed warm-up decode)" if have_img
                        else "the timed steps"),
        "value_window": round(tps_window, 2), "n_gpus": world, "steps": stats.timed_nfe, "warmup": args.warmup,
        "ms_per_step": round(t_max / max(stats.timed_nfe, 1) * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dt_name, "data": "synthetic",
        "tokens_per_step": round(tok_per_step, 4),
        "host_ms_per_step": round(getattr(stats, "timed_host_seconds", stats.host_seconds) / max(stats.timed_nfe, 1) * 1e3, 4),
        "sync_wait_ms_per_step": round(getattr(stats, "timed_sync_seconds", stats.sync_seconds) / max(stats.timed_nfe, 1) * 1e3, 4),
        "config": {"workload": workload + f", random-init synthetic weights (embed_token_scale={args.embed_token_scale})",
                   "prompt_len": P, "image_tokens": n_img, "kv_len_start": stats.kv_len_start, "kv_len_end": stats.kv_len,
                   "lead_in_steps": 0,
                   "prompts": world * args.prompts_per_gpu, "parallelism": f"prompt-parallel x{world}"},
    }
    if have_img:
        out["whole_image_per_rank"] = [{"tokens": int(r[3]) + 1, "nfe": int(r[4]) + 1, "seconds": round(r[5], 4)} for r in rep]
    if pinned:
        out["cpu_affinity_rank0"] = f"{len(pinned)} cores of the GPU's NUMA node"
    if iter_log:       # untimed real SJD iterations before the W warm-up steps (kv_len grows strictly, so the region's first step is unique)
        i_start = next((i for i, r in enumerate(iter_log) if r[0] == stats.kv_len_start), args.warmup)
        out["config"]["lead_in_steps"] = max(0, i_start - args.warmup)
    if whole_image and iter_log:
        # the SAME decode, continued to the end of the image: measured steps-to-converge and whole-image rate (host wall clock of the
        # whole decode, which includes the prefill iteration, the hipGraph captures and the two barrier brackets)
        n_tok, nfe = stats.total_tokens, stats.nfe
        t_img = iter_log[-1][3] - iter_log[0][3]                   # from the end of the prefill iteration to the last iteration
        durs = sorted(iter_log[i][3] - iter_log[i - 1][3] for i in range(1, len(iter_log)))
        cut = durs[:max(1, len(durs) - max(8, len(durs) // 100))]  # steady state: without the ~1 % slowest iterations, i.e. the one-off
        steady = sum(cut) / len(cut)                               # hipGraph captures and the two barrier brackets of the timed region
        out["whole_image"] = {"tokens": n_tok, "nfe": nfe, "tokens_per_step": round(n_tok / max(nfe, 1), 4),
                              "seconds": round(t_img, 4), "ms_per_step": round(t_img / max(nfe - 1, 1) * 1e3, 4),
                              "tokens_per_s": round((n_tok - 1) / t_img, 2),
                              "steady_ms_per_step": round(steady * 1e3, 4), "steady_tokens_per_s": round((n_tok - 1) / (steady * (nfe - 1)), 2),
                              "capture_seconds": round(capture_s, 4),        # the untimed warm-up decodes that captured the image's hipGraphs
                              "tokens_per_s_incl_capture": round((n_tok - 1) / (t_img + capture_s), 2),
                              "finished": bool(seq[-1] in cfg.eos_token_ids),
                              "reference_published_nfe": "1009-1115 (hardware unstated, BASELINE.md)" if args.model == "lumina7b" else None}
        pts = [P, P + n_img // 2 - 24, P + n_img - 112] if args.model != "lumina7b" else [64, 1216, 2368]
        out["per_kv"] = per_kv_table(iter_log, pts)
        for S in pts:
            if str(S) in out["per_kv"]:
                k1 = measure_k1(args, model, attn, device, kv_len=S)
                out["per_kv"][str(S)].update({"k1_us": round(k1["avg_ms"] * 1e3, 2), "k1_GBps": round(k1["gbps"], 1)})

        # (rounds 1-3 counted the hipGraph captures inside `value`; the old basis stays readable next to the new one)
        out["value_incl_capture"] = out["whole_image"]["tokens_per_s_incl_capture"] if world == 1 else None
    out["schema"] = 5           # 5: value_incl_capture, roofline.rocprof_avg_us, torch_baseline over >= 64 steps
    r_main, r_k1 = roofline_blocks(args, prof, prof_g1, pair)
    if r_main is not None:
        out["roofline"] = r_main
    if r_k1 is not None:
        out["roofline_k1"] = r_k1
    side_legs = (world == 1 and PP == 1)
    gpu_sched_ms = None
    if side_legs and not args.no_graph:
        try:
            gpu_sched_ms = measure_scheduler_gpu(eng, grammar0, prompt, args.window)
        except Exception:
            gpu_sched_ms = None
    if side_legs and not args.no_ar_baseline and args.model in ("lumina7b", "lumina_tiny"):
        # (measured right behind the headline, before the other configurations and side legs: a leg's position in a three-minute run moves it by 1-3 %,
        #  and this one is a ratio against the headline's own step time)
        # the reference's AUTOREGRESSIVE baseline on the same kernels (round 6; IS:417-450 + HF _sample, one token per forward: a ONE-row window whose
        # uncond row sees the prompt from the image-start token on -- sjd_amd.inference_solver.FlexARInferenceSolver without renew_pipeline_sampler,
        # pinned on reference runs by tests/test_oracle_golden.py::test_loop_lumina_autoregressive_baseline).  Decoded from the engine's own accepted
        # sequence up to the start of the timed region: the same KV length as the headline.  `value_window` / its tokens/s = the AR-vs-SJD ratio
        # the reference publishes (2.05-2.16 x for Lumina-mGPT), here on one MI355X.
        from sjd_amd.engine import WindowSpec
        ctx = seq[:stats.kv_len_start + 1] if stats.kv_len_start >= P else list(prompt)
        u0, Pc = P - 3, len(ctx)
        spec_ar = WindowSpec(first_tokens=torch.tensor([ctx, ctx], dtype=torch.long, device=device),
                             first_positions=torch.stack([torch.arange(Pc), torch.tensor([1] * u0 + list(range(Pc - u0)))]).to(device),
                             key_start=torch.tensor([0, u0], dtype=torch.int32), pos_offset=torch.tensor([0, -u0], dtype=torch.long), kv_base=0)
        cfg_ar = copy.copy(cfg)
        cfg_ar.jacobi_loop_interval_l, cfg_ar.jacobi_loop_interval_r, cfg_ar.max_num_new_tokens = 1, 1 << 20, 1
        eng_ar = SJDEngine(model, margs.vocab_size, device, max_window=1, use_graph=not args.no_graph)
        _, st_a = eng_ar.decode(ctx, spec_ar, copy.deepcopy(grammar0), cfg_ar, warmup_iters=8, timed_iters=args.ar_steps, on_timed_start=sync_all,
                                on_timed_end=sync_all)
        ms_a = st_a.seconds / max(st_a.timed_nfe, 1) * 1e3
        out["ar_baseline"] = {"ms_per_step": round(ms_a, 4), "tokens_per_step": 1.0, "tokens_per_s": round(1e3 / ms_a, 2), "steps": st_a.timed_nfe,
                              "kv_len": [st_a.kv_len_start, st_a.kv_len],
                              "sjd_speedup": round(tps_window / (1e3 / ms_a), 3), "sjd_step_reduction": round(tok_per_step, 3),
                              "what": "autoregressive decoding (window 1) on the same kernels, weights and KV length; sjd_speedup = value_window / tokens_per_s "
                                      "(the reference publishes 2.05-2.16 x latency and ~2.3 x steps for Lumina-mGPT, hardware unstated)"}
        eng.reset_graphs()              # (the two engines share the backbone's workspaces: the headline engine re-captures if it runs again)
    if side_legs and not args.no_other_configs and args.model == "lumina7b":
        # BASELINE.json configs 3 and 5 in the same driver-visible line (bounded: ~64 timed steps each after a real lead-in)
        # Round 6: measured BEFORE the headline's own side legs, with the headline engine kept alive (288 GB: both models fit).  Behind the side legs the
        # same legs read 1-3.5 % slower (Emu3 bf16 3.84-3.87 ms alone or right behind the headline, 3.98 at the end of a default run; each side leg adds
        # a little -- profiles/r6_other_configs_leg_ab.txt): a three-minute run warms the part, and these are the BASELINE configurations.
        torch.cuda.empty_cache()
        out["other_configs"] = {}
        # (config 3 twice: fp16 as BASELINE.json words it, and bf16 -- what the reference's own test_emu3.py:27 runs -- where the lossless
        #  12-bit weight stream G1z / G1sz applies)
        # (config 5 twice as well: the fp8 KV cache BASELINE.json names, and the same workload on the bf16 cache -- the reference's own precision,
        #  JA:137-272 / MC:567 -- so that what the fp8 path buys or costs in tokens/s sits in one driver-run line)
        for key, name, win, dt_, kv_ in (("emu3_8b", "emu3_8b", 32, None, "auto"), ("emu3_8b_bf16", "emu3_8b", 32, "bf16", "auto"),
                                         ("anole7b", "anole7b", 16, None, "fp8"), ("anole7b_bf16kv", "anole7b", 16, None, "16bit")):
            try:
                out["other_configs"][key] = other_config(args, name, win, device, dtype=dt_, kv=kv_)
            except Exception as e:       # a side leg must not cost the headline line
                out["other_configs"][key] = {"error": repr(e)[:300]}
    if side_legs and not args.no_floor:
        # floor regime: plain random embeddings -> the next-token distribution depends almost only on the previous
        # token -> ~1 accepted token per step.  Same engine, same graphs (the embedding table is re-drawn in place).
        e_before = model.model.embed_tokens.weight[:8, :8].float().clone()
        synthetic.refill_embeddings_device(model, seed=0, embed_token_scale=1.0)
        redrawn = not torch.equal(e_before, model.model.embed_tokens.weight[:8, :8].float())
        cfg_f = copy.copy(cfg)
        _, st_f = eng.decode(prompt, spec, copy.deepcopy(grammar0), cfg_f, warmup_iters=8, timed_iters=args.floor_steps,
                             on_timed_start=sync_all, on_timed_end=sync_all)
        synthetic.refill_embeddings_device(model, seed=0, embed_token_scale=args.embed_token_scale)
        tpf = st_f.tokens / max(st_f.timed_nfe, 1)
        # (round 3 called this leg `floor`; it measures 2.2 tokens/step, so it is an acceptance-SENSITIVITY probe, not a floor)
        out["acceptance_probe_unscaled_embeddings"] = {
            "embed_token_scale": 1.0, "embeddings_redrawn": redrawn, "tokens_per_step": round(tpf, 4), "steps": st_f.timed_nfe,
            "ms_per_step": round(st_f.seconds / max(st_f.timed_nfe, 1) * 1e3, 4), "kv_len": [st_f.kv_len_start, st_f.kv_len],
            "tokens_per_s_at_headline_ms_per_step": round(tpf / (t_max / max(stats.timed_nfe, 1)), 2)}
    if side_legs:
        # the floor of the metric: SJD accepts at least ONE token per step whatever the weights are (the first draft is always verified against its
        # own distribution), so ms_per_step -- the only hardware fact on this line -- bounds tokens/s from below at 1 token/step
        ms_h = t_max / max(stats.timed_nfe, 1) * 1e3
        out["floor"] = {"tokens_per_step": 1.0, "tokens_per_s": round(1e3 / ms_h, 2),
                        "what": "1 / ms_per_step: the rate at the algorithm's minimum acceptance of one token per step (= plain AR decoding at this step time); "
                                "`value` / this = the measured tokens per step"}
    if side_legs and not args.no_torch_baseline and args.model in ("lumina7b", "lumina_tiny"):
        # PyTorch-ROCm SJD: the reference's data flow with ATen ops on the SAME weights, prefilled with the engine's
        # own accepted sequence up to the start of the timed region, so that both run at the same KV length
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import torch_sjd_baseline as TB
        ctx = seq[:stats.kv_len_start + 1] if stats.kv_len_start >= P else list(prompt)
        tb = TB.run_on_engine_model(model, ctx, P, grid, args.torch_baseline_steps, 4, seed=1234, window=args.window)
        tb["what"] = ("reference data flow in PyTorch-ROCm ops (torch.cat KV cache, masked SDPA, torch.topk, torch.multinomial, Python "
                      "accept loop with a sync per draft), same weights, same KV length; tools/torch_sjd_baseline.py")
        # Both run the same algorithm on the same weights, so their expected accepted tokens/step are equal; the baseline's own 24-step
        # sample of it is noisy, so the ratio is taken per STEP: PyTorch-ROCm SJD ms/step over engine ms/step at the same KV length
        tb["tokens_per_s_at_engine_acceptance"] = round(tok_per_step / (tb["ms_per_step"] / 1e3), 2)
        out["torch_baseline"] = tb
        out["vs_baseline"] = round(tps_window / tb["tokens_per_s_at_engine_acceptance"], 3)
        out["vs_baseline_raw_tokens_per_s"] = round(tps_window / tb["tokens_per_s"], 3) if tb["tokens_per_s"] > 0 else None
        out["vs_baseline_kind"] = ("value_window / torch_baseline.tokens_per_s_at_engine_acceptance = PyTorch-ROCm SJD ms/step over engine ms/step "
                                   "(same GPU, weights, KV length; the reference publishes no number on stated hardware)")
    if side_legs and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, gpu_sched_ms)
    out["bench_wall_s"] = {"decode": round(decode_wall, 2)}
    if dist.is_initialized():
        dist.destroy_process_group()
    _print_line(out)


def _print_line(out):
    try:                                   # RCCL prints its version banner through C stdio: flush it so that the JSON line is the LAST line
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
