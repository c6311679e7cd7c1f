#!/usr/bin/env python3
"""Where the time of ONE launch goes: in-kernel phase timestamps of k1_partial and g1_skinny_gemm.

Builds csrc/sjd_attention.hip and csrc/sjd_gemm.hip with -DSJD_TRACE (thread 0 of every workgroup records the 100 MHz wall clock at the
phase boundaries marked SJD_TR(i) in the sources) into accelerating-t2i-ar-with-sjd_amd/libsjd_hip_trace.so, re-executes itself with
SJD_HIP_LIB pointing at it, replays the kernels in a hipGraph and prints per-phase means over the workgroups of the LAST launch.
  python tools/phase_trace.py            (on the MI355X box through gpurun; the instrumented build is made on the fly, hipcc is in the image)
This is the tool that found K1's 5.3 us merge epilogue and the serialised second key tile (round 2)."""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "accelerating-t2i-ar-with-sjd_amd")
TRACE_SO = os.path.join(PKG, "libsjd_hip_trace.so")
PER_WG = None


def build():
    csrc = os.path.join(PKG, "csrc")
    subprocess.check_call(["make", "-C", csrc], stdout=subprocess.DEVNULL)
    objs = []
    for f in ("sjd_attention", "sjd_gemm", "sjd_glue", "sjd_sampling"):
        o = os.path.join("/tmp", f + "_trace.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DSJD_TRACE", "-Wno-unused-value", "-mllvm", "-amdgpu-kernarg-preload-count=16"]
                              + os.environ.get("SJD_TRACE_FLAGS", "").split()
                              + (["-ffp-contract=off"] if f == "sjd_sampling" else []) + ["-c", os.path.join(csrc, f + ".hip"), "-o", o])
        objs.append(o)
    rest = [os.path.join(csrc, f + ".o") for f in ("sjd_capi",)]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", TRACE_SO] + objs + rest)


def us(x):
    return round(float(x) * 10e-3, 2)          # 100 MHz ticks -> us


def trace_k1(lib, torch, ops, np):
    dev = torch.device("cuda:0")
    B, n, H, D, layers = 2, 16, 32, 128, 32
    lib.sjd_debug_trace_k1.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for kv in (64, 448, 1216, 2368):
        s_max = ((kv + n + 64 + 31) // 32) * 32
        kc = torch.randn(layers, B, H, s_max, D, device=dev).to(torch.bfloat16)
        vc = torch.randn(layers, B, H, s_max, D, device=dev).to(torch.bfloat16)
        q = torch.randn(B, n, H, D, device=dev).to(torch.bfloat16)
        out = torch.empty_like(q)
        ks = torch.tensor([0, 63], dtype=torch.int32, device=dev)
        ws = ops.attention_workspace(B, H, n, D, 4, dev)
        one = lambda i: ops.draft_window_attention(q, kc[i], vc[i], out, ks, None, kv, 4, ws)
        with torch.cuda.stream(torch.cuda.Stream()):
            one(0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(layers):
                one(i)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        nwg = 4 * H * B
        buf = np.zeros((nwg, 8), dtype=np.uint64)
        assert lib.sjd_debug_trace_k1(buf.ctypes.data, nwg) == 0
        t = buf.astype(np.int64)
        t = t[t[:, 7] > 0]                      # splits beyond the effective count exit before the first stamp that matters
        t0 = t[:, 0].min()
        d = lambda a, b: us((t[:, b] - t[:, a]).mean())
        print(json.dumps(dict(kernel="k1_partial<bf16,128,8>", kv_len=kv, n_split=4, workgroups=int(len(t)),
                              start_skew_us=us((t[:, 0] - t0).max()),
                              phase_us=dict(kv_len_key_start=d(0, 1), first_tile=d(1, 2), key_loop=d(2, 3), wait_for_waves=d(3, 4),
                                            merge_buffers=d(4, 5), merge_publish=d(5, 6), store_ack=d(6, 7)),
                              end_us=dict(mean=us((t[:, 7] - t0).mean()), max=us((t[:, 7] - t0).max())))), flush=True)


def trace_k1_shared(lib, torch, ops, np):
    """Emu3's shape: GQA 32 / 8, draft window 32 (two 16-row chunks), 16 key splits: k1_partial_shared"""
    dev = torch.device("cuda:0")
    B, n, H, Hkv, D, layers, ns = 2, 32, 32, 8, 128, 32, 16
    lib.sjd_debug_trace_k1.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for kv in (1024, 4096, 8192):
        s_max = ((kv + n + 64 + 31) // 32) * 32
        kc = torch.randn(layers, B, Hkv, s_max, D, device=dev).to(torch.float16)
        vc = torch.randn(layers, B, Hkv, s_max, D, device=dev).to(torch.float16)
        q = torch.randn(B, n, H, D, device=dev).to(torch.float16)
        out = torch.empty_like(q)
        ks = torch.tensor([0, 0], dtype=torch.int32, device=dev)
        ws = ops.attention_workspace(B, H, n, D, ns, dev)
        one = lambda i: ops.draft_window_attention(q, kc[i], vc[i], out, ks, None, kv, ns, ws)
        with torch.cuda.stream(torch.cuda.Stream()):
            one(0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(layers):
                one(i)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        nwg = ns * Hkv * B
        buf = np.zeros((nwg, 8), dtype=np.uint64)
        assert lib.sjd_debug_trace_k1(buf.ctypes.data, nwg) == 0
        t = buf.astype(np.int64)
        t = t[t[:, 5] > 0]
        t0 = t[:, 0].min()
        d = lambda a, b: us((t[:, b] - t[:, a]).mean())
        tiles = (kv + n + 31) // 32 / ns
        print(json.dumps(dict(kernel=("k1_partial_shared<f16,128,8 waves>" if os.environ.get("SJD_K1_RING") == "0" else "k1_partial_ring<f16,128,8 waves>"), kv_len=kv, n_split=ns, workgroups=int(len(t)), tiles_per_workgroup=round(tiles, 1),
                              phase_us=dict(tile_ranges=d(0, 1), first_tile=d(1, 2), key_loop=d(2, 3), publish=d(3, 5)),
                              key_loop_us_per_tile=round(d(2, 3) / max(tiles - 1, 1), 2),
                              end_us=dict(mean=us((t[:, 5] - t0).mean()), max=us((t[:, 5] - t0).max())))), flush=True)


def trace_g1(lib, torch, ops, np, z=False):
    """z: the same launches over the 12-bit lossless weight stream (g1z_skinny_gemm; stamp 3 = the unit's header arrived)"""
    import sjd_amd._lib as L
    import sjd_amd.backbones as BB
    dev = torch.device("cuda:0")
    lib.sjd_debug_trace_g1.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for name, (N, K) in dict(qkv=(12288, 4096), o=(4096, 4096), gate_up=(22016, 4096), down=(4096, 11008)).items():
        KC, waves, sm = BB.ChameleonBackbone.G1_CFG[name]
        x = torch.randn(32, K, device=dev).to(torch.bfloat16)
        wps = [(ops.pack_weight_z if z else ops.pack_weight)((torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16), KC, sm) for _ in range(6)]
        nc = (K + KC - 1) // KC
        out = torch.empty(nc, 32, N, dtype=torch.float32, device=dev)

        def g1(i):
            if z:
                L.check(lib.sjd_skinny_gemm_z(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wps[i % 6].data.data_ptr()), ctypes.c_void_p(wps[i % 6].exc.data_ptr()),
                                              wps[i % 6].cap, ctypes.c_void_p(out.data_ptr()), 32, N, K, KC, waves, int(sm), 0, N, 0,
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "g1z")
                return
            L.check(lib.sjd_skinny_gemm(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wps[i % 6].data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                        32, N, K, KC, waves, int(sm), 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "g1")
        with torch.cuda.stream(torch.cuda.Stream()):
            g1(0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(12):
                g1(i)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        nwg = ((N // 32 + waves - 1) // waves) * nc
        buf = np.zeros((nwg, 8), dtype=np.uint64)
        assert lib.sjd_debug_trace_g1(buf.ctypes.data, nwg) == 0
        t = buf.astype(np.int64)
        t0 = t[:, 0].min()
        d = lambda a, b: us((t[:, b] - t[:, a]).mean())
        loop = t[:, 4] - t[:, 3]
        if PER_WG is not None:        # --per-wg FILE: every workgroup's stamps and where it ran (slot 7: XCC_ID << 32 | HW_ID), us from the first entry
            gx = (N // 32 + waves - 1) // waves
            PER_WG.write(json.dumps(dict(shape=name, KC=KC, waves=waves, grid=[gx, nc], wg=[
                dict(x=i % gx, y=i // gx, xcc=int(t[i, 7] >> 32) & 15, cu=int(t[i, 7] >> 8) & 15, sh=int(t[i, 7] >> 12) & 1, se=int(t[i, 7] >> 13) & 7,
                     t=[us(t[i, k] - t0) for k in range(7)]) for i in range(nwg)])) + "\n")
            PER_WG.flush()
        print(json.dumps(dict(kernel=("g1z_skinny_gemm" if z else "g1_skinny_gemm") + "<bf16, 32 rows>", shape=name, KC=KC, waves=waves, workgroups=nwg,
                              start_skew_us=us((t[:, 0] - t0).max()),
                              phase_us=dict(stage_activation=d(0, 1), wait_for_waves=d(1, 2), first_weight_group=d(2, 3), main_loop=d(3, 4),
                                            store_issue=d(4, 5), store_ack=d(5, 6)),
                              main_loop_min_max_us=[us(loop.min()), us(loop.max())],
                              end_us=dict(mean=us((t[:, 6] - t0).mean()), max=us((t[:, 6] - t0).max())))), flush=True)


def trace_g1s(lib, torch, ops, np):
    """G1s: gate|up + SiLU * up in one launch (two staging phases)"""
    dev = torch.device("cuda:0")
    lib.sjd_debug_trace_g1.argtypes = [ctypes.c_void_p, ctypes.c_int]
    I, K = 11008, 4096
    x = torch.randn(32, K, device=dev).to(torch.bfloat16)
    wps = [ops.pack_weight((torch.randn(2 * I, K, device=dev) / K ** 0.5).to(torch.bfloat16), K // 2, True) for _ in range(6)]
    rn = (ops.residual_sumsq(x.clone(), None), K, 1e-5)
    with torch.cuda.stream(torch.cuda.Stream()):
        ops.gateup_silu(x, wps[0], I, K, True, row_norm=rn)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(12):
            ops.gateup_silu(x, wps[i % 6], I, K, True, row_norm=rn)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    nwg = I // 64
    buf = np.zeros((nwg, 8), dtype=np.uint64)
    assert lib.sjd_debug_trace_g1(buf.ctypes.data, nwg) == 0
    t = buf.astype(np.int64)
    t0 = t[:, 0].min()
    d = lambda a, b: us((t[:, b] - t[:, a]).mean())
    print(json.dumps(dict(kernel="g1_gateup_silu<bf16, 64>", workgroups=nwg, us_per_launch_in_graph=round(e0.elapsed_time(e1) * 1000 / 60, 2),
                          phase_us=dict(stage_phase0=d(0, 1), wait_for_waves=d(1, 2), first_weight_group=d(2, 3), phase0_and_restage=d(3, 4),
                                        phase1=d(4, 5), epilogue=d(5, 6)),
                          end_us=dict(mean=us((t[:, 6] - t0).mean()), max=us((t[:, 6] - t0).max())))), flush=True)


def trace_g1_tiled(lib, torch, ops, np):
    """128 window rows (four prompts per forward): g1_skinny_gemm_tiled"""
    import sjd_amd._lib as L
    import sjd_amd.backbones as BB
    dev = torch.device("cuda:0")
    for name, (N, K) in dict(qkv=(12288, 4096), gate_up=(22016, 4096), down=(4096, 11008)).items():
        KC, waves, sm = BB.ChameleonBackbone.G1_CFG_128ROW[name]
        x = torch.randn(128, K, device=dev).to(torch.bfloat16)
        wps = [ops.pack_weight((torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16), KC, sm) for _ in range(6)]
        nc = (K + KC - 1) // KC
        out = torch.empty(nc, 128, N, dtype=torch.float32, device=dev)

        def g1(i):
            L.check(lib.sjd_skinny_gemm(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wps[i % 6].data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                        128, N, K, KC, waves, int(sm), 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "g1")
        with torch.cuda.stream(torch.cuda.Stream()):
            g1(0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(12):
                g1(i)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        nwg = ((N // 32 + waves - 1) // waves) * nc
        buf = np.zeros((nwg, 8), dtype=np.uint64)
        assert lib.sjd_debug_trace_g1(buf.ctypes.data, nwg) == 0
        t = buf.astype(np.int64)
        t = t[(t[:, 7] > 0) & (t[:, 3] > t[:, 0])]            # workgroups with at least two full sub-tiles
        t0 = t[:, 0].min()
        d = lambda a, b: us((t[:, b] - t[:, a]).mean())
        n_sub = min(KC, K) // 256
        print(json.dumps(dict(kernel="g1_skinny_gemm_tiled<bf16, 128 rows>", shape=name, KC=KC, waves=waves, workgroups=int(len(t)), sub_tiles=n_sub,
                              phase_us=dict(first_sub_tile_staged=d(0, 1), first_sub_tile=d(1, 4), loop_total=d(1, 3)),
                              second_sub_tile_us=dict(group_A=d(4, 5), group_B=d(5, 6), stage_next=d(6, 7)),
                              per_sub_tile_us=round(d(1, 3) / max(n_sub, 1), 2), end_us=us((t[:, 3] - t0).max()))), flush=True)


def trace_in_situ(lib, torch, ops, np, kv_target=1216):
    """the LAST launch of every instrumented kernel inside a real decode (Lumina-7B shapes, hipGraph): layer 31 of the last iteration --
    inputs just written by the previous kernel on other XCDs, i.e. what the stage costs where it runs"""
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
    prompt = lumina_prompt(kv_target - 60, grid, grid, seed=5)
    spec = lumina_window_spec(prompt, dev)
    cfg = SJDConfig(jacobi_loop_interval_l=0, jacobi_loop_interval_r=grid * grid + grid - 13, max_num_new_tokens=window, guidance_scale=3.0,
                    seed=5, max_length=len(prompt) + 60, eos_token_ids=(8196,))
    model.setup_cache(batch=2, s_max=((len(prompt) + 60 + 2 * window + 64 + 31) // 32) * 32)
    eng = SJDEngine(model, margs.vocab_size, dev, max_window=window, use_graph=True)
    seq, stats = eng.decode(prompt, spec, LuminaGrammar(2000, 10), cfg)
    torch.cuda.synchronize()
    for f in ("sjd_debug_trace_k1", "sjd_debug_trace_k1c"):
        getattr(lib, f).argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.sjd_debug_trace_glue.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]

    def table(name, buf, cols, labels):
        t = buf[:, :cols].astype(np.int64)
        t = t[(t[:, 0] > 0) & np.all(np.diff(t, axis=1) >= 0, axis=1)]      # workgroups that stamped every phase in THIS launch
        t = t[t[:, 0] >= t[:, 0].max() - 2000]                               # (rows left over from an earlier launch with a bigger grid)
        t0 = t[:, 0].min()
        ph = {labels[i]: us((t[:, i + 1] - t[:, i]).mean()) for i in range(cols - 1)}
        print(json.dumps(dict(kernel=name, in_situ_kv_len=int(stats.kv_len), workgroups=int(len(t)), start_skew_us=us((t[:, 0] - t0).max()),
                              phase_us=ph, end_us=dict(mean=us((t[:, cols - 1] - t0).mean()), max=us((t[:, cols - 1] - t0).max())))), flush=True)

    b = np.zeros((256, 8), dtype=np.uint64)
    assert lib.sjd_debug_trace_k1(b.ctypes.data, 256) == 0
    table("k1_partial (layer 31, in situ)", b, 8, ["kv_len_key_start", "first_tile", "key_loop", "wait_for_waves", "merge_buffers", "merge_publish", "store_ack"])
    b = np.zeros((64, 4), dtype=np.uint64)
    assert lib.sjd_debug_trace_k1c(b.ctypes.data, 64) == 0
    table("k1_combine (layer 31, in situ)", b, 3, ["partials_arrive", "normalise_store"])
    lib.sjd_debug_trace_k2.argtypes = [ctypes.c_void_p, ctypes.c_int]
    b = np.zeros((16, 16), dtype=np.uint64)
    assert lib.sjd_debug_trace_k2(b.ctypes.data, 16) == 0
    table("k2_logits_to_probs_sample (one workgroup per window row, last iteration), in situ", b, 10,
          ["first_batch_rule_stats_state", "zero_outside_window", "head_partials_cfg_mask_stage", "max_and_count", "top_k_select", "exp_and_sum",
           "normalise_store_list", "dense_draw", "argmax_pair"])
    b = np.zeros((33, 16), dtype=np.uint64)
    assert lib.sjd_debug_trace_k2(b.ctypes.data, 33) == 0
    t = b[32, :7].astype(np.int64)
    if t[0] > 0 and np.all(np.diff(t) >= 0):       # (the last iteration rejected a draft: every K4 phase ran)
        labels = ["accept_tests", "residual_row_staged", "count_topk_sum", "list_and_dense_draw", "argmax_token", "mirror_to_host"]
        print(json.dumps(dict(kernel="k4_verify_accept (one workgroup, last iteration, with a rejection), in situ",
                              phase_us={labels[i]: us(t[i + 1] - t[i]) for i in range(6)}, end_us=us(t[6] - t[0]))), flush=True)
    for kind, name, nwg in ((0, "f1r_residual_sumsq (last launch: after down of layer 31, 13 partial planes)", 256),
                            (1, "f2_qknorm_rope_append (layer 31)", 768), (2, "f3_silu_mul (layer 31)", 172)):
        b = np.zeros((nwg, 4), dtype=np.uint64)
        assert lib.sjd_debug_trace_glue(kind, b.ctypes.data, nwg) == 0
        table(name + ", in situ", b, 3, ["inputs_arrive", "compute_store"])


def trace_k2_emu3(lib, torch, ops, np, P=600):
    """K2's phase stamps at Emu3's shape (32 rows x 32768-column windows staged in LDS, top-k 2048), last iteration of a short decode"""
    import sjd_amd.backbones as BB
    import sjd_amd.synthetic as synthetic
    from sjd_amd.engine import SJDEngine, SJDConfig
    from sjd_amd.frontends import emu3_window_spec
    from sjd_amd.grammar import Emu3Grammar
    dev = torch.device("cuda:0")
    margs, window = BB.EMU3_8B, 32
    with torch.device(dev):
        model = BB.ChameleonBackbone(margs, attn=ops.HipWindowAttention()).to(torch.bfloat16).eval()
    model.G1_CFG = dict(model.G1_CFG_EMU3)
    synthetic.fill_state_dict_device(model, seed=0, embed_token_scale=0.7)
    model.enable_fused(ops, gemm="sjd")
    tok = dict(img_token=151851, eoi_token=151853, eos_token=151850, eol_token=151846, eof_token=151847, pad_token=151643)
    pos = synthetic.synthetic_prompt(P - 1, 17, lo=1000, hi=150000)[0].tolist() + [tok["img_token"]]
    neg = synthetic.synthetic_prompt(11, 18, lo=1000, hi=150000)[0].tolist() + [tok["img_token"]]
    spec = emu3_window_spec(pos, neg, tok["pad_token"], dev)
    grammar = Emu3Grammar(90, 90, 151854, 32768, top_k=2048, **tok)
    cfg = SJDConfig(jacobi_loop_interval_l=1, jacobi_loop_interval_r=90 * 90 - 1, max_num_new_tokens=window, guidance_scale=3.0,
                    seed=17, max_length=P + 70, eos_token_ids=(tok["eos_token"],))
    model.setup_cache(batch=2, s_max=1024)
    eng = SJDEngine(model, margs.vocab_size, dev, max_window=window, use_graph=True)
    seq, stats = eng.decode(spec.first_tokens[0].tolist(), spec, grammar, cfg)
    torch.cuda.synchronize()
    lib.sjd_debug_trace_k2.argtypes = [ctypes.c_void_p, ctypes.c_int]
    b = np.zeros((32, 16), dtype=np.uint64)
    assert lib.sjd_debug_trace_k2(b.ctypes.data, 32) == 0
    t = b[:, :10].astype(np.int64)
    t = t[(t[:, 0] > 0) & np.all(np.diff(t, axis=1) >= 0, axis=1)]
    t = t[t[:, 0] >= t[:, 0].max() - 2000]
    labels = ["first_batch_rule_stats_state", "zero_outside_window", "head_partials_cfg_mask_stage", "max_and_count", "top_k_select", "exp_and_sum",
              "normalise_store_list", "dense_draw", "argmax_pair"]
    t0 = t[:, 0].min()
    print(json.dumps(dict(kernel="k2_logits_to_probs_sample at Emu3's shape, in situ (rows that ran every phase; forced rows return early)", workgroups=int(len(t)),
                          phase_us={labels[i]: us((t[:, i + 1] - t[:, i]).mean()) for i in range(9)},
                          end_us=dict(mean=us((t[:, 9] - t0).mean()), max=us((t[:, 9] - t0).max())))), flush=True)


def main():
    if os.environ.get("SJD_HIP_LIB") != TRACE_SO:
        build()
        os.execve(sys.executable, [sys.executable] + sys.argv, dict(os.environ, SJD_HIP_LIB=TRACE_SO))
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import sjd_amd._lib as L
    import sjd_amd.ops as ops
    lib = L.load()
    global PER_WG
    if "--per-wg" in sys.argv:
        PER_WG = open(sys.argv[sys.argv.index("--per-wg") + 1], "w")
        trace_g1(lib, torch, ops, np)
        return
    if "--k2-emu3" in sys.argv:
        return trace_k2_emu3(lib, torch, ops, np)
    if "--k1s" in sys.argv:          # the shared-tile K1 shapes only (round 4: the LDS-DMA ring kernel; SJD_K1_RING=0 the round-3 kernel)
        trace_k1_shared(lib, torch, ops, np)
        return
    if "--g1s" in sys.argv:
        trace_g1s(lib, torch, ops, np)
        return
    if "--in-situ" in sys.argv:
        trace_in_situ(lib, torch, ops, np)
        return
    if "--g1z" in sys.argv:
        trace_g1(lib, torch, ops, np)
        trace_g1(lib, torch, ops, np, z=True)
        return
    trace_k1(lib, torch, ops, np)
    trace_k1_shared(lib, torch, ops, np)
    trace_g1(lib, torch, ops, np)
    trace_g1s(lib, torch, ops, np)
    trace_g1_tiled(lib, torch, ops, np)
    trace_in_situ(lib, torch, ops, np)


if __name__ == "__main__":
    main()
