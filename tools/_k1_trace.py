import ctypes, json, sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import sjd_amd._lib as L
import sjd_amd.ops as ops
lib = L.load()
dev = torch.device("cuda:0")
B, n, H, D, layers = 2, 16, 32, 128, 32
for kv in (64, 448, 1216, 2368):
    s_max = ((kv + n + 64 + 31) // 32) * 32
    kc = torch.randn(layers, B, H, s_max, D, device=dev).to(torch.bfloat16)
    vc = torch.randn(layers, B, H, s_max, D, device=dev).to(torch.bfloat16)
    q = torch.randn(B, n, H, D, device=dev).to(torch.bfloat16)
    out = torch.empty_like(q)
    ks = torch.tensor([0, 63], dtype=torch.int32, device=dev)
    ws = ops.attention_workspace(B, H, n, D, 4, dev)
    def one(i): ops.draft_window_attention(q, kc[i], vc[i], out, ks, None, kv, 4, ws)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side): one(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(layers): one(i)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    nwg = 4 * 32 * 2
    buf = np.zeros((nwg, 8), dtype=np.uint64)
    lib.sjd_debug_k1_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.sjd_debug_k1_trace(buf.ctypes.data, nwg) == 0
    t = buf[:, :8].astype(np.int64)
    live = t[:, 5] > 0
    t = t[live]
    t0 = t[:, 0].min()
    rel = (t - t0) * 10e-3      # us (100 MHz)
    print(json.dumps(dict(kv=kv, workgroups=int(live.sum()), start_skew_us=[round(float(rel[:, 0].mean()), 2), round(float(rel[:, 0].max()), 2)],
                          phase_us_mean=dict(params=round(float((t[:, 1] - t[:, 0]).mean() * 10e-3), 2), first_tile=round(float((t[:, 2] - t[:, 1]).mean() * 10e-3), 2),
                                             loop=round(float((t[:, 3] - t[:, 2]).mean() * 10e-3), 2), barrier=round(float((t[:, 4] - t[:, 3]).mean() * 10e-3), 2),
                                             lds_write=round(float((t[:, 6] - t[:, 4]).mean() * 10e-3), 2), merge=round(float((t[:, 7] - t[:, 6]).mean() * 10e-3), 2), store_ack=round(float((t[:, 5] - t[:, 7]).mean() * 10e-3), 2)),
                          end_us=[round(float(rel[:, 5].mean()), 2), round(float(rel[:, 5].max()), 2)])))
