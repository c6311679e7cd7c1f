import sys, os, ctypes, json
sys.path.insert(0, os.getcwd())
import torch
import sjd_amd._lib as L, sjd_amd.ops as ops
lib = L.load_exp()
dev = torch.device("cuda:0")
N, K, KC = 12288, 4096, 512
x = torch.randn(32, K, device=dev).to(torch.bfloat16)
ws = [ops.pack_weight_z((torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16), KC, False) for _ in range(6)]
for i in range(12):
    ops.skinny_gemm_engine(x, ws[i % 6], n_wg=256)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib.sjd_debug_engine_trace(buf)
v = list(buf)
print(json.dumps(dict(lib=os.environ.get("SJD_HIP_LIB", "")[-30:], issue_cycles=v[0], wait_cycles=v[1], idle=v[2], total_cycles=v[3], slots=v[4], waits=v[5],
                      per_slot_issue=round(v[0] / max(v[4], 1)), per_wait=round(v[1] / max(v[5], 1)))))
