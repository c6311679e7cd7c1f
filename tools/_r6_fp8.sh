#!/bin/bash
# config 5: the fp8 model-level bound under (a) round 5's arithmetic, (b) hi / lo operands, (c) hi / lo + calibrated per-layer scales
cd ${GRAFT_REPO_ROOT:-.}
T="python -m pytest tests/test_gpu_fp8_model_bound.py -x -q -m gpu"
SJD_HIP_LIB=tools/_exp/k1_nohilo/libsjd_hip.so SJD_FP8_CALIBRATE=0 SJD_FP8_TAG="round-5 arithmetic: Q and P rounded once to e4m3, scales (1, 1)" SJD_FP8_FILE_TAG=_a_r5 $T 2>&1 | tail -3
SJD_FP8_CALIBRATE=0 SJD_FP8_TAG="hi/lo e4m3 operands for Q and P, scales (1, 1)" SJD_FP8_FILE_TAG=_b_hilo $T 2>&1 | tail -3
SJD_FP8_FILE_TAG=_c_hilo_calibrated $T 2>&1 | tail -3
python - <<PY
import json
for t in ("_a_r5", "_b_hilo", "_c_hilo_calibrated"):
    try:
        d = json.load(open("gpurun_out/r6_fp8_model_bound%s.json" % t))
        print(t, d["summary"], d["decode_256_steps"], d.get("kv_scales_first_and_last_layer"))
    except Exception as e:
        print(t, "missing", e)
PY
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fp8" 2>&1 | tail -3
