#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6_full_gpu_suite_mid.txt
cat gpurun_out/r6_full_gpu_suite_mid.txt
