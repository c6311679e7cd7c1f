#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_glue.py -x -q -m gpu -k "five_to_eight or wide_chunk or three_and_four" 2>&1 | tail -15
