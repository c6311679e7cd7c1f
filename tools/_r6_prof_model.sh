#!/bin/bash
# by-shape kernel times of a short decode of another configuration.  usage: _r6_prof_model.sh TAG <bench.py arguments...>
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out
B="python bench.py --steps 48 --warmup 8 --no-floor --no-torch-baseline --no-ar-baseline --no-cpu-baseline --no-whole-image --no-other-configs $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- $B > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof_bench.err
python tools/trace_by_grid.py $O/prof_$TAG 200 > $O/${TAG}_by_shape.txt
rm -rf $O/prof_$TAG
head -16 $O/${TAG}_by_shape.txt
