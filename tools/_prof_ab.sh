cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out
B="python bench.py --steps 64 --warmup 8 --no-floor --no-torch-baseline --no-cpu-baseline --no-whole-image --no-other-configs"
(cd tools/_exp/old && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_old -- $B > $O/r3_prof_old.json 2> $O/r3_prof_old.err)
python tools/trace_by_grid.py $O/prof_old 200 > $O/r3_old_by_shape.txt
SJD_K1_NO_MERGE=1 SJD_REDUCE_FUSED=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_n00 -- $B > $O/r3_prof_n00.json 2> $O/r3_prof_n00.err
python tools/trace_by_grid.py $O/prof_n00 200 > $O/r3_n00_by_shape.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_n11 -- $B > $O/r3_prof_n11.json 2> $O/r3_prof_n11.err
python tools/trace_by_grid.py $O/prof_n11 200 > $O/r3_n11_by_shape.txt
rm -rf $O/prof_old $O/prof_n00 $O/prof_n11
head -12 $O/r3_old_by_shape.txt; head -12 $O/r3_n00_by_shape.txt; head -12 $O/r3_n11_by_shape.txt
for f in old n00 n11; do python -c "
import json; d=json.loads(open('$O/r3_prof_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'])"; done
