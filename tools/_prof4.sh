cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
B="python bench.py --prompts-per-gpu 4 --steps 1200 --warmup 4 --no-floor --no-torch-baseline --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_4p -- $B > $O/r3_4p_prof_bench.json 2> $O/r3_4p_prof.err
python tools/trace_by_grid.py $O/prof_4p 200 > $O/r3_4prompts_by_shape.txt
head -28 $O/r3_4prompts_by_shape.txt
rm -rf $O/prof_4p
