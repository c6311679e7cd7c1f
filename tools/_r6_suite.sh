#!/bin/bash
# round-end evidence: the whole GPU suite plain and with the poisoned allocator pool, the loop fuzz, the default bench line, smoke()
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
TAG=${1:-r6_final}
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${TAG}_smoke.txt 2>&1; tail -1 $O/${TAG}_smoke.txt
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; tail -c 600 $O/${TAG}_bench.json
python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/${TAG}_gpu_suite_plain.txt; tail -3 $O/${TAG}_gpu_suite_plain.txt
SJD_TEST_POISON=1 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/${TAG}_gpu_suite_poisoned.txt; tail -3 $O/${TAG}_gpu_suite_poisoned.txt
python tools/fuzz_loops.py --n 24 > $O/${TAG}_fuzz.txt 2>&1; tail -3 $O/${TAG}_fuzz.txt
python tools/fuzz_loops.py --n 24 --seed 100 --poison > $O/${TAG}_fuzz_poisoned.txt 2>&1; tail -3 $O/${TAG}_fuzz_poisoned.txt
