#!/bin/bash
# first GPU session of round 2: base-size parity, x3 bench + kernel stats
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_base_parity.py -x -q -m gpu -s > gpurun_out/a_parity.log 2>&1; echo "parity rc=$?" 
tail -5 gpurun_out/a_parity.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/a_bench_x3.json 2> gpurun_out/a_bench_x3.err; echo "bench rc=$?"
cat gpurun_out/a_bench_x3.json
rocprofv3 --kernel-trace --stats -d gpurun_out/a_prof_x3 -o x3 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/a_prof_x3.log 2>&1; echo "prof rc=$?"
python scripts/prof_summary.py $(dirname $(find gpurun_out/a_prof_x3 -name "*kernel_stats.csv" | head -1)) 16 > gpurun_out/a_prof_x3_summary.md 2>&1
head -40 gpurun_out/a_prof_x3_summary.md
