#!/bin/bash
# interleaved three-product layout: op tests, encoder/system tests, bench + kernel stats
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_interleaved.py tests/test_gpu_ops.py tests/test_gpu_encoders.py -x -q -m gpu > gpurun_out/b_ops.log 2>&1; echo "ops rc=$?"; tail -15 gpurun_out/b_ops.log
python -m pytest tests/test_gpu_base_parity.py tests/test_gpu_system.py -x -q -m gpu -s > gpurun_out/b_sys.log 2>&1; echo "sys rc=$?"; tail -8 gpurun_out/b_sys.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_bench_x3.json 2> gpurun_out/b_bench_x3.err; echo "bench rc=$?"
cat gpurun_out/b_bench_x3.json; tail -3 gpurun_out/b_bench_x3.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b_prof_x3 -o x3 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b_prof_x3.log 2>&1; echo "prof rc=$?"
python scripts/prof_summary.py $(dirname $(find gpurun_out/b_prof_x3 -name "*kernel_stats.csv" | head -1)) 16 > gpurun_out/b_prof_x3_summary.md 2>&1
head -30 gpurun_out/b_prof_x3_summary.md
