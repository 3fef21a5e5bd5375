#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/i_all.log 2>&1; echo "all rc=$?"; tail -5 gpurun_out/i_all.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/i_bench_x3.json 2> gpurun_out/i_bench_x3.err; echo "bench rc=$?"
cat gpurun_out/i_bench_x3.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision bf16 > gpurun_out/i_bench_bf16.json 2> gpurun_out/i_bench_bf16.err; echo "bench rc=$?"
cat gpurun_out/i_bench_bf16.json
