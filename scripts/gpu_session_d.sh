#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_interleaved.py tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/d_ops.log 2>&1; echo "ops rc=$?"; tail -15 gpurun_out/d_ops.log
timeout -k 10 600 python -m pytest tests/test_gpu_encoders.py tests/test_gpu_base_parity.py tests/test_gpu_system.py -x -q -m gpu -s > gpurun_out/d_sys.log 2>&1; echo "sys rc=$?"; tail -5 gpurun_out/d_sys.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/d_bench_x3.json 2> gpurun_out/d_bench_x3.err; echo "bench rc=$?"
cat gpurun_out/d_bench_x3.json
