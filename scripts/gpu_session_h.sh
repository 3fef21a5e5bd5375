#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_interleaved.py -x -q -m gpu -k "512_thread or split_k" > gpurun_out/h_ops.log 2>&1; echo "ops rc=$?"; tail -5 gpurun_out/h_ops.log
timeout -k 10 400 python scripts/gemm_il_probe.py --cfgs --wide > gpurun_out/h_probe_cfgs.txt 2>&1; echo "probe rc=$?"
cat gpurun_out/h_probe_cfgs.txt
