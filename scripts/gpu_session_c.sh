#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
python scripts/gemm_il_probe.py > gpurun_out/c_probe_il.txt 2>&1; echo "rc=$?"
python scripts/gemm_il_probe.py --resident > gpurun_out/c_probe_res.txt 2>&1; echo "rc=$?"
python scripts/gemm_il_probe.py --stages > gpurun_out/c_probe_stages.txt 2>&1; echo "rc=$?"
python scripts/gemm_il_probe.py --plain > gpurun_out/c_probe_plain.txt 2>&1; echo "rc=$?"
cat gpurun_out/c_probe_il.txt gpurun_out/c_probe_res.txt
