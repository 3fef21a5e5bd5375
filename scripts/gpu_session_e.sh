#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e_prof -o x3 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/e_prof.log 2>&1; echo "prof rc=$?"
python scripts/trace_encoder_pass.py gpurun_out/e_prof/x3_kernel_trace.csv 2 > gpurun_out/e_pass_eager.txt 2>&1
python scripts/trace_step.py gpurun_out/e_prof/x3_kernel_trace.csv > gpurun_out/e_step.txt 2>&1
tail -60 gpurun_out/e_step.txt
rm -f gpurun_out/e_prof/x3_kernel_trace.csv.keep
