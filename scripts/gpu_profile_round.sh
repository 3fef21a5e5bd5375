#!/bin/bash
# Round profile: the driver's bench line (with CPU baseline), rocprofv3 kernel stats + per-step summary of the same command
# (graph-replay windows only), PMC traffic passes over the same command, the extra lines (front end, config 3).
# usage: gpu_profile_round.sh <tag> [group]      writes gpurun_out/<tag>_*
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r03_x}
G=${2:-4}
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
cat gpurun_out/${TAG}_bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -o k -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_prof.err; echo "prof rc=$?"
cp gpurun_out/${TAG}_prof/k_kernel_stats.csv gpurun_out/${TAG}_kernel_stats_whole_process.csv 2>/dev/null
python scripts/step_summary.py gpurun_out/${TAG}_prof/k_kernel_trace.csv $G > gpurun_out/${TAG}_summary.md 2>&1
python scripts/trace_step.py gpurun_out/${TAG}_prof/k_kernel_trace.csv > gpurun_out/${TAG}_step_queues.txt 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_f -o f -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-inference > /dev/null 2> gpurun_out/${TAG}_pmc_f.err; echo "pmc f rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_w -o w -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-inference > /dev/null 2> gpurun_out/${TAG}_pmc_w.err; echo "pmc w rc=$?"
python scripts/pmc_traffic.py gpurun_out/${TAG}_pmc_f/f_counter_collection.csv gpurun_out/${TAG}_pmc_w/w_counter_collection.csv gpurun_out/${TAG}_pmc_traffic.json > /dev/null 2>&1; echo "traffic rc=$?"
cat gpurun_out/${TAG}_pmc_traffic.json | head -40
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --front-end > gpurun_out/${TAG}_bench_front_end.json 2> gpurun_out/${TAG}_bench_front_end.err; echo "front-end rc=$?"
cut -c1-300 gpurun_out/${TAG}_bench_front_end.json
timeout -k 10 400 python bench.py --unfreeze --batch 8 --steps 12 --warmup 3 > gpurun_out/${TAG}_bench_config3_full_finetune.json 2> gpurun_out/${TAG}_bench_config3.err; echo "config3 rc=$?"
cut -c1-400 gpurun_out/${TAG}_bench_config3_full_finetune.json
timeout -k 10 300 python bench.py --unfreeze --batch 8 --steps 12 --warmup 3 --precision bf16 --no-cpu-baseline > gpurun_out/${TAG}_bench_config3_amp.json 2> gpurun_out/${TAG}_bench_config3_amp.err; echo "config3 amp rc=$?"
cut -c1-400 gpurun_out/${TAG}_bench_config3_amp.json
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_c3prof -o k -- python3 scripts/finetune_steps.py bf16 12 graph > /dev/null 2> gpurun_out/${TAG}_c3prof.err; echo "config3 prof rc=$?"
python scripts/burst_summary.py gpurun_out/${TAG}_c3prof/k_kernel_trace.csv 8 > gpurun_out/${TAG}_config3_amp_kernel_summary.md 2>&1
rm -rf gpurun_out/${TAG}_c3prof
rm -rf gpurun_out/${TAG}_pmc_f gpurun_out/${TAG}_pmc_w gpurun_out/${TAG}_prof/k_kernel_trace.csv
head -30 gpurun_out/${TAG}_summary.md
