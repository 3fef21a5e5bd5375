#!/bin/bash
# usage: r3_sweep.sh <tag> "<ENV=.. ENV=..>|<bench args>" ...   one bench line per spec into gpurun_out/<tag>_<i>.json
TAG=$1; shift
i=0
for spec in "$@"; do
  envs=${spec%%|*}; args=${spec#*|}
  env $envs python bench.py $args --no-cpu-baseline > gpurun_out/${TAG}_$i.json 2> gpurun_out/${TAG}_$i.err || { echo "spec $i failed: $spec"; tail -3 gpurun_out/${TAG}_$i.err; exit 1; }
  python - "$spec" gpurun_out/${TAG}_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r=d['roofline']
print(sys.argv[1],'->',d['ms_per_step'],'ms',d['value'],'utt/s frac',r['frac'],'gemm_ms/step',r.get('gemm_ms_per_step'),d['gemm_plans'],flush=True)
PY
  i=$((i+1))
done
