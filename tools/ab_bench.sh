#!/bin/bash
# A/B runs on ONE box (boxes differ by a few per cent): tools/ab_bench.sh "c3 c4" 2 "label=ENV1=.. ENV2=.." ...
# Each variant is an environment for bench.py (e.g. RT_AMD_LIB=rust-tracing_amd/lib/librt_amd_base.so for a library built from
# another revision); the variants alternate, REPS times each, and the Msamples/s are printed per run.
WORKLOADS="$1"; REPS="$2"; shift 2
for w in $WORKLOADS; do
  for r in $(seq 1 "$REPS"); do
    for v in "$@"; do
      label="${v%%=*}"; envs="${v#*=}"
      val=$(env $envs timeout -k 10 200 python bench.py --workload "$w" --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['value'])") || exit 1
      echo "$w rep$r $label $val"
    done
  done
done
