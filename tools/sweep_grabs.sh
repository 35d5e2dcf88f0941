#!/bin/bash
# sweep of the job hand-out knobs (RT_GRAB_TAPER, RT_JOBS_PER_GRAB, RT_SAMPLE_BUFFER_MB) on the launch-bound workloads
out=gpurun_out/sweep; mkdir -p $out
for w in c1 c2; do
  for taper in 0 1 2 8; do
    for mb in 2048 16384; do
      [ $w = c1 ] && [ $mb = 16384 ] && continue
      steps=$([ $w = c1 ] && echo 50 || echo 3)
      RT_SAMPLE_BUFFER_MB=$mb RT_GRAB_TAPER=$taper timeout -k 10 120 python3 bench.py --workload $w --no-cpu-baseline --steps $steps > $out/x.json 2>/dev/null || exit 1
      python3 -c "import json; d=json.load(open('$out/x.json')); print('$w taper $taper buffer $mb MB', d['value'], d['ms_per_step'])"
    done
  done
done
