#!/bin/bash
# sweep of the pool kernel's knobs (service waves, exchange / box thresholds, patience) on one workload
#   usage: tools/sweep_pool.sh c2|c3 [spp]
w=${1:-c2}; spp=${2:-0}
out=gpurun_out/sweep_pool; mkdir -p $out
run() { # name, env...
  name=$1; shift
  env RT_POOL=1 "$@" timeout -k 10 120 python3 bench.py --workload $w --no-cpu-baseline --steps 2 ${spp:+--spp $spp} > $out/x.json 2>/dev/null || { echo "$w $name FAILED"; return; }
  python3 -c "import json; d=json.load(open('$out/x.json')); print('$w', '$name', d['value'], d['ms_per_step'])"
}
RT_POOL=0 timeout -k 10 120 python3 bench.py --workload $w --no-cpu-baseline --steps 2 ${spp:+--spp $spp} > $out/x.json 2>/dev/null && python3 -c "import json; d=json.load(open('$out/x.json')); print('$w', 'path_kernel', d['value'], d['ms_per_step'])"
for svc in 2 3 4; do
  for thx in 8 16 32; do
    for thbox in 8 16 32; do
      run "svc=$svc th_x=$thx th_box=$thbox" RT_POOL_SERVICE=$svc RT_POOL_TH_X=$thx RT_POOL_TH_BOX=$thbox
    done
  done
done
run "svc=3 th_x=16 th_box=16 aux=1" RT_POOL_SERVICE=3 RT_POOL_AUX=1
run "svc=3 th_x=16 th_box=16 aux=0" RT_POOL_SERVICE=3 RT_POOL_AUX=0
run "svc=3 patience=0" RT_POOL_SERVICE=3 RT_POOL_PATIENCE=0
run "svc=3 patience=8" RT_POOL_SERVICE=3 RT_POOL_PATIENCE=8
run "svc=3 full=48" RT_POOL_SERVICE=3 RT_POOL_FULL=48
run "svc=3 th_prim=16" RT_POOL_SERVICE=3 RT_POOL_TH_PRIM=16
run "svc=3 th_prim=4" RT_POOL_SERVICE=3 RT_POOL_TH_PRIM=4
