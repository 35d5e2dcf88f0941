#!/bin/bash
# SQ counters of the render kernel for one workload at reduced spp, path kernel vs pool kernel (two rocprofv3 passes each)
w=${1:-c2}; spp=${2:-100}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcq; rm -rf $out; mkdir -p $out
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
SQ2="SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA"
for pool in 0 1; do
  export RT_POOL=$pool
  timeout -k 10 200 rocprofv3 --pmc $SQ1 --output-format csv -d $out/a$pool -- python3 bench.py --workload $w --no-cpu-baseline --steps 1 --warmup 0 --spp $spp > $out/a$pool.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc $SQ2 --output-format csv -d $out/b$pool -- python3 bench.py --workload $w --no-cpu-baseline --steps 1 --warmup 0 --spp $spp > $out/b$pool.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob
for pool in (0, 1):
    tot = {}
    for run in ("a", "b"):
        f = sorted(glob.glob(f"gpurun_out/pmcq/{run}{pool}/*/*_counter_collection.csv"))[-1]
        for row in csv.DictReader(open(f)):
            if ("pool_kernel" in row["Kernel_Name"]) if pool else ("path_kernel<false" in row["Kernel_Name"]):
                tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    print("pool" if pool else "path", {k: f"{v:.3e}" for k, v in sorted(tot.items())})
    print("   lane util", tot["SQ_THREAD_CYCLES_VALU"] / (tot["SQ_ACTIVE_INST_VALU"] * 64), "wait share", tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"],
          "valu busy (x4/wave cycles x waves/simd)", tot["SQ_ACTIVE_INST_VALU"] * 4 / tot["SQ_WAVE_CYCLES"])
PY
find $out -name "*.db" -delete
