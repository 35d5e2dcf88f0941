#!/bin/bash
# One GPU-box pass that produces everything profiles/ holds for a round (run through gpurun from the repo root):
#   bench lines of all four workloads, rocprofv3 kernel trace of the bench workload, PMC passes (HBM traffic, SQ),
#   stage profiles.  Raw output lands in gpurun_out/prof/; tools/collect_profiles.py condenses it into profiles/.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT" || exit 1
export TMPDIR=/tmp
for w in c2 c1 c3 c4; do
  timeout -k 10 600 python3 bench.py --workload $w > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err" || exit 1
  echo "bench $w done"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/kt.log" 2>&1 || exit 1
echo "kernel trace done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > "$OUT/fetch.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > "$OUT/write.log" 2>&1 || exit 1
echo "hbm counters done"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/sq1" -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > "$OUT/sq1.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/sq2" -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > "$OUT/sq2.log" 2>&1 || exit 1
echo "sq counters done"
for w in c2 c3 c4; do
  timeout -k 10 300 python3 tools/stage_profile.py --workload $w --spp $([ $w = c4 ] && echo 20 || echo 50) > "$OUT/stage_$w.txt" 2>&1 || exit 1
done
# keep the merged-back payload small: only the csv summaries
find "$OUT" -name "*.db" -delete 2>/dev/null
find "$OUT" -type f -size +8M -delete 2>/dev/null
du -sh "$OUT"
