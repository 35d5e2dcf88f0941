#!/bin/bash
# One GPU-box pass that produces everything profiles/ holds for a round (run through gpurun from the repo root):
#   bench lines of the four single-GPU workloads, and for each of c1..c4: rocprofv3 kernel trace, PMC passes (HBM traffic:
#   FETCH_SIZE and WRITE_SIZE in separate runs; SQ counters in two runs), stage profiles of the instrumented kernel.
# Raw output lands in gpurun_out/prof/; tools/collect_profiles.py <tag> condenses it into the tracked profiles/ files.
# (rocprofv3 gets the program itself after `--`: python3 bench.py ..., never a wrapper.)
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof"
# gpurun allows 20 minutes per call: the pass runs in two calls — `profile_round.sh a` (bench lines, c2, c3), then `profile_round.sh b`
# (c4, c1, c5, register usage) — or in one (`profile_round.sh`, no argument) where there is no such limit
PART="${1:-ab}"
if [[ "$PART" == *a* ]]; then rm -rf "$OUT"; fi
mkdir -p "$OUT"
cd "$ROOT" || exit 1
export TMPDIR=/tmp
python3 tools/source_hash.py > "$OUT/source_hash.txt" || exit 1   # which sources these profiles are of (no .git on the box)
if [[ "$PART" == *a* ]]; then
for w in c2 c1 c3 c4; do
  timeout -k 10 600 python3 bench.py --workload $w > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err" || exit 1
  echo "bench $w done"
done
fi
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
SQ2="SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA"
WORKLOADS=""
[[ "$PART" == *a* ]] && WORKLOADS="c2 c3"
[[ "$PART" == *b* ]] && WORKLOADS="$WORKLOADS c4 c1"
for w in $WORKLOADS; do
  steps=$([ $w = c1 ] && echo 20 || echo 3)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$w" -- python3 bench.py --workload $w --no-cpu-baseline --steps $steps --warmup 1 > "$OUT/kt_$w.log" 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_$w" -- python3 bench.py --workload $w --no-cpu-baseline --steps 1 --warmup 0 > "$OUT/fetch_$w.log" 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_$w" -- python3 bench.py --workload $w --no-cpu-baseline --steps 1 --warmup 0 > "$OUT/write_$w.log" 2>&1 || exit 1
  if [ $w != c1 ]; then
    timeout -k 10 300 rocprofv3 --pmc $SQ1 --output-format csv -d "$OUT/sq1_$w" -- python3 bench.py --workload $w --no-cpu-baseline --steps 1 --warmup 0 > "$OUT/sq1_$w.log" 2>&1 || exit 1
    timeout -k 10 300 rocprofv3 --pmc $SQ2 --output-format csv -d "$OUT/sq2_$w" -- python3 bench.py --workload $w --no-cpu-baseline --steps 1 --warmup 0 > "$OUT/sq2_$w.log" 2>&1 || exit 1
    timeout -k 10 300 python3 tools/stage_profile.py --workload $w --spp $([ $w = c4 ] && echo 20 || echo 50) > "$OUT/stage_$w.txt" 2>&1 || exit 1
  fi
  echo "profiles $w done"
done
if [[ "$PART" != *b* ]]; then du -sh "$OUT"; exit 0; fi
# C4's dispatch timeline (the last frame of the kernel trace above): render kernels back to back, sum_samples hidden
python3 tools/kernel_gaps.py "$OUT/kt_c4" > "$OUT/timeline_c4.txt" 2> "$OUT/timeline_c4.err" || rm -f "$OUT/timeline_c4.txt"
# C5 (final_scene 1600x1600, 10000 spp, depth 50) on this one GPU: the bench line and the HBM traffic of one frame (26 s each)
timeout -k 10 400 python3 bench.py --workload c5 --steps 1 --warmup 0 > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err" || exit 1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_c5" -- python3 bench.py --workload c5 --no-cpu-baseline --steps 1 --warmup 0 > "$OUT/fetch_c5.log" 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_c5" -- python3 bench.py --workload c5 --no-cpu-baseline --steps 1 --warmup 0 > "$OUT/write_c5.log" 2>&1 || exit 1
echo "c5 done"
python3 tools/kernel_usage.py "path_kernel<false" > "$OUT/kernel_usage.txt" 2>&1 || true
# the nine reference scenes at their in-code cameras, both walks
timeout -k 10 300 python3 tools/scene_speed.py 2>&1 | grep -v amdgpu.ids > "$OUT/scene_speed.txt" || true
# keep the merged-back payload small: only the csv summaries
find "$OUT" -name "*.db" -delete 2>/dev/null
find "$OUT" -type f -size +8M -delete 2>/dev/null
du -sh "$OUT"
