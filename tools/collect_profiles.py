#!/usr/bin/env python3
"""Condenses gpurun_out/prof/ (written by tools/profile_round.sh on the GPU box) into the tracked profiles/ files.
Usage: python tools/collect_profiles.py <tag>      e.g. r02

  profiles/<tag>_bench_<w>.json            bench.py's line per workload (c1..c4)
  profiles/<tag>_<w>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary of `bench.py --workload <w>`
  profiles/<tag>_<w>_pmc.json              SQ counters of the render kernel's dispatches of ONE frame (two --pmc passes) + derived
                                           lane utilisation / wait share, and the HBM counters of their own passes
  profiles/<tag>_stage_profile_<w>.txt     per-stage rounds / lanes / cycles of the instrumented kernel
  profiles/hbm_traffic.json                per workload: HBM bytes of one frame (what bench.py prints as roofline.traffic)
"""
import csv, glob, json, shutil, subprocess, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "gpurun_out" / "prof"
DST = ROOT / "profiles"
tag = sys.argv[1]
KERNEL = "path_kernel<false"  # the un-instrumented render kernel (bench.py's counting passes use path_kernel<true)
SUM_KERNEL = "sum_samples_kernel"  # the per-pixel summation behind it: reads the whole sample buffer back
sys.path.insert(0, str(ROOT / "tools"))
from source_hash import source_hash  # noqa: E402

# a profile is filed only for the sources it was measured on: the hash the box recorded must be the working tree's
measured = (SRC / "source_hash.txt").read_text().strip() if (SRC / "source_hash.txt").exists() else None
if measured != source_hash() and "--force" not in sys.argv:
    sys.exit(f"collect_profiles.py: gpurun_out/prof/ was measured on sources {measured}, the working tree is {source_hash()}: "
             "re-run tools/profile_round.sh on this build (or pass --force and say so in DESIGN.md)")


def one(pattern):
    hits = sorted(glob.glob(str(SRC / pattern)), key=lambda f: Path(f).stat().st_mtime)
    assert hits, pattern
    return Path(hits[-1])  # (gpurun merges into gpurun_out/ without clearing it: take the latest run's file)


def counters(run, kernel=KERNEL):
    """Sum of each counter over the dispatches of `kernel` (one frame = one bench step: --steps 1 --warmup 0; a frame that
    needs several launches has several dispatches)."""
    out, dispatches = {}, set()
    with open(one(f"{run}/*/*_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            if kernel in row["Kernel_Name"]:
                out[row["Counter_Name"]] = out.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                dispatches.add(row["Dispatch_Id"])
                out["_kernel"] = row["Kernel_Name"]
                out["_rocprof_vgpr_field"], out["_rocprof_lds_field"], out["_scratch"] = row["VGPR_Count"], row["LDS_Block_Size"], row["Scratch_Size"]
    out["_dispatches_per_frame"] = len(dispatches)
    return out


commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
hbm = {}
for w in ("c1", "c2", "c3", "c4", "c5"):
    if w == "c5":  # one GPU's frame of the 8-GPU configuration: bench line and HBM traffic only
        if not (SRC / "bench_c5.json").exists():
            continue
        shutil.copy(SRC / "bench_c5.json", DST / f"{tag}_bench_c5_1gpu.json")
        fetch, write = counters("fetch_c5"), counters("write_c5")
        sfetch, swrite = counters("fetch_c5", SUM_KERNEL), counters("write_c5", SUM_KERNEL)
        render_bytes = fetch["FETCH_SIZE"] * 1024 * 2 + write["WRITE_SIZE"] * 1024
        sum_bytes = sfetch.get("FETCH_SIZE", 0.0) * 1024 * 2 + swrite.get("WRITE_SIZE", 0.0) * 1024
        traffic = render_bytes + sum_bytes
        bench = json.load(open(SRC / "bench_c5.json"))
        hbm[w] = {"bytes_per_launch": traffic, "render_kernel_bytes": render_bytes, "sum_samples_kernel_bytes": sum_bytes,
                  "FETCH_SIZE_KB": fetch["FETCH_SIZE"], "WRITE_SIZE_KB": write["WRITE_SIZE"],
                  "dispatches_per_frame": fetch["_dispatches_per_frame"],
                  "achieved_GBps": round(traffic / (bench["roofline"]["kernel_ms"] * 1e-3) / 1e9, 1),
                  "formula": "as c1..c4", "kernel": fetch["_kernel"], "tag": f"{tag} ({commit}, sources {measured})", "source_hash": measured,
                  "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline"}
        print("c5", bench["value"], "Msamples/s; HBM", round(traffic / 1e9, 2), "GB/frame")
        continue
    shutil.copy(SRC / f"bench_{w}.json", DST / f"{tag}_bench_{w}.json")
    if (SRC / f"stage_{w}.txt").exists():
        text = "\n".join(l for l in (SRC / f"stage_{w}.txt").read_text().splitlines() if "amdgpu.ids" not in l)
        (DST / f"{tag}_stage_profile_{w}.txt").write_text(text + "\n")
    shutil.copy(one(f"kt_{w}/*/*_kernel_stats.csv"), DST / f"{tag}_{w}_kernel_stats.csv")
    fetch, write = counters(f"fetch_{w}"), counters(f"write_{w}")
    # MI355X_MICROARCH.md, HBM / rocprofv3: FETCH_SIZE and WRITE_SIZE are in KiB, each in its own pass; gfx950 reports half
    # of the bytes fetched (x2 correction)
    sfetch, swrite = counters(f"fetch_{w}", SUM_KERNEL), counters(f"write_{w}", SUM_KERNEL)
    render_bytes = fetch["FETCH_SIZE"] * 1024 * 2 + write["WRITE_SIZE"] * 1024
    sum_bytes = sfetch.get("FETCH_SIZE", 0.0) * 1024 * 2 + swrite.get("WRITE_SIZE", 0.0) * 1024
    traffic = render_bytes + sum_bytes  # what the timed step moves: the render kernel(s) and the per-pixel summation of their samples
    bench = json.load(open(SRC / f"bench_{w}.json"))
    hbm[w] = {"bytes_per_launch": traffic, "render_kernel_bytes": render_bytes, "sum_samples_kernel_bytes": sum_bytes,
              "FETCH_SIZE_KB": fetch["FETCH_SIZE"], "WRITE_SIZE_KB": write["WRITE_SIZE"],
              "sum_samples_FETCH_SIZE_KB": sfetch.get("FETCH_SIZE", 0.0), "sum_samples_WRITE_SIZE_KB": swrite.get("WRITE_SIZE", 0.0),
              "dispatches_per_frame": fetch["_dispatches_per_frame"],
              "achieved_GBps": round(traffic / (bench["roofline"]["kernel_ms"] * 1e-3) / 1e9, 1),
              "formula": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024, summed over the frame's render-kernel AND sum_samples_kernel dispatches (separate --pmc passes; "
                         "gfx950 FETCH_SIZE x2 correction; access widths here are 8 B per lane, outside the calibrated 16 B streaming case)",
              "command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --workload {w} --steps 1 --warmup 0 --no-cpu-baseline",
              "kernel": fetch["_kernel"], "tag": f"{tag} ({commit}, sources {measured})", "source_hash": measured}
    if w == "c1":
        continue
    sq = {}
    for run in (f"sq1_{w}", f"sq2_{w}"):
        sq.update(counters(run))
    sq["valu_lane_utilisation"] = sq["SQ_THREAD_CYCLES_VALU"] / (sq["SQ_ACTIVE_INST_VALU"] * 64.0)
    # share of the cycles in which a SIMD's VALU is issuing: ACTIVE_INST_VALU is counted per wave, a SIMD holds waves_per_simd of them
    waves_per_simd = {"c2": 4, "c3": 4, "c4": 3}[w]
    sq["valu_busy"] = min(1.0, sq["SQ_ACTIVE_INST_VALU"] * waves_per_simd / sq["SQ_WAVE_CYCLES"])
    sq["_tag"] = f"{tag} ({commit}, sources {measured})"
    sq["source_hash"] = measured
    sq["wait_inst_any_share_of_wave_cycles"] = sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"]
    sq["lds_bank_conflict_share_of_lds_active"] = sq["SQ_LDS_BANK_CONFLICT"] / max(1.0, sq["SQ_LDS_IDX_ACTIVE"])
    sq["hbm_bytes_per_frame"] = traffic
    sq["_note"] = (f"rocprofv3 --pmc (two passes of 8 SQ counters), the render kernel's dispatches of one frame of bench.py --workload {w}; "
                   "tools/profile_round.sh.  _rocprof_vgpr_field is rocprofv3's VGPR_Count column — for this wave64 kernel half the "
                   "allocated registers (56 <-> 112..119; the compiler's figure is in kernel_usage.txt) — and _rocprof_lds_field its "
                   "LDS_Block_Size, the STATIC group segment only: the kernel's LDS is all dynamic (bench line: scene stats lds_bytes)")
    (DST / f"{tag}_{w}_pmc.json").write_text(json.dumps(sq, indent=1) + "\n")
    print(w, bench["value"], "Msamples/s; HBM", round(traffic / 1e9, 2), "GB/frame =", hbm[w]["achieved_GBps"], "GB/s; VALU lane utilisation",
          round(sq["valu_lane_utilisation"], 3), "WAIT_INST_ANY share", round(sq["wait_inst_any_share_of_wave_cycles"], 3))
(DST / "hbm_traffic.json").write_text(json.dumps(hbm, indent=1) + "\n")
if (SRC / "timeline_c4.txt").exists():
    shutil.copy(SRC / "timeline_c4.txt", DST / f"{tag}_c4_timeline.txt")
if (SRC / "kernel_usage.txt").exists():
    shutil.copy(SRC / "kernel_usage.txt", DST / f"{tag}_kernel_usage.txt")
if (SRC / "scene_speed.txt").exists():
    shutil.copy(SRC / "scene_speed.txt", DST / f"{tag}_scene_speed.txt")
