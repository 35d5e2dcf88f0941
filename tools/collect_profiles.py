#!/usr/bin/env python3
"""Condenses gpurun_out/prof/ (written by tools/profile_round.sh on the GPU box) into the tracked profiles/ files.
Usage: python tools/collect_profiles.py <tag>      e.g. r01_ordered"""
import csv, glob, json, shutil, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "gpurun_out" / "prof"
DST = ROOT / "profiles"
tag = sys.argv[1]


def one(pattern):
    hits = sorted(glob.glob(str(SRC / pattern)), key=lambda f: Path(f).stat().st_mtime)
    assert hits, pattern
    return Path(hits[-1])  # (gpurun merges into gpurun_out/ without clearing it: take the latest run's file)


def counters(run, kernel_prefix="void (anonymous namespace)::path_kernel<false"):
    """Sum of each counter over the dispatches of the un-instrumented render kernel (one per bench step)."""
    out, dispatches = {}, set()
    with open(one(f"{run}/*/*_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            if row["Kernel_Name"].startswith(kernel_prefix):
                out[row["Counter_Name"]] = out.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                dispatches.add(row["Dispatch_Id"])
                out["_kernel"] = row["Kernel_Name"].split("(")[1 if row["Kernel_Name"].startswith("void (") else 0]
                out["_kernel"] = row["Kernel_Name"]
                out["_vgprs"], out["_lds_bytes"], out["_scratch"] = row["VGPR_Count"], row["LDS_Block_Size"], row["Scratch_Size"]
    out["_dispatches"] = len(dispatches)
    return out


for w in ("c1", "c2", "c3", "c4"):
    shutil.copy(SRC / f"bench_{w}.json", DST / f"{tag}_bench_{w}.json")
    if (SRC / f"stage_{w}.txt").exists():
        text = "\n".join(l for l in (SRC / f"stage_{w}.txt").read_text().splitlines() if "amdgpu.ids" not in l)
        (DST / f"{tag}_stage_profile_{w}.txt").write_text(text + "\n")
shutil.copy(one("kt/*/*_kernel_stats.csv"), DST / f"{tag}_c2_kernel_stats.csv")

fetch, write = counters("fetch"), counters("write")
assert fetch["_dispatches"] == 1 and write["_dispatches"] == 1
# MI355X_MICROARCH.md, HBM / rocprofv3: FETCH_SIZE and WRITE_SIZE are in KiB, each in its own pass; gfx950 reports half
# of the bytes fetched (x2 correction)
traffic = fetch["FETCH_SIZE"] * 1024 * 2 + write["WRITE_SIZE"] * 1024
hbm = {"c2": {"bytes_per_launch": traffic, "FETCH_SIZE_KB": fetch["FETCH_SIZE"], "WRITE_SIZE_KB": write["WRITE_SIZE"],
              "formula": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (separate --pmc passes; gfx950 FETCH_SIZE x2 correction; access "
                         "widths here are 8 B per lane, outside the calibrated 16 B streaming case)",
              "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline",
              "kernel": fetch["_kernel"], "tag": tag}}
(DST / "hbm_traffic.json").write_text(json.dumps(hbm, indent=1) + "\n")

sq = {}
for run in ("sq1", "sq2"):
    c = counters(run)
    assert c["_dispatches"] == 1
    sq.update({k: v for k, v in c.items()})
sq["valu_lane_utilisation"] = sq["SQ_THREAD_CYCLES_VALU"] / (sq["SQ_ACTIVE_INST_VALU"] * 64.0)
sq["wait_inst_any_share_of_wave_cycles"] = sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"]
sq["_note"] = ("rocprofv3 --pmc (two passes of 8 SQ counters), one dispatch of the render kernel of bench.py's default workload "
               "(C2, 480 M samples); tools/profile_round.sh")
(DST / f"{tag}_c2_pmc.json").write_text(json.dumps(sq, indent=1) + "\n")
b = json.load(open(SRC / "bench_c2.json"))
print("c2", b["value"], "Msamples/s; HBM traffic per launch", traffic / 1e9, "GB;", "VALU lane utilisation", round(sq["valu_lane_utilisation"], 3),
      "WAIT_INST_ANY share", round(sq["wait_inst_any_share_of_wave_cycles"], 3))
