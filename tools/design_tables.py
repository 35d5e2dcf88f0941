#!/usr/bin/env python3
"""CPU.  Prints DESIGN.md's two measured tables (section 5 "Roofline") as markdown from the tracked profiles of a round:
    python tools/design_tables.py r04
so that every number in them can be traced to a file under profiles/ of that tag."""
import json, re, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
P = ROOT / "profiles"
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
prev = f"r{int(tag[1:]) - 1:02d}"
hbm = json.loads((P / "hbm_traffic.json").read_text())
names = {"c1": "C1 random-spheres 400×225×10 d10", "c2": "**C2 random-spheres 1200×800×500 d50**", "c3": "C3 Cornell 600×600×1000 d50",
         "c4": "C4 final_scene 800×800×5000 d40", "c5": "C5 final_scene 1600×1600×10000 d50 on ONE GPU"}


def bench(t, w):
    f = P / (f"{t}_bench_{w}.json" if w != "c5" else f"{t}_bench_c5_1gpu.json")
    return json.loads(f.read_text().splitlines()[-1]) if f.exists() else None


print(f"| Config (1 GPU, {tag}) | Msamples/s ({prev}) | step ms (kernels) | algorithmic flop / sample | executed: record visits / prim tests | TFLOP/s (frac of 78.6) | HBM GB / step (of 8 TB/s) | lane utilisation / VALU busy | CPU restatement, {bench(tag, 'c2')['cpu_baseline']['cores']} cores: reference / tight |")
print("|---|---|---|---|---|---|---|---|---|")
for w in ("c1", "c2", "c3", "c4", "c5"):
    b, old = bench(tag, w), bench(prev, w)
    if not b:
        continue
    r = b["roofline"]
    ex = r["executed_events_per_sample"]
    h = hbm.get(w, {})
    pmc = P / f"{tag}_{w}_pmc.json"
    sq = json.loads(pmc.read_text()) if pmc.exists() else {}
    cpu = b.get("cpu_baseline")
    gb = h.get("bytes_per_launch", 0) / 1e9
    frac_hbm = h.get("bytes_per_launch", 0) / (r["kernel_ms"] * 1e-3) / 8e12 if h else 0
    print(f"| {names[w]} | {b['value']:.0f} ({old['value']:.0f}) | {b['ms_per_step']:.2f} ({r['kernel_ms']:.2f}) | {r['algorithmic_flops_per_sample']:.0f} | "
          f"{ex['node_visits']:.1f} / {ex['sphere_tests'] + ex['quad_tests']:.1f} | {r['achieved']:.2f} ({100 * r['frac']:.1f} %) | "
          f"{gb:.1f} ({100 * frac_hbm:.1f} %) | " + (f"{sq['valu_lane_utilisation']:.3f} / {sq['valu_busy']:.2f}" if sq else "–") + " | " +
          (f"{cpu['value']:.2f} / {cpu['tight_box_test']['value']:.2f}" if cpu else "–") + " |")
print()
print("| | box | sphere | quad | other | shade | path end (separate rounds) |")
print("|---|---|---|---|---|---|---|")
for w in ("c2", "c3", "c4"):
    f = P / f"{tag}_stage_profile_{w}.txt"
    if not f.exists():
        continue
    rows = {}
    for line in f.read_text().splitlines():
        m = re.match(r"(\w+)\s+rounds per 64 samples\s+([\d.]+)\s+mean active lanes\s+([\d.]+)\s+cycles\s+([\d.]+) %", line)
        if m:
            rows[m.group(1)] = (float(m.group(2)), float(m.group(3)), float(m.group(4)))
    cell = lambda k: "–" if rows[k][0] == 0 else f"{rows[k][0]:.1f} / {rows[k][1]:.0f} / {rows[k][2]:.1f} %"
    print(f"| {w.upper()} | " + " | ".join(cell(k) for k in ("box", "sphere", "quad", "other", "shade", "newjob")) + " |")
