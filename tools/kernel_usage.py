#!/usr/bin/env python3
"""`make -C rust-tracing_amd/csrc usage` condensed: registers, spills, scratch and occupancy of every render kernel."""
import re, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
out = subprocess.run(["make", "-C", str(ROOT / "rust-tracing_amd" / "csrc"), "usage"], capture_output=True, text=True)
text = out.stdout + out.stderr
name, d = None, {}
for line in text.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1); d[name] = {}; continue
    m = re.search(r"remark:\s+(VGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
    if m and name:
        d[name][m.group(1)] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(d), capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':70s} VGPR spillV spillS scratch waves/SIMD")
for mangled, dn in zip(d, names):
    v = d[mangled]
    dn = dn.replace("(anonymous namespace)::", "").replace("(rtk::KParams)", "").replace("void ", "")
    if len(sys.argv) > 1 and sys.argv[1] not in dn:
        continue
    print(f"{dn[:70]:70s} {v.get('VGPRs', 0):4d} {v.get('VGPRs Spill', 0):6d} {v.get('SGPRs Spill', 0):6d} {v.get('ScratchSize [bytes/lane]', 0):7d} {v.get('Occupancy [waves/SIMD]', 0):6d}")
