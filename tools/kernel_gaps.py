#!/usr/bin/env python3
"""Dispatch timeline of one frame from a rocprofv3 kernel trace: python tools/kernel_gaps.py <trace dir> [frames]

For the LAST frame of the run (the dispatches behind the last-but-`frames` tiles_to_frame / the last render burst): every dispatch
with start (relative to the frame's first dispatch), duration, and the overlap with the dispatch before it on the OTHER queue —
what shows that a chunked frame's render kernels run back to back across the two internal streams while sum_samples_kernel
hides behind the next render kernel."""
import csv, glob, sys

traces = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))
if not traces:
    sys.exit(f"kernel_gaps.py: no *_kernel_trace.csv under {sys.argv[1]}")
f = traces[-1]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
render = [i for i, r in enumerate(rows) if 'path_kernel<false' in r['Kernel_Name']]
if not render:
    sys.exit(f"kernel_gaps.py: no dispatch of the render kernel ('path_kernel<false') in {f}: "
             f"kernel names seen: {sorted({r['Kernel_Name'][:60] for r in rows})[:8]}")
# the last frame: walk back from the last render dispatch while the gaps between render dispatches stay under 5 ms
last = render[-1]
first = last
for i in reversed(render[:-1]):
    if int(rows[first]['Start_Timestamp']) - int(rows[i]['End_Timestamp']) > 5_000_000:
        break
    first = i
frame = [r for r in rows[first:] if int(r['Start_Timestamp']) <= int(rows[last]['End_Timestamp']) + 50_000_000]
t0 = int(frame[0]['Start_Timestamp'])
busy_until, idle = t0, 0
print(f"{len(frame)} dispatches; times in ms from the frame's first dispatch")
print(f"{'kernel':28s} {'queue':>6s} {'start':>10s} {'dur':>9s} {'device idle before':>19s}")
short = lambda n: 'path_kernel' if 'path_kernel' in n else ('sum_samples' if 'sum_samples' in n else n.split('(')[0][-24:])
render_busy = 0
for r in frame:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = max(0, s - busy_until)
    idle += gap
    busy_until = max(busy_until, e)
    if 'path_kernel' in r['Kernel_Name']:
        render_busy += e - s
    print(f"{short(r['Kernel_Name']):28s} {r.get('Queue_Id', '?'):>6s} {(s - t0) / 1e6:10.3f} {(e - s) / 1e6:9.3f} {gap / 1e6:19.3f}")
span = busy_until - t0
print(f"frame span {span / 1e6:.3f} ms; device idle (no kernel running) {idle / 1e6:.3f} ms = {100 * idle / span:.2f} %; "
      f"render kernels' durations sum to {render_busy / 1e6:.3f} ms = {render_busy / span:.2f} x the span (they overlap across the two streams)")
