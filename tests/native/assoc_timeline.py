"""where a frame of the association loop goes on the DEVICE's clock, from a rocprofv3 kernel trace of perf_assoc_fast.py (batched mode last):
   python3 assoc_timeline.py <kernel_trace.csv> [frames to analyse from the end = 150]
A frame = the launches from one trackwin_append / sq_fit (surface) kernel to the next frame's; per frame: span (first start -> last end), sum of
kernel durations, gaps inside the frame, and the gap to the next frame's first kernel (= the host's critical path: Hungarian step, thresholds,
the next native call's first launch)."""
import csv, sys, re, collections
path = sys.argv[1]
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rows = [r for r in csv.DictReader(open(path))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:60]
# frames end with a sinkhorn kernel
frames, cur = [], []
for r in rows:
    cur.append(r)
    if "sinkhorn" in r["Kernel_Name"]:
        frames.append(cur); cur = []
frames = [f for f in frames if any("gnn_" in name(r) for r in f)][-nlast:]
span = sum(int(f[-1]["End_Timestamp"]) - int(f[0]["Start_Timestamp"]) for f in frames) / len(frames) / 1e3
busy = sum(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in f) for f in frames) / len(frames) / 1e3
between = sum(int(b[0]["Start_Timestamp"]) - int(a[-1]["End_Timestamp"]) for a, b in zip(frames, frames[1:])) / (len(frames) - 1) / 1e3
period = (int(frames[-1][0]["Start_Timestamp"]) - int(frames[0][0]["Start_Timestamp"])) / (len(frames) - 1) / 1e3
print(f"{len(frames)} frames: period {period:.1f} us = device span {span:.1f} (kernels {busy:.1f} + gaps inside {span - busy:.1f}) + between frames {between:.1f}")
agg = collections.OrderedDict()
for f in frames:
    prev_end = None
    for r in f:
        k = name(r)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if prev_end is not None:
            a[2] += (int(r["Start_Timestamp"]) - prev_end) / 1e3
        prev_end = int(r["End_Timestamp"])
print(f"{'kernel':62s} {'per frame':>9s} {'us each':>8s} {'us/frame':>9s} {'gap before, us/frame':>22s}")
for k, (n, us, gap) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:62s} {n / len(frames):9.2f} {us / n:8.1f} {us / len(frames):9.1f} {gap / len(frames):22.1f}")
