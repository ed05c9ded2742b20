"""dev: mean per-frame GPU timeline of the odometry_loop leg from a rocprofv3 kernel trace (csv): for the frames of the
timed pass, every kernel launch in order with its start offset from the frame's first launch, its duration and the idle
gap in front of it."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_distort"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void ", "").replace("icp::", "")
    return n.split("(")[0][:60]
frames, cur = [], None
for r in rows:
    n = short(r["Kernel_Name"])
    if anchor in n:
        cur = []
        frames.append(cur)
    if cur is not None:
        cur.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
frames = frames[len(frames) // 2 + 1:-1]  # the timed pass
print(f"{len(frames)} frames; launches per frame: {sorted(set(len(f) for f in frames))}")
L = max(set(len(f) for f in frames), key=[len(f) for f in frames].count)
sel = [f for f in frames if len(f) == L]
# only frames with the same kernel sequence
seq0 = [k[0] for k in sel[0]]
sel = [f for f in sel if [k[0] for k in f] == seq0]
print(f"{len(sel)} frames with the modal sequence of {L} launches")
period = [b[0][1] - a[0][1] for a, b in zip(frames[:-1], frames[1:])]
print(f"frame period mean {sum(period)/len(period)/1e3:.1f} us")
tot_busy = 0
for i, name in enumerate(seq0):
    st = sum(f[i][1] - f[0][1] for f in sel) / len(sel) / 1e3
    du = sum(f[i][2] - f[i][1] for f in sel) / len(sel) / 1e3
    gap = sum((f[i][1] - f[i - 1][2]) if i else 0 for f in sel) / len(sel) / 1e3
    tot_busy += du
    print(f"{i:3d} {name:60s} start {st:8.1f} dur {du:7.1f} gap {gap:7.1f}")
print(f"busy {tot_busy:.1f} us per frame")
# outlier frames of the timed pass: the launches that differ from the modal frame
allf = frames
med = sorted(period)[len(period) // 2]
print(f"median period {med/1e3:.1f} us; outliers:")
for j, (a, b) in enumerate(zip(allf[:-1], allf[1:])):
    per = b[0][1] - a[0][1]
    if per > 1.3 * med:
        print(f"-- frame +{j}: period {per/1e3:.1f} us, {len(a)} launches")
        prev_end = a[0][1]
        for (n, s0, e0) in a:
            gap = (s0 - prev_end) / 1e3
            du = (e0 - s0) / 1e3
            if du > 30 or gap > 30:
                print(f"     {n:58s} start {(s0 - a[0][1])/1e3:8.1f} dur {du:7.1f} gap {gap:7.1f}")
            prev_end = e0
