"""Developer tool: from a rocprofv3 kernel trace of bench.py, busy time vs span of the last N steps (GPU idle inside a step),
and the mean duration of the dominant kernel over exactly those steps (what bench.py's roofline uses)."""
import csv, sys, collections
path, steps = sys.argv[1], int(sys.argv[2])
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
flow = [i for i, r in enumerate(rows) if "k_cl_solve" in r[2] or "k_solve_color" in r[2]]
# a step = from one k_build_colliders to the next
starts = [i for i, r in enumerate(rows) if "k_build_colliders" in r[2]]
starts = starts[-(steps + 1):]
busy = 0; span = rows[starts[-1]][0] - rows[starts[0]][0]
per = collections.defaultdict(float)
gaps = collections.defaultdict(float)   # idle time by the kernel that FOLLOWS the gap
for i in range(starts[0], starts[-1]):
    s, e, n = rows[i]
    busy += e - s
    key = n.split("(")[0].split("<")[0][-40:]
    per[key] += (e - s)
    if i > starts[0]:
        g = s - max(r[1] for r in rows[max(starts[0], i - 4):i])
        if g > 0:
            gaps[key] += g
print("last %d steps: span %.3f ms/step, kernels busy %.3f ms/step, idle %.3f ms/step (%.1f %%)" % (steps, span / steps / 1e6, busy / steps / 1e6, (span - busy) / steps / 1e6, 100.0 * (span - busy) / span))
for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:40]:
    print("  %-42s %8.1f us/step" % (k, v / steps / 1e3))
print("idle time by the kernel that follows the gap (us/step):")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1])[:12]:
    print("  %-42s %8.1f" % (k, v / steps / 1e3))
