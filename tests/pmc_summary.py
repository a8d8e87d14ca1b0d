"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: mean counter value per dispatch (last N dispatches only)."""
import csv, re, sys, collections
path, last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = list(csv.DictReader(open(path)))
by = collections.defaultdict(list)
for r in rows:
    m = re.search(r"(k_\w+(<\d+>)?)", r["Kernel_Name"])
    by[(m.group(1) if m else r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("%-28s %-14s %10s %16s %16s" % ("kernel", "counter", "dispatches", "mean/dispatch", "sum"))
for (k, c), v in sorted(by.items()):
    if last:
        v = v[-last:]
    print("%-28s %-14s %10d %16.1f %16.1f" % (k, c, len(v), sum(v) / len(v), sum(v)))
