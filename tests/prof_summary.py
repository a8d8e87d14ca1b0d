"""Compact view of a rocprofv3 kernel_stats.csv: short kernel name, calls, average us, total ms, percent."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%-44s %8s %10s %10s %6s" % ("kernel", "calls", "avg_us", "total_ms", "%"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    name = r["Name"]
    m = re.search(r"(k_\w+(<\d+>)?)", name)
    short = m.group(1) if m else re.sub(r".*detail::(\w+)<.*", r"rocprim::\1", name)[:44]
    print("%-44s %8s %10.2f %10.2f %6.2f" % (short[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
