"""Print a rocprofv3 kernel_stats.csv as a short table (name, calls, average us, total ms)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:top]:
    name = r["Name"]
    if name.startswith("_ZN3smt"):
        name = name[7:].lstrip("0123456789")
    print(f"{name[:64]:64s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs']) / 1e3:9.1f} total_ms {float(r['TotalDurationNs']) / 1e6:9.2f} {r['Percentage']:>6s}%")
