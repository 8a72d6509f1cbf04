"""Dev tool: per-step view of a rocprofv3 kernel_stats.csv:  python tools/show_stats.py FILE STEPS"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total per step {tot / steps / 1e3:.1f} us")
for r in rows:
    per = float(r['TotalDurationNs']) / steps / 1e3
    if per < 0.5: continue
    print(f"{r['Name'][:100]:100s} calls/step {float(r['Calls'])/steps:5.2f} avg {float(r['AverageNs'])/1e3:7.1f} us  per step {per:7.1f} us {float(r['TotalDurationNs'])/tot*100:5.1f}%")
