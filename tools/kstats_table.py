import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{name[:70]:70s} calls/step {int(r['Calls'])/steps:5.1f} avg {float(r['AverageNs'])/1e3:8.1f} us  per-step {float(r['TotalDurationNs'])/steps/1e3:8.1f} us")
print("total per step ms", tot / steps / 1e6)
