#!/bin/bash
# One LTS / PDRA step's dispatches in start order with the idle time in front of each (rocprofv3 kernel trace):
#   gpurun -- 'bash tools/trace_lts.sh [bench args, default --config C5]'   -> gpurun_out/trace_lts.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ARGS=${*:---config C5}
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
rm -rf "$ROOT/gpurun_out/tr_lts"
rocprofv3 --kernel-trace --output-format csv -d "$ROOT/gpurun_out/tr_lts" -o run -- python3 "$ROOT/bench.py" $ARGS --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-optimizer > "$ROOT/gpurun_out/tr_lts.log" 2>&1
cd "$ROOT" && python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tr_lts/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "march_kernel<0" in r["Kernel_Name"]]
ts = [int(rows[i]["Start_Timestamp"]) for i in starts]
gaps = [ts[i + 2] - ts[i] for i in range(0, len(ts) - 2, 2)]          # two marches per step
k = min(range(len(gaps)), key=lambda i: abs(gaps[i] - sorted(gaps)[len(gaps) // 2]))   # a median step
a, b = starts[2 * k], starts[2 * k + 2]
t0, busy_end, idle, out = int(rows[a]["Start_Timestamp"]), int(rows[a]["Start_Timestamp"]), 0.0, []
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = max(0, s - busy_end) / 1e3
    idle += g
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    out.append(f"+{(s - t0) / 1e3:8.1f} dur {(e - s) / 1e3:7.1f} idle {g:6.1f} q{r.get('Queue_Id', '?')} {n[:70]}")
    busy_end = max(busy_end, e)
out.append(f"step wall {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, idle {idle:.1f} us, {b - a} dispatches; steps (us): {[round(g / 1e3) for g in gaps]}")
open("gpurun_out/trace_lts.txt", "w").write("\n".join(out) + "\n")
print(out[-1])
for l in out:
    if float(l.split("idle")[1].split()[0]) > 8: print(l)
PY
find "$ROOT/gpurun_out/tr_lts" -type f -size +5M -delete
