#!/usr/bin/env python3
"""Read one bench.py JSON line on stdin, print ms_per_step and the per-call kernel times whose name contains argv[1]."""
import json
import sys
d = json.loads(sys.stdin.read())
k = (d.get("kernel_ms_per_step_instrumented") or d.get("kernel_ms_per_step_warmup", {}))
sub = sys.argv[1] if len(sys.argv) > 1 else ""
print(round(d["ms_per_step"], 4), {n: v for n, v in k.items() if sub in n})
