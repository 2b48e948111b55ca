#!/usr/bin/env python3
"""Read bench.py's JSON line on stdin and print ms per step + each roofline (for quick A/B runs)."""
import json
import sys

for line in sys.stdin:
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    items = [d] + d.get("secondary", [])
    for it in items:
        roofs = [it["roofline"]] + it.get("roofline_other", []) if it.get("roofline") else []
        print(it.get("workload", it.get("config", {}).get("workload", "?"))[:40], "ms/step", it.get("ms_per_step"))
        for r in roofs:
            print("   ", {k: r[k] for k in r if k in ("kernel", "ms_per_launch", "achieved", "frac", "edges_out")})
