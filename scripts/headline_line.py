#!/usr/bin/env python3
"""stdin: bench.py's JSON line -> value, ms per step and the kernel classes' ms per step (A/B runs)."""
import json, sys
d = json.loads(sys.stdin.read())
print(" ".join(sys.argv[1:]), d["value"], d["ms_per_step"],
      {k: v.get("ms_per_step", v.get("avg_ms")) for k, v in d["kernels"].items()})
