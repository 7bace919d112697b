#!/usr/bin/env python3
"""bench.py's coherent_motion figure next to the headline's, once per process (so that a library switch read at
start-up applies): prints one line.  scripts/ab_env.sh-style loops call it with the experiments library."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import schroedinger_amd as sa  # noqa: E402

ctx = sa.Context(0)
wl = bench.Workload(ctx, 8, seed=1, queues=2)
for _ in range(40):
    wl.step()
ctx.synchronize()
ctx.profile_enable(True)
ctx.profile_reset()
for _ in range(4):
    wl.step(alone=True)
ctx.synchronize()
head = ctx.profile_read()["obmc"][0] / 4
ctx.profile_enable(False)
co = bench.coherent_motion(wl)
print(json.dumps({"switches": {k: v for k, v in os.environ.items() if k.startswith("SCHRO_HIP_OBMC")},
                  "headline_obmc_ms": round(head, 4), "coherent_obmc_ms": co["obmc_ms_per_step"], "coherent_step_ms": co["ms_per_step"]}))
