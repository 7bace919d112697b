#!/usr/bin/env python3
"""The 1 / 8 / 32-picture table of BASELINE config 2 (3-level DD(9,7) inverse wavelet, 1920x1080 4:2:0 s16):
one launch set per batch, one batch in flight and two.  Run it once per library / switch:

    python scripts/iiwt_1080p_table.py                                   # the product library
    SCHRO_HIP_LIB=schroedinger_amd/libschro_hip_exp.so SCHRO_HIP_IIWT_CHAIN=1 python scripts/iiwt_1080p_table.py
    SCHRO_HIP_LIB=schroedinger_amd/libschro_hip_exp.so SCHRO_HIP_IIWT_FUSE=2 python scripts/iiwt_1080p_table.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench                      # noqa: E402
import schroedinger_amd as sa     # noqa: E402

ctx = sa.Context(0)

rows = {}
for frames in (1, 8, 32):
    r = bench.iiwt_1080p(ctx, frames=frames, steps=48)
    rows[str(frames)] = {"one_batch_ms": r["median_ms"], "one_batch_frac_of_8TBs": r["frac_of_8TBs"],
                         "two_batches_ms": r["two_batches_in_flight"]["ms"],
                         "two_batches_frac_of_8TBs": r["two_batches_in_flight"]["frac_of_8TBs"]}
print(json.dumps({"iiwt_1080p_table": rows,
                  "switches": {k: v for k, v in os.environ.items() if k.startswith("SCHRO_HIP_")}}))
