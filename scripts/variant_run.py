#!/usr/bin/env python3
"""One of bench.py's decode variants by itself (GPU box; for rocprofv3 --kernel-trace --stats / --pmc passes):
   python3 scripts/variant_run.py prec=0            full pel, 8 x 2160p
   python3 scripts/variant_run.py w=1920 h=1080 frames=32
   python3 scripts/variant_run.py xblen=24 xbsep=16 steps=40"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

if __name__ == "__main__":
    kw = {k: (tuple(int(x) for x in v.split(",")) if "," in v else int(v)) for k, v in (a.split("=") for a in sys.argv[1:])}
    print(json.dumps(bench.decode_variant(0, **kw)))
