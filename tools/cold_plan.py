#!/usr/bin/env python3
"""What a NEW plan shape costs: the hiprtc compilation behind createPlan with an EMPTY code-object cache, and the same call again in a
fresh process over the now-warm disk cache (the timed region of bench.py only ever sees the warm cache, and the
reference has no such cliff — so the cliff is reported).  Each measurement is its own process (the in-memory cache dies with it);
COMET_JIT_CACHE_DIR points at a temporary directory, the cache that ships next to libcomet.so is not touched.

  python tools/cold_plan.py [--query q1|q6] [--out f.json]      one JSON line"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, time, json
sys.path.insert(0, %r)
from datafusion_comet_amd import native, tpch
native.lib()
plans = {"q1": lambda: [tpch.q1_plan()], "q6": lambda: [tpch.q6_plan()]}[%r]()
bs = [p.encode() for p in plans]
t0 = time.perf_counter()
for b in bs:
    native.compile_plan(b)
print(json.dumps({"ms": (time.perf_counter() - t0) * 1e3, "plans": len(bs)}))
"""


def measure(query: str, cache_dir: str) -> dict:
    # (ROCm 7.2's code object manager keeps a cache of its own under ~/.cache: off, or a plan some earlier process compiled would not be cold — 16 ms instead of 550)
    env = dict(os.environ, COMET_JIT_CACHE_DIR=cache_dir, AMD_COMGR_CACHE="0")
    p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, query)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    if p.returncode != 0:
        raise RuntimeError(p.stderr[-600:])
    return json.loads(p.stdout.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--query", default="q1", choices=["q1", "q6"])
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    d = tempfile.mkdtemp(prefix="comet_cold_jit_")
    try:
        cold = measure(a.query, d)
        objs = len([f for f in os.listdir(d) if f.endswith(".hsaco")])
        warm = measure(a.query, d)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    line = {"query": a.query, "plans": cold["plans"], "cold_create_plan_ms": cold["ms"], "warm_disk_cache_create_plan_ms": warm["ms"], "code_objects": objs,
            "note": "hiprtc compilation of the plan's fused kernels for gfx950 with an empty code-object cache (own process), then the same call "
                    "in a fresh process over the disk cache it left; in-process repeats hit the in-memory cache (µs)"}
    s = json.dumps(line)
    print(s, flush=True)
    if a.out:
        with open(a.out, "w") as f:
            f.write(s + "\n")


if __name__ == "__main__":
    main()
