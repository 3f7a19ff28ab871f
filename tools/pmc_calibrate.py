#!/usr/bin/env python3
"""Calibrates rocprofv3's FETCH_SIZE counter on this GPU for the load widths the fused pipelines issue.  MI355X_MICROARCH.md states that on
gfx950 FETCH_SIZE reports HALF the bytes of a coalesced streaming read and calibrates that for 16 B/lane loads only; the Q1 / Q6 kernels also
issue 4 B/lane (Date32) and 8 B/lane (the low half of a narrow Decimal128, every second 8 bytes of the buffer) loads.

  python tools/pmc_calibrate.py --run        # what rocprofv3 profiles: three streaming reads of a 4 GiB buffer per width
  python tools/pmc_calibrate.py              # runs rocprofv3 --pmc FETCH_SIZE over itself and prints counter vs truth

Output (kept under profiles/): per width the counter (KiB → bytes), the true byte count and their ratio."""
import argparse
import csv
import ctypes
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NBYTES = 4 << 30


def run():
    import datafusion_comet_amd  # noqa: F401 — before torch: the JIT then compiles with the installed ROCm's compiler (see that module)
    import torch
    from datafusion_comet_amd import native
    lib = native.lib()
    lib.comet_calib_read.restype = ctypes.c_int32
    lib.comet_calib_read.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    buf = torch.randint(0, 255, (NBYTES,), dtype=torch.uint8, device="cuda:0")
    sink = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    for w in (4, 8, 16, 816):     # 816 = the low 8 bytes of every 16 (how narrow Decimal128 columns are read)
        for _ in range(3):
            assert lib.comet_calib_read(buf.data_ptr(), NBYTES, w, sink.data_ptr(), st) == 0
        torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--run", action="store_true")
    a = ap.parse_args()
    if a.run:
        return run()
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    d = tempfile.mkdtemp(prefix="comet_calib_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.check_call([rp, "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "c", "--", sys.executable, os.path.abspath(__file__), "--run"],
                          env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    vals = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "calib_read" in k and r["Counter_Name"] == "FETCH_SIZE":
                w = 816 if "lo8" in k else 4 if "<unsigned int>" in k else 16 if "CalibB16>" in k else 8
                vals.setdefault(w, []).append(float(r["Counter_Value"]))
    shutil.rmtree(d, ignore_errors=True)
    print(f"# FETCH_SIZE calibration: streaming read of {NBYTES} bytes, 256 x 8 blocks, rocprofv3 --pmc FETCH_SIZE (KiB)")
    print("# lane_bytes (816 = the low 8 bytes of every 16)  launches  FETCH_SIZE_bytes(avg)  true_bytes  true/counter")
    for w in sorted(vals):
        v = vals[w][1:] or vals[w]
        c = 1024.0 * sum(v) / len(v)
        print(f"{w:10d}  {len(v):8d}  {c:21.0f}  {NBYTES:10d}  {NBYTES / c:12.4f}")


if __name__ == "__main__":
    main()
