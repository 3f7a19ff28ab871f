#!/usr/bin/env python3
"""HBM bytes of a program's kernels from the SIZE-CLASSED request counters of gfx950's L2 (TCC → EA), next to what FETCH_SIZE says.

MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies every read request at 64 B, and a coalesced streaming read issues 128-B requests — so it
is doubled; "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".  The join
kernels mix streaming reads (the probe columns) with gathers (table entries, rows of a run), whose requests need not be 128 B.  rocprofv3's
counter_defs.yaml lists, for gfx950, counters that carry the request size:

  TCC_EA0_RDREQ_128B / _64B / _32B      read requests by size
  TCC_EA0_RDREQ_DRAM_32B                32-byte units of the read requests that went to DRAM (a 64-B request counts 2, a 128-B request 4)
  TCC_EA0_WRREQ_WRITE_DRAM_32B          the same for writes

  python tools/pmc_sizes.py <out.txt> [--calib] -- <program …>      # six rocprofv3 passes (--pmc … --kernel-trace) over the program

--calib: the program is tools/pmc_calibrate.py --run (streaming reads of a known 4 GiB): prints counter ÷ truth per load width.
Per kernel name, averaged over its dispatches: requests by size, read bytes by the size classes, DRAM read / write bytes, and 2 × FETCH_SIZE's
formula (= 128 B × all read requests) for comparison, and FETCH_SIZE itself (doubled) with TCC_BUBBLE, the term of its gfx950 formula that is not a
plain request count: FETCH_SIZE = 128·BUBBLE + 64·(RDREQ − BUBBLE − RDREQ_32B) + 32·RDREQ_32B."""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile
from collections import defaultdict

PASSES = (("TCC_EA0_RDREQ_DRAM_32B_sum", "TCC_EA0_RDREQ_sum"),
          ("TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_32B_sum"),
          ("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"),
          ("FETCH_SIZE",), ("TCC_BUBBLE_sum", "TCC_EA0_RDREQ_sum"), ("WRITE_SIZE",))


def main():
    out = sys.argv[1]
    calib = "--calib" in sys.argv
    cmd = sys.argv[sys.argv.index("--") + 1:]
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    env = dict(os.environ, TMPDIR="/tmp")
    per = defaultdict(lambda: defaultdict(float))     # kernel → counter → sum over dispatches
    launches = defaultdict(int)
    errors = []
    for pi, counters in enumerate(PASSES):
        d = tempfile.mkdtemp(prefix="comet_sizes_", dir="/tmp")
        p = subprocess.run([rp, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "s", "--", *cmd], env=env, cwd="/tmp",
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if p.returncode != 0:
            errors.append(f"pass {pi} ({' '.join(counters)}): exit code {p.returncode}: {p.stdout[-300:]}")
        seen = set()
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                if calib:
                    kn = r["Kernel_Name"]
                    if "calib_read" not in kn:
                        continue
                    k = "calib lo8of16" if "lo8" in kn else "calib 4B" if "<unsigned int>" in kn else "calib 16B" if "CalibB16>" in kn else "calib 8B"
                per[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if pi == 0 and (k, r["Dispatch_Id"]) not in seen:
                    seen.add((k, r["Dispatch_Id"]))
                    launches[k] += 1
        shutil.rmtree(d, ignore_errors=True)
    with open(out, "w") as o:
        o.write("# " + " ".join(cmd) + "\n# rocprofv3 --pmc passes: " + " | ".join(" ".join(c) for c in PASSES) + "\n")
        for e in errors:
            o.write("# ERROR " + e + "\n")
        o.write("# per kernel, AVERAGE PER DISPATCH.  rd_by_class = 128·n128 + 64·n64 + 32·n32; rd_dram = 32 · RDREQ_DRAM_32B; wr_dram = 32 · WRREQ_WRITE_DRAM_32B;\n"
                "# fetch_x2 = 128 B · RDREQ (what doubling FETCH_SIZE assumes).  MB = 1e6 bytes.\n")
        o.write(f"{'kernel':28s} {'n':>4s} {'RDREQ':>12s} {'n128':>12s} {'n64':>12s} {'n32':>12s} {'rd_by_class_MB':>15s} {'rd_dram_MB':>11s} {'fetch_x2_MB':>12s} {'x2/dram':>8s} {'wr_dram_MB':>11s} {'WRREQ':>12s} {'wr64':>12s} {'2xFETCH_MB':>11s} {'BUBBLE':>12s} {'WRITE_SZ_MB':>11s}\n")
        rows = []
        for k, c in per.items():
            n = max(launches.get(k, 0), 1)
            g = lambda x: c.get(x, 0.0) / n
            rd_class = 128 * g("TCC_EA0_RDREQ_128B_sum") + 64 * g("TCC_EA0_RDREQ_64B_sum") + 32 * g("TCC_EA0_RDREQ_32B_sum")
            rd_dram = 32 * g("TCC_EA0_RDREQ_DRAM_32B_sum")
            rdreq = g("TCC_EA0_RDREQ_sum") / 2          # (collected in two of the passes)
            x2 = 128 * rdreq
            wr = 32 * g("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum")
            rows.append((-(rd_dram + wr) * n, f"{k[:28]:28s} {n:4d} {rdreq:12.0f} {g('TCC_EA0_RDREQ_128B_sum'):12.0f} {g('TCC_EA0_RDREQ_64B_sum'):12.0f} {g('TCC_EA0_RDREQ_32B_sum'):12.0f} "
                         f"{rd_class / 1e6:15.1f} {rd_dram / 1e6:11.1f} {x2 / 1e6:12.1f} {x2 / max(rd_dram, 1):8.3f} {wr / 1e6:11.1f} {g('TCC_EA0_WRREQ_sum'):12.0f} {g('TCC_EA0_WRREQ_64B_sum'):12.0f} {2 * 1024 * g('FETCH_SIZE') / 1e6:11.1f} {g('TCC_BUBBLE_sum'):12.0f} {1024 * g('WRITE_SIZE') / 1e6:11.1f}\n"))
        for _, line in sorted(rows)[:40]:
            o.write(line)
        if calib:
            o.write(f"# truth: every calib dispatch reads {4 << 30} bytes = {(4 << 30) / 1e6:.1f} MB (lo8of16 touches every line of it)\n")
    sys.stdout.write(open(out).read())


if __name__ == "__main__":
    main()
