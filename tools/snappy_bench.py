#!/usr/bin/env python3
"""Time the device snappy decompressors alone (the multi-kernel pipeline comet_snappy2_inflate_pages and the one-wave-per-page kernel comet_snappy_inflate_pages) on Parquet-like pages: 1 MiB PLAIN pages of decimal(12,2)-as-INT64
(TPC-H l_extendedprice: ~4 output bytes per snappy element), of doubles (incompressible: 64 KiB literals) and of low-cardinality int32.
One JSON line: pages, decompressed bytes, kernel ms, GB/s of output."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=480)
    ap.add_argument("--page-bytes", type=int, default=1 << 20)
    ap.add_argument("--out", default="")
    ap.add_argument("--skip-one-wave", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--kinds", default="decimal_int64,double,int32_lowcard")
    ap.add_argument("--codec", default="snappy", choices=["snappy", "zstd"], help="zstd: the zstd pipeline (comet_zstd2_inflate_pages) on the same pages")
    ap.add_argument("--level", type=int, default=1, help="zstd compression level of the pages (Spark's parquet.compression.codec.zstd.level default is 3; pyarrow's 1)")
    a = ap.parse_args()
    import numpy as np
    import pyarrow as pa
    from datafusion_comet_amd import native
    rng = np.random.default_rng(1)
    n8 = a.page_bytes // 8
    kinds = {
        "decimal_int64": lambda: rng.integers(90_000, 10_000_000, n8).astype(np.int64).tobytes(),
        "double": lambda: rng.standard_normal(n8).tobytes(),
        "int32_lowcard": lambda: rng.integers(0, 50, a.page_bytes // 4).astype(np.int32).tobytes(),
    }
    def dict_indices(card, bw):
        """the index section of a dictionary-encoded page as parquet-cpp writes it: bit-packed runs of 63 groups of eight random indices"""
        def gen():
            out = bytearray()
            while len(out) < a.page_bytes:
                v = rng.integers(0, card, 504).astype(np.uint64)
                bits = np.zeros(504 * bw, np.uint8)
                for b in range(bw):
                    bits[b::bw] = (v >> np.uint64(b)) & np.uint64(1)
                out += bytes([(63 << 1) | 1]) + np.packbits(bits, bitorder="little").tobytes()
            return bytes(out[:a.page_bytes])
        return gen
    kinds["dict_idx_bw4"] = dict_indices(11, 4)        # l_discount / l_tax: 11 / 9 distinct values
    kinds["dict_idx_bw6"] = dict_indices(50, 6)        # l_quantity
    kinds["dict_idx_bw12"] = dict_indices(2526, 12)    # l_shipdate
    res = {"pages": a.pages, "page_bytes": a.page_bytes, "codec": a.codec}
    if a.codec == "zstd":
        res["level"] = a.level
    for name, gen in kinds.items():
        if name not in a.kinds.split(","):
            continue
        distinct = [gen() for _ in range(8)]
        comp = [pa.compress(p, codec="snappy", asbytes=True) if a.codec == "snappy" else pa.Codec("zstd", compression_level=a.level).compress(p, asbytes=True) for p in distinct]
        pages = [distinct[i % 8] for i in range(a.pages)]
        streams = [comp[i % 8] for i in range(a.pages)]
        total = sum(map(len, pages))
        res[name] = {"compressed_ratio": sum(map(len, streams)) / total}
        runs = (("pipeline", native.snappy2_inflate_pages), ("one_wave_per_page", native.snappy_inflate_pages)) if a.codec == "snappy" else (("pipeline", native.zstd2_inflate_pages),)
        for label, fn in runs:
            if label == "one_wave_per_page" and a.skip_one_wave:
                continue
            best = None
            for _ in range(3):
                r = fn(streams, [len(p) for p in pages])
                got, ms = r[0], r[1]
                best = ms if best is None else min(best, ms)
            assert a.no_check or all(g == w for g, w in zip(got, pages))
            res[name][label] = {"kernel_ms": best, "out_GBps": total / best / 1e6}
            if label == "pipeline":
                res[name][label]["pages_to_fallback"] = sum(1 for x in r[2] if x == 1)
    line = json.dumps(res)
    print(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
