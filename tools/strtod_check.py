"""Strings → Double on the GPU against Python's float(): which inputs come back NULL (a diagnostic for a compiler-dependent result)."""
import importlib.util
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import datafusion_comet_amd  # noqa: F401
import numpy as np
import pyarrow as pa
from datafusion_comet_amd import native, serde as S

spec = importlib.util.spec_from_file_location("sd_cpu", os.path.join(ROOT, "tests", "test_strtod_cpu.py"))
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
vals = [v for v in m._numbers(random.Random(5), 6000) if len(v) < 400]
t = pa.table({"s": pa.array(vals, pa.utf8()), "k": pa.array(np.zeros(len(vals), np.int32))})
STR = S.T_STRING
plan = S.project(S.scan([STR, S.T_INT32]), [S.cast(S.col(0, STR), S.T_DOUBLE, S.LEGACY)])
out = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], 1, plan.encode()))
g = out.column(0).combine_chunks()
print("toolchain:", native.jit_toolchain())
bad = [i for i in range(len(vals)) if not g[i].is_valid]
print(len(vals), "inputs,", len(bad), "NULL")
import collections
feat = collections.Counter()
for i in bad:
    v = vals[i]
    feat[("len%8=" + str(len(v) % 8), )] += 1
print(sorted(feat.items()))
lens_bad = collections.Counter(len(vals[i]) for i in bad)
lens_all = collections.Counter(len(v) for v in vals)
print("NULL by length:", sorted(lens_bad.items())[:40])
print("all  by length:", sorted(lens_all.items())[:40])
def feats(v):
    return ("exp" if ("e" in v.lower().rstrip("df")) and not v.strip().lower().lstrip("+-").startswith(("inf", "nan")) else "noexp", "dot" if "." in v else "nodot", "sign" if v.strip()[:1] in "+-" else "nosign")
badset = set(bad)
tab = collections.Counter()
for i, v in enumerate(vals):
    tab[feats(v) + (("NULL" if i in badset else "ok"),)] += 1
for k in sorted(tab):
    print(k, tab[k])
print("NULL without exponent:", [vals[i] for i in bad if feats(vals[i])[0] == "noexp"][:30])
print("ok with exponent:", [vals[i] for i in range(len(vals)) if i not in badset and feats(vals[i])[0] == "exp"][:30])
for i in bad[:12]:
    print(repr(vals[i]))
ok = [i for i in range(len(vals)) if g[i].is_valid][:15]
print("valid examples:", [vals[i] for i in ok])
