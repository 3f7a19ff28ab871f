"""CPU oracle: evaluates a Comet plan (the ``serde`` Python objects the plan bytes were made from) over
pyarrow tables, operator at a time, the way the reference's DataFusion plan does.

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product (datafusion-comet_amd/) never imports this module.

Heavy per-row arithmetic is in ``comet_oracle.c`` (plain C restatement, each function citing the
reference); this file restates the operator semantics:
  FilterExec        keep rows whose predicate is TRUE and valid     (planner.rs:1230-1247)
  ProjectionExec    one materialised array per expression           (operators/projection.rs:37-74)
  AggregateExec     Partial → state columns, Final → values         (planner.rs:1248-1384, agg_funcs/*.rs)
  expressions       planner.rs:976-1132 (decimal path selection), :600-649 (CheckOverflow fusion)
Exact Python ``int`` versions of the decimal rules live in ``pyint.py`` and cross-check the C code.

Pinned: against the reference's known-answer vectors (tests/golden/reference_kats.json), its Parquet fixture files, and — end to end —
its own TPC-H scale-factor-1 answers (tests/golden/tpch_sf1/q{1,3,4,5,6,7,8,11,12,14,17,18,19,21,22}.sql.out, over tables regenerated with dbgen's random streams:
tests/test_tpch_golden_cpu.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import pyarrow as pa

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcomet_oracle.so")

DEC128 = np.dtype([("lo", "<u8"), ("hi", "<i8")])


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "comet_oracle.c")):
        build()
    return ctypes.CDLL(_SO)


C = _lib()
_vp = ctypes.c_void_p


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_vp)


class SumDecState(ctypes.Structure):
    # C: struct { __int128 sum; int32_t has_sum; int32_t is_empty; } → 16-byte aligned, sizeof == 32
    _fields_ = [("sum", ctypes.c_uint64 * 2), ("has_sum", ctypes.c_int32), ("is_empty", ctypes.c_int32), ("_pad", ctypes.c_uint64)]


class AvgDecState(ctypes.Structure):
    _fields_ = [("sum", ctypes.c_uint64 * 2), ("count", ctypes.c_int64), ("is_not_null", ctypes.c_int32), ("pad", ctypes.c_int32)]


def dec_to_int(a: np.ndarray, i: int) -> int:
    return (int(a["hi"][i]) << 64) | int(a["lo"][i])


def ints_to_dec(vals) -> np.ndarray:
    out = np.zeros(len(vals), dtype=DEC128)
    for i, v in enumerate(vals):
        v = int(v)
        out["lo"][i] = v & 0xFFFFFFFFFFFFFFFF
        out["hi"][i] = v >> 64
    return out


def i64_to_dec(v: np.ndarray) -> np.ndarray:
    out = np.zeros(len(v), dtype=DEC128)
    out["lo"] = v.astype(np.int64).view(np.uint64)
    out["hi"] = v.astype(np.int64) >> 63
    return out


def _limbs_to_int(limbs) -> int:
    v = (int(limbs[1]) << 64) | int(limbs[0])
    return v - (1 << 128) if v >> 127 else v


# --------------------------------------------------------------------------- columns


@dataclass
class Col:
    dtype: object                 # serde.DataType
    values: np.ndarray
    valid: Optional[np.ndarray]   # bool array, None = all valid

    def __len__(self):
        return len(self.values)

    def ok(self) -> np.ndarray:
        if self.valid is not None:
            return self.valid
        # (all valid: one read-only array of ones per column, not one per call — row loops ask for every row)
        o = self.__dict__.get("_all_ok")
        if o is None or len(o) != len(self.values):
            o = np.ones(len(self.values), bool)
            o.flags.writeable = False
            self.__dict__["_all_ok"] = o
        return o


def _np_dtype(S, t):
    return {S.BOOL: np.bool_, S.INT8: np.int8, S.INT16: np.int16, S.INT32: np.int32, S.INT64: np.int64, S.FLOAT: np.float32,
            S.DOUBLE: np.float64, S.DATE: np.int32, S.TIMESTAMP: np.int64, S.TIMESTAMP_NTZ: np.int64}.get(t.type_id)


def col_from_arrow(S, arr: pa.Array, t) -> Col:
    if isinstance(arr, pa.ChunkedArray):
        arr = arr.combine_chunks()
    n = len(arr)
    valid = None
    if arr.null_count:
        valid = np.array(arr.is_valid().to_numpy(zero_copy_only=False), dtype=bool)
    if t.type_id == getattr(S, "LIST", -1):      # lists of flat elements as Python lists (size / element_at / array_contains; split's results)
        vals = np.empty(n, dtype=object)
        for i, v in enumerate(arr.to_pylist()):
            vals[i] = v
        return Col(t, vals, valid)
    if t.type_id == S.DECIMAL:
        buf = arr.buffers()[1]
        vals = np.frombuffer(buf, dtype=DEC128)[arr.offset:arr.offset + n].copy()
    elif t.type_id == S.BOOL:
        vals = np.array(arr.fill_null(False).to_numpy(zero_copy_only=False), dtype=bool)
    elif t.type_id == S.STRING:
        vals = arr.to_numpy(zero_copy_only=False)       # an object array of str (None in NULL slots)
        if vals.dtype != object:
            vals = vals.astype(object)
    else:
        nt = _np_dtype(S, t)
        buf = arr.buffers()[1]
        vals = np.frombuffer(buf, dtype=nt)[arr.offset:arr.offset + n].copy()
    return Col(t, vals, valid)


def col_to_arrow(S, c: Col) -> pa.Array:
    n = len(c)
    t = c.dtype
    mask = None if c.valid is None or c.valid.all() else ~c.valid
    if t.type_id == getattr(S, "LIST", -1):      # (split's result: lists of strings as Python lists)
        et = {S.STRING: pa.utf8(), S.INT32: pa.int32(), S.INT64: pa.int64(), S.DOUBLE: pa.float64(), S.DATE: pa.date32()}[t.element.type_id]
        return pa.array([None if (mask is not None and mask[i]) else list(c.values[i]) for i in range(n)], type=pa.list_(pa.field("item", et, nullable=t.contains_null)))
    if t.type_id == S.DECIMAL:
        vals = c.values.copy()
        if mask is not None:
            vals[mask] = np.zeros(1, DEC128)[0]
        vb = None
        nulls = 0
        if mask is not None:
            vb = pa.py_buffer(np.packbits(~mask, bitorder="little").tobytes())
            nulls = int(mask.sum())
        return pa.Array.from_buffers(pa.decimal128(t.precision, t.scale), n, [vb, pa.py_buffer(vals.tobytes())], null_count=nulls)
    pt = {S.BOOL: pa.bool_(), S.INT8: pa.int8(), S.INT16: pa.int16(), S.INT32: pa.int32(), S.INT64: pa.int64(),
          S.FLOAT: pa.float32(), S.DOUBLE: pa.float64(), S.DATE: pa.date32(), S.STRING: pa.utf8(), S.TIMESTAMP: pa.timestamp("us", tz="UTC"),
          S.TIMESTAMP_NTZ: pa.timestamp("us")}[t.type_id]
    if t.type_id in (S.TIMESTAMP, S.TIMESTAMP_NTZ):
        return pa.array(c.values.astype(np.int64), type=pa.int64(), mask=mask).cast(pt)
    if t.type_id == S.DATE:
        return pa.array(c.values.astype(np.int32), type=pa.int32(), mask=mask).cast(pa.date32())
    if t.type_id == S.STRING:
        return pa.array([None if (mask is not None and mask[i]) else c.values[i] for i in range(n)], type=pa.utf8())
    return pa.array(c.values, type=pt, mask=mask)


_LIBM_LIB = None


def _libm_fn(name: str, nargs: int = 1):
    """a function of the platform's libm (glibc here as on the reference's hosts): C semantics for NaN, infinities and domain errors, no Python exceptions"""
    global _LIBM_LIB
    import ctypes.util
    if _LIBM_LIB is None:
        _LIBM_LIB = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    fn = getattr(_LIBM_LIB, name)
    fn.restype = ctypes.c_double
    fn.argtypes = [ctypes.c_double] * nargs
    return fn


def crate_pattern_to_python(pat: str):
    """the crate's pattern in Python's `re` syntax, for the part both read alike: $ / \\z are "at the very end" there (\\Z here) unless (?m)"""
    import re
    multiline = re.match(r"\(\?[is]*m[ism]*\)", pat) is not None
    return re.compile((pat if multiline else re.sub(r"(?<!\\)\$", r"\\Z", pat)).replace("\\z", "\\Z"))


def find_iter_like_the_crate(rx, text: str):
    """regex::Regex::find_iter (regex-automata util::iter::Searcher::advance): successive leftmost matches, each search starting where the last
    match ended; an EMPTY match that ends where the previous match ended is dropped and the search repeated one character on."""
    at, last_end = 0, None
    while at <= len(text):
        m = rx.search(text, at)
        if m is None:
            return
        if m.start() == m.end() and m.end() == last_end:
            at += 1
            if at > len(text):
                return
            m = rx.search(text, at)
            if m is None:
                return
        yield m
        at = last_end = m.end()


def split_like_the_crate(pat: str, text: str, limit: int):
    """push_split_parts (string_funcs/split.rs:434-472) over Regex::split / find_iter as above"""
    rx = crate_pattern_to_python(pat)
    parts, last = [], 0
    for count, m in enumerate(find_iter_like_the_crate(rx, text)):
        if limit > 0 and count >= limit - 1:
            break
        parts.append(text[last:m.start()])
        last = m.end()
    parts.append(text[last:])
    if limit == 0:
        while parts and parts[-1] == "":
            parts.pop()
        if not parts:
            parts = [""]
    return parts


# --------------------------------------------------------------------------- expression evaluation
SUBQUERIES = {}      # scalar subquery id → Python value (None = NULL): what a test registered for the plan it runs (CometScalarSubquery.setSubquery's part)



class OracleError(Exception):
    """A Spark-semantic error the reference would raise (ANSI overflow, …)."""


class Evaluator:
    def __init__(self, S):
        self.S = S

    # ---- helpers
    def _and_valid(self, a, b):
        if a is None:
            return b
        if b is None:
            return a
        return a & b

    def _as_dec(self, c: Col) -> np.ndarray:
        return c.values

    def eval(self, e, cols: List[Col], n: int) -> Col:
        S = self.S
        k = e.kind
        if k == "bound":
            return cols[e.index]
        if k == "literal":
            return self._literal(e, n)
        if k in ("eq", "neq", "lt", "lt_eq", "gt", "gt_eq"):
            a, b = self.eval(e.children[0], cols, n), self.eval(e.children[1], cols, n)
            return Col(S.T_BOOL, self._compare(k, a, b), self._and_valid(a.valid, b.valid))
        if k == "eq_null_safe":
            a, b = self.eval(e.children[0], cols, n), self.eval(e.children[1], cols, n)
            eq = self._compare("eq", a, b)
            return Col(S.T_BOOL, (a.ok() & b.ok() & eq) | (~a.ok() & ~b.ok()), None)
        if k in ("and_", "or_"):
            a, b = self.eval(e.children[0], cols, n), self.eval(e.children[1], cols, n)
            av, bv, ao, bo = a.values.astype(bool), b.values.astype(bool), a.ok(), b.ok()
            if k == "and_":   # Kleene: FALSE dominates NULL
                val = ao & av & bo & bv
                ok = (ao & ~av) | (bo & ~bv) | (ao & bo)
            else:
                val = (ao & av) | (bo & bv)
                ok = (ao & av) | (bo & bv) | (ao & bo)
            return Col(S.T_BOOL, val, None if ok.all() else ok)
        if k in ("hour", "minute", "second"):
            # SparkHour / SparkMinute / SparkSecond (datetime_funcs/extract_date_part.rs:83-110): of the session zone's wall clock
            from . import strcast as C
            a = self.eval(e.children[0], cols, n)
            tz = getattr(e, "timezone", None) or "UTC"
            out = np.zeros(n, np.int32)
            for i in range(n):
                if a.ok()[i]:
                    us = int(a.values[i])
                    sec = (C.utc_to_local_us(tz, us) if a.dtype.type_id == S.TIMESTAMP else us) % 86_400_000_000 // 1_000_000
                    out[i] = {"hour": sec // 3600, "minute": sec // 60 % 60, "second": sec % 60}[k]
            return Col(S.T_INT32, out, a.valid)
        if k == "subquery":
            # Subquery{id, datatype} (expressions/subquery.rs:72-180): the value the JVM holds for the id — here the evaluator's `subqueries` table — as a scalar
            v = SUBQUERIES.get(int(e.value))
            lit = S.lit(v, e.dtype)
            return self.eval(lit, cols, n)
        if k == "list_extract":
            # ListExtract (array_funcs/list_extract.rs:229-320): GetArrayItem counts from 0, element_at from 1 (negative: from the end; 0: INVALID_INDEX_OF_ZERO);
            # outside the list: NULL, or the error under ANSI; a NULL list / ordinal / element: NULL
            a, o = self.eval(e.children[0], cols, n), self.eval(e.children[1], cols, n)
            one = bool(getattr(e, "one_based", False))
            out, ok = [None] * n, np.zeros(n, bool)
            for i in range(n):
                if not (a.ok()[i] and o.ok()[i]):
                    continue
                lst, idx = a.values[i], int(o.values[i])
                if one:
                    if idx == 0:
                        raise OracleError("INVALID_INDEX_OF_ZERO")
                    pos = idx - 1 if 0 < idx <= len(lst) else (len(lst) + idx if idx < 0 and -idx <= len(lst) else None)
                else:
                    pos = idx if 0 <= idx < len(lst) else None
                if pos is None:
                    if e.fail_on_error:
                        raise OracleError(("INVALID_ARRAY_INDEX_IN_ELEMENT_AT" if one else "INVALID_ARRAY_INDEX") + ' "indexValue":%d,"arraySize":%d' % (idx, len(lst)))
                    continue
                if lst[pos] is not None:
                    out[i], ok[i] = lst[pos], True
            et = a.dtype.element
            if et.type_id == S.STRING:
                return Col(et, np.array(out, dtype=object), ok)
            if et.type_id == S.DATE:
                import datetime
                out = [(v - datetime.date(1970, 1, 1)).days if v is not None else 0 for v in out]
            return Col(et, np.array([0 if v is None else v for v in out], dtype=_np_dtype(S, et)), ok)
        if k == "unix_timestamp":
            # SparkUnixTimestamp (datetime_funcs/unix_timestamp.rs:70-150): floor(µs / 10^6) of an instant or a TIMESTAMP_NTZ; a date's midnight in the zone as an instant
            from . import strcast as C
            a = self.eval(e.children[0], cols, n)
            tz = getattr(e, "timezone", None) or "UTC"
            out = np.zeros(n, np.int64)
            for i in range(n):
                if a.ok()[i]:
                    out[i] = (C.local_to_utc_us(tz, int(a.values[i]) * 86_400_000_000) if a.dtype.type_id == S.DATE else int(a.values[i])) // 1_000_000
            return Col(S.T_INT64, out, a.valid)
        if k == "trunc_timestamp":
            # timestamp_trunc (kernels/temporal.rs:179-270, 587-625): the zone's wall clock cut to the unit (fixed-offset zones and UTC here)
            import datetime
            from . import strcast as C
            a = self.eval(e.children[0], cols, n)
            tz = getattr(e, "timezone", None) or "UTC"
            u = e.children[1].value.upper()
            U = datetime.datetime(1970, 1, 1)
            out = np.zeros(n, np.int64)
            for i in range(n):
                if not a.ok()[i]:
                    continue
                us = int(a.values[i])
                loc = C.utc_to_local_us(tz, us) if a.dtype.type_id == S.TIMESTAMP else us
                t = U + datetime.timedelta(microseconds=loc)
                z = dict(hour=0, minute=0, second=0, microsecond=0)
                if u in ("YEAR", "YYYY", "YY"): t = t.replace(month=1, day=1, **z)
                elif u == "QUARTER": t = t.replace(month=(t.month - 1) // 3 * 3 + 1, day=1, **z)
                elif u in ("MONTH", "MON", "MM"): t = t.replace(day=1, **z)
                elif u == "WEEK": t = (t - datetime.timedelta(days=t.weekday())).replace(**z)
                elif u in ("DAY", "DD"): t = t.replace(**z)
                elif u == "HOUR": t = t.replace(minute=0, second=0, microsecond=0)
                elif u == "MINUTE": t = t.replace(second=0, microsecond=0)
                elif u == "SECOND": t = t.replace(microsecond=0)
                elif u == "MILLISECOND": t = t.replace(microsecond=t.microsecond // 1000 * 1000)
                elif u != "MICROSECOND": raise OracleError("Unsupported format: %r for function 'timestamp_trunc'" % u)
                out[i] = (t - U) // datetime.timedelta(microseconds=1) - (loc - us)
            return Col(a.dtype, out, a.valid)
        if k == "not_":
            a = self.eval(e.children[0], cols, n)
            return Col(S.T_BOOL, ~a.values.astype(bool), a.valid)
        if k == "is_null":
            a = self.eval(e.children[0], cols, n)
            return Col(S.T_BOOL, ~a.ok(), None)
        if k == "is_not_null":
            a = self.eval(e.children[0], cols, n)
            return Col(S.T_BOOL, a.ok().copy(), None)
        if k in ("add", "subtract", "multiply", "divide", "remainder", "integral_divide"):
            a, b = self.eval(e.children[0], cols, n), self.eval(e.children[1], cols, n)
            return self._arith(e, a, b)
        if k == "check_overflow":
            return self._check_overflow(e, cols, n)
        if k == "cast":
            return self._cast(e, self.eval(e.children[0], cols, n))
        if k == "if_":
            c, t, f = (self.eval(x, cols, n) for x in e.children)
            sel = c.ok() & c.values.astype(bool)
            vals = np.where(sel, t.values, f.values) if t.dtype.type_id != S.DECIMAL else np.where(sel, t.values, f.values)
            ok = np.where(sel, t.ok(), f.ok())
            return Col(t.dtype, vals, None if ok.all() else ok)
        if k in ("bit_and", "bit_or", "bit_xor", "shift_left", "shift_right"):
            # Spark BitwiseAnd / Or / Xor, ShiftLeft / ShiftRight with Java semantics (count modulo the value's width, arithmetic >>)
            a, b = self.eval(e.children[0], cols, n), self.eval(e.children[1], cols, n)
            bits = {S.INT8: 8, S.INT16: 16, S.INT32: 32, S.INT64: 64}[a.dtype.type_id]
            wbits = 64 if bits == 64 else 32
            out = []
            for x, y in zip(a.values.tolist(), b.values.tolist()):
                if k == "bit_and":
                    v = x & y
                elif k == "bit_or":
                    v = x | y
                elif k == "bit_xor":
                    v = x ^ y
                elif k == "shift_left":
                    v = x << (y & (wbits - 1))
                else:
                    v = x >> (y & (wbits - 1))
                v &= (1 << wbits) - 1                      # int / long arithmetic first …
                v = v - (1 << wbits) if v >> (wbits - 1) else v
                v &= (1 << bits) - 1                       # … then the narrowing cast of Byte / Short results
                out.append(v - (1 << bits) if v >> (bits - 1) else v)
            return Col(a.dtype, np.array(out, dtype=_np_dtype(S, a.dtype)), self._and_valid(a.valid, b.valid))
        if k == "like":
            # Expr.like → DataFusion LikeExpr → arrow-string `like` (SQL LIKE, escape `\`, `_` = one character, `%` = any run,
            # newlines included); restated with Python's regex engine over code points — a different algorithm from the device matcher
            import re
            a, pat = self.eval(e.children[0], cols, n), e.children[1].value
            rx, i = "", 0
            while i < len(pat):
                ch = pat[i]
                if ch == "\\" and i + 1 < len(pat):
                    rx += re.escape(pat[i + 1])
                    i += 2
                    continue
                rx += ".*" if ch == "%" else "." if ch == "_" else re.escape(ch)
                i += 1
            cre = re.compile(rx, re.DOTALL)
            vals = np.array([bool(cre.fullmatch(v)) if v is not None else False for v in a.values], dtype=bool)
            return Col(S.T_BOOL, vals & a.ok(), a.valid)
        if k == "rlike":
            # predicate_funcs/rlike.rs:47-90: regex::Regex::is_match — an unanchored search over Unicode scalar values; `$` is the end of the
            # text only (Python's `$` would also match before a trailing newline, hence \Z).  Restated with Python's backtracking engine
            # for the constructs both engines read alike; the device walks a DFA built by csrc/regex.cpp
            import re
            a, pat = self.eval(e.children[0], cols, n), e.children[1].value
            rx, i, in_class = "", 0, False
            while i < len(pat):
                ch = pat[i]
                if ch == "\\" and i + 1 < len(pat):
                    rx += pat[i:i + 2]
                    i += 2
                    continue
                if ch == "[":
                    in_class = True
                elif ch == "]":
                    in_class = False
                rx += "\\Z" if (ch == "$" and not in_class) else ch
                i += 1
            cre = re.compile(rx)
            vals = np.array([bool(cre.search(v)) if v is not None else False for v in a.values], dtype=bool)
            return Col(S.T_BOOL, vals & a.ok(), a.valid)
        if k == "scalar_func":
            return self._scalar_func(e, cols, n)
        if k == "case_when":
            # planner.rs:677-704 → DataFusion CaseExpr (no base expression): first WHEN that is TRUE wins, else ELSE / NULL
            nw = e.index
            thens = [self.eval(x, cols, n) for x in e.children[nw:2 * nw]]
            if len(e.children) == 2 * nw + 1:
                r = self.eval(e.children[2 * nw], cols, n)
                vals, ok = r.values.copy(), r.ok().copy()
            else:
                vals, ok = thens[0].values.copy(), np.zeros(n, bool)
            for i in range(nw - 1, -1, -1):
                c = self.eval(e.children[i], cols, n)
                sel = c.ok() & c.values.astype(bool)
                vals = np.where(sel, thens[i].values, vals)
                ok = np.where(sel, thens[i].ok(), ok)
            return Col(thens[0].dtype, vals, None if ok.all() else ok)
        if k == "in_":
            v = self.eval(e.children[0], cols, n)
            hit = np.zeros(n, bool)
            anynull = np.zeros(n, bool)
            for item in e.children[1:]:
                li = self.eval(item, cols, n)
                hit |= li.ok() & self._compare("eq", v, li)
                anynull |= ~li.ok()
            ok = v.ok() & (hit | ~anynull)
            return Col(S.T_BOOL, ~hit if e.negated else hit, None if ok.all() else ok)
        if k == "unary_minus":
            # NegativeExpr (math_funcs/negative.rs:100-160): integers wrap in LEGACY (the minimum negates onto itself) and raise ARITHMETIC_OVERFLOW
            # with fail_on_error — only for VALID rows —, floats and decimals change sign
            a = self.eval(e.children[0], cols, n)
            tid = a.dtype.type_id
            if tid in (S.INT8, S.INT16, S.INT32, S.INT64):
                bits = {S.INT8: 8, S.INT16: 16, S.INT32: 32, S.INT64: 64}[tid]
                lo = -(1 << (bits - 1))
                x = a.values.astype(np.int64)
                if e.fail_on_error and ((x == lo) & a.ok()).any():
                    raise OracleError("ARITHMETIC_OVERFLOW")
                with np.errstate(over="ignore"):
                    r = np.where(x == lo, lo, -x)
                return Col(a.dtype, r.astype(_np_dtype(S, a.dtype)), a.valid)
            if tid in (S.FLOAT, S.DOUBLE):
                return Col(a.dtype, -a.values, a.valid)
            if tid == S.DECIMAL:
                return Col(a.dtype, ints_to_dec([-dec_to_int(a.values, i) for i in range(n)]), a.valid)
            raise NotImplementedError(f"oracle: unary minus over {a.dtype}")
        raise NotImplementedError(f"oracle: expression {k}")

    def _scalar_func(self, e, cols, n) -> Col:
        """The exactly-defined ScalarFunc subset (comet_scalar_funcs.rs → math_funcs/{ceil,floor,abs}.rs, predicate_funcs/is_nan.rs,
        DataFusion sqrt / signum / date_part)."""
        S = self.S
        f = e.value
        if f in ("datepart", "date_part"):
            import datetime
            part = e.children[0].value.lower()
            a = self.eval(e.children[1], cols, n)
            out = np.zeros(n, np.int32)
            for i in range(n):
                if a.ok()[i]:
                    d = datetime.date(1970, 1, 1) + datetime.timedelta(days=int(a.values[i]))
                    out[i] = {"year": d.year, "month": d.month, "day": d.day, "quarter": (d.month - 1) // 3 + 1, "dow": (d.weekday() + 1) % 7,
                              "doy": d.timetuple().tm_yday, "isodow": d.isoweekday(), "week": d.isocalendar()[1]}[part]
            return Col(S.T_INT32, out, a.valid)
        if f == "xxhash64":
            # spark_xxhash64 (hash_funcs/xxhash64.rs:31-82): XXH64 (the `xxhash` package here, twox-hash there) of each non-NULL value's
            # little-endian bytes, seeded with the running hash; value encodings as for murmur3 (hash_funcs/utils.rs:50-225)
            import xxhash
            seed = e.children[-1]
            if seed.kind != "literal" or seed.value is None or seed.dtype.type_id != S.INT64:
                raise OracleError("The seed of function xxhash64 must be an Int64 scalar value")
            h = [seed.value & 0xFFFFFFFFFFFFFFFF] * n
            for ch in e.children[:-1]:
                a = self.eval(ch, cols, n)
                ok = a.ok()
                tid = a.dtype.type_id
                for i in range(n):
                    if not ok[i]:
                        continue
                    if tid == S.DECIMAL:
                        v = dec_to_int(a.values, i)
                        b = (v & (2**64 - 1)).to_bytes(8, "little") if a.dtype.precision <= 18 else (v & (2**128 - 1)).to_bytes(16, "little")
                    elif tid in (S.INT64, S.TIMESTAMP, S.TIMESTAMP_NTZ):
                        b = (int(a.values[i]) & (2**64 - 1)).to_bytes(8, "little")
                    elif tid in (S.INT8, S.INT16, S.INT32, S.DATE, S.BOOL):
                        b = (int(a.values[i]) & (2**32 - 1)).to_bytes(4, "little")
                    elif tid == S.FLOAT:
                        x = np.float32(a.values[i])
                        b = (0).to_bytes(4, "little") if x == 0 else x.tobytes()
                    elif tid == S.DOUBLE:
                        x = np.float64(a.values[i])
                        b = (0).to_bytes(8, "little") if x == 0 else x.tobytes()
                    else:
                        raise OracleError(f"xxhash64 over {a.dtype} is not restated")
                    h[i] = xxhash.xxh64_intdigest(b, h[i])
            return Col(S.T_INT64, np.array(h, dtype=np.uint64).view(np.int64).copy() if n else np.zeros(0, np.int64), None)
        if f == "murmur3_hash":
            # spark_murmur3_hash (hash_funcs/murmur3.rs:24-70): seed literal last, NULLs skipped, result never NULL
            C = _lib()
            seed = e.children[-1]
            if seed.kind != "literal" or seed.value is None or seed.dtype.type_id != S.INT32:
                raise OracleError("The seed of function murmur3_hash must be an Int32 scalar value")
            h = np.full(max(n, 1), seed.value & 0xFFFFFFFF, np.uint32)
            for ch in e.children[:-1]:
                a = self.eval(ch, cols, n)
                vb = None if a.valid is None else np.ascontiguousarray(a.valid.astype(np.uint8))
                tid = a.dtype.type_id
                if tid == S.DECIMAL:
                    C.o_murmur3_decimal(_p(np.ascontiguousarray(a.values)), ctypes.c_int32(a.dtype.precision), _p(vb), ctypes.c_int64(n), _p(h))
                elif tid in (S.INT64, S.TIMESTAMP, S.TIMESTAMP_NTZ):
                    C.o_murmur3_i64(_p(np.ascontiguousarray(a.values, dtype=np.int64)), _p(vb), ctypes.c_int64(n), _p(h))
                elif tid in (S.INT8, S.INT16, S.INT32, S.DATE, S.BOOL):
                    C.o_murmur3_i32(_p(np.ascontiguousarray(a.values.astype(np.int32))), _p(vb), ctypes.c_int64(n), _p(h))
                elif tid == S.FLOAT:
                    C.o_murmur3_f32(_p(np.ascontiguousarray(a.values, dtype=np.float32)), _p(vb), ctypes.c_int64(n), _p(h))
                elif tid == S.DOUBLE:
                    C.o_murmur3_f64(_p(np.ascontiguousarray(a.values, dtype=np.float64)), _p(vb), ctypes.c_int64(n), _p(h))
                else:
                    raise OracleError(f"murmur3_hash over {a.dtype} is not restated")
            return Col(S.T_INT32, h[:n].view(np.int32).copy(), None)
        if f in ("date_add", "date_sub", "date_diff", "datediff"):
            # wrapping 32-bit day arithmetic (datafusion-spark SparkDateAdd / SparkDateSub; datetime_funcs/date_diff.rs:72-110)
            a, b = self.eval(e.children[0], cols, n), self.eval(e.children[1], cols, n)
            x, y = a.values.astype(np.int64), b.values.astype(np.int64)
            z = x + y if f == "date_add" else x - y
            z = ((z + 2**31) % 2**32 - 2**31).astype(np.int32)
            return Col(S.T_INT32 if f in ("date_diff", "datediff") else S.T_DATE, z, self._and_valid(a.valid, b.valid))
        if f == "round":
            # spark_round (math_funcs/round.rs:160-260)
            a = self.eval(e.children[0], cols, n)
            point = int(e.children[1].value)
            trunc_div = lambda p, q: abs(p) // q * (1 if p >= 0 else -1)
            if a.dtype.type_id == S.DECIMAL:
                scale, out = a.dtype.scale, []
                for i in range(n):
                    x = dec_to_int(a.values, i)
                    sg = (x > 0) - (x < 0)
                    if point < 0:
                        ex = -point + scale
                        v = 0 if ex >= 39 else trunc_div(x + sg * (10**ex // 2), 10**ex) * 10**(-point)
                    else:
                        div = 10**(scale - min(scale, point))
                        v = trunc_div(x + sg * (div // 2), div)
                    out.append(v)
                return Col(e.dtype, ints_to_dec(out), a.valid)
            bits = 64 if a.dtype.type_id == S.INT64 else 32
            div = 10**(-point)
            half, out, ovf = div // 2, [], np.zeros(n, bool)
            for i in range(n):
                x = int(a.values[i])
                rem = abs(x) % div * (1 if x >= 0 else -1)       # Rust `%`: sign of the dividend
                v = x - rem + (-div if rem <= -half else div if rem >= half else 0)
                ovf[i] = not (-(1 << (bits - 1)) <= v < (1 << (bits - 1)))
                v &= (1 << bits) - 1
                out.append(v - (1 << bits) if v >> (bits - 1) else v)
            if e.fail_on_error and (ovf & a.ok()).any():
                raise OracleError("ARITHMETIC_OVERFLOW")
            return Col(a.dtype, np.array(out, dtype=_np_dtype(S, a.dtype)), a.valid)
        if f in ("substring", "substr"):
            # Spark UTF8String.substringSQL: 1-based character positions, 0 like 1, negative from the end, window clipped to the string
            a = self.eval(e.children[0], cols, n)
            pos = int(e.children[1].value)
            ln = int(e.children[2].value) if len(e.children) > 2 else 2**31 - 1

            def sub(v):
                start = pos - 1 if pos > 0 else (len(v) + pos if pos < 0 else 0)
                until = start + ln
                if until <= start or start >= len(v.encode()):
                    return ""
                return v[max(start, 0):max(until, 0)]
            return Col(S.T_STRING, np.array([sub(v) if v is not None else None for v in a.values], dtype=object), a.valid)
        if f in ("trim", "btrim", "ltrim", "rtrim") and len(e.children) == 1:
            # Spark's trim without a trim string removes the space character U+0020 only (UTF8String.trim / trimLeft / trimRight)
            a = self.eval(e.children[0], cols, n)
            fn = {"trim": lambda v: v.strip(" "), "btrim": lambda v: v.strip(" "), "ltrim": lambda v: v.lstrip(" "), "rtrim": lambda v: v.rstrip(" ")}[f]
            return Col(S.T_STRING, np.array([fn(v) if v is not None else None for v in a.values], dtype=object), a.valid)
        if f in ("rpad", "lpad", "read_side_padding"):
            # static_invoke/char_varchar_utils/read_side_padding.rs:35-47,345-420: pad to `length` CHARACTERS with the pattern repeated;
            # rpad / lpad cut a longer value to `length` characters, read-side padding of CHAR(n) columns leaves it alone
            a = self.eval(e.children[0], cols, n)
            length = max(int(e.children[1].value), 0)
            pat = e.children[2].value if len(e.children) > 2 else " "

            def pad(v):
                if len(v) >= length:
                    return v[:length] if f != "read_side_padding" else v
                fill = (pat * (length // max(len(pat), 1) + 1))[:length - len(v)] if pat else ""
                return fill + v if f == "lpad" else v + fill
            return Col(S.T_STRING, np.array([pad(v) if v is not None else None for v in a.values], dtype=object), a.valid)
        if f in ("upper", "lower"):
            # DataFusion's upper / lower = Rust's str::to_uppercase / to_lowercase: full Unicode case mapping without locale and the Final_Sigma rule —
            # which is what Python's str.upper() / str.lower() implement too (for the Unicode version Python carries; the device tables are checked
            # against Rust's own standard library in tests/test_case_map_cpu.py)
            a = self.eval(e.children[0], cols, n)
            out = np.empty(n, dtype=object)
            for i in range(n):
                out[i] = (a.values[i].upper() if f == "upper" else a.values[i].lower()) if a.ok()[i] else None
            return Col(S.T_STRING, out, a.valid)
        # ---- the Float64 functions the reference hands to DataFusion / datafusion-spark (QueryPlanSerde.scala:117-174): numpy's libm where Rust's std stands
        # there — both within an ulp or two of the exact value, the tests compare with a stated tolerance
        _LIBM = {"acos": "acos", "acosh": "acosh", "asin": "asin", "asinh": "asinh", "atan": "atan", "atanh": "atanh", "cbrt": "cbrt", "cos": "cos", "cosh": "cosh", "exp": "exp",
                 "expm1": "expm1", "ln": "log", "log2": "log2", "log10": "log10", "sin": "sin", "sinh": "sinh", "tan": "tan", "tanh": "tanh", "cot": "tan", "csc": "sin", "sec": "cos",
                 "rint": "rint"}
        if f in _LIBM or f in ("degrees", "radians"):
            a = self.eval(e.children[0], cols, n)
            x = a.values.astype(np.float64)
            if f == "degrees":      # f64::to_degrees: x · (180 / π as the constant 57.29577951308232…)
                v = x * (180.0 / np.pi)
            elif f == "radians":    # f64::to_radians: x · (π / 180)
                v = x * (np.pi / 180.0)
            else:
                fn = _libm_fn(_LIBM[f])       # the platform's libm: what Rust's f64 methods call in the reference
                v = np.array([fn(float(t)) for t in x], dtype=np.float64)
                if f in ("cot", "csc", "sec"):      # datafusion-spark: 1 / tan, 1 / sin, 1 / cos
                    with np.errstate(all="ignore"):
                        v = 1.0 / v
            return Col(S.T_DOUBLE, v, a.valid)
        if f == "pi":
            return Col(S.T_DOUBLE, np.full(n, np.pi), None)
        if f in ("atan2", "pow", "power", "spark_log"):
            a, b = self.eval(e.children[0], cols, n), self.eval(e.children[1], cols, n)
            x, y = a.values.astype(np.float64), b.values.astype(np.float64)
            ok = a.ok() & b.ok()
            with np.errstate(all="ignore"):
                if f == "atan2":
                    fn = _libm_fn("atan2", 2)
                    v = np.array([fn(float(p), float(q)) for p, q in zip(x, y)], dtype=np.float64)
                elif f == "spark_log":      # math_funcs/log.rs:30-38: ln(value) / ln(base), NULL when base <= 0 or value <= 0
                    ok = ok & (x > 0) & (y > 0)
                    ln = _libm_fn("log")
                    v = np.array([ln(float(q)) if q > 0 else 0.0 for q in y], dtype=np.float64) / np.array([ln(float(p)) if p > 0 else 1.0 for p in x], dtype=np.float64)
                else:                       # math_funcs/pow.rs:24-29: Java's Math.pow — |base| = 1 with an infinite / NaN exponent is NaN
                    fn = _libm_fn("pow", 2)
                    v = np.where((np.abs(x) == 1.0) & ~np.isfinite(y), np.nan, np.array([fn(float(p), float(q)) for p, q in zip(x, y)], dtype=np.float64))
            return Col(S.T_DOUBLE, v, None if ok.all() else ok)
        if f == "factorial":                # Spark's Factorial: 0..20, NULL outside
            import math
            a = self.eval(e.children[0], cols, n)
            inside = (a.values >= 0) & (a.values <= 20)
            ok = a.ok() & inside
            return Col(S.T_INT64, np.array([math.factorial(int(v)) if i else 0 for v, i in zip(a.values, inside)], dtype=np.int64), None if ok.all() else ok)
        if f in ("bitwise_not", "bit_count", "bit_get", "getbit", "shiftrightunsigned"):
            a = self.eval(e.children[0], cols, n)
            if f == "bitwise_not":
                return Col(a.dtype, ~a.values, a.valid)
            if f == "bit_count":            # Java's Long.bitCount of the value widened to a long
                v = a.values.astype(np.int64).view(np.uint64)
                return Col(S.T_INT32, np.array([bin(int(x)).count("1") for x in v], dtype=np.int32), a.valid)
            b = self.eval(e.children[1], cols, n)
            ok = a.ok() & b.ok()
            if f == "shiftrightunsigned":   # Java's >>>: the count modulo the width
                if a.values.dtype == np.int64:
                    v = (a.values.view(np.uint64) >> (b.values.astype(np.int64) & 63).astype(np.uint64)).view(np.int64)
                else:
                    v = (a.values.astype(np.int32).view(np.uint32) >> (b.values.astype(np.int64) & 31).astype(np.uint32)).view(np.int32)
                return Col(a.dtype, v, None if ok.all() else ok)
            return Col(S.T_INT8, ((a.values.astype(np.int64) >> b.values.astype(np.int64)) & 1).astype(np.int8), None if ok.all() else ok)
        if f in ("greatest", "least"):
            # NULL arguments are skipped, NULL only when every argument is; NaN is the greatest double (Spark's ordering, DataFusion's total order)
            args = [self.eval(c, cols, n) for c in e.children]
            vals = args[0].values.copy()
            ok = args[0].ok().copy()
            for b in args[1:]:
                bv, bok = b.values, b.ok()
                if vals.dtype.kind == "f":
                    an, bn = np.isnan(vals), np.isnan(bv)
                    with np.errstate(all="ignore"):
                        better = (bn | (~an & (bv > vals))) if f == "greatest" else (an | (~bn & (bv < vals)))
                else:
                    better = (bv > vals) if f == "greatest" else (bv < vals)
                take = bok & (~ok | better)
                vals = np.where(take, bv, vals)
                ok = ok | bok
            return Col(args[0].dtype, vals, None if ok.all() else ok)
        if f in ("last_day", "date_from_unix_date", "date_trunc", "trunc", "next_day", "make_date"):
            import datetime
            E = datetime.date(1970, 1, 1)
            to_date = lambda v: E + datetime.timedelta(days=int(v))
            a = self.eval(e.children[0], cols, n)
            if f == "date_from_unix_date":  # date_from_unix_date.rs:52-60: the Int32 is the date
                return Col(S.T_DATE, a.values.astype(np.int32), a.valid)
            out = np.zeros(n, np.int32)
            ok = a.ok().copy()
            if f == "make_date":            # make_date.rs:87-96: chrono's from_ymd_opt — what it refuses is NULL
                m, d = self.eval(e.children[1], cols, n), self.eval(e.children[2], cols, n)
                ok &= m.ok() & d.ok()
                for i in range(n):
                    if ok[i]:
                        try:
                            out[i] = (datetime.date(int(a.values[i]), int(m.values[i]), int(d.values[i])) - E).days
                        except ValueError:
                            ok[i] = False       # (years beyond 1..9999 are not generated by the tests: Python's calendar ends there)
                return Col(S.T_DATE, out, None if ok.all() else ok)
            arg = e.children[1].value if len(e.children) > 1 else None
            if f == "next_day":             # next_day.rs:49-68
                names = {"MO": 0, "MON": 0, "MONDAY": 0, "TU": 1, "TUE": 1, "TUESDAY": 1, "WE": 2, "WED": 2, "WEDNESDAY": 2, "TH": 3, "THU": 3, "THURSDAY": 3, "FR": 4, "FRI": 4,
                         "FRIDAY": 4, "SA": 5, "SAT": 5, "SATURDAY": 5, "SU": 6, "SUN": 6, "SUNDAY": 6}
                target = names.get(arg.upper()) if arg is not None else None
                if target is None:
                    return Col(S.T_DATE, out, np.zeros(n, bool))
            for i in range(n):
                if not ok[i]:
                    continue
                dd = to_date(a.values[i])
                if f == "last_day":
                    import calendar
                    r = dd.replace(day=calendar.monthrange(dd.year, dd.month)[1])
                elif f == "next_day":
                    r = dd + datetime.timedelta(days=7 - (dd.weekday() - target) % 7)
                else:                       # kernels/temporal.rs:63-100, 326-337
                    u = arg.upper()
                    if u in ("YEAR", "YYYY", "YY"):
                        r = datetime.date(dd.year, 1, 1)
                    elif u == "QUARTER":
                        r = datetime.date(dd.year, (dd.month - 1) // 3 * 3 + 1, 1)
                    elif u in ("MONTH", "MON", "MM"):
                        r = datetime.date(dd.year, dd.month, 1)
                    elif u == "WEEK":
                        r = dd - datetime.timedelta(days=dd.weekday())
                    else:
                        raise OracleError("Unsupported format: %r for function 'date_trunc'" % arg)
                out[i] = (r - E).days
            return Col(S.T_DATE, out, None if ok.all() else ok)
        if f in ("seconds_to_timestamp", "timestamp_seconds"):
            # seconds_to_timestamp.rs:63-105: integers · 10^6 (an Int64 product beyond i64: "long overflow"), floats (s · 10^6) as i64 saturating, NaN / ±inf NULL
            a = self.eval(e.children[0], cols, n)
            if a.values.dtype.kind == "f":
                x = a.values.astype(np.float64)
                ok = a.ok() & np.isfinite(x)
                with np.errstate(all="ignore"):
                    p = np.where(np.isfinite(x), x, 0.0) * 1e6
                v = np.where(p >= 9.223372036854775807e18, np.iinfo(np.int64).max, np.where(p <= -9.223372036854775808e18, np.iinfo(np.int64).min, np.trunc(np.clip(p, -9.2e18, 9.2e18)).astype(np.int64)))
                return Col(S.T_TIMESTAMP, v.astype(np.int64), None if ok.all() else ok)
            big = a.ok() & (np.abs(a.values.astype(np.float64)) > 9223372036854.0)
            if big.any():
                raise OracleError("long overflow")
            return Col(S.T_TIMESTAMP, a.values.astype(np.int64) * 1000000, a.valid)
        if f in ("reverse", "repeat", "replace", "substring_index", "substr_index", "md5", "sha1", "sha2", "instr", "strpos", "ascii", "crc32"):
            # DataFusion / datafusion-spark string functions and digests (QueryPlanSerde.scala:208-249): reverse = the scalar values reversed; repeat; replace = Rust's
            # str::replace (an empty search string matches before every character and at the end — what Python's does too); substring_index = DataFusion's substr_index
            # (split / rsplit without overlap); md5 / sha1 / sha2 = lower-case hexadecimal digits of the UTF-8 bytes' digest; instr = 1-based character position of
            # the first occurrence (0: none); ascii = the first scalar value (0: empty); crc32 = zlib's
            import hashlib
            import zlib
            c0 = e.children[0]
            if c0.kind == "cast" and c0.dtype.type_id in (S.BYTES, S.STRING):      # Spark's Md5 / Sha1 / Sha2 / Crc32 take binary: Cast(s AS BINARY) around the column
                c0 = c0.children[0]
            a = self.eval(c0, cols, n)
            lit = lambda i: e.children[i].value

            def sub_index(v, d, k):      # DataFusion's substr_index: split (k > 0) / rsplit (k < 0) pieces, Rust's non-overlapping scans from that side
                if k == 0 or d == "":
                    return ""
                if k > 0:
                    return d.join(v.split(d)[:k])
                pieces, rest = [], v      # from the right
                for _ in range(-k):
                    at = rest.rfind(d)
                    if at < 0:
                        return v
                    pieces.append(rest[at + len(d):])
                    rest = rest[:at]
                return d.join(reversed(pieces))
            fn = {"reverse": lambda v: v[::-1], "repeat": lambda v: v * max(int(lit(1)), 0), "replace": lambda v: v.replace(lit(1), lit(2) if len(e.children) > 2 else ""),
                  "substring_index": lambda v: sub_index(v, lit(1), int(lit(2))), "substr_index": lambda v: sub_index(v, lit(1), int(lit(2))),
                  "md5": lambda v: hashlib.md5(v.encode()).hexdigest(), "sha1": lambda v: hashlib.sha1(v.encode()).hexdigest(),
                  "sha2": lambda v: hashlib.new({224: "sha224", 256: "sha256", 0: "sha256", 384: "sha384", 512: "sha512"}[int(lit(1))], v.encode()).hexdigest() if f == "sha2" else None,
                  "instr": lambda v: v.find(lit(1)) + 1, "strpos": lambda v: v.find(lit(1)) + 1, "ascii": lambda v: ord(v[0]) if v else 0, "crc32": lambda v: zlib.crc32(v.encode())}[f]
            if f in ("instr", "strpos", "ascii"):
                return Col(S.T_INT32, np.array([fn(v) if okv else 0 for v, okv in zip(a.values, a.ok())], dtype=np.int32), a.valid)
            if f == "crc32":
                return Col(S.T_INT64, np.array([fn(v) if okv else 0 for v, okv in zip(a.values, a.ok())], dtype=np.int64), a.valid)
            return Col(S.T_STRING, np.array([fn(v) if okv else None for v, okv in zip(a.values, a.ok())], dtype=object), a.valid)
        if f in ("size", "cardinality"):
            # SparkSizeFunc (array_funcs/size.rs:79-125): the element count, -1 for a NULL list; never NULL
            a = self.eval(e.children[0], cols, n)
            return Col(S.T_INT32, np.array([len(v) if okv else -1 for v, okv in zip(a.values, a.ok())], dtype=np.int32), None)
        if f == "array_contains":
            # Spark's ArrayContains (datafusion-spark SparkArrayContains): NULL array / key: NULL; found: true; else NULL if the array holds a NULL, else false
            a, key = self.eval(e.children[0], cols, n), self.eval(e.children[1], cols, n)
            import datetime
            out, ok = np.zeros(n, bool), np.zeros(n, bool)
            for i in range(n):
                if not (a.ok()[i] and key.ok()[i]):
                    continue
                kv = key.values[i]
                if a.dtype.element.type_id == S.DATE:
                    kv = datetime.date(1970, 1, 1) + datetime.timedelta(days=int(kv))
                found = any(v is not None and v == kv for v in a.values[i])
                out[i] = found
                ok[i] = found or all(v is not None for v in a.values[i])
            return Col(S.T_BOOL, out, None if ok.all() else ok)
        if f == "regexp_extract_all":
            # spark_regexp_extract_all (string_funcs/regexp_extract_all.rs:32-108): group idx (default 1) of EVERY match — the crate's captures_iter, i.e. find_iter's
            # matches —, the empty string where the group took no part; no match: an empty list; NULL subject: NULL
            a = self.eval(e.children[0], cols, n)
            pat = e.children[1].value
            idx = int(e.children[2].value) if len(e.children) > 2 else 1
            rx = crate_pattern_to_python(pat)
            if idx < 0 or idx > rx.groups:
                raise OracleError("The value of parameter `idx` in `regexp_extract_all` is invalid: Expects group index between 0 and %d, but got %d." % (rx.groups, idx))
            out = np.empty(n, dtype=object)
            for i in range(n):
                out[i] = [m.group(idx) or "" for m in find_iter_like_the_crate(rx, a.values[i])] if a.ok()[i] else None
            return Col(S.list_type(S.T_STRING, True), out, a.valid)
        if f == "split":
            # spark_split (string_funcs/split.rs:32-97, 434-472): the pieces between the pattern's matches — limit > 0: at most limit - 1 cuts;
            # limit = 0: trailing empty pieces dropped (nothing left: one empty piece); limit < 0 (the default): every piece.  NULL subject → NULL list.
            a = self.eval(e.children[0], cols, n)
            pat = e.children[1].value
            limit = int(e.children[2].value) if len(e.children) > 2 else -1
            out = np.empty(n, dtype=object)
            for i in range(n):
                out[i] = split_like_the_crate(pat, a.values[i], limit) if a.ok()[i] else None
            return Col(S.list_type(S.T_STRING, False), out, a.valid)
        if f == "regexp_extract":
            # spark_regexp_extract (string_funcs/regexp_extract.rs:38-108; arguments regexp_extract_common.rs:36-107): group `idx` (default 1) of the
            # FIRST match of the pattern (the crate's Regex::captures_read: leftmost, alternatives / repetitions preferred in pattern order); the empty
            # string without a match or when the group took no part in it; NULL for a NULL subject, pattern or idx; idx outside 0..groups is an error.
            # The crate is not here: Python's `re` — a backtracking engine, i.e. the same leftmost, preference-ordered match — stands in for it on the
            # syntax both read alike ($ is "at the very end" in the crate: \Z here; \z likewise; \d \w \s Unicode-aware in both).
            import re
            a = self.eval(e.children[0], cols, n)
            pat = e.children[1].value
            idx = e.children[2].value if len(e.children) > 2 else 1
            if pat is None or idx is None:
                return Col(S.T_STRING, np.array([None] * n, dtype=object), np.zeros(n, bool))
            try:
                rx = crate_pattern_to_python(pat)
            except re.error as err:
                raise OracleError("The value of parameter `regexp` in `regexp_extract` is invalid: '%s' (%s)" % (pat, err))
            if idx < 0 or idx > rx.groups:
                raise OracleError("The value of parameter `idx` in `regexp_extract` is invalid: Expects group index between 0 and %d, but got %d." % (rx.groups, idx))
            out = np.empty(n, dtype=object)
            for i in range(n):
                if a.ok()[i]:
                    m = rx.search(a.values[i])
                    out[i] = (m.group(int(idx)) or "") if m else ""
                else:
                    out[i] = None
            return Col(S.T_STRING, out, a.valid)
        if f == "concat":
            # Spark's Concat (datafusion-spark's SparkConcat, jni_api.rs:70): the arguments' bytes one after the other; NULL as soon as one is NULL
            args = [self.eval(c, cols, n) for c in e.children]
            ok = np.ones(n, bool)
            for a in args:
                ok &= a.ok()
            out = np.empty(n, dtype=object)
            for i in range(n):
                out[i] = "".join(a.values[i] for a in args) if ok[i] else None
            return Col(S.T_STRING, out, None if ok.all() else ok)
        if f == "coalesce":
            # the first non-NULL argument
            args = [self.eval(c, cols, n) for c in e.children]
            out_vals = args[-1].values.copy()
            out_ok = args[-1].ok().copy()
            for a in reversed(args[:-1]):
                ok = a.ok()
                out_vals = np.where(ok, a.values, out_vals) if a.values.dtype != object else np.array([x if k else y for x, y, k in zip(a.values, out_vals, ok)], dtype=object)
                out_ok = ok | out_ok
            return Col(args[0].dtype, out_vals, None if out_ok.all() else out_ok)
        if f in ("starts_with", "ends_with", "contains"):
            # byte-wise on the UTF-8 encodings (UTF8_BINARY collation; strings.scala:343-360)
            a, lit = self.eval(e.children[0], cols, n), e.children[1].value.encode()
            fn = {"starts_with": bytes.startswith, "ends_with": bytes.endswith, "contains": lambda v, l: l in v}[f]
            vals = np.array([fn(v.encode(), lit) if v is not None else False for v in a.values], dtype=bool)
            return Col(S.T_BOOL, vals & a.ok(), a.valid)
        if f in ("length", "char_length", "character_length", "octet_length", "bit_length"):
            a = self.eval(e.children[0], cols, n)
            g = (lambda v: len(v)) if f in ("length", "char_length", "character_length") else (lambda v: len(v.encode()) * (8 if f == "bit_length" else 1))
            return Col(S.T_INT32, np.array([g(v) if v is not None else 0 for v in a.values], dtype=np.int32), a.valid)
        a = self.eval(e.children[0], cols, n)
        tid = a.dtype.type_id
        if f in ("ceil", "floor"):
            if tid in (S.FLOAT, S.DOUBLE):
                with np.errstate(all="ignore"):
                    x = (np.ceil if f == "ceil" else np.floor)(a.values.astype(np.float64))
                out = np.zeros(n, np.int64)
                for i in range(n):                                   # Rust `as i64`: saturating, NaN → 0
                    v = x[i]
                    out[i] = 0 if v != v else (2**63 - 1 if v >= 2.0**63 else (-2**63 if v <= -2.0**63 else int(v)))
                return Col(S.T_INT64, out, a.valid)
            if tid == S.INT64:
                return a
            if tid == S.DECIMAL and a.dtype.scale > 0:
                d = 10 ** a.dtype.scale
                vals = [(-((-dec_to_int(a.values, i)) // d) if f == "ceil" else dec_to_int(a.values, i) // d) for i in range(n)]
                return Col(e.dtype, ints_to_dec(vals), a.valid)
            raise OracleError(f"{f} over {a.dtype}")
        if f == "abs":
            fail = e.fail_on_error or (len(e.children) == 2 and bool(e.children[1].value))
            if tid in (S.FLOAT, S.DOUBLE):
                return Col(a.dtype, np.abs(a.values), a.valid)
            if tid == S.DECIMAL:
                return Col(a.dtype, ints_to_dec([abs(dec_to_int(a.values, i)) for i in range(n)]), a.valid)
            info = np.iinfo(a.values.dtype)
            if fail and ((a.values == info.min) & a.ok()).any():
                raise OracleError("ARITHMETIC_OVERFLOW integer")
            with np.errstate(over="ignore"):
                return Col(a.dtype, np.where(a.values < 0, (0 - a.values.astype(np.int64)).astype(a.values.dtype), a.values), a.valid)   # wrapping_abs
        if f == "sqrt":
            with np.errstate(all="ignore"):
                return Col(S.T_DOUBLE, np.sqrt(a.values.astype(np.float64)), a.valid)
        if f == "signum":
            x = a.values.astype(np.float64)
            return Col(S.T_DOUBLE, np.where(x != x, x, np.where(x > 0, 1.0, np.where(x < 0, -1.0, 0.0))), a.valid)
        if f == "isnan":
            x = a.values.astype(np.float64)
            return Col(S.T_BOOL, a.ok() & (x != x), None)
        raise NotImplementedError(f"oracle: scalar function {f}")

    def _literal(self, e, n) -> Col:
        S = self.S
        t = e.dtype
        valid = None if e.value is not None else np.zeros(n, bool)
        if t.type_id == S.DECIMAL:
            vals = ints_to_dec([e.value or 0]).repeat(n)
        elif t.type_id == S.STRING:
            vals = np.array([e.value] * n, dtype=object)
        else:
            vals = np.full(n, e.value if e.value is not None else 0, dtype=_np_dtype(S, t))
        return Col(t, vals, valid)

    def _compare(self, k, a: Col, b: Col) -> np.ndarray:
        S = self.S
        if a.dtype.type_id == S.DECIMAL:
            assert a.dtype.scale == b.dtype.scale, "decimal comparison needs equal scales"
            op = {"eq": 0, "neq": 1, "lt": 2, "lt_eq": 3, "gt": 4, "gt_eq": 5}[k]
            out = np.zeros(len(a), np.uint8)
            C.o_cmp_i128(op, _p(np.ascontiguousarray(a.values)), _p(np.ascontiguousarray(b.values)), _p(out), ctypes.c_int64(len(a)))
            return out.astype(bool)
        x, y = a.values, b.values
        if a.dtype.type_id in (S.STRING, S.BYTES):
            # byte-wise unsigned lexicographic order of the UTF-8 encodings (Spark UTF8String.compareTo, arrow-ord string kernels);
            # NULL slots hold None: compare as empty, the validity mask removes them later
            if a.dtype.type_id == S.STRING and len(x) and isinstance(x[0], (str, type(None))) and isinstance(y[0], (str, type(None))):
                # str against str: Python orders strings by code point, which is the byte order of their UTF-8 encodings — numpy's
                # object loops instead of one Python call per row (the TPC-H golden tests compare tens of millions of strings)
                xn, yn = np.equal(x, None), np.equal(y, None)
                xs = np.where(xn, "", x) if xn.any() else x
                ys = np.where(yn, "", y) if yn.any() else y
                try:
                    r = {"eq": np.equal, "neq": np.not_equal, "lt": np.less, "lt_eq": np.less_equal, "gt": np.greater, "gt_eq": np.greater_equal}[k](xs, ys)
                    return np.asarray(r, dtype=bool)
                except TypeError:
                    pass               # (a column that mixes str and bytes: the general path below)
            enc = lambda v: b"" if v is None else (v.encode() if isinstance(v, str) else bytes(v))
            xs, ys = [enc(v) for v in x], [enc(v) for v in y]
            f = {"eq": lambda p, q: p == q, "neq": lambda p, q: p != q, "lt": lambda p, q: p < q, "lt_eq": lambda p, q: p <= q,
                 "gt": lambda p, q: p > q, "gt_eq": lambda p, q: p >= q}[k]
            return np.array([f(p, q) for p, q in zip(xs, ys)], dtype=bool)
        if a.dtype.type_id in (S.FLOAT, S.DOUBLE):
            # arrow-ord compares floats with IEEE totalOrder
            it = np.int64 if a.dtype.type_id == S.DOUBLE else np.int32
            bits = 63 if it is np.int64 else 31

            def key(v):
                b_ = v.view(it)
                return b_ ^ ((b_ >> bits).astype(it).view(np.uint64 if it is np.int64 else np.uint32) >> 1).view(it)
            x, y = key(np.ascontiguousarray(x)), key(np.ascontiguousarray(y))
        return {"eq": x == y, "neq": x != y, "lt": x < y, "lt_eq": x <= y, "gt": x > y, "gt_eq": x >= y}[k]

    def _arith(self, e, a: Col, b: Col) -> Col:
        S = self.S
        n = len(a)
        valid = self._and_valid(a.valid, b.valid)
        if a.dtype.type_id == S.DECIMAL and b.dtype.type_id == S.DECIMAL:
            p1, s1, p2, s2 = a.dtype.precision, a.dtype.scale, b.dtype.precision, b.dtype.scale
            mul = e.kind == "multiply"
            addsub = e.kind in ("add", "subtract")
            if e.kind in ("divide", "integral_divide"):
                # decimal_div / decimal_integral_div: spark-expr/src/math_funcs/div.rs:40-165, exact Python integers
                integral = e.kind == "integral_divide"
                s3 = e.dtype.scale
                l_exp, r_exp = max(0, s2 + s3 + 1 - s1), max(0, s1 - (s2 + s3 + 1))
                live = np.ones(n, bool) if valid is None else valid
                res = []
                for i in range(n):
                    L, R = dec_to_int(a.values, i) * 10 ** l_exp, dec_to_int(b.values, i) * 10 ** r_exp
                    if R == 0:
                        if e.eval_mode == S.ANSI and live[i]:
                            raise OracleError("DIVIDE_BY_ZERO")
                        res.append(0)
                        continue
                    div = abs(L) // abs(R) * (-1 if (L < 0) != (R < 0) else 1)      # BigInt division truncates toward zero
                    q = div if integral else (div - 5 if div < 0 else div + 5)
                    q = abs(q) // 10 * (-1 if q < 0 else 1)
                    if integral and getattr(e, "check_divide_overflow", False) and e.eval_mode == S.ANSI and live[i] and not -2**63 <= q < 2**63:
                        raise OracleError("ARITHMETIC_OVERFLOW")     # quotient_to_i128, div.rs:57-68
                    res.append(q if -2**127 <= q < 2**127 else 2**127 - 1)           # to_i128().unwrap_or(i128::MAX)
                return Col(e.dtype, ints_to_dec(res), valid)
            if e.kind == "remainder":
                # create_modulo_expr → arrow-arith 58.4's decimal `rem` (math_funcs/modulo_expr.rs:137-206): both operands at the larger scale
                # (in 256 bits where the reference casts to Decimal256), Rust `%` (sign of the dividend); a zero divisor is NULL (ANSI: error)
                sm = max(s1, s2)
                live = np.ones(n, bool) if valid is None else valid
                res, nz = [], np.ones(n, bool)
                for i in range(n):
                    L, R = dec_to_int(a.values, i) * 10 ** (sm - s1), dec_to_int(b.values, i) * 10 ** (sm - s2)
                    if R == 0:
                        if e.eval_mode == S.ANSI and live[i]:
                            raise OracleError("REMAINDER_BY_ZERO")
                        nz[i] = False
                        res.append(0)
                        continue
                    res.append(abs(L) % abs(R) * (-1 if L < 0 else 1))
                v2 = live & nz
                return Col(e.dtype, ints_to_dec(res), None if v2.all() else v2)
            assert mul or addsub, f"decimal {e.kind} not in oracle"
            wide = (addsub and max(s1, s2) + max(p1 - s1, p2 - s2) >= 38) or (mul and p1 + p2 >= 38)  # planner.rs:1000-1008
            av, bv = np.ascontiguousarray(a.values), np.ascontiguousarray(b.values)
            out = np.zeros(n, DEC128)
            if wide:
                ok = np.zeros(n, np.uint8)
                op = {"add": 0, "subtract": 1, "multiply": 2}[e.kind]
                C.o_wide_decimal(op, _p(av), s1, _p(bv), s2, e.dtype.precision, e.dtype.scale, _p(out), _p(ok), ctypes.c_int64(n))
                okb = ok.astype(bool)
                if e.eval_mode == S.ANSI and (~okb & (np.ones(n, bool) if valid is None else valid)).any():
                    raise OracleError("ARITHMETIC_OVERFLOW")
                valid2 = okb if valid is None else (valid & okb)
                c = Col(e.dtype, out, None if valid2.all() else valid2)
                c.wide = True
                return c
            if mul:
                C.o_dec_mul(_p(av), _p(bv), _p(out), ctypes.c_int64(n))
                return Col(S.decimal(min(38, p1 + p2 + 1), s1 + s2), out, valid)
            C.o_dec_addsub(_p(av), s1, _p(bv), s2, 1 if e.kind == "subtract" else 0, _p(out), ctypes.c_int64(n))
            sm = max(s1, s2)
            return Col(S.decimal(min(38, max(p1 - s1, p2 - s2) + sm + 1), sm), out, valid)
        rt = e.dtype
        if e.kind == "remainder" and rt.type_id in (S.INT8, S.INT16, S.INT32, S.INT64, S.FLOAT, S.DOUBLE):
            # create_modulo_expr (math_funcs/modulo_expr.rs): zero divisor → NULL (ANSI: error); sign of the dividend; MIN % -1 = 0
            nt = _np_dtype(S, rt)
            x, y = a.values.astype(nt), b.values.astype(nt)
            zero = y == 0
            live = np.ones(n, bool) if valid is None else valid
            if e.eval_mode == S.ANSI and (zero & live).any():
                raise OracleError("REMAINDER_BY_ZERO")
            with np.errstate(all="ignore"):
                if rt.type_id in (S.FLOAT, S.DOUBLE):
                    r = np.fmod(x, np.where(zero, 1, y)).astype(nt)
                else:
                    ys = np.where(zero | (y == -1), 1, y)
                    r = np.where(y == -1, 0, np.fmod(x, ys)).astype(nt)
            v2 = live & ~zero
            return Col(rt, np.where(v2, r, 0).astype(nt), None if v2.all() else v2)
        if rt.type_id in (S.INT8, S.INT16, S.INT32, S.INT64):
            nt = _np_dtype(S, rt)
            x, y = a.values.astype(np.int64), b.values.astype(np.int64)
            with np.errstate(over="ignore"):
                if rt.type_id == S.INT64:
                    ux, uy = x.view(np.uint64), y.view(np.uint64)
                    r = {"add": ux + uy, "subtract": ux - uy, "multiply": ux * uy}[e.kind].view(np.int64)
                    if e.eval_mode != S.LEGACY:
                        big = {"add": x.astype(object) + y.astype(object), "subtract": x.astype(object) - y.astype(object),
                               "multiply": x.astype(object) * y.astype(object)}[e.kind]
                        ovf = np.array([not (-2**63 <= int(v) < 2**63) for v in big], bool)
                    else:
                        ovf = None
                else:
                    full = {"add": x + y, "subtract": x - y, "multiply": x * y}[e.kind]
                    r = full.astype(nt)
                    ovf = (r.astype(np.int64) != full) if e.eval_mode != S.LEGACY else None
            if ovf is not None:
                live = ovf & (np.ones(n, bool) if valid is None else valid)
                if e.eval_mode == S.ANSI and live.any():
                    raise OracleError("ARITHMETIC_OVERFLOW integer")
                if e.eval_mode == S.TRY:
                    v2 = ~ovf if valid is None else (valid & ~ovf)
                    r = np.where(v2, r, 0).astype(nt)   # checked_arithmetic.rs:54-124 zeroes NULL slots
                    return Col(rt, r, None if v2.all() else v2)
            return Col(rt, r.astype(nt), valid)
        if rt.type_id in (S.FLOAT, S.DOUBLE):
            nt = _np_dtype(S, rt)
            x, y = a.values.astype(nt), b.values.astype(nt)
            with np.errstate(all="ignore"):
                r = {"add": x + y, "subtract": x - y, "multiply": x * y, "divide": x / y}[e.kind]
            return Col(rt, r.astype(nt), valid)
        raise NotImplementedError(f"oracle arithmetic on {rt}")

    def _check_overflow(self, e, cols, n) -> Col:
        S = self.S
        child_e = e.children[0]
        # planner.rs:615-633: Cast(dec→dec) + CheckOverflow with the same target type → fused rescale+check
        if child_e.kind == "cast" and child_e.dtype == e.dtype:
            inner = self.eval(child_e.children[0], cols, n)
            if inner.dtype.type_id == S.DECIMAL:
                return self._rescale(inner, e.dtype, e.fail_on_error)
        c = self.eval(child_e, cols, n)
        assert c.dtype.type_id == S.DECIMAL
        if getattr(c, "wide", False) and c.dtype == e.dtype:   # planner.rs:606-613
            return c
        assert c.dtype.scale == e.dtype.scale, "CheckOverflow never rescales (checkoverflow.rs:36-39)"
        ok = np.zeros(n, np.uint8)
        C.o_check_overflow(_p(np.ascontiguousarray(c.values)), e.dtype.precision, _p(ok), ctypes.c_int64(n))
        okb = ok.astype(bool)
        if e.fail_on_error and (~okb & c.ok()).any():
            raise OracleError("NUMERIC_VALUE_OUT_OF_RANGE")
        v2 = okb if c.valid is None else (c.valid & okb)
        return Col(e.dtype, c.values, None if v2.all() else v2)

    def _rescale(self, c: Col, to, fail_on_error) -> Col:
        n = len(c)
        out = np.zeros(n, DEC128)
        ok = np.zeros(n, np.uint8)
        C.o_rescale_check(_p(np.ascontiguousarray(c.values)), c.dtype.scale, to.precision, to.scale, _p(out), _p(ok), ctypes.c_int64(n))
        okb = ok.astype(bool)
        if fail_on_error and (~okb & c.ok()).any():
            raise OracleError("NUMERIC_VALUE_OUT_OF_RANGE")
        v2 = okb if c.valid is None else (c.valid & okb)
        return Col(to, out, None if v2.all() else v2)

    def _cast(self, e, c: Col) -> Col:
        S = self.S
        to, frm = e.dtype, c.dtype
        if to == frm:
            return c
        ints = (S.INT8, S.INT16, S.INT32, S.INT64)
        def try_result(vals, fits):
            # TRY (try_cast) of a number to an integer type does not reach Comet's own kernels (cast.rs:284-293, 311-326: `if eval_mode != Try`) but
            # arrow's cast with `safe: true` (:236-241, 401-407; arrow-cast 58.4, third party — its published rule restated, parity unpinned): the
            # value truncated toward zero, NULL when that does not fit the target type (a NaN never does)
            v2 = fits if c.valid is None else (c.valid & fits)
            return Col(to, np.where(fits, vals, 0).astype(_np_dtype(S, to)), None if v2.all() else v2)
        if frm.type_id in ints and to.type_id in ints:
            r = c.values.astype(np.int64).astype(_np_dtype(S, to))   # LEGACY wraps
            if e.eval_mode == S.ANSI and ((r.astype(np.int64) != c.values.astype(np.int64)) & c.ok()).any():
                raise OracleError("CAST_OVERFLOW")
            if e.eval_mode == S.TRY:
                return try_result(r, r.astype(np.int64) == c.values.astype(np.int64))
            return Col(to, r, c.valid)
        if frm.type_id in ints + (S.FLOAT,) and to.type_id == S.DOUBLE:
            return Col(to, c.values.astype(np.float64), c.valid)
        if frm.type_id in ints and to.type_id == S.FLOAT:
            return Col(to, c.values.astype(np.float32), c.valid)
        if frm.type_id in ints and to.type_id == S.DECIMAL:
            vals = [int(v) * 10 ** to.scale for v in c.values]
            bound = 10 ** to.precision - 1
            ok = np.array([abs(v) <= bound for v in vals], bool)
            if e.eval_mode == S.ANSI and (~ok & c.ok()).any():
                raise OracleError("NUMERIC_VALUE_OUT_OF_RANGE")
            v2 = ok if c.valid is None else (c.valid & ok)
            return Col(to, ints_to_dec([v if o else 0 for v, o in zip(vals, ok)]), None if v2.all() else v2)
        if frm.type_id == S.DECIMAL and to.type_id == S.DECIMAL:
            return self._rescale(c, to, e.eval_mode == S.ANSI)
        n = len(c)
        wrap = {S.INT8: 8, S.INT16: 16, S.INT32: 32, S.INT64: 64}

        def as_int(v, bits):       # Rust `as` between integers: two's-complement truncation
            v &= (1 << bits) - 1
            return v - (1 << bits) if v >> (bits - 1) else v

        def sat(x, bits):          # Rust float `as iN`: saturating, NaN → 0
            if x != x:
                return 0
            lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
            return lo if x <= lo else (hi if x >= hi + 1 else int(x))
        if frm.type_id in (S.FLOAT, S.DOUBLE) and to.type_id in ints:
            # conversion_funcs/numeric.rs:311-425
            out, ovf = [], np.zeros(n, bool)
            narrow = to.type_id in (S.INT8, S.INT16)
            for i, x in enumerate(c.values.astype(np.float64)):
                w = 64 if to.type_id == S.INT64 else 32
                v = sat(float(x), w)
                ovf[i] = (x != x) or sat(abs(float(x)), w) == (1 << (w - 1)) - 1
                if narrow:
                    nv = as_int(v, wrap[to.type_id])
                    ovf[i] |= nv != v
                    v = nv
                out.append(v)
            if e.eval_mode == S.ANSI and (ovf & c.ok()).any():
                raise OracleError("CAST_OVERFLOW")
            if e.eval_mode == S.TRY:
                bits = wrap[to.type_id]
                lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
                xs = c.values.astype(np.float64)
                fits = np.array([x == x and abs(x) != float("inf") and lo <= int(x) <= hi for x in xs], bool)
                return try_result(np.array([int(x) if f else 0 for x, f in zip(xs, fits)], np.int64), fits)
            return Col(to, np.array(out, dtype=_np_dtype(S, to)), c.valid)
        if frm.type_id == S.DECIMAL and to.type_id in ints:
            # numeric.rs:426-560: truncate toward zero by 10^scale, then `as`
            d = 10 ** frm.scale
            out, ovf = [], np.zeros(n, bool)
            for i in range(n):
                v = dec_to_int(c.values, i)
                t = abs(v) // d * (-1 if v < 0 else 1)
                if to.type_id == S.INT64:
                    ovf[i] = abs(t) > 2**63 - 1
                    out.append(as_int(t, 64))
                else:
                    ovf[i] = abs(t) > 2**31 - 1
                    v32 = as_int(t, 32)
                    nv = as_int(v32, wrap[to.type_id])
                    ovf[i] |= nv != v32
                    out.append(nv)
            if e.eval_mode == S.ANSI and (ovf & c.ok()).any():
                raise OracleError("CAST_OVERFLOW")
            if e.eval_mode == S.TRY:
                bits = wrap[to.type_id]
                lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
                ts = [abs(dec_to_int(c.values, i)) // d * (-1 if dec_to_int(c.values, i) < 0 else 1) for i in range(n)]
                fits = np.array([lo <= t <= hi for t in ts], bool)
                return try_result(np.array([t if f else 0 for t, f in zip(ts, fits)], np.int64), fits)
            return Col(to, np.array(out, dtype=_np_dtype(S, to)), c.valid)
        if frm.type_id in (S.FLOAT, S.DOUBLE) and to.type_id in (S.TIMESTAMP, S.TIMESTAMP_NTZ):
            # cast_float_to_timestamp (numeric.rs:87-135): NaN / ±Infinity → NULL (ANSI: CAST_INVALID_INPUT); micros = val · 10^6 in double arithmetic,
            # kept when floor(micros) ≤ i64::MAX as f64 and ceil(micros) ≥ i64::MIN as f64 (`micros as i64`, saturating), else NULL (ANSI: CAST_OVERFLOW)
            import math
            out, ok = [], np.zeros(n, bool)
            for i, x in enumerate(c.values.astype(np.float64)):
                x = float(x)
                if x != x or math.isinf(x):
                    if e.eval_mode == S.ANSI and c.ok()[i]:
                        raise OracleError("CAST_INVALID_INPUT")
                    out.append(0)
                    continue
                m = x * 1000000.0
                if not math.isinf(m) and math.floor(m) <= 9223372036854775808.0 and math.ceil(m) >= -9223372036854775808.0:
                    ok[i] = True
                    out.append(max(-2**63, min(2**63 - 1, int(m))))
                else:
                    if e.eval_mode == S.ANSI and c.ok()[i]:
                        raise OracleError("CAST_OVERFLOW")
                    out.append(0)
            v2 = ok if c.valid is None else (c.valid & ok)
            return Col(to, np.array(out, np.int64), None if v2.all() else v2)
        if frm.type_id == S.DECIMAL and to.type_id in (S.TIMESTAMP, S.TIMESTAMP_NTZ):
            # cast_decimal_to_timestamp (numeric.rs:1184-1208): value · 10^6 / 10^scale in 256 bits, truncated toward zero, `as_i128() as i64`
            out = []
            for i in range(n):
                v = dec_to_int(c.values, i) * 1_000_000
                q = abs(v) // 10 ** frm.scale * (-1 if v < 0 else 1)
                out.append(as_int(q, 64))
            return Col(to, np.array(out, np.int64), c.valid)
        if frm.type_id == S.DECIMAL and to.type_id in (S.FLOAT, S.DOUBLE):
            div = float(10.0 ** frm.scale)
            vals = np.array([float(dec_to_int(c.values, i)) / div for i in range(n)], np.float64)     # int → f64 rounds to nearest even, like `as f64`
            return Col(to, vals.astype(_np_dtype(S, to)), c.valid)
        if frm.type_id == S.DOUBLE and to.type_id == S.FLOAT:
            with np.errstate(all="ignore"):
                return Col(to, c.values.astype(np.float32), c.valid)
        if frm.type_id == S.BOOL and to.type_id in ints + (S.FLOAT, S.DOUBLE):
            return Col(to, c.values.astype(_np_dtype(S, to)), c.valid)
        if frm.type_id in ints + (S.FLOAT, S.DOUBLE) and to.type_id == S.BOOL:
            return Col(to, c.values != 0, c.valid)
        if frm.type_id == S.DATE and to.type_id == S.INT32:      # cast.rs:273-276: the days since the epoch, reinterpreted
            return Col(to, c.values.astype(np.int32), c.valid)
        if frm.type_id == S.DECIMAL and to.type_id == S.BOOL:      # spark_cast_decimal_to_boolean (numeric.rs:853-864): value != 0
            return Col(to, np.array([dec_to_int(c.values, i) != 0 for i in range(n)], bool), c.valid)
        if frm.type_id in (S.FLOAT, S.DOUBLE) and to.type_id == S.DECIMAL:
            from . import strcast as C
            vals, valid = [], np.zeros(n, bool)
            for i in range(n):
                v = None
                if c.ok()[i]:
                    v, err = C.float_to_decimal(float(c.values[i]), to.precision, to.scale)
                    if err and e.eval_mode == S.ANSI:
                        raise OracleError("NUMERIC_VALUE_OUT_OF_RANGE")
                valid[i] = v is not None
                vals.append(v or 0)
            return Col(to, ints_to_dec(vals), None if valid.all() else valid)
        if frm.type_id == S.STRING or to.type_id == S.STRING:
            return self._string_cast(e, c)
        temporal = (S.TIMESTAMP, S.TIMESTAMP_NTZ)
        if frm.type_id in temporal + (S.DATE,) or to.type_id in temporal:
            # conversion_funcs/temporal.rs:37-78, cast.rs:395-415, utils.rs:62-87,269-297, numeric.rs:252-267, boolean.rs:33-50
            from . import strcast as C
            tz = getattr(e, "timezone", None) or "UTC"
            ok = c.ok()
            out = np.zeros(n, np.int64)
            for i in range(n):
                if not ok[i]:
                    continue
                v = int(c.values[i])
                if frm.type_id in temporal and to.type_id == S.DATE:
                    out[i] = (C.utc_to_local_us(tz, v) if frm.type_id == S.TIMESTAMP else v) // 86_400_000_000
                elif frm.type_id == S.DATE and to.type_id in temporal:
                    out[i] = C.local_to_utc_us(tz, v * 86_400_000_000) if to.type_id == S.TIMESTAMP else v * 86_400_000_000
                elif frm.type_id in temporal and to.type_id == S.INT64:
                    out[i] = v // 1_000_000
                elif frm.type_id in ints and to.type_id in temporal:
                    out[i] = max(-2**63, min(2**63 - 1, v * 1_000_000))
                elif frm.type_id == S.BOOL and to.type_id in temporal:
                    out[i] = 1 if c.values[i] else 0
                elif frm.type_id == S.TIMESTAMP and to.type_id == S.TIMESTAMP_NTZ:
                    out[i] = C.utc_to_local_us(tz, v)
                elif frm.type_id == S.TIMESTAMP_NTZ and to.type_id == S.TIMESTAMP:
                    out[i] = C.local_to_utc_us(tz, v)
                else:
                    raise NotImplementedError(f"oracle cast {frm} → {to}")
            return Col(to, out.astype(_np_dtype(S, to)), c.valid)
        raise NotImplementedError(f"oracle cast {frm} → {to}")

    def _string_cast(self, e, c: Col) -> Col:
        """Casts from and to strings: oracle/strcast.py restates conversion_funcs/string.rs and numeric.rs:593-704 value by value."""
        from . import strcast as C
        S = self.S
        to, frm = e.dtype, c.dtype
        n = len(c)
        ok = c.ok()
        mode = {S.LEGACY: C.LEGACY, S.ANSI: C.ANSI, S.TRY: C.TRY}[e.eval_mode]
        ints = {S.INT8: 8, S.INT16: 16, S.INT32: 32, S.INT64: 64}
        if frm.type_id == S.STRING:
            vals, valid = [], np.zeros(n, bool)
            for i in range(n):
                v = None
                if ok[i]:
                    b = c.values[i].encode() if isinstance(c.values[i], str) else bytes(c.values[i])
                    if to.type_id == S.BOOL:
                        v, err = C.string_to_bool(b, mode)
                    elif to.type_id in ints:
                        v, err = C.string_to_int(b, mode, ints[to.type_id])
                    elif to.type_id == S.DECIMAL:
                        v, err = C.string_to_decimal(b, to.precision, to.scale, mode)
                    elif to.type_id == S.DATE:
                        v, err = C.string_to_date(b, mode)
                    elif to.type_id in (S.FLOAT, S.DOUBLE):
                        v, err = C.string_to_float(b, mode, to.type_id == S.FLOAT)
                    elif to.type_id == S.TIMESTAMP:
                        v, err = C.string_to_timestamp(b, mode, getattr(e, "timezone", None) or "UTC", bool(getattr(e, "is_spark4_plus", False)))
                    elif to.type_id == S.TIMESTAMP_NTZ:
                        v, err = C.string_to_timestamp_ntz(b, mode)
                    else:
                        raise NotImplementedError(f"oracle cast string → {to}")
                    if err:
                        raise OracleError(err)
                valid[i] = v is not None
                vals.append(0 if v is None else v)
            out = ints_to_dec(vals) if to.type_id == S.DECIMAL else np.array(vals, dtype=_np_dtype(S, to))
            return Col(to, out, None if valid.all() else valid)
        tz = getattr(e, "timezone", None) or "UTC"
        out = np.empty(n, dtype=object)
        for i in range(n):
            if not ok[i]:
                out[i] = None
            elif frm.type_id in ints:
                out[i] = C.int_to_string(int(c.values[i]))
            elif frm.type_id == S.BOOL:
                out[i] = C.bool_to_string(bool(c.values[i]))
            elif frm.type_id == S.DECIMAL:
                out[i] = C.decimal_to_string(dec_to_int(c.values, i), frm.scale, mode)
            elif frm.type_id in (S.FLOAT, S.DOUBLE):
                out[i] = C.float_to_string(c.values[i], frm.type_id == S.FLOAT)
            elif frm.type_id == S.DATE:
                out[i] = C.date_to_string(int(c.values[i]))
            elif frm.type_id in (S.TIMESTAMP, S.TIMESTAMP_NTZ):
                out[i] = C.timestamp_to_string(C.utc_to_local_us(tz, int(c.values[i])) if frm.type_id == S.TIMESTAMP else int(c.values[i]))
            else:
                raise NotImplementedError(f"oracle cast {frm} → string")
        return Col(to, out, c.valid)


# --------------------------------------------------------------------------- operators


def _take(c: Col, idx: np.ndarray) -> Col:
    return Col(c.dtype, c.values[idx], None if c.valid is None else c.valid[idx])


def run_plan(S, op, table) -> List[Col]:
    """Evaluate the operator tree.  `table` is the Scan input, or a list of tables consumed by the Scan leaves in
    depth-first, left-before-right order (planner.rs:1726).  Returns the output columns."""
    if table is None:
        table = []
    if isinstance(table, pa.Table):
        table = [table]
    if not isinstance(table, _ScanQueue):
        table = _ScanQueue(list(table))
    ev = Evaluator(S)
    k = op.kind
    if k in ("scan", "shuffle_scan"):   # ShuffleScan: the decoded blocks of its input, concatenated (shuffle_scan.rs:139-190)
        t = table.pop()
        assert t.num_columns == len(op.fields)
        return [col_from_arrow(S, t.column(i), ty) for i, ty in enumerate(op.fields)]
    if k == "native_scan":
        # independent decoder: Arrow C++ / parquet-cpp through pyarrow (parity for K18 is pinned on it)
        import pyarrow.parquet as papq
        parts = []
        for path, start, length, size in op.files:
            pf = papq.ParquetFile(path)
            md = pf.metadata
            for g in range(md.num_row_groups):
                rg = md.row_group(g)
                c0 = rg.column(0)
                first = c0.dictionary_page_offset if (c0.has_dictionary_page and c0.dictionary_page_offset < c0.data_page_offset) else c0.data_page_offset
                comp = sum(rg.column(i).total_compressed_size for i in range(rg.num_columns))
                mid = first + comp // 2
                if start <= mid < start + length:
                    parts.append(pf.read_row_group(g, columns=list(op.field_names)))
        t = pa.concat_tables(parts) if parts else pa.table({n: pa.array([], type=pa.null()) for n in op.field_names})
        t = t.select(list(op.field_names))
        return [col_from_arrow(S, t.column(i), ty) for i, ty in enumerate(op.fields)]
    if k in ("hash_join", "sort_merge_join", "bnlj"):
        left = run_plan(S, op.children[0], table)
        right = run_plan(S, op.children[1], table)
        out = _hash_join(S, ev, op, left, right)
        if k == "sort_merge_join" and out:
            # SortMergeJoinExec emits in join-key order (planner.rs:2126-2191): order the pairs by the preserved side's keys
            import functools
            n = len(out[0])
            by_right = op.join_type == S.RIGHT_OUTER
            keys = op.right_keys if by_right else op.left_keys
            nl = len(left)
            cols = out[nl:] if by_right else out
            kc = [(ev.eval(e, cols, n), so[1], so[2]) for e, so in zip(keys, op.sort_orders)]

            def cmp(i, j):
                for c, desc, nulls_last in kc:
                    ni, nj = not c.ok()[i], not c.ok()[j]
                    if ni or nj:
                        if ni and nj:
                            continue
                        r = -1 if ni else 1
                        return r if not nulls_last else -r
                    a, b = c.values[i], c.values[j]
                    if a != b:
                        r = -1 if a < b else 1
                        return -r if desc else r
                return 0
            order = np.array(sorted(range(n), key=functools.cmp_to_key(cmp)), dtype=np.int64)
            out = [_take(c, order) for c in out]
        return out
    child = run_plan(S, op.children[0], table)
    n = len(child[0]) if child else 0
    if k == "filter":
        p = ev.eval(op.predicate, child, n)
        keep = p.ok() & p.values.astype(bool)      # only TRUE and valid survives
        idx = np.nonzero(keep)[0]
        return [_take(c, idx) for c in child]
    if k == "projection":
        return [ev.eval(e, child, n) for e in op.exprs]
    if k == "window":
        # WindowAggExec over sorted input (planner.rs:2267-2379): partitions / peer groups are runs of equal keys in the given order
        pk = [ev.eval(e, child, n) for e in op.partition_by]
        okeys = [ev.eval(e, child, n) for e, _, _ in op.sort_orders]

        def cell(c, i):
            if not c.ok()[i]:
                return None
            return dec_to_int(c.values, i) if c.dtype.type_id == S.DECIMAL else (c.values[i].item() if hasattr(c.values[i], "item") else c.values[i])
        ptuple = [tuple(cell(c, i) for c in pk) for i in range(n)]
        otuple = [tuple(cell(c, i) for c in okeys) for i in range(n)]
        ps, gs = [0] * n, [0] * n          # partition start / peer-group start of every row
        for i in range(n):
            newp = i == 0 or ptuple[i] != ptuple[i - 1]
            ps[i] = i if newp else ps[i - 1]
            gs[i] = i if (newp or otuple[i] != otuple[i - 1]) else gs[i - 1]
        pe, ge = [0] * n, [0] * n          # one past the partition / peer-group end
        for i in range(n - 1, -1, -1):
            pe[i] = i + 1 if (i == n - 1 or ps[i + 1] != ps[i]) else pe[i + 1]
            ge[i] = i + 1 if (i == n - 1 or gs[i + 1] != gs[i]) else ge[i + 1]
        def frame_rows(i, frame):
            """rows [start, end) of row i's frame, clipped to its partition: a bound is the partition edge, the current row (ROWS) / its peer group
            (RANGE), the current row ± rows (negative = PRECEDING), or — ("value", literal) — a RANGE value offset"""
            ftype, lo, up = frame
            if isinstance(lo, tuple) or isinstance(up, tuple):
                # RANGE with value offsets (DataFusion WindowFrameStateRange as the reference configures it, planner.rs:3031-3037,
                # 3090-3096): target = key ∓ offset in SORT order, computed in the key's width with wrapping arithmetic, NULL for a
                # NULL key; lower bound = first row that does not sort before its target, upper = first row that sorts after it
                kcol, (_, kdesc, knl) = okeys[0], op.sort_orders[0]
                bits = 8 * kcol.values.dtype.itemsize

                def wrapk(x):
                    x &= (1 << bits) - 1
                    return x - (1 << bits) if x >> (bits - 1) else x
                kv = lambda r: int(kcol.values[r]) if kcol.ok()[r] else None

                def sorts_before(r, t):
                    a = kv(r)
                    if a is None or t is None:
                        return False if (a is None and t is None) else ((a is None) == (not knl))
                    return a > t if kdesc else a < t

                def sorts_after(r, t):
                    a = kv(r)
                    if a is None or t is None:
                        return False if (a is None and t is None) else ((a is None) == bool(knl))
                    return a < t if kdesc else a > t
                cur = kv(i)
            if isinstance(lo, tuple):
                t = None if cur is None else wrapk(cur + int(lo[1].value) if kdesc else cur - int(lo[1].value))
                start = next((r for r in range(ps[i], pe[i]) if not sorts_before(r, t)), pe[i])
            else:
                start = ps[i] if lo == "unbounded" else ((i if ftype == "rows" else gs[i]) if lo == "current" else i + int(lo))
            if isinstance(up, tuple):
                t = None if cur is None else wrapk(cur - int(up[1].value) if kdesc else cur + int(up[1].value))
                end = next((r for r in range(ps[i], pe[i]) if sorts_after(r, t)), pe[i])
            else:
                end = pe[i] if up == "unbounded" else ((i + 1 if ftype == "rows" else ge[i]) if up == "current" else i + int(up) + 1)
            start, end = max(start, ps[i]), min(end, pe[i])
            return start, max(end, start)

        def pick_rows(src, frame, mode, nth, ignore_nulls):
            """first_value / last_value / nth_value: the row of the frame whose value is returned (−1: none), NULLs skipped on request"""
            idx = np.full(n, -1, np.int64)
            okf = src.ok()
            for i in range(n):
                start, end = frame_rows(i, frame)
                rows = [r for r in range(start, end) if okf[r]] if ignore_nulls else list(range(start, end))
                k = 0 if mode == "first" else len(rows) - 1 if mode == "last" else nth - 1
                if 0 <= k < len(rows):
                    idx[i] = rows[k]
            okv = (idx >= 0) & okf[np.maximum(idx, 0)]
            vals = src.values[np.maximum(idx, 0)]
            if src.values.dtype == object:
                vals = np.array([v if o else None for v, o in zip(vals, okv)], dtype=object)
            return Col(src.dtype, vals, None if okv.all() else okv)
        out = list(child)
        for wf in op.window_fns:
            if wf[0] == "agg":
                # SUM / COUNT / AVG over [partition start, frame end): exact integer arithmetic, evaluated like the aggregates'
                # Final step (sum_decimal.rs:264-279, sum_int.rs, avg_decimal.rs:597-689)
                _, a, rtype, (ftype, lo, up) = wf
                arg = ev.eval(a.children[0], child, n)
                if a.kind in ("first", "last"):
                    out.append(pick_rows(arg, (ftype, lo, up), a.kind, 0, a.ignore_nulls))
                    continue
                isdec = arg.dtype.type_id == S.DECIMAL
                ival = [(dec_to_int(arg.values, i) if isdec else int(arg.values[i])) if arg.ok()[i] else None for i in range(n)]
                vals, oks = [], []
                for i in range(n):
                    start, end = frame_rows(i, (ftype, lo, up))
                    win = [v for v in ival[start:max(end, start)] if v is not None]
                    if a.kind in ("min", "max"):
                        vals.append((min(win) if a.kind == "min" else max(win)) if win else 0); oks.append(bool(win)); continue
                    if a.kind == "count":
                        vals.append(len(win)); oks.append(True); continue
                    if not win:
                        vals.append(0); oks.append(False); continue
                    tot = sum(win)
                    if a.kind == "sum" and not isdec:
                        tot &= (1 << 64) - 1
                        vals.append(tot - (1 << 64) if tot >> 63 else tot); oks.append(True); continue
                    st = a.sum_dtype if a.kind == "avg" else a.dtype
                    if abs(tot) > 10**st.precision - 1:
                        vals.append(0); oks.append(False); continue
                    if a.kind == "sum":
                        vals.append(tot); oks.append(True); continue
                    v = tot * 10**max(0, a.dtype.scale - st.scale)
                    c = len(win)
                    q, r = abs(v) // c, abs(v) % c
                    q = q + 1 if r >= (c + 1) // 2 else q
                    q = q if v >= 0 else -q
                    okv = abs(q) <= 10**a.dtype.precision - 1
                    vals.append(q if okv else 0); oks.append(okv)
                okn = np.array(oks, bool)
                if a.kind in ("min", "max"):
                    out.append(Col(arg.dtype, ints_to_dec(vals) if isdec else np.array(vals, dtype=arg.values.dtype), None if okn.all() else okn))
                    continue
                if a.kind == "count" or (a.kind == "sum" and not isdec):
                    out.append(Col(S.T_INT64, np.array(vals, np.int64), None if okn.all() else okn))
                else:
                    out.append(Col(a.dtype, ints_to_dec(vals), None if okn.all() else okn))
                continue
            name, args, rtype = wf[:3]
            if name == "nth_value":
                out.append(pick_rows(ev.eval(args[0], child, n), wf[3], "nth", int(args[1].value), bool(wf[4]) if len(wf) > 4 else False))
                continue
            if name in ("lag", "lead"):
                src = ev.eval(args[0], child, n)
                kk = int(args[1].value) if len(args) > 1 else 1
                sh = -kk if name == "lag" else kk
                if len(wf) > 4 and wf[4] and kk != 0:
                    # IGNORE NULLS: the |kk|-th non-NULL row before (lag) / after (lead) the current one, inside the partition; a negative
                    # offset looks the other way (DataFusion's WindowShift keeps one signed shift: lag(x, -k) = lead(x, k))
                    okf = src.ok()
                    idx = np.full(n, -1, np.int64)
                    back = (name == "lag") == (kk > 0)
                    for i in range(n):
                        rows = [r for r in (range(i - 1, ps[i] - 1, -1) if back else range(i + 1, pe[i])) if okf[r]]
                        if len(rows) >= abs(kk):
                            idx[i] = rows[abs(kk) - 1]
                else:
                    idx = np.array([i + sh if ps[i] <= i + sh < pe[i] else -1 for i in range(n)], dtype=np.int64)
                okv = (idx >= 0) & src.ok()[np.maximum(idx, 0)]
                vals = src.values[np.maximum(idx, 0)]
                if len(args) > 2 and args[2].value is not None:      # rows whose offset row is outside the partition take the default
                    dflt = ev.eval(args[2], child, n)
                    vals = np.where(idx < 0, dflt.values, vals) if vals.dtype != object else vals
                    okv = okv | (idx < 0)
                if src.values.dtype == object:
                    vals = np.array([v if o else None for v, o in zip(vals, okv)], dtype=object)
                out.append(Col(src.dtype, vals, None if okv.all() else okv))
                continue
            rows = [pe[i] - ps[i] for i in range(n)]
            rank = [gs[i] - ps[i] + 1 for i in range(n)]
            if name == "row_number":
                v = np.array([i - ps[i] + 1 for i in range(n)], np.int32)
            elif name == "rank":
                v = np.array(rank, np.int32)
            elif name == "dense_rank":
                dr, cur = [0] * n, 0
                for i in range(n):
                    cur = 1 if ps[i] == i else (cur + 1 if gs[i] == i else cur)
                    dr[i] = cur
                v = np.array(dr, np.int32)
            elif name == "percent_rank":
                v = np.array([(rank[i] - 1) / (rows[i] - 1) if rows[i] > 1 else 0.0 for i in range(n)], np.float64)
            elif name == "cume_dist":
                v = np.array([(ge[i] - ps[i]) / rows[i] for i in range(n)], np.float64)
            elif name == "ntile":
                kb = int(args[0].value)
                res = []
                for i in range(n):
                    i0, q, r = i - ps[i], rows[i] // kb, rows[i] % kb
                    thr = r * (q + 1)
                    res.append(i0 // (q + 1) + 1 if i0 < thr else (i0 + 1 if q == 0 else (i0 - thr) // q + r + 1))
                v = np.array(res, np.int32)
            else:
                raise NotImplementedError(f"oracle: window function {name}")
            out.append(Col(S.T_DOUBLE if v.dtype == np.float64 else S.T_INT32, v, None))
        return out
    if k == "expand":
        # ExpandExec (operators/expand.rs; planner.rs:1913-1948): each projection over the input, results stacked.  Row order between
        # projections is an implementation detail (the reference interleaves per input batch); tests compare multisets.
        parts = []
        for proj in op.projections:
            row = []
            for e in proj:
                if e.kind == "literal" and e.value is None:
                    row.append(None)                      # untyped NULL: takes the column's type from another projection
                else:
                    row.append(ev.eval(e, child, n))
            parts.append(row)
        out = []
        for j in range(len(op.projections[0])):
            proto = next(r[j] for r in parts if r[j] is not None)
            vals, oks = [], []
            for r in parts:
                if r[j] is None:
                    vals.append(np.zeros(n, dtype=proto.values.dtype) if proto.values.dtype != object else np.array([None] * n, dtype=object))
                    oks.append(np.zeros(n, bool))
                else:
                    vals.append(r[j].values)
                    oks.append(r[j].ok())
            ok = np.concatenate(oks)
            out.append(Col(proto.dtype, np.concatenate(vals), None if ok.all() else ok))
        return out
    if k == "hash_agg":
        return _hash_agg(S, ev, op, child, n)
    if k == "limit":
        # LocalLimitExec / GlobalLimitExec (planner.rs:1436-1470): rows [offset, limit)
        lo = min(max(0, op.offset), n)
        hi = n if op.limit < 0 else min(n, op.limit)
        return [_take(c, np.arange(lo, max(lo, hi))) for c in child]
    if k == "sort":
        # SortExec (planner.rs:1488-1522): lexicographic over the sort orders; NULLS FIRST/LAST per key; floats in IEEE totalOrder
        # (arrow-ord); ties in arbitrary order (tests compare the keys and the multiset); fetch, then skip
        import functools
        keys = [(ev.eval(e, child, n), desc, nl) for e, desc, nl in op.sort_orders]

        def total(c, i):
            v = c.values[i]
            if c.dtype.type_id == S.DECIMAL:
                return dec_to_int(c.values, i)
            if c.dtype.type_id in (S.FLOAT, S.DOUBLE):
                bits = np.array([v], dtype=np.float64 if c.dtype.type_id == S.DOUBLE else np.float32).view(np.int64 if c.dtype.type_id == S.DOUBLE else np.int32)[0]
                w = 63 if c.dtype.type_id == S.DOUBLE else 31
                return int(bits) ^ (((1 << w) - 1) if bits < 0 else 0)
            return v.item() if hasattr(v, "item") else v

        def cmp(i, j):
            for c, desc, nulls_last in keys:
                ni, nj = not c.ok()[i], not c.ok()[j]
                if ni or nj:
                    if ni and nj:
                        continue
                    first = i if ni else j            # the NULL row
                    r = -1 if first == i else 1
                    return r if not nulls_last else -r
                a, b = total(c, i), total(c, j)
                if a != b:
                    r = -1 if a < b else 1
                    return -r if desc else r
            return 0
        order = sorted(range(n), key=functools.cmp_to_key(cmp))
        if op.fetch is not None and op.fetch >= 0:
            order = order[:op.fetch]
        if op.skip:
            order = order[op.skip:]
        return [_take(c, np.array(order, dtype=np.int64)) for c in child]
    raise NotImplementedError(k)


class _ScanQueue:
    def __init__(self, tables):
        self.tables = tables

    def pop(self):
        return self.tables.pop(0)


def _key_tuple(S, cols: List[Col], i: int):
    t = []
    for c in cols:
        if c.valid is not None and not c.valid[i]:
            return None                         # NULL never matches (NullEqualsNothing, planner.rs:2225-2227)
        if c.dtype.type_id == S.DECIMAL:
            t.append(dec_to_int(c.values, i))
        else:
            v = c.values[i]
            v = v.item() if hasattr(v, "item") else v
            if isinstance(v, float) and v == 0.0:
                v = 0.0                          # -0.0 == 0.0 for join keys (NormalizeNaNAndZero)
            t.append(v)
    return tuple(t)


def _join_pairs_sorted(lk: List[Col], rk: List[Col], nl: int, nr: int):
    """the matching (left row, right row) pairs of an equi-join on integer-like keys, in the order the dictionary loop below gives them
    (left rows in order, each one's matches by ascending right row) — by a stable sort of the right keys and two binary searches per left
    row instead of one Python call per row (the TPC-H SF1 tests join millions of rows); → None for other key types.  NULL never matches."""
    if not lk or not all(c.values.dtype.kind in "iub" for c in lk + rk):
        return None
    lok, rok = np.ones(nl, bool), np.ones(nr, bool)
    for c in lk:
        lok &= c.ok()
    for c in rk:
        rok &= c.ok()
    if len(lk) == 1:
        a, b = lk[0].values.astype(np.int64), rk[0].values.astype(np.int64)
    else:           # several keys: one id per distinct key tuple over both sides
        rows = np.concatenate([np.stack([c.values.astype(np.int64) for c in lk], axis=1), np.stack([c.values.astype(np.int64) for c in rk], axis=1)])
        _, inv = np.unique(rows, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        a, b = inv[:nl], inv[nl:]
    rrows = np.nonzero(rok)[0]
    order = rrows[np.argsort(b[rrows], kind="stable")]
    bs = b[order]
    lo, hi = np.searchsorted(bs, a, "left"), np.searchsorted(bs, a, "right")
    cnt = np.where(lok, hi - lo, 0)
    total = int(cnt.sum())
    li = np.repeat(np.arange(nl, dtype=np.int64), cnt)
    within = np.arange(total, dtype=np.int64) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    ri = order[np.repeat(lo, cnt) + within].astype(np.int64) if total else np.zeros(0, np.int64)
    return li, ri


def _hash_join(S, ev: "Evaluator", op, left: List[Col], right: List[Col]) -> List[Col]:
    """HashJoinExec restated (planner.rs:2192-2266): Inner / LeftSemi / LeftAnti, optional residual condition over
    left ++ right.  Output order is unspecified in the reference; here probe order (tests compare multisets)."""
    nl = len(left[0]) if left else 0
    nr = len(right[0]) if right else 0
    lk = [ev.eval(e, left, nl) for e in op.left_keys]
    rk = [ev.eval(e, right, nr) for e in op.right_keys]
    fast = _join_pairs_sorted(lk, rk, nl, nr)
    if fast is not None:
        li, ri = fast
    else:
        index = {}
        for i in range(nr):
            k = _key_tuple(S, rk, i)
            if k is not None:
                index.setdefault(k, []).append(i)
        li, ri = [], []
        for i in range(nl):
            k = _key_tuple(S, lk, i)
            for j in (index.get(k, []) if k is not None else []):
                li.append(i)
                ri.append(j)
        li, ri = np.array(li, np.int64), np.array(ri, np.int64)
    pairs = [_take(c, li) for c in left] + [_take(c, ri) for c in right]
    if op.condition is not None and len(li):
        c = ev.eval(op.condition, pairs, len(li))
        keep = c.ok() & c.values.astype(bool)
        li, ri = li[keep], ri[keep]
        pairs = [_take(c2, np.nonzero(keep)[0]) for c2 in pairs]
    if op.join_type == S.INNER:
        return pairs
    matched = np.zeros(nl, bool)
    matched[li] = True
    if op.join_type in (S.LEFT_OUTER, S.RIGHT_OUTER, S.FULL_OUTER):
        # the preserved side's unmatched rows follow, the other side NULL (planner.rs:2448-2460 → DataFusion HashJoinExec)
        rmatched = np.zeros(nr, bool)
        rmatched[ri] = True
        out = pairs

        def null_cols(cols, k):
            return [Col(c.dtype, np.zeros(k, c.values.dtype) if c.values.dtype != object else np.array([""] * k, dtype=object), np.zeros(k, bool)) for c in cols]

        def concat(a: Col, b: Col) -> Col:
            va = a.ok() if (a.valid is not None or b.valid is not None) else None
            return Col(a.dtype, np.concatenate([a.values, b.values]), None if va is None else np.concatenate([a.ok(), b.ok()]))
        if op.join_type in (S.LEFT_OUTER, S.FULL_OUTER):
            sel = np.nonzero(~matched)[0]
            extra = [_take(c, sel) for c in left] + null_cols(right, len(sel))
            out = [concat(a, b) for a, b in zip(out, extra)]
        if op.join_type in (S.RIGHT_OUTER, S.FULL_OUTER):
            sel = np.nonzero(~rmatched)[0]
            extra = null_cols(left, len(sel)) + [_take(c, sel) for c in right]
            out = [concat(a, b) for a, b in zip(out, extra)]
        return out
    sel = np.nonzero(matched if op.join_type == S.LEFT_SEMI else ~matched)[0]
    return [_take(c, sel) for c in left]


def _group_ids(S, keys: List[Col], n: int):
    """Row → group index in first-seen order (DataFusion GroupValues; output order is unspecified in the
    reference, tests compare as multisets)."""
    if not keys:
        return np.zeros(n, np.int64), 1, []
    if n and all(k.values.dtype.kind in "iub" for k in keys):
        # fast path (integer-like keys): np.unique over (is_null, value) pairs; group order is irrelevant
        mat = np.zeros((n, 2 * len(keys)), np.int64)
        for j, k in enumerate(keys):
            ok = k.ok()
            mat[:, 2 * j] = ~ok
            mat[:, 2 * j + 1] = np.where(ok, k.values.astype(np.int64), 0)
        _, first, inv = np.unique(mat, axis=0, return_index=True, return_inverse=True)
        return inv.reshape(-1).astype(np.int64), len(first), None
    tuples = []
    for i in range(n):
        t = []
        for kcol in keys:
            if kcol.valid is not None and not kcol.valid[i]:
                t.append(None)
            elif kcol.dtype.type_id == S.DECIMAL:
                t.append(dec_to_int(kcol.values, i))
            else:
                v = kcol.values[i]
                t.append(v.item() if hasattr(v, "item") else v)
        tuples.append(tuple(t))
    index = {}
    gid = np.zeros(n, np.int64)
    for i, t in enumerate(tuples):
        g = index.get(t)
        if g is None:
            g = len(index)
            index[t] = g
        gid[i] = g
    return gid, len(index), list(index.keys())


def _hash_agg(S, ev: Evaluator, op, child: List[Col], n: int) -> List[Col]:
    grouped = len(op.exprs) > 0
    keys = [ev.eval(e, child, n) for e in op.exprs]
    if op.mode == S.PARTIAL:
        gid, ng, key_vals = _group_ids(S, keys, n)
    else:
        gid, ng, key_vals = _group_ids(S, child[:len(op.exprs)], n) if grouped else (np.zeros(n, np.int64), 1, [])
    out: List[Col] = []
    for ki, kc in enumerate(op.exprs):
        src = keys[ki] if op.mode == S.PARTIAL else child[ki]
        # first row of each group carries the key
        first = np.full(ng, n, np.int64)
        np.minimum.at(first, gid, np.arange(n, dtype=np.int64))
        out.append(_take(src, first))
    # per-expression modes (planner.rs:1274-1345): PartialMerge aggregates of a Partial operator merge the state columns that
    # start at initial_input_buffer_offset (MergeAsPartial, execution/merge_as_partial.rs)
    modes = list(getattr(op, "expr_modes", None) or [])
    state_col = op.initial_input_buffer_offset if modes else len(op.exprs)
    for i, a in enumerate(op.aggs):
        m = modes[i] if modes else op.mode
        if m == S.PARTIAL:
            out += _agg_partial(S, ev, a, child, n, gid, ng, grouped)
        else:
            cols, used = _agg_final(S, a, child, state_col, n, gid, ng, grouped, emit_state=m == S.PARTIAL_MERGE)
            state_col += used
            out += cols
    return out


def _filter_valid(S, ev, a, child, n, base_valid):
    if a.filter is None:
        return base_valid
    f = ev.eval(a.filter, child, n)
    keep = f.ok() & f.values.astype(bool)
    return keep if base_valid is None else (base_valid & keep)


def _agg_partial(S, ev, a, child, n, gid, ng, grouped) -> List[Col]:
    gidc = np.ascontiguousarray(gid)
    if a.kind == "count":
        ok = np.ones(n, bool)
        for ce in a.children:
            ok &= ev.eval(ce, child, n).ok()
        ok = _filter_valid(S, ev, a, child, n, ok)
        cnt = np.bincount(gid[ok], minlength=ng).astype(np.int64) if n else np.zeros(ng, np.int64)
        return [Col(S.T_INT64, cnt, None)]
    v = ev.eval(a.children[0], child, n)
    valid = _filter_valid(S, ev, a, child, n, v.valid)
    vb = None if valid is None else np.ascontiguousarray(valid.astype(np.uint8))
    if a.kind in ("sum", "avg") and a.dtype.type_id == S.DECIMAL:
        st = a.dtype if a.kind == "sum" else a.sum_dtype
        vals = np.ascontiguousarray(v.values)
        if a.kind == "sum":
            states = (SumDecState * ng)()
            for g in range(ng):
                C.o_sumdec_init(ctypes.byref(states[g]))
            if grouped:
                rc = C.o_sumdec_update_groups(states, _p(vals), _p(vb), _p(gidc), ctypes.c_int64(n), st.precision, int(a.eval_mode == S.ANSI))
            else:
                # ungrouped accumulator sees the input batch by batch; is_empty logic is per batch (sum_decimal.rs:246-251)
                rc = 0
                for base in range(0, max(n, 1), 8192):
                    ln = min(8192, n - base)
                    if ln <= 0:
                        break
                    sl = vals[base:base + ln]
                    vbs = None if vb is None else vb[base:base + ln]
                    rc |= C.o_sumdec_update_batch(ctypes.byref(states[0]), _p(np.ascontiguousarray(sl)),
                                                  _p(None if vbs is None else np.ascontiguousarray(vbs)), ctypes.c_int64(ln),
                                                  st.precision, int(a.eval_mode == S.ANSI))
            if rc:
                raise OracleError("ARITHMETIC_OVERFLOW sum")
            sums = ints_to_dec([_limbs_to_int(s.sum) if s.has_sum else 0 for s in states])
            has = np.array([bool(s.has_sum) for s in states], bool)
            empty = np.array([bool(s.is_empty) for s in states], bool)
            return [Col(st, sums, None if has.all() else has), Col(S.T_BOOL, empty, None)]
        states = (AvgDecState * ng)()
        for g in range(ng):
            C.o_avgdec_init(ctypes.byref(states[g]))
        C.o_avgdec_update_groups(states, _p(vals), _p(vb), _p(gidc) if grouped else None, ctypes.c_int64(n), st.precision)
        if grouped:
            nn = np.array([bool(s.is_not_null) for s in states], bool)
            sums = ints_to_dec([_limbs_to_int(s.sum) for s in states])
            cnts = np.array([s.count for s in states], np.int64)
            vv = None if nn.all() else nn
            return [Col(st, sums, vv), Col(S.T_INT64, cnts, vv)]   # both arrays share the null mask (avg_decimal.rs:638-653)
        s = states[0]
        has = s.count > 0 and bool(s.is_not_null)   # ungrouped: sum is None until the first value (avg_decimal.rs:225)
        return [Col(st, ints_to_dec([_limbs_to_int(s.sum) if has else 0]), None if has else np.array([False])),
                Col(S.T_INT64, np.array([s.count], np.int64), None)]
    if a.kind == "sum" and a.dtype.type_id in (S.INT8, S.INT16, S.INT32, S.INT64):
        sums = np.zeros(ng, np.int64)
        has = np.zeros(ng, np.uint8)
        C.o_sumint_update_groups(_p(sums), _p(has), _p(np.ascontiguousarray(v.values.astype(np.int64))), _p(vb), _p(gidc), ctypes.c_int64(n))
        hb = has.astype(bool)
        return [Col(S.T_INT64, np.where(hb, sums, 0), None if hb.all() else hb)]
    if a.kind in ("sum", "avg"):   # float
        sums = np.zeros(ng, np.float64)
        cnts = np.zeros(ng, np.int64)
        C.o_avgf64_update_groups(_p(sums), _p(cnts), _p(np.ascontiguousarray(v.values.astype(np.float64))), _p(vb), _p(gidc), ctypes.c_int64(n))
        if a.kind == "avg":
            if grouped:
                return [Col(S.T_DOUBLE, sums, None), Col(S.T_INT64, cnts, None)]
            some = n > 0
            return [Col(S.T_DOUBLE, sums, None if some else np.array([False])), Col(S.T_INT64, cnts, None)]
        hb = cnts > 0
        return [Col(S.T_DOUBLE, sums, None if hb.all() else hb)]
    if a.kind in ("min", "max") and v.values.dtype.kind in "iuf":
        ok = np.ones(n, bool) if valid is None else valid
        is_f = v.values.dtype.kind == "f"
        init = (np.inf if a.kind == "min" else -np.inf) if is_f else (np.iinfo(np.int64).max if a.kind == "min" else np.iinfo(np.int64).min)
        acc = np.full(ng, init, np.float64 if is_f else np.int64)
        vals_ok = v.values[ok].astype(acc.dtype)
        (np.minimum if a.kind == "min" else np.maximum).at(acc, gid[ok], vals_ok)
        hb = np.bincount(gid[ok], minlength=ng) > 0
        return [Col(v.dtype, np.where(hb, acc, 0).astype(_np_dtype(S, v.dtype)), None if hb.all() else hb)]
    if a.kind in ("min", "max"):
        ok = np.ones(n, bool) if valid is None else valid
        res, has = [], []
        for g in range(ng):
            sel = np.nonzero((gid == g) & ok)[0]
            if len(sel) == 0:
                res.append(0)
                has.append(False)
                continue
            if v.dtype.type_id == S.DECIMAL:
                xs = [dec_to_int(v.values, i) for i in sel]
            else:
                xs = [v.values[i].item() for i in sel]
            res.append(min(xs) if a.kind == "min" else max(xs))
            has.append(True)
        hb = np.array(has, bool)
        vals = ints_to_dec(res) if v.dtype.type_id == S.DECIMAL else np.array(res, dtype=_np_dtype(S, v.dtype))
        return [Col(v.dtype, vals, None if hb.all() else hb)]
    raise NotImplementedError(f"oracle aggregate {a.kind}")


def _agg_final(S, a, child, state_col, n, gid, ng, grouped, emit_state=False):
    """merge_batch of each accumulator over Partial state rows, then evaluate (Final) or state (PartialMerge)."""
    if a.kind == "count":
        c = child[state_col]
        out = np.zeros(ng, np.int64)
        np.add.at(out, gid, c.values.astype(np.int64))
        return [Col(S.T_INT64, out, None)], 1
    if a.kind == "sum" and a.dtype.type_id == S.DECIMAL:
        sc, ec = child[state_col], child[state_col + 1]
        states = (SumDecState * ng)()
        for g in range(ng):
            C.o_sumdec_init(ctypes.byref(states[g]))
        for i in range(n):
            rc = C.o_sumdec_merge(ctypes.byref(states[gid[i]]), ctypes.byref(_i128(dec_to_int(sc.values, i))), int(sc.ok()[i]), int(bool(ec.values[i])),
                                  a.dtype.precision, int(a.eval_mode == S.ANSI))
            if rc:
                raise OracleError("ARITHMETIC_OVERFLOW sum")
        if emit_state:   # state(): (sum Option, is_empty) — sum_decimal.rs:281-295
            vals = [_limbs_to_int(st_.sum) if st_.has_sum else 0 for st_ in states]
            okb = np.array([bool(st_.has_sum) for st_ in states], bool)
            emp = np.array([bool(st_.is_empty) for st_ in states], bool)
            return [Col(a.dtype, ints_to_dec(vals), None if okb.all() else okb), Col(S.T_BOOL, emp, None)], 2
        vals, ok = [], []
        for s in states:
            out = (ctypes.c_uint64 * 2)()
            has = C.o_sumdec_evaluate(ctypes.byref(s), a.dtype.precision, out)
            vals.append(_limbs_to_int(out) if has else 0)
            ok.append(bool(has))
        okb = np.array(ok, bool)
        return [Col(a.dtype, ints_to_dec(vals), None if okb.all() else okb)], 2
    if a.kind == "avg" and a.dtype.type_id == S.DECIMAL and not grouped:
        # AvgDecimalAccumulator::merge_batch (avg_decimal.rs:331-356): arrow sum() skips NULLs, one precision check on the batch total
        sc, cc = child[state_col], child[state_col + 1]
        count = sum(int(cc.values[i]) for i in range(n) if cc.ok()[i])
        parts = [dec_to_int(sc.values, i) for i in range(n) if sc.ok()[i]]
        total = None
        if parts:
            t = sum(parts)
            t = (t + 2**127) % 2**128 - 2**127                       # wrapping i128 add
            total = t if abs(t) <= 10 ** a.sum_dtype.precision - 1 else None
        if emit_state:   # state(): (sum Option, count) — :301-306
            return [Col(a.sum_dtype, ints_to_dec([total or 0]), None if total is not None else np.array([False])),
                    Col(S.T_INT64, np.array([count], np.int64), None)], 2
        out = (ctypes.c_uint64 * 2)()
        has = total is not None and count != 0 and C.o_avgdec_avg_p(ctypes.byref(_i128(total)), ctypes.c_int64(count), a.dtype.precision, a.dtype.scale, a.sum_dtype.scale, out)
        return [Col(a.dtype, ints_to_dec([_limbs_to_int(out) if has else 0]), None if has else np.array([False]))], 2
    if a.kind == "avg" and a.dtype.type_id == S.DECIMAL:
        sc, cc = child[state_col], child[state_col + 1]
        states = (AvgDecState * ng)()
        for g in range(ng):
            C.o_avgdec_init(ctypes.byref(states[g]))
        for i in range(n):
            rc = C.o_avgdec_merge(ctypes.byref(states[gid[i]]), ctypes.byref(_i128(dec_to_int(sc.values, i))), int(sc.ok()[i]), ctypes.c_int64(int(cc.values[i])),
                                  int(cc.ok()[i]), a.sum_dtype.precision, int(a.eval_mode == S.ANSI))
            if rc:
                raise OracleError("ARITHMETIC_OVERFLOW avg")
        if emit_state:   # AvgDecimalGroupsAccumulator::state (avg_decimal.rs:638-653): sums and counts share the is_not_null mask
            okb = np.array([bool(st_.is_not_null) for st_ in states], bool)
            vals = [_limbs_to_int(st_.sum) if st_.is_not_null else 0 for st_ in states]
            cnts = np.array([st_.count if st_.is_not_null else 0 for st_ in states], np.int64)
            v = None if okb.all() else okb
            return [Col(a.sum_dtype, ints_to_dec(vals), v), Col(S.T_INT64, cnts, v)], 2
        vals, ok = [], []
        for s in states:
            # evaluate (avg_decimal.rs:597-636): under ANSI an overflowed sum below a count is an error (:610-616), not NULL
            if a.eval_mode == S.ANSI and not s.is_not_null and s.count > 0:
                raise OracleError("ARITHMETIC_OVERFLOW avg")
            out = (ctypes.c_uint64 * 2)()
            has = C.o_avgdec_evaluate(ctypes.byref(s), a.dtype.precision, a.dtype.scale, a.sum_dtype.scale, out)
            vals.append(_limbs_to_int(out) if has else 0)
            ok.append(bool(has))
        okb = np.array(ok, bool)
        return [Col(a.dtype, ints_to_dec(vals), None if okb.all() else okb)], 2
    if a.kind == "sum" and a.dtype.type_id in (S.INT8, S.INT16, S.INT32, S.INT64):
        c = child[state_col]
        sums = np.zeros(ng, np.int64)
        has = np.zeros(ng, np.uint8)
        C.o_sumint_update_groups(_p(sums), _p(has), _p(np.ascontiguousarray(c.values.astype(np.int64))),
                                 _p(None if c.valid is None else np.ascontiguousarray(c.valid.astype(np.uint8))),
                                 _p(np.ascontiguousarray(gid)), ctypes.c_int64(n))
        hb = has.astype(bool)
        return [Col(S.T_INT64, np.where(hb, sums, 0), None if hb.all() else hb)], 1
    if a.kind == "avg":
        sc, cc = child[state_col], child[state_col + 1]
        sums = np.zeros(ng, np.float64)
        cnts = np.zeros(ng, np.int64)
        for i in range(n):
            if sc.ok()[i]:
                sums[gid[i]] += sc.values[i]
            cnts[gid[i]] += int(cc.values[i])
        if emit_state:   # (sum, count) — avg.rs:139-176; the ungrouped sum is Some once a non-NULL partial sum arrived
            anysum = np.zeros(ng, bool)
            for i in range(n):
                if sc.ok()[i]:
                    anysum[gid[i]] = True
            v = None if (grouped or anysum.all()) else anysum
            return [Col(S.T_DOUBLE, sums, v), Col(S.T_INT64, cnts, None)], 2
        hb = cnts > 0
        with np.errstate(all="ignore"):
            res = np.where(hb, sums / np.where(hb, cnts, 1), 0.0)
        return [Col(S.T_DOUBLE, res, None if hb.all() else hb)], 2
    if a.kind in ("min", "max"):
        c = child[state_col]
        res, has = [], []
        for g in range(ng):
            sel = np.nonzero((gid == g) & c.ok())[0]
            if len(sel) == 0:
                res.append(0)
                has.append(False)
                continue
            xs = [dec_to_int(c.values, i) for i in sel] if c.dtype.type_id == S.DECIMAL else [c.values[i].item() for i in sel]
            res.append(min(xs) if a.kind == "min" else max(xs))
            has.append(True)
        hb = np.array(has, bool)
        vals = ints_to_dec(res) if c.dtype.type_id == S.DECIMAL else np.array(res, dtype=_np_dtype(S, c.dtype))
        return [Col(c.dtype, vals, None if hb.all() else hb)], 1
    raise NotImplementedError(f"oracle final aggregate {a.kind}")


class _I128(ctypes.Structure):
    _fields_ = [("lo", ctypes.c_uint64), ("hi", ctypes.c_uint64)]


def _i128(v: int) -> _I128:
    v &= (1 << 128) - 1
    return _I128(v & 0xFFFFFFFFFFFFFFFF, v >> 64)


def partition_starts_and_indices(pids: np.ndarray, num_partitions: int):
    """multi_partition.rs:54-103 → (partition_starts[P+1] uint32, partition_row_indices[n] uint32)."""
    pids = np.ascontiguousarray(pids, dtype=np.int32)
    starts = np.zeros(num_partitions + 1, np.uint32)
    idx = np.zeros(max(len(pids), 1), np.uint32)
    _lib().o_partition_starts_and_indices(_p(pids), ctypes.c_int64(len(pids)), ctypes.c_int32(num_partitions), _p(starts), _p(idx))
    return starts, idx[:len(pids)]


def hash_partition_ids(S, table: pa.Table, key_cols, num_partitions: int) -> np.ndarray:
    """Spark HashPartitioning ids: murmur3 seed 42 chained over the key columns (hash_funcs/utils.rs:573-760), pmod
    (comet_partitioning.rs:51-57) — the shuffle writer's computation at multi_partition.rs:296-312."""
    C = _lib()
    n = table.num_rows
    h = np.full(max(n, 1), 42, np.uint32)
    for c in key_cols:
        arr = table.column(c).combine_chunks() if isinstance(table.column(c), pa.ChunkedArray) else table.column(c)
        t = arr.type
        vb = None
        if arr.null_count:
            vb = np.asarray(arr.is_valid()).astype(np.uint8)
        if pa.types.is_decimal(t):
            col = col_from_arrow(S, arr, S.from_arrow_type(t))
            C.o_murmur3_decimal(_p(np.ascontiguousarray(col.values)), ctypes.c_int32(t.precision), _p(vb), ctypes.c_int64(n), _p(h))
        elif t in (pa.int64(),) or pa.types.is_timestamp(t):
            a = np.ascontiguousarray(np.asarray(arr.cast(pa.int64()).fill_null(0)), dtype=np.int64)
            C.o_murmur3_i64(_p(a), _p(vb), ctypes.c_int64(n), _p(h))
        elif t in (pa.int32(), pa.date32(), pa.int16(), pa.int8()):
            a = np.ascontiguousarray(np.asarray(arr.cast(pa.int32()).fill_null(0)), dtype=np.int32)
            C.o_murmur3_i32(_p(a), _p(vb), ctypes.c_int64(n), _p(h))
        elif pa.types.is_boolean(t):   # booleans hash as i32 0 / 1 (hash_funcs/utils.rs:573-600)
            a = np.ascontiguousarray(np.asarray(arr.fill_null(False)).astype(np.int32))
            C.o_murmur3_i32(_p(a), _p(vb), ctypes.c_int64(n), _p(h))
        elif t == pa.float32():
            a = np.ascontiguousarray(np.asarray(arr.fill_null(0)), dtype=np.float32)
            C.o_murmur3_f32(_p(a), _p(vb), ctypes.c_int64(n), _p(h))
        elif t == pa.float64():
            a = np.ascontiguousarray(np.asarray(arr.fill_null(0)), dtype=np.float64)
            C.o_murmur3_f64(_p(a), _p(vb), ctypes.c_int64(n), _p(h))
        elif pa.types.is_string(t) or pa.types.is_binary(t):
            enc = [b"" if v is None else (v.encode() if isinstance(v, str) else v) for v in arr.to_pylist()]
            offs = np.zeros(n + 1, np.int32)
            offs[1:] = np.cumsum([len(e) for e in enc])
            data = np.frombuffer(b"".join(enc) + b"\0", np.uint8).copy()
            C.o_murmur3_utf8(_p(offs), _p(data), _p(vb), ctypes.c_int64(n), _p(h))
        else:
            raise OracleError(f"hash partitioning on {t} is not restated")
    out = np.zeros(max(n, 1), np.int32)
    C.o_pmod_array(_p(h), ctypes.c_int64(n), ctypes.c_int32(num_partitions), _p(out))
    return out[:n]


def run_plan_to_arrow(S, op, table) -> pa.Table:
    cols = run_plan(S, op, table)
    return pa.table([col_to_arrow(S, c) for c in cols], names=[f"col_{i}" for i in range(len(cols))])


# --------------------------------------------------------------------------- CPU baseline (bench.py)


def q6_reference_pipeline(table: pa.Table, d0: int, d1: int, disc_lo: int, disc_hi: int, qty_lt: int, batch: int = 8192):
    """Times nothing itself: runs the operator-at-a-time Q6 stage-1 pipeline of comet_oracle.c over the table."""
    n = table.num_rows
    cols = [table.column(i).combine_chunks() for i in range(4)]
    qty, price, disc = (np.frombuffer(c.buffers()[1], dtype=DEC128)[c.offset:c.offset + n] for c in cols[:3])
    ship = np.frombuffer(cols[3].buffers()[1], dtype=np.int32)[cols[3].offset:cols[3].offset + n]
    out = (ctypes.c_uint64 * 2)()
    has, empty = ctypes.c_int32(), ctypes.c_int32()
    lits = (_I128 * 3)(_i128(disc_lo), _i128(disc_hi), _i128(qty_lt))
    C.o_q6_reference_pipeline(_p(np.ascontiguousarray(qty)), _p(np.ascontiguousarray(price)), _p(np.ascontiguousarray(disc)),
                              _p(np.ascontiguousarray(ship)), ctypes.c_int64(n), ctypes.c_int32(d0), ctypes.c_int32(d1),
                              lits, ctypes.c_int64(batch), out, ctypes.byref(has), ctypes.byref(empty))
    return _limbs_to_int(out), bool(has.value), bool(empty.value)


def q1_reference_pipeline(table: pa.Table, cutoff: int, batch: int = 8192, max_groups: int = 64) -> dict:
    """Runs comet_oracle.c's operator-at-a-time Q1 stage-1 pipeline over a lineitem_q1-shaped table (columns l_quantity,
    l_extendedprice, l_discount, l_tax, l_returnflag, l_linestatus, l_shipdate).  Returns {(returnflag, linestatus):
    {"sums": [7 exact ints: sum_qty, sum_price, sum_disc_price, sum_charge, avg sums ×3], "present": [7 bools],
    "counts": [avg counts ×3, count(1)]}} — the Partial states of q1_plan()."""
    n = table.num_rows
    cols = [table.column(i).combine_chunks() if isinstance(table.column(i), pa.ChunkedArray) else table.column(i) for i in range(7)]
    dec = [np.ascontiguousarray(np.frombuffer(c.buffers()[1], dtype=DEC128)[c.offset:c.offset + n]) for c in cols[:4]]

    def utf8(c):
        off = np.ascontiguousarray(np.frombuffer(c.buffers()[1], dtype=np.int32)[c.offset:c.offset + n + 1])
        data = np.frombuffer(c.buffers()[2], dtype=np.uint8) if c.buffers()[2] is not None else np.zeros(1, np.uint8)
        return off, data
    rf_off, rf_b = utf8(cols[4])
    ls_off, ls_b = utf8(cols[5])
    ship = np.ascontiguousarray(np.frombuffer(cols[6].buffers()[1], dtype=np.int32)[cols[6].offset:cols[6].offset + n])
    keys = np.zeros(max_groups * 32, np.uint8)
    sums = np.zeros(max_groups * 7, DEC128)
    flags = np.zeros(max_groups * 7, np.uint8)
    counts = np.zeros(max_groups * 4, np.int64)
    C.o_q1_reference_pipeline.restype = ctypes.c_int64
    ng = C.o_q1_reference_pipeline(_p(dec[0]), _p(dec[1]), _p(dec[2]), _p(dec[3]), _p(rf_off), _p(rf_b), _p(ls_off), _p(ls_b), _p(ship),
                                   ctypes.c_int64(n), ctypes.c_int32(cutoff), ctypes.c_int64(batch), ctypes.c_int64(max_groups),
                                   _p(keys), _p(sums), _p(flags), _p(counts))
    if ng < 0:
        raise OracleError("more groups than max_groups")
    out = {}
    for g in range(ng):
        k = keys[g * 32:(g + 1) * 32]
        key = tuple(bytes(k[16 * j + 1:16 * j + 1 + int(k[16 * j])]).decode() for j in range(2))
        out[key] = {"sums": [_limbs_to_int((sums["lo"][g * 7 + a], int(sums["hi"][g * 7 + a]) & 0xFFFFFFFFFFFFFFFF)) for a in range(7)],
                    "present": [bool(flags[g * 7 + a]) for a in range(7)],
                    "counts": [int(counts[g * 4 + a]) for a in range(4)]}
    return out
