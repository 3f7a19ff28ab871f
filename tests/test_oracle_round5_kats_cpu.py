"""The oracle's round-5 restatements pinned on the reference's own test vectors (tests/golden/reference_kats.json: "regexp_and_split" transcribed from
string_funcs/{regexp_extract,regexp_extract_all,split}.rs, "dates_and_math" from datetime_funcs/{next_day,make_date}.rs, math_funcs/{pow,log}.rs and Spark's
documented answers) — and the device's own code, run on the host, on the same vectors."""
import datetime as dt
import json
import math
import os

import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S
from oracle import oracle as O

K = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
STR, I32, F64, D = S.T_STRING, S.T_INT32, S.T_DOUBLE, S.T_DATE
E = dt.date(1970, 1, 1)


def _eval(expr, types, table):
    return O.run_plan_to_arrow(S, S.project(S.scan(types), [expr]), table).column(0).to_pylist()


def test_regexp_extract_and_all_and_split():
    k = K["regexp_and_split"]
    for c in k["regexp_extract"]:
        t = pa.table({"s": pa.array(c["values"], pa.utf8())})
        e = S.scalar_func("regexp_extract", [S.col(0, STR), S.lit(c["pattern"], STR), S.lit(c["idx"], I32)], STR)
        assert _eval(e, [STR], t) == c["expected"], c
        assert [None if v is None else native.regexp_extract_host(c["pattern"], c["idx"], v)[1] for v in c["values"]] == c["expected"], c
    for c in k["regexp_extract_all"]:
        t = pa.table({"s": pa.array(c["values"], pa.utf8())})
        e = S.scalar_func("regexp_extract_all", [S.col(0, STR), S.lit(c["pattern"], STR), S.lit(c["idx"], I32)], S.list_type(STR, True))
        assert _eval(e, [STR], t) == c["expected"], c
        assert [None if v is None else native.extract_all_host(c["pattern"], c["idx"], v) for v in c["values"]] == c["expected"], c
    for c in k["split"]:
        assert O.split_like_the_crate(c["pattern"], c["value"], c["limit"]) == c["expected"], c
        assert native.split_host(c["pattern"], c["limit"], c["value"]) == c["expected"], c
    for c in k["errors"]:
        t = pa.table({"s": pa.array(["abc"])})
        e = S.scalar_func(c["fn"], [S.col(0, STR), S.lit(c["pattern"], STR), S.lit(c["idx"], I32)], STR)
        with pytest.raises(O.OracleError, match=c["message"]):
            _eval(e, [STR], t)
        with pytest.raises(native.CometNativeException, match=c["message"]):
            native.regexp_extract_host(c["pattern"], c["idx"], "abc")


def _days(s):
    return (dt.date.fromisoformat(s) - E).days


def test_dates():
    k = K["dates_and_math"]
    names = {"MO": 0, "MON": 0, "MONDAY": 0, "TU": 1, "TUE": 1, "TUESDAY": 1, "WE": 2, "WED": 2, "WEDNESDAY": 2, "TH": 3, "THU": 3, "THURSDAY": 3, "FR": 4, "FRI": 4, "FRIDAY": 4, "SA": 5, "SAT": 5,
             "SATURDAY": 5, "SU": 6, "SUN": 6, "SUNDAY": 6}
    for c in k["next_day"]:
        t = pa.table({"d": pa.array([_days(c["date"])], pa.int32()).cast(pa.date32())})
        got = _eval(S.scalar_func("next_day", [S.col(0, D), S.lit(c["day"], STR)], D), [D], t)[0]
        assert got == (None if c["expected"] is None else dt.date.fromisoformat(c["expected"])), c
        if c["day"].upper() in names:
            assert native.date_fn_host(5, _days(c["date"]), names[c["day"].upper()]) == _days(c["expected"]), c
    for c in k["make_date"]:
        y, m, d = c["ymd"]
        t = pa.table({"y": pa.array([y], pa.int32()), "m": pa.array([m], pa.int32()), "d": pa.array([d], pa.int32())})
        got = _eval(S.scalar_func("make_date", [S.col(0, I32), S.col(1, I32), S.col(2, I32)], D), [I32, I32, I32], t)[0]
        assert (None if got is None else (got - E).days) == c["expected"], c
        assert native.date_fn_host(6, y, m, d) == c["expected"], c
    for c in k["documented"]:
        x = _days(c["date"])
        t = pa.table({"d": pa.array([x], pa.int32()).cast(pa.date32())})
        if c["fn"] == "weekday":      # CometWeekDay: datepart('isodow') − 1
            assert _eval(S.date_part("isodow", S.col(0, D)), [D], t)[0] - 1 == c["expected"] and native.date_fn_host(1, x) - 1 == c["expected"]
        elif c["fn"] == "weekofyear":
            assert _eval(S.date_part("week", S.col(0, D)), [D], t)[0] == c["expected"] and native.date_fn_host(2, x) == c["expected"]
        elif c["fn"] == "last_day":
            assert _eval(S.scalar_func("last_day", [S.col(0, D)], D), [D], t)[0] == dt.date.fromisoformat(c["expected"]) and native.date_fn_host(4, x) == _days(c["expected"])
        else:
            unit = {"YEAR": 0, "QUARTER": 1, "MM": 2, "WEEK": 3}[c["fmt"].upper()]
            assert _eval(S.scalar_func("date_trunc", [S.col(0, D), S.lit(c["fmt"], STR)], D), [D], t)[0] == dt.date.fromisoformat(c["expected"]) and native.date_fn_host(3, x, unit) == _days(c["expected"])


def test_pow_and_log():
    k = K["dates_and_math"]
    f = lambda v: float(v) if isinstance(v, str) else v
    for c in k["pow"]:
        t = pa.table({"b": pa.array([f(c["base"])], pa.float64()), "e": pa.array([f(c["exp"])], pa.float64())})
        got = _eval(S.scalar_func("pow", [S.col(0, F64), S.col(1, F64)], F64), [F64, F64], t)[0]
        want = f(c["expected"])
        assert (math.isnan(got) and math.isnan(want)) or got == want, c
    for c in k["spark_log"]:
        t = pa.table({"b": pa.array([c["base"]], pa.float64()), "v": pa.array([c["value"]], pa.float64())})
        got = _eval(S.scalar_func("spark_log", [S.col(0, F64), S.col(1, F64)], F64), [F64, F64], t)[0]
        assert got == c["expected"] or (got is not None and abs(got - c["expected"]) < 1e-12), c


def test_size_and_list_extract():
    k = K["lists"]
    LI = S.list_type(I32, True)
    for c in k["size"]:
        t = pa.table({"l": pa.array(c["lists"], pa.list_(pa.int32()))})
        assert _eval(S.scalar_func("size", [S.col(0, LI)], I32), [LI], t) == c["expected"], c
    for c in k["list_extract"]:
        t = pa.table({"l": pa.array(c["lists"], pa.list_(pa.int32()))})
        assert _eval(S.list_extract(S.col(0, LI), S.lit(c["index"], I32), one_based=c["one_based"]), [LI], t) == c["expected"], c
    with pytest.raises(O.OracleError, match="INVALID_INDEX_OF_ZERO"):
        _eval(S.list_extract(S.col(0, LI), S.lit(0, I32), one_based=True), [LI], pa.table({"l": pa.array([[1]], pa.list_(pa.int32()))}))
