"""The Spark error JSON of errors that name the offending value (csrc/err_sites.cpp through comet_error_json): the keys the JVM side reads back
(spark/src/main/spark-3.5/…/ShimSparkErrorConverter.scala: params("value"), params("precision"), params("scale"), params("fromType"),
params("toType") — a missing one is a NoSuchElementException there) and the value formats of the reference's raise sites
(common/src/error.rs:318-380, 769-775; conversion_funcs/numeric.rs:282-305, 335-349, 440-585, 755-765, 938-948; string.rs:39-70, 219, 359, 1117)."""
import struct

from datafusion_comet_amd import native

NVOOR = ("NumericValueOutOfRange", "NUMERIC_VALUE_OUT_OF_RANGE.WITH_SUGGESTION")
OVF = ("CastOverFlow", "CAST_OVERFLOW")


def _f64(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def _f32(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def test_numeric_value_out_of_range():
    # decimal_overflow_error (error.rs:769-775): the UNSCALED i128, to_string(); error.rs:941-953's own test expects the keys value / precision / scale
    j = native.error_json(*NVOOR, 0, lo=99999, precision=5, scale=2)
    assert j == {"errorType": "NumericValueOutOfRange", "errorClass": "NUMERIC_VALUE_OUT_OF_RANGE.WITH_SUGGESTION", "params": {"value": "99999", "precision": 5, "scale": 2}}
    big = -(10**30 + 7)
    j = native.error_json(*NVOOR, 0, lo=big & (2**64 - 1), hi=(big >> 64) & (2**64 - 1), precision=28, scale=10)
    assert j["params"] == {"value": str(big), "precision": 28, "scale": 10}
    # cast_int_to_decimal128 (numeric.rs:760): the input integer; numeric.rs:1396-1400's test: 1000 → decimal(3,2)
    assert native.error_json(*NVOOR, 7, lo=1000, precision=3, scale=2)["params"] == {"value": "1000", "precision": 3, "scale": 2}
    assert native.error_json(*NVOOR, 7, lo=-5 & (2**64 - 1), precision=3, scale=2)["params"]["value"] == "-5"
    # cast_float_to_decimal128 (numeric.rs:943): Rust's Display of the f64; numeric.rs:1566-1571 and :1739-1744 expect "4242.42" and "99.995"
    for x, want in [(4242.42, "4242.42"), (99.995, "99.995"), (1e21, "1000000000000000000000"), (-0.5, "-0.5"), (1e-7, "0.0000001"), (123456789.0, "123456789")]:
        assert native.error_json(*NVOOR, 6, lo=_f64(x), precision=5, scale=2)["params"]["value"] == want, x


def test_cast_overflow():
    # cast_int_to_int_macro: value.to_string() + the source type's literal suffix (numeric.rs:296-300, 828-845)
    j = native.error_json(*OVF, 1, lo=2147483648, from_type="BIGINT", to_type="INT", suffix="L")
    assert j == {"errorType": "CastOverFlow", "errorClass": "CAST_OVERFLOW", "params": {"value": "2147483648L", "fromType": "BIGINT", "toType": "INT"}}
    assert native.error_json(*OVF, 1, lo=-129 & (2**64 - 1), from_type="SMALLINT", to_type="TINYINT", suffix="S")["params"]["value"] == "-129S"
    assert native.error_json(*OVF, 1, lo=40000, from_type="INT", to_type="SMALLINT")["params"]["value"] == "40000"
    # cast_float_to_int*: format!("{:e}D", v).replace("e", "E") for doubles, "{:e}" for floats (numeric.rs:335-349, 1028-1118)
    for x, want in [(1e10, "1E10D"), (3.0e9, "3E9D"), (-2.5e19, "-2.5E19D"), (1.5e-7, "1.5E-7D"), (float("nan"), "NaND"), (float("inf"), "infD"), (123456789012.5, "1.234567890125E11D")]:
        assert native.error_json(*OVF, 2, lo=_f64(x), from_type="DOUBLE", to_type="INT")["params"]["value"] == want, x
    for x, want in [(3.0e9, "3E9"), (1.5e10, "1.5E10"), (float("nan"), "NaN")]:
        assert native.error_json(*OVF, 3, lo=_f32(x), from_type="FLOAT", to_type="INT")["params"]["value"] == want, x
    # cast_decimal_to_int*: format_decimal_str(value, p, s) + "BD", from "DECIMAL(p,s)" (numeric.rs:440-585)
    for unscaled, p, s, want in [(1234567890123, 15, 2, "12345678901.23BD"), (-1234567890123, 15, 2, "-12345678901.23BD"), (5, 10, 3, "0.005BD"), (-5, 10, 3, "-0.005BD"),
                                 (123456, 6, 0, "123456BD"), (10**20, 38, 0, "1" + "0" * 20 + "BD")]:
        j = native.error_json(*OVF, 5, lo=unscaled & (2**64 - 1), hi=(unscaled >> 64) & (2**64 - 1), from_type=f"DECIMAL({p},{s})", to_type="INT", precision=p, scale=s)
        assert j["params"] == {"value": want, "fromType": f"DECIMAL({p},{s})", "toType": "INT"}, (unscaled, p, s)


def test_strings_that_do_not_parse():
    # invalid_value(raw value, "STRING", type name) (string.rs:1117; error.rs:961-970's test expects value "abc", fromType STRING, toType INT)
    j = native.error_json("CastInvalidValue", "CAST_INVALID_INPUT", 4, lo=3, from_type="STRING", to_type="INT", string=b"abc")
    assert j == {"errorType": "CastInvalidValue", "errorClass": "CAST_INVALID_INPUT", "params": {"value": "abc", "fromType": "STRING", "toType": "INT"}}
    # the raw value with its whitespace, quotes and backslashes survives JSON (InvalidInputInCastToDatetime, string.rs:53-63)
    raw = ' 2020-13-01 "x"\\\t'.encode()
    j = native.error_json("InvalidInputInCastToDatetime", "CAST_INVALID_INPUT", 4, lo=len(raw), from_type="STRING", to_type="TIMESTAMP_NTZ", string=raw)
    assert j["params"] == {"value": raw.decode(), "fromType": "STRING", "toType": "TIMESTAMP_NTZ"}
    # a value longer than the error block keeps its first bytes
    long = ("x" * 300).encode()
    j = native.error_json("CastInvalidValue", "CAST_INVALID_INPUT", 4, lo=300, from_type="STRING", to_type="DECIMAL(10,2)", string=long[:224])
    assert j["params"]["value"] == "x" * 224 + "..." and j["params"]["toType"] == "DECIMAL(10,2)"
    assert native.error_json("CastInvalidValue", "CAST_INVALID_INPUT", 4, lo=0, from_type="STRING", to_type="BOOLEAN", string=b"")["params"]["value"] == ""
    assert native.error_json("CastInvalidValue", "CAST_INVALID_INPUT", 4, lo=2, from_type="STRING", to_type="INT", string="é".encode())["params"]["value"] == "é"


def test_arithmetic_overflow_names_the_type():
    # ArithmeticOverflow { from_type } (error.rs:369-373): negative.rs:136-150 says "byte" / "short" / "integer" / "long", abs.rs:205-255 "Int8" … "Int64"
    j = native.error_json("ArithmeticOverflow", "ARITHMETIC_OVERFLOW", 8, from_type="long")
    assert j == {"errorType": "ArithmeticOverflow", "errorClass": "ARITHMETIC_OVERFLOW", "params": {"fromType": "long"}}


def _ctx_plan(ctx, pool=None):
    from datafusion_comet_amd import serde as S
    fields = [S.T_STRING, S.T_INT32]
    e = S.with_context(S.cast(S.col(0, S.T_STRING), S.T_INT32, S.ANSI), 17, **ctx)
    plan = S.project(S.scan(fields), [e, S.math("remainder", S.col(1, S.T_INT32), S.col(1, S.T_INT32), S.T_INT32, S.ANSI)])
    if pool is not None:
        plan.sql_text_pool = pool
    return plan.encode()


def test_errors_carry_the_query_context():
    """SparkErrorWithContext::to_json (error.rs:806-831) and QueryContext::format_summary (query_context.rs:104-158; its tests :303-330 expect
    "== SQL of VIEW v1 (line 1, position 8) ==", the text and three carets for "a/b"): the expression's context, registered under its expr_id
    (planner.rs:302-316), travels with the error its raise site raises"""
    sql = "SELECT a/b FROM t"
    j = native.plan_error_json(_ctx_plan(dict(sql_text=sql, start_index=7, stop_index=9, object_type="VIEW", object_name="v1", line=1, start_position=7)), 0, lo=3, string=b"abc")
    assert j["errorType"] == "CastInvalidValue" and j["params"] == {"value": "abc", "fromType": "STRING", "toType": "INT"}
    assert j["context"] == {"sqlText": sql, "startIndex": 7, "stopIndex": 9, "objectType": "VIEW", "objectName": "v1", "line": 1, "startPosition": 7}
    assert j["summary"] == "== SQL of VIEW v1 (line 1, position 8) ==\nSELECT a/b FROM t\n       ^^^"
    # without an object; the text from the root operator's pool (expr.proto:137-141, operator.proto:39-47); a fragment with a two-byte character
    # (query_context.rs:396-404: "café" is characters 7..10)
    sql2 = "SELECT café FROM t"
    j = native.plan_error_json(_ctx_plan(dict(sql_text_idx=1, start_index=7, stop_index=10, line=1, start_position=7), pool=["SELECT 1", sql2]), 0, lo=1, string=b"x")
    assert j["context"] == {"sqlText": sql2, "startIndex": 7, "stopIndex": 10, "objectType": None, "objectName": None, "line": 1, "startPosition": 7}
    assert j["summary"] == "== SQL (line 1, position 8) ==\nSELECT café FROM t\n       ^^^^"
    # only expressions that carry a context have one: the remainder beside the cast does not
    import pytest
    with pytest.raises(native.CometNativeException, match="1 raise sites with a QueryContext"):
        native.plan_error_json(_ctx_plan(dict(sql_text=sql, start_index=7, stop_index=9, line=1, start_position=7)), 1)


def test_the_kernel_text_does_not_depend_on_the_sql_text(tmp_path, monkeypatch):
    """the code-object cache is keyed by the generated text: two queries that differ only in their SQL text (every query, under Spark 4's ANSI
    default) must share their kernels — site ids come from the raise site's content and ordinal, the context is kept beside the pipeline"""
    import os
    from datafusion_comet_amd import serde as S

    def plan(sql):
        e = S.with_context(S.cast(S.col(0, S.T_STRING), S.decimal(17, 3), S.ANSI), 5, sql_text=sql, start_index=7, stop_index=20, line=1, start_position=7)
        return S.project(S.scan([S.T_STRING]), [e]).encode()
    monkeypatch.setenv("COMET_JIT_DUMP_DIR", str(tmp_path))
    native.compile_plan(plan("SELECT CAST(s AS DECIMAL(17,3)) FROM t"))
    first = sorted(os.listdir(tmp_path))
    assert first and any("err_detail_str" in open(tmp_path / f).read() for f in first)
    native.compile_plan(plan("select cast(s as decimal(17,3)) from another_table -- v2"))
    assert sorted(os.listdir(tmp_path)) == first          # nothing new was generated: the second plan found its kernels compiled


def test_decimal_sum_overflow_carries_the_aggregates_context():
    """sum_decimal.rs / avg_decimal.rs wrap_error_with_context: the aggregate's own QueryContext (AggExpr.query_context = 90 under expr_id = 91); the
    flag a kernel raises does not say WHICH sum overflowed, so the context is attached only when the pipeline's ANSI sums agree on it"""
    from datafusion_comet_amd import serde as S
    D = S.decimal(10, 2)
    sql = "SELECT sum(v), sum(w) FROM t"
    ctx = dict(sql_text=sql, start_index=7, stop_index=12, line=1, start_position=7)
    one = S.hash_agg(S.scan([D, D]), [], [S.with_context(S.sum_(S.col(0, D), D, S.ANSI), 3, **ctx)], S.PARTIAL)
    j = native.plan_error_json(one.encode(), -1)
    assert j["errorType"] == "DecimalSumOverflow" and j["errorClass"] == "ARITHMETIC_OVERFLOW" and j["params"] == {"functionName": "sum"}
    assert j["context"]["sqlText"] == sql and j["summary"] == "== SQL (line 1, position 8) ==\n" + sql + "\n" + " " * 7 + "^" * 6
    two = S.hash_agg(S.scan([D, D]), [], [S.with_context(S.sum_(S.col(0, D), D, S.ANSI), 3, **ctx),
                                           S.with_context(S.sum_(S.col(1, D), D, S.ANSI), 4, sql_text=sql, start_index=15, stop_index=20, line=1, start_position=15)], S.PARTIAL)
    j = native.plan_error_json(two.encode(), -1)
    assert j["params"] == {"functionName": "sum"} and "context" not in j
    assert native.plan_error_json(S.hash_agg(S.scan([D, S.T_INT64]), [], [S.avg(S.col(0, D), S.decimal(14, 6), D, S.ANSI)], S.FINAL).encode(), -2) == \
        {"errorType": "DecimalSumOverflow", "errorClass": "ARITHMETIC_OVERFLOW", "params": {"functionName": "avg"}}
