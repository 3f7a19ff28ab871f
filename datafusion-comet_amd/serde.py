"""Plan-bytes builder: produces the protobuf bytes the JVM side would send for a plan.

This is the Python stand-in for Spark's ``QueryPlanSerde`` (spark/src/main/scala/org/apache/comet/serde/
QueryPlanSerde.scala, ``arithmetic.scala``, ``aggregates.scala``, ``literals.scala``) used by tests and
bench.py: it emits wire-format bytes for ``spark.spark_operator.Operator`` exactly as protoc-generated
code would (field numbers from native/proto/src/proto/{operator,expr,types,literal}.proto), without
needing protoc.  The same Python plan objects are interpreted by ``oracle/`` so that the product (which
only ever sees the bytes) and the oracle are driven from one description.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

# --------------------------------------------------------------------------- wire format


def _varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _tag(field_no: int, wt: int) -> bytes:
    return _varint((field_no << 3) | wt)


def _context_fields(e) -> bytes:
    """query_context = 90 (QueryContext, expr.proto:109-141) and expr_id = 91 of an Expr or an AggExpr (set by with_context)"""
    out = b""
    qc = getattr(e, "query_context", None)
    if qc is not None:
        c = (_f_bytes(1, qc["sql_text"].encode()) if qc.get("sql_text") else b"") + _f_varint(2, qc.get("start_index", 0)) + _f_varint(3, qc.get("stop_index", 0))
        if qc.get("object_type") is not None:
            c += _f_bytes(4, qc["object_type"].encode())
        if qc.get("object_name") is not None:
            c += _f_bytes(5, qc["object_name"].encode())
        c += _f_varint(6, qc.get("line", 0)) + _f_varint(7, qc.get("start_position", 0))
        if qc.get("sql_text_idx") is not None:
            c += _f_varint(8, qc["sql_text_idx"])
        out += _f_msg(90, c)
    if getattr(e, "expr_id", None) is not None:
        out += _f_varint(91, e.expr_id)
    return out


def _f_varint(field_no: int, v: int) -> bytes:
    return _tag(field_no, 0) + _varint(v)


def _f_bytes(field_no: int, b: bytes) -> bytes:
    return _tag(field_no, 2) + _varint(len(b)) + b


def _f_msg(field_no: int, b: bytes) -> bytes:
    return _f_bytes(field_no, b)


# --------------------------------------------------------------------------- types (types.proto:43-114)

BOOL, INT8, INT16, INT32, INT64, FLOAT, DOUBLE, STRING, BYTES, TIMESTAMP, DECIMAL, TIMESTAMP_NTZ, DATE, NULL = range(14)


LIST, MAP, STRUCT = 14, 15, 16


@dataclass(frozen=True)
class DataType:
    type_id: int
    precision: int = 0
    scale: int = 0
    fields: tuple = ()              # STRUCT: ((name, DataType, nullable), …) — types.proto StructInfo
    element: Optional["DataType"] = None   # LIST: ListInfo.element_type
    contains_null: bool = True      # LIST: ListInfo.contains_null

    def encode(self) -> bytes:
        out = b""
        if self.type_id != 0:
            out += _f_varint(1, self.type_id)
        if self.type_id == DECIMAL:
            dec = _f_varint(1, self.precision) + (_f_varint(2, self.scale) if self.scale else b"")
            out += _f_msg(2, _f_msg(2, dec))
        elif self.type_id == LIST:
            li = _f_msg(1, self.element.encode()) + (_f_varint(2, 1) if self.contains_null else b"")
            out += _f_msg(2, _f_msg(3, li))
        elif self.type_id == MAP:      # MapInfo{key_type = 1, value_type = 2, value_contains_null = 3}; fields = (("key", K, False), ("value", V, nullable))
            mi = _f_msg(1, self.fields[0][1].encode()) + _f_msg(2, self.fields[1][1].encode()) + (_f_varint(3, 1) if self.fields[1][2] else b"")
            out += _f_msg(2, _f_msg(4, mi))
        elif self.type_id == STRUCT:
            si = b"".join(_f_bytes(1, n.encode()) for n, _, _ in self.fields)
            si += b"".join(_f_msg(2, t.encode()) for _, t, _ in self.fields)
            si += _f_bytes(3, bytes(1 if nl else 0 for _, _, nl in self.fields))      # repeated bool: packed, as protobuf-java writes it
            out += _f_msg(2, _f_msg(5, si))
        return out

    def __repr__(self):
        names = ["bool", "int8", "int16", "int32", "int64", "float", "double", "string", "bytes", "timestamp",
                 "decimal", "timestamp_ntz", "date", "null"]
        if self.type_id == DECIMAL:
            return f"decimal({self.precision},{self.scale})"
        if self.type_id == LIST:
            return f"list<{self.element!r}>"
        if self.type_id == MAP:
            return f"map<{self.fields[0][1]!r}, {self.fields[1][1]!r}>"
        if self.type_id == STRUCT:
            return "struct<" + ", ".join(f"{n}: {t!r}" for n, t, _ in self.fields) + ">"
        return names[self.type_id]


def struct_type(fields) -> DataType:
    """fields: [(name, DataType, nullable)]"""
    return DataType(STRUCT, fields=tuple((n, t, bool(nl)) for n, t, nl in fields))


def map_type(key: DataType, value: DataType, value_contains_null: bool = True) -> DataType:
    return DataType(MAP, fields=(("key", key, False), ("value", value, bool(value_contains_null))))


def list_type(element: DataType, contains_null: bool = True) -> DataType:
    return DataType(LIST, element=element, contains_null=contains_null)


def decimal(p: int, s: int) -> DataType:
    return DataType(DECIMAL, p, s)


T_BOOL, T_INT8, T_INT16, T_INT32, T_INT64 = (DataType(t) for t in (BOOL, INT8, INT16, INT32, INT64))
T_FLOAT, T_DOUBLE, T_STRING, T_DATE, T_TIMESTAMP = (DataType(t) for t in (FLOAT, DOUBLE, STRING, DATE, TIMESTAMP))

LEGACY, TRY, ANSI = 0, 1, 2


def from_arrow_type(t) -> DataType:
    """pyarrow DataType → spark.spark_expression.DataType (inverse of to_arrow_datatype, execution/serde.rs:71-200)."""
    import pyarrow as pa
    if pa.types.is_decimal(t):
        return decimal(t.precision, t.scale)
    if pa.types.is_struct(t):
        return struct_type([(t.field(i).name, from_arrow_type(t.field(i).type), t.field(i).nullable) for i in range(t.num_fields)])
    if pa.types.is_list(t):
        return list_type(from_arrow_type(t.value_type), t.value_field.nullable)
    if pa.types.is_map(t):
        return map_type(from_arrow_type(t.key_type), from_arrow_type(t.item_type), t.item_field.nullable)
    if pa.types.is_timestamp(t):
        return T_TIMESTAMP if t.tz else DataType(TIMESTAMP_NTZ)
    m = {pa.bool_(): T_BOOL, pa.int8(): T_INT8, pa.int16(): T_INT16, pa.int32(): T_INT32, pa.int64(): T_INT64, pa.float32(): T_FLOAT,
         pa.float64(): T_DOUBLE, pa.utf8(): T_STRING, pa.binary(): DataType(BYTES), pa.date32(): T_DATE}
    if t not in m:
        raise ValueError(f"no Spark type for Arrow {t}")
    return m[t]


def arrow_type_id(t):
    """(DataTypeId, precision) as comet_murmur3_column expects them."""
    d = from_arrow_type(t)
    return d.type_id, d.precision

# --------------------------------------------------------------------------- expressions (expr.proto)


@dataclass
class Expr:
    kind: str                       # 'literal', 'bound', 'add', ...
    children: List["Expr"] = field(default_factory=list)
    dtype: Optional[DataType] = None  # literal/bound/cast/check_overflow datatype, math return_type
    value: object = None            # literal python value (None = NULL)
    index: int = -1                 # bound
    eval_mode: int = LEGACY
    fail_on_error: bool = False
    negated: bool = False
    check_divide_overflow: bool = False   # MathExpr field 6 (integral_divide only)

    # field numbers of Expr.expr_struct (expr.proto:30-107)
    TAGS = dict(subquery=50, list_extract=56, trunc_timestamp=47, unix_timestamp=65, hour=22, minute=23, second=24, literal=2, bound=3, add=4, subtract=5, multiply=6, divide=7, cast=8, eq=9, neq=10, gt=11, gt_eq=12,
                lt=13, lt_eq=14, is_null=15, is_not_null=16, and_=17, or_=18, check_overflow=25, like=26, rlike=30, scalar_func=31, eq_null_safe=32,
                neq_null_safe=33, bit_and=34, bit_or=35, bit_xor=36, shift_right=42, shift_left=43, integral_divide=59, remainder=37, case_when=38, in_=39, not_=40, unary_minus=41, if_=44, normalize_nan_and_zero=45,
                unbound=51, get_struct_field=54)

    def encode(self) -> bytes:
        k = self.kind
        tag = self.TAGS[k]
        if k == "literal":
            body = self._encode_literal()
        elif k == "bound":
            body = (_f_varint(1, self.index) if self.index else b"") + _f_msg(2, self.dtype.encode())
        elif k == "unbound":
            body = _f_bytes(1, b"col") + _f_msg(2, self.dtype.encode())
        elif k in ("add", "subtract", "multiply", "divide", "remainder", "integral_divide"):
            body = _f_msg(1, self.children[0].encode()) + _f_msg(2, self.children[1].encode())
            body += _f_msg(4, self.dtype.encode())
            if self.eval_mode:
                body += _f_varint(5, self.eval_mode)
            if getattr(self, "check_divide_overflow", False):
                body += _f_varint(6, 1)
        elif k in ("hour", "minute", "second", "unix_timestamp"):      # expr.proto:436-458: child = 1, timezone = 2
            body = _f_msg(1, self.children[0].encode()) + _f_bytes(2, (getattr(self, "timezone", None) or "UTC").encode())
        elif k == "subquery":                                          # expr.proto:513-516: id = 1, datatype = 2
            body = _f_varint(1, int(self.value)) + _f_msg(2, self.dtype.encode())
        elif k == "list_extract":                                      # expr.proto:533-539
            body = _f_msg(1, self.children[0].encode()) + _f_msg(2, self.children[1].encode())
            if len(self.children) > 2:
                body += _f_msg(3, self.children[2].encode())
            if getattr(self, "one_based", False):
                body += _f_varint(4, 1)
            if self.fail_on_error:
                body += _f_varint(5, 1)
        elif k == "trunc_timestamp":                                   # expr.proto:507-511: format = 1, child = 2, timezone = 3
            body = _f_msg(1, self.children[1].encode()) + _f_msg(2, self.children[0].encode()) + _f_bytes(3, (getattr(self, "timezone", None) or "UTC").encode())
        elif k == "cast":
            body = _f_msg(1, self.children[0].encode()) + _f_msg(2, self.dtype.encode()) + _f_bytes(3, (getattr(self, "timezone", None) or "UTC").encode())
            if self.eval_mode:
                body += _f_varint(4, self.eval_mode)
            if getattr(self, "is_spark4_plus", False):
                body += _f_varint(6, 1)
        elif k == "check_overflow":
            body = _f_msg(1, self.children[0].encode()) + _f_msg(2, self.dtype.encode())
            if self.fail_on_error:
                body += _f_varint(3, 1)
        elif k == "normalize_nan_and_zero":
            body = _f_msg(1, self.children[0].encode()) + _f_msg(2, self.dtype.encode())
        elif k == "unary_minus":
            body = _f_msg(1, self.children[0].encode()) + (_f_varint(2, 1) if self.fail_on_error else b"")
        elif k == "in_":
            body = _f_msg(1, self.children[0].encode())
            for c in self.children[1:]:
                body += _f_msg(2, c.encode())
            if self.negated:
                body += _f_varint(3, 1)
        elif k == "if_":
            body = b"".join(_f_msg(i + 1, c.encode()) for i, c in enumerate(self.children))
        elif k == "get_struct_field":      # expr.proto:528-531: child = 1, ordinal = 2
            body = _f_msg(1, self.children[0].encode()) + (_f_varint(2, self.index) if self.index else b"")
        elif k == "scalar_func":   # ScalarFunc{func=1, args=2, return_type=3, fail_on_error=4}; value = function name
            body = _f_bytes(1, self.value.encode()) + b"".join(_f_msg(2, c.encode()) for c in self.children)
            if self.dtype is not None:
                body += _f_msg(3, self.dtype.encode())
            if self.fail_on_error:
                body += _f_varint(4, 1)
        elif k == "case_when":   # children = when* then* [else]; index = number of WHEN branches
            n = self.index
            body = b"".join(_f_msg(2, c.encode()) for c in self.children[:n]) + b"".join(_f_msg(3, c.encode()) for c in self.children[n:2 * n])
            if len(self.children) == 2 * n + 1:
                body += _f_msg(4, self.children[2 * n].encode())
        else:  # BinaryExpr / UnaryExpr
            body = b"".join(_f_msg(i + 1, c.encode()) for i, c in enumerate(self.children))
        return _f_msg(tag, body) + _context_fields(self)

    def _encode_literal(self) -> bytes:
        t = self.dtype
        out = b""
        v = self.value
        if v is not None:
            tid = t.type_id
            if tid == BOOL:
                out += _f_varint(1, 1 if v else 0)
            elif tid == INT8:
                out += _f_varint(2, int(v))
            elif tid == INT16:
                out += _f_varint(3, int(v))
            elif tid in (INT32, DATE):
                out += _f_varint(4, int(v))
            elif tid in (INT64, TIMESTAMP, TIMESTAMP_NTZ):
                out += _f_varint(5, int(v))
            elif tid == FLOAT:
                out += _tag(6, 5) + struct.pack("<f", float(v))
            elif tid == DOUBLE:
                out += _tag(7, 1) + struct.pack("<d", float(v))
            elif tid == STRING:
                out += _f_bytes(8, v.encode() if isinstance(v, str) else bytes(v))
            elif tid == BYTES:
                out += _f_bytes(9, bytes(v))
            elif tid == DECIMAL:
                # BigInteger.toByteArray: minimal big-endian two's complement (literals.scala:92-95)
                iv = int(v)
                n = max(1, (iv.bit_length() + 8) // 8)
                out += _f_bytes(10, iv.to_bytes(n, "big", signed=True))
            else:
                raise ValueError(f"literal of {t}")
        out += _f_msg(12, t.encode())
        if v is None:
            out += _f_varint(13, 1)
        return out


def lit(value, dtype: DataType) -> Expr:
    """Decimal literals carry the UNSCALED integer (literal.proto:38); a decimal.Decimal is converted with the type's scale."""
    import decimal as _d
    if dtype.type_id == DECIMAL and isinstance(value, _d.Decimal):
        value = int(value.scaleb(dtype.scale).to_integral_exact())
    return Expr("literal", dtype=dtype, value=value)


def col(index: int, dtype: DataType) -> Expr:
    return Expr("bound", dtype=dtype, index=index)


def get_struct_field(child: Expr, ordinal: int) -> Expr:
    return Expr("get_struct_field", [child], index=ordinal)


def _bin(kind):
    def f(a: Expr, b: Expr) -> Expr:
        return Expr(kind, [a, b])
    return f


eq, neq, gt, gt_eq, lt, lt_eq = (_bin(k) for k in ("eq", "neq", "gt", "gt_eq", "lt", "lt_eq"))
and_, or_ = _bin("and_"), _bin("or_")
eq_null_safe = _bin("eq_null_safe")
bit_and, bit_or, bit_xor, shift_left, shift_right = (_bin(k) for k in ("bit_and", "bit_or", "bit_xor", "shift_left", "shift_right"))
like = _bin("like")            # Expr.like = 26: BinaryExpr(string, pattern)
rlike = _bin("rlike")          # Expr.rlike = 30: BinaryExpr(string, pattern)


def not_(a: Expr) -> Expr:
    return Expr("not_", [a])


def is_null(a: Expr) -> Expr:
    return Expr("is_null", [a])


def is_not_null(a: Expr) -> Expr:
    return Expr("is_not_null", [a])


def math(kind: str, a: Expr, b: Expr, return_type: DataType, eval_mode: int = LEGACY) -> Expr:
    return Expr(kind, [a, b], dtype=return_type, eval_mode=eval_mode)


def integral_divide(a: Expr, a_type: DataType, b: Expr, b_type: DataType, eval_mode: int = LEGACY, check_divide_overflow: bool = False) -> Expr:
    """Spark's `a div b` exactly as CometIntegralDivide serialises it (serde/arithmetic.scala:283-345): integral operands are cast to
    Decimal(19,0), a zero divisor becomes NULL outside ANSI mode (nullIfWhenPrimitive), decimal_integral_div produces Decimal(intDig, 0),
    CheckOverflow bounds it and a LEGACY cast brings it to Long."""
    def as_dec(x, t):
        return (x, t) if t.type_id == DECIMAL else (cast(x, decimal(19, 0)), decimal(19, 0))
    (l, lt), (r, rt) = as_dec(a, a_type), as_dec(b, b_type)
    if eval_mode != ANSI:
        r = if_(eq(r, lit(0, rt)), lit(None, rt), r)
    int_dig = lt.precision - lt.scale + rt.scale
    out_t = decimal(min(int_dig if int_dig else 1, 38), 0)
    d = Expr("integral_divide", [l, r], dtype=out_t, eval_mode=eval_mode, check_divide_overflow=check_divide_overflow)
    return cast(check_overflow(d, out_t, fail_on_error=eval_mode == ANSI), T_INT64)


def check_overflow(child: Expr, dtype: DataType, fail_on_error: bool = False) -> Expr:
    return Expr("check_overflow", [child], dtype=dtype, fail_on_error=fail_on_error)


def cast(child: Expr, dtype: DataType, eval_mode: int = LEGACY, timezone: str = "UTC", is_spark4_plus: bool = False) -> Expr:
    e = Expr("cast", [child], dtype=dtype, eval_mode=eval_mode)
    e.timezone = timezone
    e.is_spark4_plus = is_spark4_plus
    return e


def with_context(e, expr_id: int, **qc):
    """attach Spark's SQLQueryContext to an expression or an aggregate (sql_text | sql_text_idx, start_index, stop_index, line, start_position, object_type,
    object_name) under its expr_id: the errors it raises carry it"""
    e.query_context = qc
    e.expr_id = expr_id
    return e


def time_part(kind: str, child: Expr, timezone: str = "UTC") -> Expr:
    """Hour / Minute / Second of a timestamp in the session time zone (expr.proto:436-453)."""
    assert kind in ("hour", "minute", "second")
    e = Expr(kind, [child])
    e.timezone = timezone
    return e


def subquery(sub_id: int, dtype: DataType) -> Expr:
    """a scalar subquery's result (ScalarSubquery → Subquery{id, datatype}, expr.proto:513-516): the native side asks for the value at the first executePlan"""
    return Expr("subquery", [], dtype=dtype, value=sub_id)


def list_extract(child: Expr, ordinal: Expr, one_based: bool = False, fail_on_error: bool = False, default: Optional[Expr] = None) -> Expr:
    """GetArrayItem (arr[i], from 0) / ElementAt on an array (element_at(arr, i), from 1, negative from the end) — expr.proto:533-539"""
    e = Expr("list_extract", [child, ordinal] + ([default] if default is not None else []), fail_on_error=fail_on_error)
    e.one_based = one_based
    return e


def trunc_timestamp(child: Expr, fmt: str, timezone: str = "UTC") -> Expr:
    """TruncTimestamp (date_trunc(fmt, ts); expr.proto:507-511): children = [timestamp, format literal]"""
    e = Expr("trunc_timestamp", [child, lit(fmt, T_STRING)])
    e.timezone = timezone
    return e


def unix_timestamp(child: Expr, timezone: str = "UTC") -> Expr:
    """UnixTimestamp of a timestamp / timestamp_ntz / date (expr.proto:455-458) → bigint seconds"""
    e = Expr("unix_timestamp", [child])
    e.timezone = timezone
    return e


def if_(c: Expr, t: Expr, f: Expr) -> Expr:
    return Expr("if_", [c, t, f])


def scalar_func(name: str, args: Sequence[Expr], return_type: Optional[DataType] = None, fail_on_error: bool = False) -> Expr:
    """ScalarFunc (expr.proto:466-471), e.g. scalar_func("ceil", [x], T_INT64)."""
    return Expr("scalar_func", list(args), dtype=return_type, value=name, fail_on_error=fail_on_error)


def date_part(field_name: str, date_expr: Expr) -> Expr:
    """What CometGetDateField emits for year()/month()/…: Cast(datepart(<field>, date) AS int) (serde/datetime.scala:36-80)."""
    return cast(scalar_func("datepart", [lit(field_name, T_STRING), date_expr]), T_INT32)


def case_when(branches: Sequence, else_: Optional[Expr] = None) -> Expr:
    """CASE WHEN c1 THEN v1 ... [ELSE e] END; branches = [(when, then), ...] (expr.proto:473-483)."""
    whens, thens = [b[0] for b in branches], [b[1] for b in branches]
    return Expr("case_when", whens + thens + ([else_] if else_ is not None else []), index=len(branches))


def in_(value: Expr, items: Sequence[Expr], negated: bool = False) -> Expr:
    return Expr("in_", [value, *items], negated=negated)


# --------------------------------------------------------------------------- aggregates (expr.proto:143-260)


@dataclass
class AggExpr:
    kind: str                       # count | sum | min | max | avg | first | last
    children: List[Expr]
    dtype: Optional[DataType] = None      # result type
    sum_dtype: Optional[DataType] = None  # avg
    eval_mode: int = LEGACY
    filter: Optional[Expr] = None
    ignore_nulls: bool = False      # first | last

    TAGS = dict(count=2, sum=3, min=4, max=5, avg=6, first=7, last=8)

    def encode(self) -> bytes:
        if self.kind == "count":
            body = b"".join(_f_msg(1, c.encode()) for c in self.children)
        elif self.kind == "sum":
            body = _f_msg(1, self.children[0].encode()) + _f_msg(2, self.dtype.encode())
            if self.eval_mode:
                body += _f_varint(3, self.eval_mode)
        elif self.kind in ("min", "max"):
            body = _f_msg(1, self.children[0].encode()) + _f_msg(2, self.dtype.encode())
        elif self.kind in ("first", "last"):      # First / Last{child=1, datatype=2, ignore_nulls=3} (expr.proto:210-220)
            body = _f_msg(1, self.children[0].encode()) + _f_msg(2, self.dtype.encode()) + (_f_varint(3, 1) if self.ignore_nulls else b"")
        elif self.kind == "avg":
            body = _f_msg(1, self.children[0].encode()) + _f_msg(2, self.dtype.encode()) + _f_msg(3, self.sum_dtype.encode())
            if self.eval_mode:
                body += _f_varint(4, self.eval_mode)
        else:
            raise ValueError(self.kind)
        out = _f_msg(self.TAGS[self.kind], body)
        if self.filter is not None:
            out += _f_msg(89, self.filter.encode())
        return out + _context_fields(self)


def count(*children: Expr) -> AggExpr:
    return AggExpr("count", list(children))


def sum_(child: Expr, dtype: DataType, eval_mode: int = LEGACY, filter: Optional[Expr] = None) -> AggExpr:
    return AggExpr("sum", [child], dtype=dtype, eval_mode=eval_mode, filter=filter)


def avg(child: Expr, dtype: DataType, sum_dtype: DataType, eval_mode: int = LEGACY) -> AggExpr:
    return AggExpr("avg", [child], dtype=dtype, sum_dtype=sum_dtype, eval_mode=eval_mode)


def min_(child: Expr, dtype: DataType) -> AggExpr:
    return AggExpr("min", [child], dtype=dtype)


def max_(child: Expr, dtype: DataType) -> AggExpr:
    return AggExpr("max", [child], dtype=dtype)


def first_(child: Expr, dtype: DataType, ignore_nulls: bool = False) -> AggExpr:
    return AggExpr("first", [child], dtype=dtype, ignore_nulls=ignore_nulls)


def last_(child: Expr, dtype: DataType, ignore_nulls: bool = False) -> AggExpr:
    return AggExpr("last", [child], dtype=dtype, ignore_nulls=ignore_nulls)


# --------------------------------------------------------------------------- operators (operator.proto)

PARTIAL, FINAL, PARTIAL_MERGE = 0, 1, 2


@dataclass
class Operator:
    kind: str                                   # scan | projection | filter | hash_agg | raw
    children: List["Operator"] = field(default_factory=list)
    fields: List[DataType] = field(default_factory=list)      # scan
    exprs: List[Expr] = field(default_factory=list)           # projection list / grouping exprs
    predicate: Optional[Expr] = None
    aggs: List[AggExpr] = field(default_factory=list)
    mode: int = PARTIAL
    expr_modes: List[int] = field(default_factory=list)      # hash_agg: per-aggregate modes (HashAggregate.expr_modes = 6)
    initial_input_buffer_offset: int = 0                     # hash_agg: first state column of the PartialMerge aggregates (= 7)
    plan_id: int = 0
    raw_tag: int = 0                            # 'raw': arbitrary op_struct tag with empty body (negative tests)
    left_keys: List[Expr] = field(default_factory=list)       # hash_join
    right_keys: List[Expr] = field(default_factory=list)
    join_type: int = 0
    build_side: int = 0
    condition: Optional[Expr] = None

    # native_scan
    field_names: List[str] = field(default_factory=list)
    case_sensitive: bool = True
    data_filters: List[Expr] = field(default_factory=list)    # pushed-down predicates (row-group pruning only)
    partition_fields: List[tuple] = field(default_factory=list)   # Hive partition columns: (name, DataType)
    partition_values: List[tuple] = field(default_factory=list)   # per file: one python value per partition column (None = NULL)
    field_ids: List[Optional[int]] = field(default_factory=list)  # per required column: Parquet field id (metadata "PARQUET:field_id") or None
    use_field_id: bool = False
    ignore_missing_field_id: bool = False
    allow_type_promotion: bool = True             # Spark 4.x behaviour (ShimCometConf); False = Spark 3.x: widening reads are rejected
    allow_timestamp_ltz_to_ntz: bool = True
    default_values: dict = field(default_factory=dict)            # required-schema position → python value of the column's default
    # sort / limit
    sort_orders: List[tuple] = field(default_factory=list)   # (expr, descending, nulls_last)
    fetch: Optional[int] = None
    skip: int = 0
    limit: int = -1
    offset: int = 0
    files: List[tuple] = field(default_factory=list)          # (path, start, length, file_size)
    # shuffle_writer
    partitioning: str = "single"                # hash | single | round_robin | range
    num_partitions: int = 1
    max_hash_columns: int = 0
    data_file: str = ""
    index_file: str = ""
    codec: int = 0                              # CompressionCodec: 0 None, 1 Zstd, 2 Lz4, 3 Snappy
    compression_level: int = 1
    bounds: List[list] = field(default_factory=list)          # range partitioning: boundary rows (lists of literal Exprs), ascending
    outer: bool = False                         # explode: explode_outer
    position: bool = False                      # explode: posexplode
    projections: List[list] = field(default_factory=list)     # expand: one list of Exprs per projection
    window_fns: List[tuple] = field(default_factory=list)     # window: (function name, argument Exprs, result DataType)
    partition_by: List[Expr] = field(default_factory=list)    # window

    TAGS = dict(shuffle_writer=106, shuffle_scan=116, expand=107, explode=114, window=110, bnlj=117, scan=100, projection=101, filter=102, sort=103, hash_agg=104, limit=105, sort_merge_join=108, hash_join=109, native_scan=111)

    def encode(self) -> bytes:
        out = b"".join(_f_msg(1, c.encode()) for c in self.children)
        if self.plan_id:
            out += _f_varint(2, self.plan_id)
        for text in getattr(self, "sql_text_pool", None) or []:      # Operator.sql_text_pool = 3 (root only, operator.proto:39-47)
            out += _f_bytes(3, text.encode())
        if self.kind == "scan":
            body = b"".join(_f_msg(1, f.encode()) for f in self.fields) + _f_bytes(2, b"test_scan")
        elif self.kind == "window":
            # Window{window_expr=1 (WindowExpr{built_in_window_function=1 (Expr.scalarFunc), spec=3, result_type=5}), order_by_list=2, partition_by_list=3}
            # (operator.proto:793-862); the frame in spec is what Spark sends for ranking functions: ROWS UNBOUNDED PRECEDING .. CURRENT ROW
            def so_expr(e, desc, nulls_last):
                return _f_msg(19, _f_msg(1, e.encode()) + (_f_varint(2, 1) if desc else b"") + (_f_varint(3, 1) if nulls_last else b""))
            frame = _f_msg(2, _f_msg(1, b"")) + _f_msg(3, _f_msg(3, b""))
            spec = b"".join(_f_msg(1, e.encode()) for e in self.partition_by) + b"".join(_f_msg(2, so_expr(*o)) for o in self.sort_orders) + _f_msg(3, frame)
            body = b""
            for wf in self.window_fns:
                if wf[0] == "agg":
                    # aggregate over a frame: ("agg", AggExpr, result type, (rows|range, unbounded|current, unbounded|current))
                    # a bound is "unbounded", "current" or an int: rows relative to the current row, negative = PRECEDING, positive = FOLLOWING —
                    # carried by Preceding.offset / Following.offset whichever side it is on (operator.proto:815-845, planner.rs:3016-3030);
                    # RANGE frames: ("value", literal of the ORDER BY key's type) = that much PRECEDING (lower) / FOLLOWING (upper)
                    _, agg, rtype, (ftype, lo, up) = wf

                    def bound(b):
                        if isinstance(b, tuple):       # ("value", literal Expr): a RANGE frame's value offset (Preceding / Following.range_offset)
                            return _f_msg(2, _f_msg(2, b[1]._encode_literal()))
                        if b == "unbounded":
                            return _f_msg(1, b"")
                        if b == "current":
                            return _f_msg(3, b"")
                        return _f_msg(2, _f_varint(1, int(b) & 0xFFFFFFFFFFFFFFFF) if int(b) != 0 else b"")
                    fr = (_f_varint(1, 1) if ftype == "range" else b"") + _f_msg(2, bound(lo)) + _f_msg(3, bound(up))
                    aspec = b"".join(_f_msg(1, e.encode()) for e in self.partition_by) + b"".join(_f_msg(2, so_expr(*o)) for o in self.sort_orders) + _f_msg(3, fr)
                    # First / Last: the JVM side repeats their ignoreNulls in WindowExpr.ignore_nulls = 4 (CometWindowExec.scala:249-256)
                    body += _f_msg(1, _f_msg(2, agg.encode()) + _f_msg(3, aspec) + (_f_varint(4, 1) if agg.ignore_nulls else b"") + _f_msg(5, rtype.encode()))
                    continue
                name, args, rtype = wf[:3]
                fn = Expr("scalar_func", list(args), value=name)
                wspec = spec
                if len(wf) > 3:      # nth_value: (name, args, result type, frame, ignore_nulls)
                    ftype, lo, up = wf[3]

                    def bound2(b):
                        if isinstance(b, tuple):
                            return _f_msg(2, _f_msg(2, b[1]._encode_literal()))
                        if b == "unbounded":
                            return _f_msg(1, b"")
                        if b == "current":
                            return _f_msg(3, b"")
                        return _f_msg(2, _f_varint(1, int(b) & 0xFFFFFFFFFFFFFFFF) if int(b) != 0 else b"")
                    fr = (_f_varint(1, 1) if ftype == "range" else b"") + _f_msg(2, bound2(lo)) + _f_msg(3, bound2(up))
                    wspec = b"".join(_f_msg(1, e.encode()) for e in self.partition_by) + b"".join(_f_msg(2, so_expr(*o)) for o in self.sort_orders) + _f_msg(3, fr)
                ign = _f_varint(4, 1) if len(wf) > 4 and wf[4] else b""
                body += _f_msg(1, _f_msg(1, fn.encode()) + _f_msg(3, wspec) + ign + _f_msg(5, rtype.encode()))
            body += b"".join(_f_msg(2, so_expr(*o)) for o in self.sort_orders)
            body += b"".join(_f_msg(3, e.encode()) for e in self.partition_by)
        elif self.kind == "expand":
            # Expand{project_list=1 (all projections back to back), num_expr_per_project=3} (operator.proto:738-741)
            body = b"".join(_f_msg(1, e.encode()) for proj in self.projections for e in proj) + _f_varint(3, len(self.projections[0]))
        elif self.kind == "explode":
            # Explode{child=1, outer=2, project_list=3, position=4} (operator.proto:743-752)
            body = _f_msg(1, self.exprs[0].encode()) + (_f_varint(2, 1) if self.outer else b"") + b"".join(_f_msg(3, e.encode()) for e in self.exprs[1:])
            body += _f_varint(4, 1) if self.position else b""
        elif self.kind == "shuffle_scan":
            # ShuffleScan{fields=1, source=2} (operator.proto:134-138)
            body = b"".join(_f_msg(1, f.encode()) for f in self.fields) + _f_bytes(2, b"CometShuffleExchangeExec [id=test]")
        elif self.kind == "shuffle_writer":
            # ShuffleWriter{partitioning=1, output_data_file=3, output_index_file=4, codec=5, compression_level=6} (operator.proto:688-707);
            # Partitioning{hash_partition=1{hash_expression=1,num_partitions=2}, single_partition=2, range_partition=3{num_partitions=2},
            # round_robin_partition=4{num_partitions=1,max_hash_columns=2}} (partitioning.proto:29-66)
            if self.partitioning == "hash":
                part = _f_msg(1, b"".join(_f_msg(1, e.encode()) for e in self.exprs) + _f_varint(2, self.num_partitions))
            elif self.partitioning == "single":
                part = _f_msg(2, b"")
            elif self.partitioning == "range":
                # RangePartition{sort_orders=1 (Expr{sort_order=19}), num_partitions=2, boundary_rows=4 (BoundaryRow{partition_bounds=1})}
                rp = b""
                for e, desc, nulls_last in self.sort_orders:
                    so = _f_msg(1, e.encode()) + (_f_varint(2, 1) if desc else b"") + (_f_varint(3, 1) if nulls_last else b"")
                    rp += _f_msg(1, _f_msg(19, so))
                rp += _f_varint(2, self.num_partitions)
                for row in self.bounds:
                    rp += _f_msg(4, b"".join(_f_msg(1, v.encode()) for v in row))
                part = _f_msg(3, rp)
            else:
                part = _f_msg(4, _f_varint(1, self.num_partitions) + (_f_varint(2, self.max_hash_columns) if self.max_hash_columns else b""))
            body = _f_msg(1, part) + _f_bytes(3, self.data_file.encode()) + _f_bytes(4, self.index_file.encode())
            if self.codec:
                body += _f_varint(5, self.codec)
            body += _f_varint(6, self.compression_level)
        elif self.kind == "projection":
            body = b"".join(_f_msg(1, e.encode()) for e in self.exprs)
        elif self.kind == "filter":
            body = _f_msg(1, self.predicate.encode())
        elif self.kind == "hash_agg":
            body = b"".join(_f_msg(1, e.encode()) for e in self.exprs)
            body += b"".join(_f_msg(2, a.encode()) for a in self.aggs)
            if self.mode:
                body += _f_varint(5, self.mode)
            body += b"".join(_f_varint(6, m) for m in self.expr_modes)
            if self.initial_input_buffer_offset:
                body += _f_varint(7, self.initial_input_buffer_offset)
        elif self.kind == "sort":
            # Sort{sort_orders=1 (Expr{sort_order=19 SortOrder{child=1,direction=2,null_ordering=3}}), fetch=3, skip=4} (operator.proto:641-645)
            body = b""
            for e, desc, nulls_last in self.sort_orders:
                so = _f_msg(1, e.encode()) + (_f_varint(2, 1) if desc else b"") + (_f_varint(3, 1) if nulls_last else b"")
                body += _f_msg(1, _f_msg(19, so))
            if self.fetch is not None:
                body += _f_varint(3, self.fetch)
            if self.skip:
                body += _f_varint(4, self.skip)
        elif self.kind == "limit":
            body = (_f_varint(1, self.limit) if self.limit else b"") + (_f_varint(2, self.offset) if self.offset else b"")
        elif self.kind == "native_scan":
            # NativeScan{common=1 NativeScanCommon{required_schema=1,data_schema=2,projection_vector=5,session_timezone=6,
            # case_sensitive=9,source=12,fields=13}, file_partition=2 SparkFilePartition{partitioned_file=1}} (operator.proto:103-190)
            def sf(n, t, fid=None):
                out = _f_bytes(1, n.encode()) + _f_msg(2, t.encode()) + _f_varint(3, 1)
                if fid is not None:   # map<string,string> metadata = 4 (CometParquetUtils.PARQUET_FIELD_ID_META_KEY)
                    out += _f_msg(4, _f_bytes(1, b"PARQUET:field_id") + _f_bytes(2, str(fid).encode()))
                return out
            ids = list(self.field_ids) + [None] * (len(self.fields) - len(self.field_ids))
            common = b"".join(_f_msg(1, sf(n, t, i)) for n, t, i in zip(self.field_names, self.fields, ids))
            common += b"".join(_f_msg(2, sf(n, t, i)) for n, t, i in zip(self.field_names, self.fields, ids))
            common += b"".join(_f_msg(3, sf(n, t)) for n, t in self.partition_fields)
            common += b"".join(_f_msg(4, e.encode()) for e in self.data_filters)
            common += b"".join(_f_varint(5, i) for i in range(len(self.fields)))
            common += _f_bytes(6, b"UTC")
            dv = sorted(self.default_values.items())
            common += b"".join(_f_msg(7, lit(v, self.fields[i]).encode()) for i, v in dv) + b"".join(_f_varint(8, i) for i, _ in dv)
            common += (_f_varint(9, 1) if self.case_sensitive else b"") + _f_bytes(12, b"parquet") + b"".join(_f_msg(13, t.encode()) for t in self.fields)
            common += (_f_varint(15, 1) if self.use_field_id else b"") + (_f_varint(16, 1) if self.ignore_missing_field_id else b"")
            common += (_f_varint(17, 1) if self.allow_type_promotion else b"") + (_f_varint(18, 1) if self.allow_timestamp_ltz_to_ntz else b"")
            part = b""
            for fi, (path, start, length, size) in enumerate(self.files):
                pf = _f_bytes(1, ("file://" + path).encode())
                if start:
                    pf += _f_varint(2, start)
                pf += _f_varint(3, length) + _f_varint(4, size)
                if self.partition_values:
                    pf += b"".join(_f_msg(5, lit(v, t).encode()) for v, (_, t) in zip(self.partition_values[fi], self.partition_fields))
                part += _f_msg(1, pf)
            body = _f_msg(1, common) + _f_msg(2, part)
        elif self.kind == "sort_merge_join":
            # SortMergeJoin{left_join_keys=1,right_join_keys=2,join_type=3,sort_options=4,condition=5} (operator.proto:765-771)
            body = b"".join(_f_msg(1, e.encode()) for e in self.left_keys) + b"".join(_f_msg(2, e.encode()) for e in self.right_keys)
            if self.join_type:
                body += _f_varint(3, self.join_type)
            for e, desc, nulls_last in self.sort_orders:
                so = _f_msg(1, e.encode()) + (_f_varint(2, 1) if desc else b"") + (_f_varint(3, 1) if nulls_last else b"")
                body += _f_msg(4, _f_msg(19, so))
            if self.condition is not None:
                body += _f_msg(5, self.condition.encode())
        elif self.kind == "bnlj":
            # BroadcastNestedLoopJoin{join_type=1, build_side=2, condition=3} (operator.proto:773-777)
            body = (_f_varint(1, self.join_type) if self.join_type else b"") + (_f_varint(2, self.build_side) if self.build_side else b"")
            if self.condition is not None:
                body += _f_msg(3, self.condition.encode())
        elif self.kind == "hash_join":
            # HashJoin{left_join_keys=1,right_join_keys=2,join_type=3,condition=4,build_side=5} (operator.proto:754-763)
            body = b"".join(_f_msg(1, e.encode()) for e in self.left_keys) + b"".join(_f_msg(2, e.encode()) for e in self.right_keys)
            if self.join_type:
                body += _f_varint(3, self.join_type)
            if self.condition is not None:
                body += _f_msg(4, self.condition.encode())
            if self.build_side:
                body += _f_varint(5, self.build_side)
        elif self.kind == "raw":
            return out + _f_msg(self.raw_tag, b"")
        else:
            raise ValueError(self.kind)
        return out + _f_msg(self.TAGS[self.kind], body)


def scan(fields: Sequence[DataType]) -> Operator:
    return Operator("scan", fields=list(fields))


def expand(child: Operator, projections: Sequence[Sequence[Expr]]) -> Operator:
    """One output row per input row and projection (grouping sets / rollup / cube)."""
    return Operator("expand", [child], projections=[list(p) for p in projections])


def explode(child: Operator, array: Expr, project_list: Sequence[Expr] = (), outer: bool = False, position: bool = False) -> Operator:
    """explode / posexplode [_outer]: output = project_list ++ [pos] ++ [element], one row per element of `array`."""
    return Operator("explode", [child], exprs=[array] + list(project_list), outer=outer, position=position)


def window(child: Operator, partition_by: Sequence[Expr], order_by: Sequence, fns: Sequence[tuple]) -> Operator:
    """fns: (name, [argument Exprs], result DataType) with name in row_number / rank / dense_rank / percent_rank / cume_dist / ntile / lag / lead;
    order_by as for sort().  The child must deliver its rows sorted by (partition_by, order_by) — Spark plans that Sort."""
    so = [(o[0], bool(o[1]), bool(o[2]) if len(o) > 2 else bool(o[1])) for o in order_by]
    return Operator("window", [child], partition_by=list(partition_by), sort_orders=so, window_fns=[tuple(f) for f in fns])


def shuffle_scan(fields: Sequence[DataType]) -> Operator:
    """Leaf fed by a stream of shuffle blocks (CometShuffleBlockIterator in the JVM, native.ShuffleBlockInput here)."""
    return Operator("shuffle_scan", fields=list(fields))


CODEC_NONE, CODEC_ZSTD, CODEC_LZ4, CODEC_SNAPPY = 0, 1, 2, 3


def shuffle_writer(child: Operator, data_file: str, index_file: str, partitioning: str = "single", hash_exprs: Sequence[Expr] = (),
                   num_partitions: int = 1, codec: int = CODEC_NONE, compression_level: int = 1, max_hash_columns: int = 0,
                   sort_orders: Sequence = (), bounds: Sequence = ()) -> Operator:
    """partitioning "range": sort_orders as for sort() — (expr, descending[, nulls_last]) — and `bounds`, the boundary rows (one
    literal per sort order, ascending under that order); a row goes to partition #(bounds ≤ row)."""
    so = [(o[0], bool(o[1]), bool(o[2]) if len(o) > 2 else bool(o[1])) for o in sort_orders]
    return Operator("shuffle_writer", [child], exprs=list(hash_exprs), partitioning=partitioning, num_partitions=num_partitions,
                    data_file=data_file, index_file=index_file, codec=codec, compression_level=compression_level,
                    max_hash_columns=max_hash_columns, sort_orders=so, bounds=[list(r) for r in bounds])


def filter_(child: Operator, predicate: Expr) -> Operator:
    return Operator("filter", [child], predicate=predicate)


def project(child: Operator, exprs: Sequence[Expr]) -> Operator:
    return Operator("projection", [child], exprs=list(exprs))


def hash_agg(child: Operator, grouping: Sequence[Expr], aggs: Sequence[AggExpr], mode: int = PARTIAL, expr_modes: Sequence[int] = (),
             initial_input_buffer_offset: int = 0) -> Operator:
    """expr_modes: per-aggregate PARTIAL / PARTIAL_MERGE for Spark's mixed-mode aggregate of the count(DISTINCT) rewrite; the
    PARTIAL_MERGE aggregates read their state columns from the child starting at initial_input_buffer_offset."""
    return Operator("hash_agg", [child], exprs=list(grouping), aggs=list(aggs), mode=mode, expr_modes=list(expr_modes),
                    initial_input_buffer_offset=initial_input_buffer_offset)


def sort(child: Operator, orders: Sequence, fetch: Optional[int] = None, skip: int = 0) -> Operator:
    """orders: (expr, descending) or (expr, descending, nulls_last); Spark's defaults are ASC NULLS FIRST / DESC NULLS LAST."""
    so = [(o[0], bool(o[1]), bool(o[2]) if len(o) > 2 else bool(o[1])) for o in orders]
    return Operator("sort", [child], sort_orders=so, fetch=fetch, skip=skip)


def limit(child: Operator, n: int, offset: int = 0) -> Operator:
    return Operator("limit", [child], limit=n, offset=offset)


def final_of(partial_plan: "Operator", state_schema) -> "Operator":
    """HashAggregate(Final) over a Scan of a Partial plan's output (group columns, then each aggregate's state columns) —
    the stage Spark plans after the exchange (planner.rs:1248-1384 with AggregateMode::Final)."""
    fields = [from_arrow_type(f.type) for f in state_schema]
    ng = len(partial_plan.exprs)
    return hash_agg(scan(fields), [col(i, fields[i]) for i in range(ng)], partial_plan.aggs, FINAL)


def native_scan(files: Sequence, names: Sequence[str], types: Sequence[DataType], case_sensitive: bool = True,
                data_filters: Sequence[Expr] = (), partition_fields: Sequence[tuple] = (), partition_values: Sequence[tuple] = (),
                field_ids: Sequence = (), use_field_id: bool = False, ignore_missing_field_id: bool = False, allow_type_promotion: bool = True,
                allow_timestamp_ltz_to_ntz: bool = True, default_values: Optional[dict] = None) -> Operator:
    """Parquet scan of `files` (paths, or (path, start, length, size) byte-range splits) producing columns `names`."""
    import os
    fl = []
    for f in files:
        if isinstance(f, str):
            sz = os.path.getsize(f)
            fl.append((f, 0, sz, sz))
        else:
            fl.append(tuple(f))
    return Operator("native_scan", fields=list(types), field_names=list(names), files=fl, case_sensitive=case_sensitive, data_filters=list(data_filters),
                    partition_fields=list(partition_fields), partition_values=list(partition_values), field_ids=list(field_ids),
                    use_field_id=use_field_id, ignore_missing_field_id=ignore_missing_field_id, allow_type_promotion=allow_type_promotion,
                    allow_timestamp_ltz_to_ntz=allow_timestamp_ltz_to_ntz, default_values=dict(default_values or {}))


INNER, LEFT_OUTER, RIGHT_OUTER, FULL_OUTER, LEFT_SEMI, LEFT_ANTI = range(6)
BUILD_LEFT, BUILD_RIGHT = 0, 1


def hash_join(left: Operator, right: Operator, left_keys: Sequence[Expr], right_keys: Sequence[Expr], join_type: int = INNER,
              build_side: int = BUILD_LEFT, condition: Optional[Expr] = None) -> Operator:
    """Keys are bound to each side's own schema; `condition` to the concatenated left ++ right schema."""
    return Operator("hash_join", [left, right], left_keys=list(left_keys), right_keys=list(right_keys), join_type=join_type,
                    build_side=build_side, condition=condition)


def nested_loop_join(left: Operator, right: Operator, join_type: int = INNER, build_side: int = BUILD_RIGHT, condition: Optional[Expr] = None) -> Operator:
    """BroadcastNestedLoopJoin: no equi-keys; `condition` (bound to left ++ right) decides, None = cross join."""
    return Operator("bnlj", [left, right], join_type=join_type, build_side=build_side, condition=condition)


def sort_merge_join(left: Operator, right: Operator, left_keys: Sequence[Expr], right_keys: Sequence[Expr], join_type: int = INNER,
                    condition: Optional[Expr] = None, descending: bool = False) -> Operator:
    """SortMergeJoin: same keys / join types as hash_join; sort_options carry the key ordering of the (sorted) inputs."""
    so = [(k, descending, descending) for k in left_keys]
    return Operator("sort_merge_join", [left, right], left_keys=list(left_keys), right_keys=list(right_keys), join_type=join_type,
                    condition=condition, sort_orders=so)


def config_map(entries: dict) -> bytes:
    """spark.spark_config.ConfigMap (config.proto:24)."""
    out = b""
    for k, v in entries.items():
        out += _f_msg(1, _f_bytes(1, k.encode()) + _f_bytes(2, str(v).encode()))
    return out


def decode_metric_node(b: bytes):
    """spark.spark_metric.NativeMetricNode → (dict, [children]) — what CometMetricNode.set_all_from_bytes parses."""
    pos = 0
    metrics, children = {}, []

    def rv():
        nonlocal pos
        v, shift = 0, 0
        while True:
            c = b[pos]
            pos += 1
            v |= (c & 0x7F) << shift
            if not c & 0x80:
                return v
            shift += 7

    while pos < len(b):
        t = rv()
        fno, wt = t >> 3, t & 7
        assert wt == 2
        n = rv()
        sub = b[pos:pos + n]
        pos += n
        if fno == 1:
            p2 = 0
            key, val = None, 0
            while p2 < len(sub):
                t2 = sub[p2]
                p2 += 1
                if t2 == 0x0A:
                    ln = sub[p2]
                    p2 += 1
                    key = sub[p2:p2 + ln].decode()
                    p2 += ln
                elif t2 == 0x10:
                    v, shift = 0, 0
                    while True:
                        c = sub[p2]
                        p2 += 1
                        v |= (c & 0x7F) << shift
                        if not c & 0x80:
                            break
                        shift += 7
                    val = v if v < (1 << 63) else v - (1 << 64)
            metrics[key] = val
        elif fno == 2:
            children.append(decode_metric_node(sub))
    return metrics, children
