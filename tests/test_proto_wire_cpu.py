"""Independent check of the plan codec.  serde.py (encoder) and csrc/proto.cpp (decoder) are both hand-written in this repo, so a
shared mis-reading of a field number would pass every other test.  The referee here is google.protobuf running on descriptors
built from the reference's own .proto files (tests/golden/make_proto_descriptors.py → tests/golden/comet_protos.desc):

  1. the committed descriptor set is what the .proto files under /root/reference say today (skipped where the reference is absent);
  2. every plan shape serde.py can emit parses under that schema with NO unknown field at any depth;
  3. a second encoder written against FIELD NAMES only (this file's to_pb: no field number appears in it) builds the same message —
     so a tag serde.py got wrong would have to be wrong in the .proto as well;
  4. protobuf's own serialization of those messages is accepted by proto.cpp and plans to the same pipeline as serde's bytes.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from google.protobuf import descriptor_pb2, descriptor_pool, message_factory, unknown_fields  # noqa: E402

from datafusion_comet_amd import serde as S, tpch, tpcds  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_PROTO_DIR = "/root/reference/native/proto/src/proto"


@pytest.fixture(scope="module")
def pool():
    fds = descriptor_pb2.FileDescriptorSet()
    with open(os.path.join(GOLDEN, "comet_protos.desc"), "rb") as f:
        fds.ParseFromString(f.read())
    p = descriptor_pool.DescriptorPool()
    for fd in fds.file:
        p.Add(fd)
    return p


def cls(pool, name):
    return message_factory.GetMessageClass(pool.FindMessageTypeByName(name))


def unknown_paths(msg, path=""):
    out = []
    if len(unknown_fields.UnknownFieldSet(msg)):
        out.append(path or "/")
    for fd, v in msg.ListFields():
        if fd.type != fd.TYPE_MESSAGE:
            continue
        if fd.is_repeated:
            vals = v.values() if hasattr(v, "values") else v
            for i, x in enumerate(vals):
                if hasattr(x, "ListFields"):
                    out += unknown_paths(x, f"{path}/{fd.name}[{i}]")
        else:
            out += unknown_paths(v, f"{path}/{fd.name}")
    return out


# --------------------------------------------------------------------------- the by-name encoder

TYPE_NAMES = {S.BOOL: "BOOL", S.INT8: "INT8", S.INT16: "INT16", S.INT32: "INT32", S.INT64: "INT64", S.FLOAT: "FLOAT", S.DOUBLE: "DOUBLE",
              S.STRING: "STRING", S.BYTES: "BYTES", S.TIMESTAMP: "TIMESTAMP", S.DECIMAL: "DECIMAL", S.TIMESTAMP_NTZ: "TIMESTAMP_NTZ", S.DATE: "DATE"}
EXPR_FIELD = {"and_": "and", "or_": "or", "in_": "in", "not_": "not", "if_": "if", "scalar_func": "scalarFunc", "eq_null_safe": "eqNullSafe",
              "neq_null_safe": "neqNullSafe", "bit_and": "bitwiseAnd", "bit_or": "bitwiseOr", "bit_xor": "bitwiseXor", "shift_right": "bitwiseShiftRight",
              "shift_left": "bitwiseShiftLeft", "case_when": "caseWhen"}
EVAL = {0: "LEGACY", 1: "TRY", 2: "ANSI"}


def set_dtype(pb, t):
    pb.type_id = pb.DESCRIPTOR.fields_by_name["type_id"].enum_type.values_by_name[TYPE_NAMES[t.type_id]].number
    if t.type_id == S.DECIMAL:
        pb.type_info.decimal.precision = t.precision
        pb.type_info.decimal.scale = t.scale


def set_expr(pb, e):
    k = e.kind
    sub = getattr(pb, EXPR_FIELD.get(k, k))
    if k == "literal":
        v, tid = e.value, e.dtype.type_id
        if v is not None:
            if tid == S.BOOL:
                sub.bool_val = bool(v)
            elif tid == S.INT8:
                sub.byte_val = int(v)
            elif tid == S.INT16:
                sub.short_val = int(v)
            elif tid in (S.INT32, S.DATE):
                sub.int_val = int(v)
            elif tid in (S.INT64, S.TIMESTAMP, S.TIMESTAMP_NTZ):
                sub.long_val = int(v)
            elif tid == S.FLOAT:
                sub.float_val = float(v)
            elif tid == S.DOUBLE:
                sub.double_val = float(v)
            elif tid == S.STRING:
                sub.string_val = v if isinstance(v, str) else bytes(v).decode()
            elif tid == S.BYTES:
                sub.bytes_val = bytes(v)
            elif tid == S.DECIMAL:
                iv = int(v)
                sub.decimal_val = iv.to_bytes(max(1, (iv.bit_length() + 8) // 8), "big", signed=True)
        set_dtype(sub.datatype, e.dtype)
        if v is None:
            sub.is_null = True
    elif k == "bound":
        sub.index = e.index
        set_dtype(sub.datatype, e.dtype)
    elif k == "unbound":
        sub.name = "col"
        set_dtype(sub.datatype, e.dtype)
    elif k in ("add", "subtract", "multiply", "divide", "remainder", "integral_divide"):
        set_expr(sub.left, e.children[0])
        set_expr(sub.right, e.children[1])
        set_dtype(sub.return_type, e.dtype)
        sub.eval_mode = sub.DESCRIPTOR.fields_by_name["eval_mode"].enum_type.values_by_name[EVAL[e.eval_mode]].number
        if e.check_divide_overflow:
            sub.check_divide_overflow = True
    elif k == "cast":
        set_expr(sub.child, e.children[0])
        set_dtype(sub.datatype, e.dtype)
        sub.timezone = "UTC"
        sub.eval_mode = sub.DESCRIPTOR.fields_by_name["eval_mode"].enum_type.values_by_name[EVAL[e.eval_mode]].number
    elif k == "check_overflow":
        set_expr(sub.child, e.children[0])
        set_dtype(sub.datatype, e.dtype)
        sub.fail_on_error = e.fail_on_error
    elif k == "normalize_nan_and_zero":
        set_expr(sub.child, e.children[0])
        set_dtype(sub.datatype, e.dtype)
    elif k == "unary_minus":
        set_expr(sub.child, e.children[0])
        sub.fail_on_error = e.fail_on_error
    elif k == "in_":
        set_expr(sub.in_value, e.children[0])
        for c in e.children[1:]:
            set_expr(sub.lists.add(), c)
        sub.negated = e.negated
    elif k == "if_":
        set_expr(sub.if_expr, e.children[0])
        set_expr(sub.true_expr, e.children[1])
        set_expr(sub.false_expr, e.children[2])
    elif k == "scalar_func":
        sub.func = e.value
        for c in e.children:
            set_expr(sub.args.add(), c)
        if e.dtype is not None:
            set_dtype(sub.return_type, e.dtype)
        sub.fail_on_error = e.fail_on_error
    elif k == "case_when":
        n = e.index
        for c in e.children[:n]:
            set_expr(sub.when.add(), c)
        for c in e.children[n:2 * n]:
            set_expr(sub.then.add(), c)
        if len(e.children) == 2 * n + 1:
            set_expr(sub.else_expr, e.children[2 * n])
    elif len(e.children) == 2:
        set_expr(sub.left, e.children[0])
        set_expr(sub.right, e.children[1])
    else:
        assert len(e.children) == 1, k
        set_expr(sub.child, e.children[0])
    sub.SetInParent()


def set_sort_order(pb_expr, e, desc, nulls_last):
    so = pb_expr.sort_order
    set_expr(so.child, e)
    so.direction = so.DESCRIPTOR.fields_by_name["direction"].enum_type.values_by_name["Descending" if desc else "Ascending"].number
    so.null_ordering = so.DESCRIPTOR.fields_by_name["null_ordering"].enum_type.values_by_name["NullsLast" if nulls_last else "NullsFirst"].number
    so.SetInParent()


def set_agg(pb, a):
    sub = getattr(pb, a.kind)
    if a.kind == "count":
        for c in a.children:
            set_expr(sub.children.add(), c)
    else:
        set_expr(sub.child, a.children[0])
        set_dtype(sub.datatype, a.dtype)
        if a.kind == "avg":
            set_dtype(sub.sum_datatype, a.sum_dtype)
        if a.kind in ("sum", "avg"):
            sub.eval_mode = sub.DESCRIPTOR.fields_by_name["eval_mode"].enum_type.values_by_name[EVAL[a.eval_mode]].number
        if a.kind in ("first", "last") and a.ignore_nulls:
            sub.ignore_nulls = True
    sub.SetInParent()
    if a.filter is not None:
        set_expr(pb.filter, a.filter)


def enum_no(msg, field_name, value_name):
    return msg.DESCRIPTOR.fields_by_name[field_name].enum_type.values_by_name[value_name].number


JOIN_TYPES = ["Inner", "LeftOuter", "RightOuter", "FullOuter", "LeftSemi", "LeftAnti"]


def set_struct_field(pb, name, t, fid=None):
    pb.name = name
    set_dtype(pb.data_type, t)
    pb.nullable = True
    if fid is not None:
        pb.metadata["PARQUET:field_id"] = str(fid)


def set_op(pb, op):
    for c in op.children:
        set_op(pb.children.add(), c)
    if op.plan_id:
        pb.plan_id = op.plan_id
    k = op.kind
    if k == "scan":
        for t in op.fields:
            set_dtype(pb.scan.fields.add(), t)
        pb.scan.source = "test_scan"
    elif k == "shuffle_scan":
        for t in op.fields:
            set_dtype(pb.shuffle_scan.fields.add(), t)
        pb.shuffle_scan.source = "CometShuffleExchangeExec [id=test]"
    elif k == "projection":
        for e in op.exprs:
            set_expr(pb.projection.project_list.add(), e)
        pb.projection.SetInParent()
    elif k == "filter":
        set_expr(pb.filter.predicate, op.predicate)
    elif k == "hash_agg":
        h = pb.hash_agg
        for e in op.exprs:
            set_expr(h.grouping_exprs.add(), e)
        for a in op.aggs:
            set_agg(h.agg_exprs.add(), a)
        h.mode = enum_no(h, "mode", ["Partial", "Final", "PartialMerge"][op.mode])
        for m in op.expr_modes:
            h.expr_modes.append(enum_no(h, "mode", ["Partial", "Final", "PartialMerge"][m]))
        h.initial_input_buffer_offset = op.initial_input_buffer_offset
        h.SetInParent()
    elif k == "sort":
        for e, desc, nl in op.sort_orders:
            set_sort_order(pb.sort.sort_orders.add(), e, desc, nl)
        if op.fetch is not None:
            pb.sort.fetch = op.fetch
        pb.sort.skip = op.skip
        pb.sort.SetInParent()
    elif k == "limit":
        pb.limit.limit = op.limit
        pb.limit.offset = op.offset
        pb.limit.SetInParent()
    elif k == "expand":
        for proj in op.projections:
            for e in proj:
                set_expr(pb.expand.project_list.add(), e)
        pb.expand.num_expr_per_project = len(op.projections[0])
    elif k == "hash_join":
        j = pb.hash_join
        for e in op.left_keys:
            set_expr(j.left_join_keys.add(), e)
        for e in op.right_keys:
            set_expr(j.right_join_keys.add(), e)
        j.join_type = enum_no(j, "join_type", JOIN_TYPES[op.join_type])
        if op.condition is not None:
            set_expr(j.condition, op.condition)
        j.build_side = enum_no(j, "build_side", ["BuildLeft", "BuildRight"][op.build_side])
        j.SetInParent()
    elif k == "sort_merge_join":
        j = pb.sort_merge_join
        for e in op.left_keys:
            set_expr(j.left_join_keys.add(), e)
        for e in op.right_keys:
            set_expr(j.right_join_keys.add(), e)
        j.join_type = enum_no(j, "join_type", JOIN_TYPES[op.join_type])
        for e, desc, nl in op.sort_orders:
            set_sort_order(j.sort_options.add(), e, desc, nl)
        if op.condition is not None:
            set_expr(j.condition, op.condition)
        j.SetInParent()
    elif k == "bnlj":
        j = pb.broadcast_nested_loop_join
        j.join_type = enum_no(j, "join_type", JOIN_TYPES[op.join_type])
        j.build_side = enum_no(j, "build_side", ["BuildLeft", "BuildRight"][op.build_side])
        if op.condition is not None:
            set_expr(j.condition, op.condition)
        j.SetInParent()
    elif k == "native_scan":
        c = pb.native_scan.common
        ids = list(op.field_ids) + [None] * (len(op.fields) - len(op.field_ids))
        for n, t, i in zip(op.field_names, op.fields, ids):
            set_struct_field(c.required_schema.add(), n, t, i)
        for n, t, i in zip(op.field_names, op.fields, ids):
            set_struct_field(c.data_schema.add(), n, t, i)
        for n, t in op.partition_fields:
            set_struct_field(c.partition_schema.add(), n, t)
        for e in op.data_filters:
            set_expr(c.data_filters.add(), e)
        c.projection_vector.extend(range(len(op.fields)))
        c.session_timezone = "UTC"
        for i, v in sorted(op.default_values.items()):
            set_expr(c.default_values.add(), S.lit(v, op.fields[i]))
            c.default_values_indexes.append(i)
        c.use_field_id, c.ignore_missing_field_id = op.use_field_id, op.ignore_missing_field_id
        c.allow_type_promotion, c.allow_timestamp_ltz_to_ntz = op.allow_type_promotion, op.allow_timestamp_ltz_to_ntz
        c.case_sensitive = op.case_sensitive
        c.source = "parquet"
        for t in op.fields:
            set_dtype(c.fields.add(), t)
        for fi, (path, start, length, size) in enumerate(op.files):
            pf = pb.native_scan.file_partition.partitioned_file.add()
            pf.file_path = "file://" + path
            pf.start, pf.length, pf.file_size = start, length, size
            if op.partition_values:
                for v, (_, t) in zip(op.partition_values[fi], op.partition_fields):
                    set_expr(pf.partition_values.add(), S.lit(v, t))
        pb.native_scan.file_partition.SetInParent()
    elif k == "shuffle_writer":
        w = pb.shuffle_writer
        p = w.partitioning
        if op.partitioning == "hash":
            for e in op.exprs:
                set_expr(p.hash_partition.hash_expression.add(), e)
            p.hash_partition.num_partitions = op.num_partitions
        elif op.partitioning == "single":
            p.single_partition.SetInParent()
        elif op.partitioning == "range":
            for e, desc, nl in op.sort_orders:
                set_sort_order(p.range_partition.sort_orders.add(), e, desc, nl)
            p.range_partition.num_partitions = op.num_partitions
            for row in op.bounds:
                br = p.range_partition.boundary_rows.add()
                for v in row:
                    set_expr(br.partition_bounds.add(), v)
        else:
            p.round_robin_partition.num_partitions = op.num_partitions
            p.round_robin_partition.max_hash_columns = op.max_hash_columns
        w.output_data_file = op.data_file
        w.output_index_file = op.index_file
        w.codec = enum_no(w, "codec", ["None", "Zstd", "Lz4", "Snappy"][op.codec])
        w.compression_level = op.compression_level
    elif k == "window":
        w = pb.window

        def spec(sp, frame):
            for e in op.partition_by:
                set_expr(sp.partitionSpec.add(), e)
            for e, desc, nl in op.sort_orders:
                set_sort_order(sp.orderSpec.add(), e, desc, nl)
            ftype, lo, up = frame
            f = sp.frameSpecification
            f.frame_type = enum_no(f, "frame_type", "Range" if ftype == "range" else "Rows")
            for side, b, unb, off in ((f.lower_bound, lo, "unboundedPreceding", "preceding"), (f.upper_bound, up, "unboundedFollowing", "following")):
                if isinstance(b, tuple):          # RANGE value offset: Preceding / Following.range_offset = the typed literal
                    tmp = w.partition_by_list.add().__class__()
                    set_expr(tmp, b[1])
                    getattr(side, off).range_offset.CopyFrom(tmp.literal)
                    del w.partition_by_list[-1]
                elif b == "unbounded":
                    getattr(side, unb).SetInParent()
                elif b == "current":
                    side.currentRow.SetInParent()
                else:                             # ROWS offset, negative = PRECEDING
                    getattr(side, off).offset = int(b)
                    getattr(side, off).SetInParent()
        for wf in op.window_fns:
            we = w.window_expr.add()
            if wf[0] == "agg":
                _, agg, rtype, frame = wf
                set_agg(we.agg_func, agg)
                spec(we.spec, frame)
                if getattr(agg, "ignore_nulls", False):
                    we.ignore_nulls = True
            else:
                name, args, rtype = wf[:3]
                set_expr(we.built_in_window_function, S.Expr("scalar_func", list(args), value=name))
                spec(we.spec, wf[3] if len(wf) > 3 else ("rows", "unbounded", "current"))
                if len(wf) > 4 and wf[4]:
                    we.ignore_nulls = True
            set_dtype(we.result_type, rtype)
        for e, desc, nl in op.sort_orders:
            set_sort_order(w.order_by_list.add(), e, desc, nl)
        for e in op.partition_by:
            set_expr(w.partition_by_list.add(), e)
        w.SetInParent()
    else:
        raise AssertionError(k)


# --------------------------------------------------------------------------- the corpus: every shape serde.py can emit

def corpus():
    I64, I32, DBL, STR, DATE, BOOL, DEC = S.T_INT64, S.T_INT32, S.T_DOUBLE, S.T_STRING, S.T_DATE, S.T_BOOL, S.decimal(12, 2)
    a, b, s, d, x = S.col(0, I64), S.col(1, DBL), S.col(2, STR), S.col(3, DATE), S.col(4, DEC)
    fields = [I64, DBL, STR, DATE, DEC]
    sc = lambda: S.scan(fields)
    plans = {f"tpch_{i}": p for i, p in enumerate(tpch.warm_plans())}
    plans["tpch_q3_single"] = tpch.q3_plan()
    plans["tpch_q6_final"] = tpch.q6_plan(S.FINAL)
    qa, qb, _ = tpcds.q95_plans()
    plans["tpcds_q95_a"], plans["tpcds_q95_b"] = qa, qb
    every_literal = [S.lit(True, BOOL), S.lit(-3, S.T_INT8), S.lit(300, S.T_INT16), S.lit(-70000, I32), S.lit(1 << 40, I64), S.lit(1.5, S.T_FLOAT),
                     S.lit(-2.25, DBL), S.lit("héllo", STR), S.lit(b"\x00\x01", S.DataType(S.BYTES)), S.lit(-12345678901234567890123, S.decimal(38, 6)),
                     S.lit(9000, DATE), S.lit(1_700_000_000_000_000, S.T_TIMESTAMP), S.lit(5, S.DataType(S.TIMESTAMP_NTZ)), S.lit(None, I64), S.lit(None, DEC)]
    plans["literals"] = S.project(sc(), every_literal)
    plans["expressions"] = S.project(S.filter_(sc(), S.and_(S.or_(S.not_(S.is_null(a)), S.is_not_null(b)), S.in_(a, [S.lit(1, I64), S.lit(2, I64)], negated=True))), [
        S.math("add", a, S.lit(1, I64), I64, S.ANSI), S.math("subtract", a, a, I64, S.TRY), S.math("divide", b, b, DBL), S.math("remainder", a, a, I64),
        S.cast(a, DBL, S.ANSI), S.check_overflow(S.math("multiply", x, x, S.decimal(25, 4)), S.decimal(25, 4), True), S.if_(S.gt(a, a), a, a),
        S.case_when([(S.lt(a, a), a), (S.lt_eq(a, a), a)], a), S.case_when([(S.gt_eq(a, a), a)]), S.scalar_func("abs", [a], I64),
        S.scalar_func("substring", [s, S.lit(1, I32), S.lit(3, I32)], STR, True), S.date_part("year", d),
        S.Expr("eq_null_safe", [a, a]), S.Expr("neq_null_safe", [a, a]), S.Expr("bit_and", [a, a]), S.Expr("bit_or", [a, a]), S.Expr("bit_xor", [a, a]),
        S.Expr("shift_left", [a, S.lit(2, I32)]), S.Expr("shift_right", [a, S.lit(2, I32)]), S.Expr("unary_minus", [a], fail_on_error=True),
        S.Expr("normalize_nan_and_zero", [b], dtype=DBL), S.Expr("like", [s, S.lit("a%", STR)]), S.neq(a, a), S.eq(s, S.lit("x", STR)),
        S.integral_divide(a, I64, a, I64, S.ANSI, True)])
    plans["aggregates"] = S.hash_agg(sc(), [s, d], [S.count(a, b), S.count(S.lit(1, I32)), S.sum_(x, S.decimal(22, 2), S.ANSI, filter=S.gt(a, S.lit(0, I64))),
                                                   S.avg(x, S.decimal(16, 6), S.decimal(22, 2), S.TRY), S.min_(b, DBL), S.max_(a, I64)], S.PARTIAL_MERGE,
                                     expr_modes=[S.PARTIAL, S.PARTIAL_MERGE, S.PARTIAL_MERGE, S.PARTIAL_MERGE, S.PARTIAL, S.PARTIAL], initial_input_buffer_offset=2)
    plans["sort_limit"] = S.limit(S.sort(sc(), [(a, True, True), (s, False, False)], fetch=10, skip=2), 5, 1)
    plans["joins"] = S.hash_join(S.sort_merge_join(sc(), sc(), [a], [a], S.LEFT_SEMI, condition=S.neq(S.col(1, DBL), S.col(6, DBL))),
                                 S.nested_loop_join(sc(), sc(), S.LEFT_OUTER, S.BUILD_RIGHT, S.gt(S.col(0, I64), S.col(5, I64))),
                                 [a], [a], S.FULL_OUTER, S.BUILD_RIGHT, S.lt(S.col(0, I64), S.col(5, I64)))
    plans["expand"] = S.expand(sc(), [[a, s], [a, S.lit(None, STR)]])
    plans["window"] = S.window(sc(), [s], [(d, False, False)], [("row_number", [], I32), ("lag", [a, S.lit(1, I32)], I64),
                                                               ("agg", S.sum_(x, S.decimal(22, 2)), S.decimal(22, 2), ("range", "unbounded", "current")),
                                                               ("agg", S.count(a), I64, ("rows", "unbounded", "unbounded")),
                                                               ("agg", S.min_(a, I64), I64, ("rows", -3, 2)), ("agg", S.max_(a, I64), I64, ("rows", 0, "current"))])
    plans["window_first_last_nth"] = S.window(sc(), [s], [(a, False, False)], [("agg", S.first_(x, DEC, True), DEC, ("rows", -1, 1)), ("agg", S.last_(s, STR), STR, ("range", "unbounded", "current")),
                                                                                ("nth_value", [x, S.lit(2, I64)], DEC, ("rows", "unbounded", "unbounded"), True),
                                                                                ("nth_value", [s, S.lit(1, I64)], STR, ("rows", -2, "current"))])
    plans["window_range_offsets"] = S.window(sc(), [s], [(a, True, True)], [("agg", S.count(x), I64, ("range", ("value", S.lit(5, I64)), ("value", S.lit(0, I64)))),
                                                                              ("agg", S.sum_(x, S.decimal(22, 2)), S.decimal(22, 2), ("range", "unbounded", ("value", S.lit(7, I64))))])
    for i, (part, kw) in enumerate([("hash", dict(hash_exprs=[a, s], num_partitions=7)), ("single", {}), ("round_robin", dict(num_partitions=5, max_hash_columns=2)),
                                    ("range", dict(sort_orders=[(a, False, False)], num_partitions=3, bounds=[[S.lit(10, I64)], [S.lit(20, I64)]]))]):
        plans[f"shuffle_writer_{part}"] = S.shuffle_writer(sc(), "/tmp/d.data", "/tmp/d.index", part, codec=i, **kw)
    plans["shuffle_scan"] = S.filter_(S.shuffle_scan(fields), S.gt(a, S.lit(0, I64)))
    plans["native_scan"] = S.native_scan([("/data/a.parquet", 0, 100, 100), ("/data/b.parquet", 4, 50, 200)], ["a", "b"], [I64, DEC], case_sensitive=False,
                                         data_filters=[S.gt(S.col(0, I64), S.lit(5, I64))], partition_fields=[("p", I32), ("q", STR)],
                                         partition_values=[(1, "x"), (None, "y")])
    plans["native_scan_field_ids"] = S.native_scan([("/data/a.parquet", 0, 100, 100)], ["a", "b", "c"], [I64, DEC, STR], field_ids=[7, None, 9], use_field_id=True,
                                                   ignore_missing_field_id=True, allow_type_promotion=False, allow_timestamp_ltz_to_ntz=False,
                                                   default_values={1: 12345, 2: "dflt"})
    return plans


def test_descriptor_set_matches_the_reference_protos():
    if not os.path.isdir(REF_PROTO_DIR):
        pytest.skip("reference checkout not present (GPU box): the committed descriptor set is used as is")
    sys.path.insert(0, GOLDEN)
    import make_proto_descriptors as M
    with open(os.path.join(GOLDEN, "comet_protos.desc"), "rb") as f:
        assert M.build(REF_PROTO_DIR).SerializeToString(deterministic=True) == f.read(), "re-run tests/golden/make_proto_descriptors.py"


def test_descriptors_carry_the_survey_field_numbers(pool):
    # spot anchors from SURVEY.md Appendix A, so that a parser bug in make_proto_descriptors.py cannot go unnoticed either
    op = pool.FindMessageTypeByName("spark.spark_operator.Operator")
    assert {f.name: f.number for f in op.fields if f.number >= 100 and f.number <= 111} == {
        "scan": 100, "projection": 101, "filter": 102, "sort": 103, "hash_agg": 104, "limit": 105, "shuffle_writer": 106, "expand": 107,
        "sort_merge_join": 108, "hash_join": 109, "window": 110, "native_scan": 111}
    ex = pool.FindMessageTypeByName("spark.spark_expression.Expr")
    assert ex.fields_by_name["check_overflow"].number == 25 and ex.fields_by_name["caseWhen"].number == 38 and ex.fields_by_name["if"].number == 44
    lit = pool.FindMessageTypeByName("spark.spark_expression.Literal")
    assert lit.fields_by_name["decimal_val"].number == 10 and lit.fields_by_name["datatype"].number == 12 and lit.fields_by_name["is_null"].number == 13
    ha = pool.FindMessageTypeByName("spark.spark_operator.HashAggregate")
    assert [ha.fields_by_name[n].number for n in ("grouping_exprs", "agg_exprs", "mode", "expr_modes", "initial_input_buffer_offset")] == [1, 2, 5, 6, 7]


@pytest.mark.parametrize("name", sorted(corpus()))
def test_serde_bytes_are_the_schemas_bytes(pool, name):
    plan = corpus()[name]
    data = plan.encode()
    Op = cls(pool, "spark.spark_operator.Operator")
    parsed = Op()
    parsed.ParseFromString(data)
    assert unknown_paths(parsed) == [], "serde.py emitted a field the reference schema does not have"
    by_name = Op()
    set_op(by_name, plan)
    assert parsed == by_name, f"serde.py and the by-name encoder disagree:\n{parsed}\n-- vs --\n{by_name}"


def test_config_map_and_metric_node(pool):
    cm = cls(pool, "spark.spark_config.ConfigMap")()
    cm.ParseFromString(S.config_map({"spark.comet.batchSize": "8192", "k": "v"}))
    assert unknown_paths(cm) == [] and dict(cm.entries) == {"spark.comet.batchSize": "8192", "k": "v"}
    node = cls(pool, "spark.spark_metric.NativeMetricNode")()
    node.metrics["output_rows"] = 42
    child = node.children.add()
    child.metrics["elapsed_compute"] = 7
    got = S.decode_metric_node(node.SerializeToString())
    assert got[0] == {"output_rows": 42} and got[1][0][0] == {"elapsed_compute": 7}


def test_proto_cpp_accepts_protobufs_own_bytes(pool):
    """proto.cpp decodes what google.protobuf serializes from by-name messages (packed repeated scalars included) and plans it to the same
    fused pipeline as serde's bytes.  compile_plan needs no GPU (hiprtc cross-compiles)."""
    from datafusion_comet_amd import native
    Op = cls(pool, "spark.spark_operator.Operator")
    plans = corpus()
    for name in ("tpch_0", "tpch_1", "tpch_q3_single", "tpcds_q95_a", "tpcds_q95_b", "sort_limit", "expand", "window", "window_range_offsets", "window_first_last_nth"):
        by_name = Op()
        set_op(by_name, plans[name])
        theirs = by_name.SerializeToString(deterministic=True)
        assert native.compile_plan(theirs) == native.compile_plan(plans[name].encode()), name
