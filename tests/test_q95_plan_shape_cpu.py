"""The plan `tpcds.q95_plans()` builds for BASELINE config 5 against the reference's OWN approved plan for TPC-DS Q95
(spark/src/test/resources/tpcds-plan-stability/approved-plans-v1_4/q95/extended.txt, kept verbatim as tests/golden/tpcds/q95_approved_plan_extended.txt — a data file of
the reference's plan-stability suite): the operator tree Comet runs natively, operator by operator.  Exchanges, the sorts under them and the broadcast exchanges are stage
boundaries / JVM-side operators (CometExchange, CometSort feeding a sort-merge join, CometBroadcastExchange, CometColumnarToRow): within one partition the native operators
between them are what a plan handed to createPlan contains, so they are dropped from the approved tree before the comparison; everything else must match in kind, arity
and order — five scans of web_sales, web_returns and the three dimensions in the same depth-first order, the ws_wh self-join under BOTH semi joins, the three broadcast
joins, the four aggregates."""
import os
import re

from datafusion_comet_amd import serde as S, tpcds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROP = {"CometExchange", "CometSort", "CometBroadcastExchange", "CometColumnarToRow"}
KIND = {"CometHashAggregate": "hash_agg", "CometProject": "projection", "CometFilter": "filter", "CometSortMergeJoin": "sort_merge_join",
        "CometBroadcastHashJoin": "hash_join", "CometNativeScan": "scan"}


def _approved_tree():
    lines = [l.rstrip("\n") for l in open(os.path.join(ROOT, "tests", "golden", "tpcds", "q95_approved_plan_extended.txt")) if l.strip() and not l.startswith("Comet accelerated")]
    nodes = []      # (depth, name, table)
    for l in lines:
        m = re.match(r"^([ :+\-]*)(Comet\w+)(?: parquet spark_catalog\.default\.(\w+))?", l)
        assert m, l
        nodes.append((len(m.group(1)) // 3, m.group(2), m.group(3)))
    root = None
    stack = []      # (depth, node)
    for depth, name, table in nodes:
        node = {"name": name, "table": table, "children": []}
        while stack and stack[-1][0] >= depth:
            stack.pop()
        if stack:
            stack[-1][1]["children"].append(node)
        else:
            root = node
        stack.append((depth, node))
    return root


def _strip(node):
    """the approved tree without stage boundaries; → list of nodes (a dropped node is replaced by its stripped children)"""
    kids = [k for c in node["children"] for k in _strip(c)]
    if node["name"] in DROP:
        return kids
    return [{"kind": KIND[node["name"]], "table": node["table"], "children": kids}]


def _ours(op, leaves):
    kind = op.kind
    if kind == "scan":
        return {"kind": "scan", "table": leaves.pop(0), "children": []}
    return {"kind": kind, "table": None, "children": [_ours(c, leaves) for c in op.children]}


def _same(a, b, path="root"):
    assert a["kind"] == b["kind"], (path, a["kind"], b["kind"])
    assert a["table"] == b["table"], (path, a["table"], b["table"])
    assert len(a["children"]) == len(b["children"]), (path, a["kind"], len(a["children"]), len(b["children"]))
    for i, (x, y) in enumerate(zip(a["children"], b["children"])):
        _same(x, y, f"{path}/{a['kind']}[{i}]")


def test_q95_plan_has_the_approved_plans_operator_tree():
    approved = _strip(_approved_tree())
    assert len(approved) == 1
    stage_a, stage_b, leaves = tpcds.q95_plans()
    assert leaves == ["web_sales", "web_sales", "web_sales", "web_returns", "web_sales", "web_sales", "date_dim", "customer_address", "web_site"]
    ours_a = _ours(stage_a, list(leaves))
    # stage B is the Final aggregate above the exchange: its Scan leaf stands for the stage boundary, stage A hangs below it
    assert stage_b.kind == "hash_agg" and stage_b.mode == S.FINAL and len(stage_b.children) == 1 and stage_b.children[0].kind == "scan"
    ours = {"kind": "hash_agg", "table": None, "children": [ours_a]}
    _same(approved[0], ours)


def test_q95_join_types_and_aggregate_modes_follow_the_approved_plan():
    stage_a, stage_b, _ = tpcds.q95_plans()
    joins, aggs = [], []

    def walk(op):
        if op.kind in ("sort_merge_join", "hash_join"):
            joins.append((op.kind, op.join_type, op.condition is not None))
        if op.kind == "hash_agg":
            aggs.append((op.mode, list(op.expr_modes), len(op.exprs)))
        for c in op.children:
            walk(c)

    walk(stage_a)
    # depth-first: three broadcast joins (inner), the two LeftSemi sort-merge joins, the ws_wh self-joins (inner, wh1 <> wh2) and web_returns ⋈ ws_wh
    assert [j[0] for j in joins[:3]] == ["hash_join"] * 3 and all(j[1] == S.INNER and not j[2] for j in joins[:3])
    smj = [j for j in joins if j[0] == "sort_merge_join"]
    assert [(j[1], j[2]) for j in smj] == [(S.LEFT_SEMI, False), (S.LEFT_SEMI, False), (S.INNER, True), (S.INNER, False), (S.INNER, True)]
    # count(DISTINCT ws_order_number) as the four-aggregate rewrite: Partial by order → PartialMerge by order → mixed (sums merge, the count is Partial) → Final
    assert [(m, em, n) for m, em, n in aggs] == [(S.PARTIAL, [S.PARTIAL_MERGE, S.PARTIAL_MERGE, S.PARTIAL], 0), (S.PARTIAL_MERGE, [], 1), (S.PARTIAL, [], 1)]
    assert stage_b.mode == S.FINAL
