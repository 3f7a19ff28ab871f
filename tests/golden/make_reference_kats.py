#!/usr/bin/env python3
"""Writes tests/golden/reference_kats.json: the known-answer vectors the reference's OWN tests hold for the hot
path, transcribed with their source location (nothing is computed here — these are the reference's asserted
values).  Re-run only when the reference's tests change.  /root/reference is NOT needed at test time."""
import json
import os

K = {}
# native/spark-expr/src/hash_funcs/murmur3.rs:208-281 (seed 42)
K["murmur3"] = {
    "i8": {"values": [1, 0, -1, 127, -128], "expected": [0xdea578e3, 0x379fae8f, 0xa0590e3d, 0x43b4d8ed, 0x422a1365]},
    "i32": {"values": [1, 0, -1, 2147483647, -2147483648], "expected": [0xdea578e3, 0x379fae8f, 0xa0590e3d, 0x07fb67e7, 0x2b1f0fc6]},
    "i64": {"values": [1, 0, -1, 9223372036854775807, -9223372036854775808],
            "expected": [0x99f0149d, 0x9c67b85d, 0xc8008529, 0xa05b5d7b, 0xcd1e64fb]},
    "f32": {"values": [1.0, 0.0, -0.0, -1.0, 99999999999.99999999999, -99999999999.99999999999],
            "expected": [0xe434cc39, 0x379fae8f, 0x379fae8f, 0xdc0da8eb, 0xcbdc340f, 0xc0361c86]},
    "f64": {"values": [1.0, 0.0, -0.0, -1.0, 99999999999.99999999999, -99999999999.99999999999],
            "expected": [0xe4876492, 0x9c67b85d, 0x9c67b85d, 0x13d81357, 0xb87e1595, 0xa0eef9f9]},
    "str": {"values": ["hello", "bar", "", "😁", "天地", "a", "ab", "abc", "abcd", "abcde"],
            "expected": [3286402344, 2486176763, 142593372, 885025535, 2395000894, 1485273170, 0xfa37157b, 1322437556,
                         0xe860e5cc, 814637928]},
}
# native/shuffle/src/comet_partitioning.rs:63-72
K["pmod"] = {"hashes": [0x99f0149d, 0x9c67b85d, 0xc8008529, 0xa05b5d7b, 0xcd1e64fb], "n": 200, "expected": [69, 5, 193, 171, 115]}
# native/spark-expr/src/math_funcs/wide_decimal_binary_expr.rs:400-590 ; null = overflow → NULL (LEGACY)
K["wide_decimal"] = [
    {"op": "add", "l": [1000000000, 2500000000], "s1": 10, "r": [2000000000, 7500000000], "s2": 10, "p_out": 38, "s_out": 10, "expected": [3000000000, 10000000000]},
    {"op": "subtract", "l": [5000, 1000], "s1": 2, "r": [3000, 2000], "s2": 2, "p_out": 38, "s_out": 2, "expected": [2000, -1000]},
    {"op": "add", "l": [150], "s1": 2, "r": [2500], "s2": 4, "p_out": 38, "s_out": 4, "expected": [17500]},
    {"op": "multiply", "l": [100000], "s1": 5, "r": [200000], "s2": 5, "p_out": 38, "s_out": 6, "expected": [2000000]},
    {"op": "multiply", "l": [15], "s1": 1, "r": [15], "s2": 1, "p_out": 38, "s_out": 1, "expected": [23]},
    {"op": "multiply", "l": [-15], "s1": 1, "r": [15], "s2": 1, "p_out": 38, "s_out": 1, "expected": [-23]},
    {"op": "add", "l": [5], "s1": 0, "r": [5], "s2": 0, "p_out": 1, "s_out": 0, "expected": [None]},
    {"op": "multiply", "l": [0], "s1": 10, "r": [0], "s2": 10, "p_out": 38, "s_out": 10, "expected": [0]},
    {"op": "add", "l": [10**38 - 1], "s1": 0, "r": [0], "s2": 0, "p_out": 38, "s_out": 0, "expected": [10**38 - 1]},
    {"op": "add", "l": [150], "s1": 2, "r": [25], "s2": 2, "p_out": 38, "s_out": 4, "expected": [17500]},
    {"op": "subtract", "l": [300], "s1": 2, "r": [100], "s2": 2, "p_out": 38, "s_out": 4, "expected": [20000]},
    {"op": "multiply", "l": [95], "s1": 2, "r": [10000], "s2": 2, "p_out": 38, "s_out": 2, "expected": [9500]},
]
# native/spark-expr/src/math_funcs/internal/checkoverflow.rs:417-512 (input Decimal128(38,0) → check precision 3)
K["check_overflow"] = [
    {"values": [999, 12, None, 5], "p": 3, "expected": [999, 12, None, 5]},
    {"values": [999, 1000, None, 5], "p": 3, "expected": [999, None, None, 5]},
    {"values": [-1000, 5], "p": 3, "expected": [None, 5]},
    {"values": [1000, 5000, -2000], "p": 3, "expected": [None, None, None]},
    {"values": [999, 9999], "p": 3, "expected": [999, None]},
]
# native/spark-expr/src/math_funcs/internal/decimal_rescale_check.rs:302-397
K["rescale_check"] = [
    {"values": [150, -300], "s_in": 2, "p_out": 10, "s_out": 4, "expected": [15000, -30000]},
    {"values": [12350, 12349, -12350], "s_in": 4, "p_out": 10, "s_out": 2, "expected": [124, 123, -124]},
    {"values": [999, 1000], "s_in": 0, "p_out": 3, "s_out": 0, "expected": [999, None]},
    {"values": [10], "s_in": 0, "p_out": 3, "s_out": 2, "expected": [None]},
    {"values": [150, 10000, None, 250], "s_in": 2, "p_out": 4, "s_out": 2, "expected": [150, None, None, 250]},
    {"values": [10000, 20000, 30000], "s_in": 2, "p_out": 4, "s_out": 2, "expected": [None, None, None]},
    {"values": [9999, 10000], "s_in": 0, "p_out": 4, "s_out": 0, "expected": [9999, None]},
]
# native/spark-expr/src/agg_funcs/sum_decimal.rs:723-802
K["sum_decimal"] = {
    "update_with_filter": {"values": [100, 200, 300, 400], "filter": [True, False, True, False], "precision": 10, "expected": 400},
    "update_filter_null_excluded": {"values": [10, 20, 30], "filter": [True, None, True], "precision": 10, "expected": 40},
    "merge_multi_row": {"sums": [100, 200, None, 300], "is_empty": [False, False, True, False], "precision": 10, "expected": 600},
}
# native/spark-expr/src/agg_funcs/sum_int.rs:919-1014
K["sum_int"] = {
    "legacy_filter": {"values": [1, 2, 3, 4, 5], "filter": [True, False, True, False, True], "expected": 9},
    "legacy_filter_null": {"values": [10, 20, 30], "filter": [True, None, True], "expected": 40},
    "no_filter": {"values": [1, 2, 3], "expected": 6},
    "merge_multi_row": {"states": [1, 2, None, 3], "expected": 6},
}
# spark/src/test/resources/tpch-query-results/q1.sql.out:6-9 — the golden SF1 answers pin AvgDecimal's final
# HALF_UP division: avg = sum/count at decimal(16,6) from a decimal(22,2) sum (avg_decimal.rs:670-689)
K["avg_decimal_golden_q1"] = [
    {"sum": "37734107.00", "count": 1478493, "avg": "25.522006"},
    {"sum": "56586554400.73", "count": 1478493, "avg": "38273.129735"},
    {"sum": "991417.00", "count": 38854, "avg": "25.516472"},
    {"sum": "1487504710.38", "count": 38854, "avg": "38284.467761"},
    {"sum": "74476040.00", "count": 2920374, "avg": "25.502227"},
    {"sum": "111701729697.74", "count": 2920374, "avg": "38249.117989"},
    {"sum": "37719753.00", "count": 1478870, "avg": "25.505794"},
    {"sum": "56568041380.90", "count": 1478870, "avg": "38250.854626"},
]
# native/core/src/execution/planner.rs:4637-4699: `col = 3` over n % 4, 100 rows → 25 rows
# native/shuffle/src/partitioners/multi_partition.rs:78-84 (the worked example in map_partition_ids_to_starts_and_indices)
K["partition_indices"] = {"partition_ids": [3, 1, 1, 1, 2, 2, 0], "num_partitions": 4,
                          "partition_row_indices": [6, 1, 2, 3, 4, 5, 0], "partition_starts": [0, 1, 4, 6, 7]}
K["planner_filter_case"] = {"rows": 100, "modulus": 4, "equals": 3, "expected_rows": 25}

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
json.dump(K, open(out, "w"), indent=1, ensure_ascii=False)
print("wrote", out)
