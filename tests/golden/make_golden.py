#!/usr/bin/env python3
"""Writes tests/golden/reference_known_answers.json.

The reference (cool-japan/pandrs) is Rust and cannot run in this image, so these vectors are
the INPUTS and EXPECTED OUTPUTS its own tests assert, transcribed as data (values only — no
reference source text).  Each case cites the reference test it comes from.  Where the reference
test is on the legacy string-keyed frame, string keys are given as `key_strings` and the test
harness maps them to string-pool codes (equal string <=> equal code,
src/column/string_column.rs:26-32).

Cases flagged "derived": true have no asserting test in the reference; the expected values are
derived by hand from the cited source lines (SURVEY.md §8c "source-derived").
"""
import json
import os

NAN = "nan"   # JSON has no NaN literal; the test loader maps "nan" -> float('nan')

cases = {
    "groupby": [
        {   # tests/groupby_test.rs:18-67
            "cite": "tests/groupby_test.rs:18-67",
            "key_strings": ["A", "B", "A", "B", "C"],
            "values_i64": [10, 20, 30, 40, 50],
            "expect": {"A": {"count": 2, "sum": 40, "mean": 20.0},
                       "B": {"count": 2, "sum": 60, "mean": 30.0},
                       "C": {"count": 1, "sum": 50, "mean": 50.0}},
        },
        {   # tests/groupby_test.rs:69-84 numeric keys
            "cite": "tests/groupby_test.rs:69-84",
            "key_i64": [1, 2, 1, 2, 3],
            "values_i64": [10, 20, 30, 40, 50],
            "expect": {"1": {"sum": 40}, "2": {"sum": 60}, "3": {"sum": 50}},
        },
        {   # src/dataframe/pandas_compat/groupby.rs:480-651 (create_test_df + tests)
            "cite": "src/dataframe/pandas_compat/groupby.rs:480-651",
            "key_strings": ["A", "B", "A", "B", "A"],
            "values_f64": [10.0, 20.0, 30.0, 40.0, 50.0],
            "expect": {"A": {"sum": 90.0, "mean": 30.0, "min": 10.0, "max": 50.0, "count": 3,
                             "std": 20.0, "first": 10.0, "last": 50.0},
                       "B": {"sum": 60.0, "mean": 30.0, "min": 20.0, "max": 40.0, "count": 2,
                             "first": 20.0, "last": 40.0}},
        },
        {   # same fixture, second value column (score) used by test_groupby_agg :721-738
            "cite": "src/dataframe/pandas_compat/groupby.rs:721-738",
            "key_strings": ["A", "B", "A", "B", "A"],
            "values_f64": [1.0, 2.0, 3.0, 4.0, 5.0],
            "expect": {"A": {"max": 5.0}, "B": {"max": 4.0}},
        },
        {   # :696-718 — the legacy frame skips NaN; on the typed path a missing value is a NULL
            # (mask bit), which the fold skips (aggregation.rs:628) — same expected sum 40.
            "cite": "src/dataframe/pandas_compat/groupby.rs:696-718",
            "key_strings": ["A", "A", "A"],
            "values_f64": [10.0, 0.0, 30.0],
            "value_nulls": [False, True, False],
            "expect": {"A": {"sum": 40.0}},
        },
        {   # examples/optimized_groupby_example.rs:22-33 data; values by hand from
            # aggregation.rs:507-556 (the example only prints)
            "cite": "examples/optimized_groupby_example.rs:22-33", "derived": True,
            "key_strings": ["A", "B", "A", "C", "B", "A"],
            "values_i64": [10, 20, 15, 30, 25, 15],
            "expect": {"A": {"count": 3, "sum": 40, "mean": 13.333333333333334, "min": 10, "max": 15},
                       "B": {"count": 2, "sum": 45, "mean": 22.5, "min": 20, "max": 25},
                       "C": {"count": 1, "sum": 30, "mean": 30.0, "min": 30, "max": 30}},
        },
    ],
    "groupby_two_keys": {   # src/dataframe/pandas_compat/groupby.rs:653-693
        "cite": "src/dataframe/pandas_compat/groupby.rs:653-693",
        "key1_strings": ["A", "A", "B", "B"], "key2_strings": ["X", "Y", "X", "Y"],
        "values_f64": [1.0, 2.0, 3.0, 4.0], "expect_n_groups": 4,
    },
    "join_string_key": {    # src/dataframe/pandas_compat/merge.rs:271-411
        "cite": "src/dataframe/pandas_compat/merge.rs:271-411",
        "left_keys": ["A", "B", "C", "D"], "left_value1": [1.0, 2.0, 3.0, 4.0],
        "right_keys": ["B", "C", "D", "E"], "right_value2": [20.0, 30.0, 40.0, 50.0],
        # expected (key, value1, value2) rows; null = the missing side (NaN in the legacy merge,
        # 0.0 fill on the optimized path join.rs:304-307 — the test checks indices, then fills)
        "inner": {"keys": ["B", "C", "D"], "left_idx": [1, 2, 3], "right_idx": [0, 1, 2]},
        "left": {"keys": ["A", "B", "C", "D"], "left_idx": [0, 1, 2, 3], "right_idx": [-1, 0, 1, 2]},
        "right": {"keys": ["B", "C", "D", "E"], "left_idx": [1, 2, 3, -1], "right_idx": [0, 1, 2, 3]},
        "outer": {"keys": ["A", "B", "C", "D", "E"], "left_idx": [0, 1, 2, 3, -1],
                  "right_idx": [-1, 0, 1, 2, 3]},
    },
    "join_numeric_key": {   # src/dataframe/pandas_compat/merge.rs:463-507
        "cite": "src/dataframe/pandas_compat/merge.rs:463-507",
        "left_keys_f64": [1.0, 2.0, 3.0], "left_names": ["Alice", "Bob", "Charlie"],
        "right_keys_f64": [2.0, 3.0, 4.0], "right_scores": [85.0, 90.0, 95.0],
        "inner": {"names": ["Bob", "Charlie"], "scores": [85.0, 90.0],
                  "left_idx": [1, 2], "right_idx": [0, 1]},
    },
    "join_optimized": {     # tests/optimized_join_test.rs:6-232 (row / column counts)
        "cite": "tests/optimized_join_test.rs:6-232",
        "left_ids": [1, 2, 3, 4], "right_ids": [1, 2, 5, 6], "right_values": [100, 200, 500, 600],
        "rows": {"inner": 2, "left": 4, "right": 4, "outer": 6}, "columns": 3,
        "disjoint": {"left_ids": [1, 2, 3], "right_ids": [4, 5, 6], "inner_rows": 0},
    },
    "reductions": [         # src/optimized/jit/simd.rs:451-506, src/optimized/jit/parallel.rs:354-408
        {"cite": "src/optimized/jit/simd.rs:451-456", "f64": [1, 2, 3, 4, 5, 6, 7, 8], "sum": 36.0},
        {"cite": "src/optimized/jit/simd.rs:459-464", "f64": [1, 2, 3, 4, 5], "mean": 3.0},
        {"cite": "src/optimized/jit/simd.rs:467-476", "f64": [3, 1, 4, 1, 5, 9, 2], "min": 1.0, "max": 9.0},
        {"cite": "src/optimized/jit/simd.rs:479-484", "i64": [1, 2, 3, 4, 5, 6, 7, 8], "sum": 36.0},
        {"cite": "src/optimized/jit/simd.rs:493-506", "f64": [], "sum": 0.0, "mean": 0.0,
         "min": "inf", "max": "-inf"},
        {"cite": "src/optimized/jit/simd.rs:493-506", "i64": [], "sum": 0.0, "mean": 0.0,
         "min": 9.223372036854775807e18, "max": -9.223372036854775808e18},
        {"cite": "src/optimized/jit/parallel.rs:354-363", "f64_range": [1, 1000], "sum": 500500.0},
        {"cite": "src/optimized/jit/parallel.rs:366-372", "f64_range": [1, 100], "mean": 50.5},
    ],
    # src/optimized/direct_aggregations.rs:363-382 (create_test_dataframe) and its tests :384-520: the *_direct family is
    # Float64Column / Int64Column::{sum, mean, min, max} (:30-120), the *_simd family simd_*_{f64,i64} over the raw slice
    # (:200-356); both are asserted to give the same numbers on this frame (:505-545)
    "direct_aggregations": {
        "cite": "src/optimized/direct_aggregations.rs:363-545",
        "float_col": [1.0, 2.0, 3.0, 4.0, 5.0], "int_col": [10, 20, 30, 40, 50],
        "expect": {"float_col": {"sum": 15.0, "mean": 3.0, "max": 5.0, "min": 1.0, "count": 5},
                   "int_col": {"sum": 150.0, "mean": 30.0, "max": 50.0, "min": 10.0, "count": 5}},
        # :548-593: 10 000 elements, float i * 0.1 and int i * 10 for i = 1..=10000; sum and mean within 1e-10 of the
        # sequential f64 sum, int max / min exact
        "large": {"cite": "src/optimized/direct_aggregations.rs:548-593", "n": 10000, "float_scale": 0.1, "int_scale": 10,
                  "abs_tolerance": 1e-10, "int_max": 100000.0, "int_min": 10.0},
    },
    # src/dataframe/pandas_compat/merge.rs:414-460: a non-key column present on both sides takes the suffixes, left then right
    "merge_suffixes": {
        "cite": "src/dataframe/pandas_compat/merge.rs:414-460",
        "left": {"key": ["A", "B"], "value": [1.0, 2.0]}, "right": {"key": ["A", "B"], "value": [10.0, 20.0]},
        "how": "inner", "suffixes": ["_left", "_right"],
        "expect": {"value_left": [1.0, 2.0], "value_right": [10.0, 20.0]},
    },
}

here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "reference_known_answers.json"), "w") as f:
    json.dump(cases, f, indent=1, sort_keys=True)
print("wrote reference_known_answers.json")
