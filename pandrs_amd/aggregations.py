"""Ready-made custom aggregations — host mirror of `jit::groupby::aggregations`
(src/optimized/jit/groupby.rs:325-437) and of the Kahan kernels behind GroupByJitExt (:69-172).

Each function maps a group's non-null values (a sequence of floats, Int64 already cast) to one float,
which is what `GroupBy.aggregate_jit` / `aggregate_custom` hand to a closure.  They run on the host:
closures are outside the device path; the groups they run over come from
pandrs_hip_groupby_indices."""
import math

import numpy as np


def weighted_mean(values):
    """Position-weighted mean, weight of the i-th value = i + 1 (groupby.rs:328-349); empty -> 0.0."""
    v = np.asarray(values, np.float64)
    if v.size == 0:
        return 0.0
    w = np.arange(1, v.size + 1, dtype=np.float64)
    return float((v * w).sum() / w.sum())


def geometric_mean(values):
    """exp(mean(ln x)) over the POSITIVE values only (groupby.rs:352-374); none -> 0.0."""
    v = np.asarray(values, np.float64)
    v = v[v > 0.0]
    return float(np.exp(np.log(v).sum() / v.size)) if v.size else 0.0


def harmonic_mean(values):
    """count / sum(1/x) over the NON-ZERO values only (groupby.rs:377-399); none -> 0.0."""
    v = np.asarray(values, np.float64)
    v = v[v != 0.0]
    return float(v.size / (1.0 / v).sum()) if v.size else 0.0


def value_range(values):
    """max - min with NaN-ignoring folds (groupby.rs:402-413, the reference's `range`); empty -> 0.0."""
    v = np.asarray(values, np.float64)
    if v.size == 0:
        return 0.0
    v = v[~np.isnan(v)]
    lo = v.min() if v.size else np.inf          # f64::min / f64::max skip NaN operands
    hi = v.max() if v.size else -np.inf
    return float(hi - lo)


def coefficient_of_variation(values):
    """sample std / |mean| (groupby.rs:416-436); fewer than 2 values or a zero mean -> 0.0."""
    v = np.asarray(values, np.float64)
    if v.size <= 1:
        return 0.0
    mean = v.sum() / v.size
    if mean == 0.0:
        return 0.0
    return float(math.sqrt(((v - mean) ** 2).sum() / (v.size - 1)) / abs(mean))


def kahan_sum(values):
    """Compensated sum, the loop inside sum_jit / mean_jit / std_jit (groupby.rs:72-82)."""
    total = comp = 0.0
    for x in values:
        y = x - comp
        t = total + y
        comp = (t - total) - y
        total = t
    return total


def kahan_mean(values):
    return kahan_sum(values) / len(values) if len(values) else 0.0


def kahan_std(values):
    """Two Kahan passes, n - 1 denominator, n <= 1 -> 0.0 (groupby.rs:131-160)."""
    n = len(values)
    if n <= 1:
        return 0.0
    mean = kahan_sum(values) / n
    return math.sqrt(kahan_sum([(x - mean) * (x - mean) for x in values]) / (n - 1))


def population_std(values):
    """parallel_std_f64_value (jit/parallel.rs:222-233): sqrt(max(E[x^2] - E[x]^2, 0)), count <= 1 -> 0.0."""
    n = len(values)
    if n <= 1:
        return 0.0
    mean = kahan_sum(values) / n
    return math.sqrt(max(kahan_sum([x * x for x in values]) / n - mean * mean, 0.0))
