// The reference's own tests for the accelerated path, replayed through the C++ host mirror
// (include/pandrs_hip.hpp) over libpandrs_hip.so — no Python, no torch in the process.
// Every test names the reference test it follows; where the reference only checks shapes
// ("at least one group exists"), the known answers of tests/golden/ are asserted as well.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "pandrs_hip.hpp"

using namespace pandrs;

static int g_failed = 0, g_run = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("    CHECK failed: %s  (%s:%d)\n", #cond, __FILE__, __LINE__); g_failed++; } } while (0)
#define RUN(fn) do { g_run++; std::printf("test %s\n", #fn); try { fn(); } catch (const std::exception &e) { std::printf("    threw: %s\n", e.what()); g_failed++; } } while (0)

static OptimizedDataFrame values_keys_frame() {      // the frame of tests/optimized_groupby_test.rs:8-23
    OptimizedDataFrame df;
    df.add_column("values", Int64Column({10, 20, 30, 40, 50}));
    df.add_column("keys", StringColumn({"A", "B", "A", "B", "C"}));
    return df;
}
static std::map<std::string, double> by_key(const OptimizedDataFrame &r, const std::string &key, const std::string &val) {
    std::map<std::string, double> m;
    auto &k = std::get<StringColumn>(r.column(key));
    auto &v = std::get<Float64Column>(r.column(val));
    for (size_t i = 0; i < r.row_count(); i++) m[k.get(i)] = v.data[i];
    return m;
}

// tests/optimized_groupby_test.rs:6-30
static void test_optimized_groupby_creation() {
    auto df = values_keys_frame();
    auto grouped = df.par_groupby({"keys"});
    CHECK(!grouped.empty());
    CHECK(grouped.size() == 3 && grouped.at("A").row_count() == 2 && grouped.at("C").row_count() == 1);   // tests/groupby_test.rs:18-40
}
// tests/optimized_groupby_test.rs:35-80 (+ the sums / means of tests/groupby_test.rs:42-82)
static void test_optimized_groupby_aggregation() {
    auto df = values_keys_frame();
    auto result = LazyFrame(df).aggregate({"keys"}, {{"values", AggregateOp::Sum, "sum"}}).execute();
    CHECK(result.row_count() > 0 && result.contains_column("keys") && result.contains_column("sum"));
    auto sums = by_key(result, "keys", "sum");
    CHECK(sums.size() == 3 && sums["A"] == 40.0 && sums["B"] == 60.0 && sums["C"] == 50.0);
    auto result_mean = LazyFrame(df).aggregate({"keys"}, {{"values", AggregateOp::Mean, "mean"}}).execute();
    CHECK(result_mean.row_count() > 0 && result_mean.contains_column("keys") && result_mean.contains_column("mean"));
    auto means = by_key(result_mean, "keys", "mean");
    CHECK(means["A"] == 20.0 && means["B"] == 30.0 && means["C"] == 50.0);
}
// tests/optimized_groupby_test.rs:85-133
static void test_optimized_groupby_multiple_aggregations() {
    auto df = values_keys_frame();
    auto result = LazyFrame(df).aggregate({"keys"}, {{"values", AggregateOp::Count, "count"}, {"values", AggregateOp::Sum, "sum"},
                                                      {"values", AggregateOp::Mean, "mean"}, {"values", AggregateOp::Min, "min"},
                                                      {"values", AggregateOp::Max, "max"}}).execute();
    CHECK(result.row_count() > 0);
    CHECK(result.column_count() == 6);
    for (auto name : {"keys", "count", "sum", "mean", "min", "max"}) CHECK(result.contains_column(name));
    CHECK((result.column_names == std::vector<std::string>{"keys", "count", "sum", "mean", "min", "max"}));
    CHECK(by_key(result, "keys", "count")["A"] == 2.0 && by_key(result, "keys", "min")["B"] == 20.0 && by_key(result, "keys", "max")["A"] == 30.0);
}
// tests/optimized_groupby_test.rs:138-186
static void test_optimized_groupby_multiple_keys() {
    OptimizedDataFrame df;
    df.add_column("category", StringColumn({"A", "A", "B", "B", "A"}));
    df.add_column("group", StringColumn({"X", "Y", "X", "Y", "X"}));
    df.add_column("values", Int64Column({10, 20, 30, 40, 50}));
    auto grouped = df.par_groupby({"category", "group"});
    CHECK(!grouped.empty() && grouped.size() == 4 && grouped.at("A_X").row_count() == 2);
    auto result = LazyFrame(df).aggregate({"category", "group"}, {{"values", AggregateOp::Sum, "sum"}}).execute();
    CHECK(result.row_count() > 0);
    CHECK(result.column_count() == 3);
    double ax = -1;
    auto &c = std::get<StringColumn>(result.column("category"));
    auto &g = std::get<StringColumn>(result.column("group"));
    auto &s = std::get<Float64Column>(result.column("sum"));
    for (size_t i = 0; i < result.row_count(); i++) if (c.get(i) == "A" && g.get(i) == "X") ax = s.data[i];
    CHECK(result.row_count() == 4 && ax == 60.0);
}
// examples/optimized_groupby_example.rs:22-33 (values derived by hand from aggregation.rs:500-754) and the
// shortcut naming "{col}_{op}" (operations.rs:515, :541)
static void test_groupby_example_and_shortcuts() {
    OptimizedDataFrame df;
    df.add_column("category", StringColumn({"A", "B", "A", "C", "B", "A"}));
    df.add_column("values", Int64Column({10, 20, 15, 30, 25, 15}));
    auto gb = df.group_by({"category"});
    auto r = gb.agg({{"values", AggregateOp::Count}, {"values", AggregateOp::Sum}, {"values", AggregateOp::Mean},
                     {"values", AggregateOp::Min}, {"values", AggregateOp::Max}});
    CHECK((r.column_names == std::vector<std::string>{"category", "values_count", "values_sum", "values_mean", "values_min", "values_max"}));
    CHECK(by_key(r, "category", "values_sum")["A"] == 40.0 && by_key(r, "category", "values_sum")["B"] == 45.0);
    CHECK(std::fabs(by_key(r, "category", "values_mean")["A"] - 40.0 / 3.0) < 1e-12);
    CHECK(by_key(r, "category", "values_min")["C"] == 30.0 && by_key(r, "category", "values_max")["B"] == 25.0);
    CHECK(by_key(gb.median("values"), "category", "values_median")["A"] == 15.0);
    CHECK(by_key(gb.median("values"), "category", "values_median")["B"] == 22.5);
    CHECK(by_key(gb.first("values"), "category", "values_first")["B"] == 20.0 && by_key(gb.last("values"), "category", "values_last")["A"] == 15.0);
    CHECK(by_key(gb.nunique("values"), "category", "values_nunique")["A"] == 2.0 && by_key(gb.nunique("values"), "category", "values_nunique")["C"] == 1.0);   // {10, 15, 15}
    auto groups = gb.groups();
    CHECK((groups[{"A"}] == std::vector<size_t>{0, 2, 5}) && (groups[{"B"}] == std::vector<size_t>{1, 4}) && (groups[{"C"}] == std::vector<size_t>{3}));
    auto big = gb.filter([](const OptimizedDataFrame &g) { return g.row_count() >= 2; });                 // operations.rs:51-74
    CHECK(big.row_count() == 5 && std::get<StringColumn>(big.column("category")).len() == 5);
    auto centered = gb.transform([](const OptimizedDataFrame &g) {                                        // operations.rs:132-276
        auto &v = std::get<Int64Column>(g.column("values")).data;
        double m = 0; for (auto x : v) m += (double)x; m /= (double)v.size();
        std::vector<double> c; for (auto x : v) c.push_back((double)x - m);
        OptimizedDataFrame out; out.add_column("category", g.column("category")); out.add_column("centered", Float64Column(c));
        return out;
    });
    CHECK(centered.row_count() == 6 && centered.column_count() == 2);
    double abs_sum = 0; for (double x : std::get<Float64Column>(centered.column("centered")).data) abs_sum += x;
    CHECK(std::fabs(abs_sum) < 1e-9);
    auto range = gb.custom("values", "range", [](const std::vector<double> &v) { return *std::max_element(v.begin(), v.end()) - *std::min_element(v.begin(), v.end()); });
    CHECK(by_key(range, "category", "range")["A"] == 5.0 && by_key(range, "category", "range")["C"] == 0.0);
}
// src/dataframe/pandas_compat/groupby.rs:480-693 fixture: sum / mean / min / max / count / std / first / last
static void test_pandas_compat_fixture_values() {
    OptimizedDataFrame df;
    df.add_column("category", StringColumn({"A", "B", "A", "B", "A"}));
    df.add_column("value", Float64Column({10, 20, 30, 40, 50}));
    auto gb = df.group_by({"category"});
    CHECK(by_key(gb.sum("value"), "category", "value_sum")["A"] == 90.0 && by_key(gb.sum("value"), "category", "value_sum")["B"] == 60.0);
    CHECK(by_key(gb.mean("value"), "category", "value_mean")["A"] == 30.0 && by_key(gb.min("value"), "category", "value_min")["B"] == 20.0);
    CHECK(by_key(gb.max("value"), "category", "value_max")["A"] == 50.0 && by_key(gb.count("value"), "category", "value_count")["A"] == 3.0);
    CHECK(std::fabs(by_key(gb.std("value"), "category", "value_std")["A"] - 20.0) < 1e-9);
    CHECK(by_key(gb.first("value"), "category", "value_first")["A"] == 10.0 && by_key(gb.last("value"), "category", "value_last")["B"] == 40.0);
}
// tests/optimized_join_test.rs:6-164, :206-232
static void test_optimized_joins() {
    OptimizedDataFrame left, right;
    left.add_column("id", Int64Column({1, 2, 3, 4}));
    left.add_column("value", StringColumn({"A", "B", "C", "D"}));
    right.add_column("id", Int64Column({1, 2, 5, 6}));
    right.add_column("score", Float64Column({85.0, 92.5, 77.0, 68.5}));
    auto inner = left.inner_join(right, "id", "id");
    CHECK(inner.row_count() == 2 && inner.column_count() == 3);
    CHECK(inner.contains_column("id") && inner.contains_column("value") && inner.contains_column("score"));
    CHECK((std::get<Int64Column>(inner.column("id")).data == std::vector<int64_t>{1, 2}));
    CHECK((std::get<Float64Column>(inner.column("score")).data == std::vector<double>{85.0, 92.5}));
    CHECK(left.left_join(right, "id", "id").row_count() == 4);
    CHECK(left.right_join(right, "id", "id").row_count() == 4);
    auto outer = left.outer_join(right, "id", "id");
    CHECK(outer.row_count() == 6 && outer.column_count() == 3);
    CHECK((std::get<Int64Column>(outer.column("id")).data == std::vector<int64_t>{1, 2, 3, 4, 5, 6}));      // left rows, then unmatched right rows
    CHECK((std::get<Float64Column>(outer.column("score")).data == std::vector<double>{85.0, 92.5, 0.0, 0.0, 77.0, 68.5}));   // misses filled with 0.0 (join.rs:304-307)
    bool threw = false;
    try { left.inner_join(right, "nope", "id"); } catch (const Error &e) { threw = e.kind == Error::ColumnNotFound; }
    CHECK(threw);
    OptimizedDataFrame l2, r2;                               // :214-232: disjoint keys => 0 rows, NON-KEY columns only (join.rs:227-284)
    l2.add_column("id", Int64Column({1, 2, 3})); l2.add_column("a", Float64Column({1, 2, 3}));
    r2.add_column("id", Int64Column({4, 5, 6})); r2.add_column("b", Float64Column({4, 5, 6}));
    auto empty = l2.inner_join(r2, "id", "id");
    CHECK(empty.row_count() == 0 && empty.column_count() == 2 && !empty.contains_column("id"));
    threw = false;                                           // join.rs:98-104
    OptimizedDataFrame r3; r3.add_column("id", Float64Column({1.0, 2.0}));
    try { left.inner_join(r3, "id", "id"); } catch (const Error &e) { threw = e.kind == Error::ColumnTypeMismatch; }
    CHECK(threw);
}
// src/dataframe/pandas_compat/merge.rs:271-411 fixtures (keys A-D / B-E), with this API's fill rules
static void test_string_key_merge_vectors() {
    OptimizedDataFrame left, right;
    left.add_column("key", StringColumn({"A", "B", "C", "D"})); left.add_column("value1", Int64Column({1, 2, 3, 4}));
    right.add_column("key", StringColumn({"B", "C", "D", "E"})); right.add_column("value2", Int64Column({20, 30, 40, 50}));
    auto inner = left.inner_join(right, "key", "key");
    auto &k = std::get<StringColumn>(inner.column("key"));
    CHECK(inner.row_count() == 3 && k.get(0) == "B" && k.get(2) == "D");
    CHECK((std::get<Int64Column>(inner.column("value1")).data == std::vector<int64_t>{2, 3, 4}));
    CHECK((std::get<Int64Column>(inner.column("value2")).data == std::vector<int64_t>{20, 30, 40}));
    auto outer = left.outer_join(right, "key", "key");
    auto &ko = std::get<StringColumn>(outer.column("key"));
    CHECK(outer.row_count() == 5 && ko.get(0) == "A" && ko.get(4) == "E");
    auto joined = LazyFrame(left).join(right, "key", "key", JoinType::Left).execute();                    // lazy.rs:405-425
    CHECK(joined.row_count() == 4 && (std::get<Int64Column>(joined.column("value2")).data == std::vector<int64_t>{0, 20, 30, 40}));
}
// src/optimized/jit/simd.rs:451-506 and split_dataframe/aggregate.rs:21-217
static void test_whole_column_reductions() {
    OptimizedDataFrame df, dn;
    df.add_column("x", Float64Column({1, 2, 3, 4, 5, 6, 7, 8}));
    dn.add_column("n", Int64Column::with_nulls({1, 9, 5, 100}, {false, false, false, true}));
    CHECK(df.sum("x") == 36.0 && df.mean("x") == 4.5 && df.min("x") == 1.0 && df.max("x") == 8.0);
    CHECK(dn.sum("n") == 15.0 && dn.mean("n") == 5.0 && dn.min("n") == 1.0 && dn.max("n") == 9.0);
    bool threw = false;                                      // core.rs: columns of one frame share their length
    try { df.add_column("short", Int64Column({1, 2})); } catch (const Error &e) { threw = e.kind == Error::InconsistentRowCount; }
    CHECK(threw);
}
// numeric / null keys are stringified like the reference ("NULL", decimal i64, shortest f64; grouping.rs:69-98);
// lazy.rs:377-382 rejects the ops outside its five; error kinds before any device work
static void test_key_strings_and_errors() {
    OptimizedDataFrame df;
    df.add_column("k", Int64Column::with_nulls({-5, 7, -5, 0}, {false, false, false, true}));
    df.add_column("f", Float64Column({0.5, -0.0, 0.5, 1e21}));
    df.add_column("v", Float64Column({1.0, 2.0, 3.0, 4.0}));
    auto r = by_key(df.group_by({"k"}).sum("v"), "k", "v_sum");
    CHECK(r.size() == 3 && r["-5"] == 4.0 && r["7"] == 2.0 && r["NULL"] == 4.0);
    auto rf = by_key(df.group_by({"f"}).count("v"), "f", "v_count");
    CHECK(rf.size() == 3 && rf["0.5"] == 2.0 && rf["-0"] == 1.0 && rf["1000000000000000000000"] == 1.0);
    bool threw = false;
    try { LazyFrame(df).aggregate({"k"}, {{"v", AggregateOp::Median, "m"}}).execute(); } catch (const Error &e) { threw = e.kind == Error::OperationFailed; }
    CHECK(threw);
    threw = false;
    try { df.group_by({"nope"}); } catch (const Error &e) { threw = e.kind == Error::ColumnNotFound; }
    CHECK(threw);
    threw = false;
    try { df.group_by({"k"}).aggregate({{"nope", AggregateOp::Sum, "s"}}); } catch (const Error &e) { threw = e.kind == Error::ColumnNotFound; }
    CHECK(threw);
    threw = false;
    try { df.add_column("v", Float64Column({1, 2, 3, 4})); } catch (const Error &e) { threw = e.kind == Error::DuplicateColumnName; }
    CHECK(threw);
}

// tests/concurrency_test.rs:351-398: four threads group the same frame at once
static void test_concurrent_callers_share_one_frame() {
    OptimizedDataFrame df;
    std::vector<std::string> keys; std::vector<int64_t> vals;
    for (int i = 0; i < 40000; i++) { keys.push_back("g" + std::to_string(i % 4)); vals.push_back(i); }
    df.add_column("k", StringColumn(keys));
    df.add_column("v", Int64Column(vals));
    std::vector<std::map<std::string, double>> got(4);
    std::vector<std::thread> th;
    for (int t = 0; t < 4; t++) th.emplace_back([&, t] { got[t] = by_key(df.group_by({"k"}).sum("v"), "k", "v_sum"); });
    for (auto &x : th) x.join();
    for (int t = 0; t < 4; t++) {
        CHECK(got[t].size() == 4);
        for (int g = 0; g < 4; g++) CHECK(got[t]["g" + std::to_string(g)] == 10000.0 * g + 4.0 * (9999.0 * 10000.0 / 2.0));
    }
}

// Parity of the whole C++ path against the CPU oracle (oracle/pandrs_oracle.c, test infrastructure) on a
// seeded random frame: nullable i64 key, nullable f64 and i64 values, every aggregate incl. Median.
extern "C" {
int oracle_groupby_agg(const pandrs_hip_column *keys, int n_keys, int64_t n_rows, const pandrs_hip_column *vals, int n_vals,
                       const pandrs_hip_agg_spec *aggs, int n_aggs, int64_t *out_n_groups, uint64_t **out_keys,
                       uint8_t **out_key_null, double **out_aggs);
void oracle_free(void *p);
}
static void test_random_frame_matches_the_oracle() {
    const size_t n = 300000;
    uint64_t state = 0x243F6A8885A308D3ull;
    auto next = [&] { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; };
    std::vector<int64_t> k(n), vi(n);
    std::vector<double> vf(n);
    std::vector<bool> kn(n), fn(n), in(n);
    for (size_t i = 0; i < n; i++) {
        k[i] = (int64_t)((next() % 5000) * 0x9E3779B97F4A7C15ull);
        vf[i] = (double)(int64_t)(next() % 200001 - 100000) / 64.0;
        vi[i] = (int64_t)(next() % 2001) - 1000;
        kn[i] = next() % 1000 == 0; fn[i] = next() % 10 == 0; in[i] = next() % 7 == 0;
    }
    OptimizedDataFrame df;
    df.add_column("k", Int64Column::with_nulls(k, kn));
    df.add_column("f", Float64Column::with_nulls(vf, fn));
    df.add_column("i", Int64Column::with_nulls(vi, in));
    const std::vector<std::pair<std::string, AggregateOp>> req = {
        {"f", AggregateOp::Sum}, {"f", AggregateOp::Mean}, {"f", AggregateOp::Min}, {"f", AggregateOp::Max}, {"f", AggregateOp::Count},
        {"f", AggregateOp::Std}, {"f", AggregateOp::Median}, {"f", AggregateOp::First}, {"i", AggregateOp::Sum}, {"i", AggregateOp::Var},
        {"i", AggregateOp::Median}, {"i", AggregateOp::Last}};
    auto r = df.group_by({"k"}).agg(req);
    pandrs_hip_column keys[1] = {detail::view(df.column("k"))};
    pandrs_hip_column vals[2] = {detail::view(df.column("f")), detail::view(df.column("i"))};
    std::vector<pandrs_hip_agg_spec> specs;
    for (auto &a : req) specs.push_back({a.first == "f" ? 0 : 1, (int32_t)a.second});
    int64_t g = 0; uint64_t *ok = nullptr; uint8_t *on = nullptr; double *oa = nullptr;
    CHECK(oracle_groupby_agg(keys, 1, (int64_t)n, vals, 2, specs.data(), (int)specs.size(), &g, &ok, &on, &oa) == 0);
    CHECK((size_t)g == r.row_count() && r.column_count() == 1 + req.size());
    std::map<std::string, size_t> row_of;
    auto &kc = std::get<StringColumn>(r.column("k"));
    for (size_t j = 0; j < r.row_count(); j++) row_of[kc.get(j)] = j;
    int bad = 0;
    for (int64_t j = 0; j < g; j++) {
        auto it = row_of.find(on[j] ? "NULL" : std::to_string((int64_t)ok[j]));
        if (it == row_of.end()) { bad++; continue; }
        for (size_t a = 0; a < req.size(); a++) {
            const double want = oa[a * (size_t)g + (size_t)j];
            const double got = std::get<Float64Column>(r.columns[1 + a]).data[it->second];
            const bool exact = req[a].second != AggregateOp::Sum && req[a].second != AggregateOp::Mean && req[a].second != AggregateOp::Std &&
                               req[a].second != AggregateOp::Var;
            if (exact ? got != want : std::fabs(got - want) > 1e-9 * std::max(1.0, std::fabs(want))) bad++;
        }
    }
    CHECK(bad == 0);
    oracle_free(ok); oracle_free(on); oracle_free(oa);
}

// Resident columns: the reference's frames Arc-clone their immutable columns into every operator
// (src/optimized/dataframe/transformations.rs:524-577, :628-694); uploaded once (make_resident), a C2-shaped frame's
// second aggregate no longer stages anything and runs at the device-resident rate.  Results are the host path's.
static void test_resident_frame_skips_the_staging() {
    const size_t n = 20'000'000, g = 200'000;
    std::vector<int64_t> keys(n);
    std::vector<std::vector<double>> vals(4, std::vector<double>(n));
    uint64_t x = 88172645463325252ull;
    auto next = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (size_t i = 0; i < n; i++) {
        keys[i] = (int64_t)((next() % g) * 0x9E3779B97F4A7C15ull);
        for (auto &v : vals) v[i] = (double)(next() % 2000001) / 1000.0 - 1000.0;
    }
    OptimizedDataFrame df;
    df.add_column("k", Int64Column(keys));
    const char *names[4] = {"a", "b", "c", "d"};
    for (int c = 0; c < 4; c++) df.add_column(names[c], Float64Column(vals[c]));
    std::vector<GroupBy::Aggregation> req;
    for (int c = 0; c < 4; c++)
        for (auto op : {AggregateOp::Sum, AggregateOp::Mean, AggregateOp::Min, AggregateOp::Max})
            req.emplace_back(names[c], op, std::string(names[c]) + "_" + std::to_string((int)op));
    auto timings = [] { pandrs_hip_timings t{}; detail::check(pandrs_hip_get_timings(detail::context(), &t)); return t; };
    auto host = df.group_by({"k"}).aggregate(req);
    host = df.group_by({"k"}).aggregate(req);                 // steady state (arenas sized)
    const pandrs_hip_timings th = timings();
    CHECK(th.phase_ms[PANDRS_HIP_PHASE_STAGE_IN] > 0.0);
    OptimizedDataFrame copy = df;                             // a copy made BEFORE the upload stays on the host path
    df.make_resident();
    OptimizedDataFrame shared = df;                           // copies made after it share the device columns
    CHECK(df.is_resident() && shared.is_resident() && !copy.is_resident());
    int64_t rb = 0, rc = 0;
    detail::check(pandrs_hip_resident_bytes(detail::context(), &rb, &rc));
    CHECK(rc == 5 && rb >= (int64_t)(n * 40));
    auto dev = shared.group_by({"k"}).aggregate(req);
    dev = shared.group_by({"k"}).aggregate(req);
    const pandrs_hip_timings td = timings();
    CHECK(td.phase_ms[PANDRS_HIP_PHASE_STAGE_IN] == 0.0);
    const double compute_host = th.total_ms - th.phase_ms[PANDRS_HIP_PHASE_STAGE_IN];
    std::printf("    20 M rows x 5 columns: host call %.2f ms (staging %.2f), resident call %.2f ms\n", th.total_ms, th.phase_ms[PANDRS_HIP_PHASE_STAGE_IN], td.total_ms);
    CHECK(td.total_ms <= 1.2 * compute_host + 0.05);
    CHECK(td.total_ms * 3 < th.total_ms);
    CHECK(dev.row_count() == g && host.row_count() == g);
    for (auto &a : req) {                                     // same numbers as the host path (group order may differ)
        auto mh = by_key(host, "k", std::get<2>(a)), md = by_key(dev, "k", std::get<2>(a));
        size_t bad = 0;
        for (auto &kv : mh) {
            const double w = kv.second, got = md[kv.first];
            const bool exact = std::get<1>(a) == AggregateOp::Min || std::get<1>(a) == AggregateOp::Max;
            if (exact ? got != w : std::fabs(got - w) > 1e-9 * std::max(1.0, std::fabs(w))) bad++;
        }
        CHECK(bad == 0);
    }
    CHECK(std::fabs(shared.sum("a") - copy.sum("a")) <= 1e-9 * std::fabs(copy.sum("a")));     // K1 on the resident column
    // a join between two resident frames
    OptimizedDataFrame right;
    std::vector<int64_t> rk(1000);
    for (size_t i = 0; i < rk.size(); i++) rk[i] = (int64_t)(i * 0x9E3779B97F4A7C15ull);
    right.add_column("k", Int64Column(rk));
    right.add_column("w", Int64Column(std::vector<int64_t>(rk.size(), 7)));
    auto jh = copy.inner_join(right, "k", "k");
    right.make_resident();
    auto jd = shared.inner_join(right, "k", "k");
    const pandrs_hip_timings tj = timings();                  // the join's last library call: one gather of a resident column through the retained pairs
    CHECK(tj.phase_ms[PANDRS_HIP_PHASE_STAGE_IN] == 0.0 && tj.phase_ms[PANDRS_HIP_PHASE_GATHER] > 0.0);      // nothing staged over PCIe, no index pairs fetched
    CHECK(jh.row_count() == jd.row_count() && jd.row_count() > 0);
    CHECK(std::get<Int64Column>(jh.column("k")).data == std::get<Int64Column>(jd.column("k")).data);
    CHECK(std::get<Float64Column>(jh.column("a")).data == std::get<Float64Column>(jd.column("a")).data);
    // adding a column invalidates the device copies of THAT frame object only; dropping every sharer frees them
    df.add_column("e", Float64Column(vals[0]));
    CHECK(!df.is_resident() && shared.is_resident());
    shared = OptimizedDataFrame();
    right = OptimizedDataFrame();
    detail::check(pandrs_hip_resident_bytes(detail::context(), &rb, &rc));
    CHECK(rc == 0 && rb == 0);
}
// group_by builds a multi-index for >= 2 keys (grouping.rs:22-28 passes as_multi_index = true): aggregate() then returns NO key
// columns and a StringMultiIndex of the key tuples (aggregation.rs:812-853); group_by_with_options(.., false) and the lazy arm
// (lazy.rs:390-394, tests/optimized_groupby_test.rs:184) keep the key columns
static void test_multi_key_group_by_builds_a_multi_index() {
    OptimizedDataFrame df;
    df.add_column("category", StringColumn({"A", "A", "B", "B", "A"}));
    df.add_column("group", StringColumn({"X", "Y", "X", "Y", "X"}));
    df.add_column("values", Int64Column({10, 20, 30, 40, 50}));
    auto gb = df.group_by({"category", "group"});
    CHECK(gb.create_multi_index);
    auto r = gb.aggregate({{"values", AggregateOp::Sum, "sum"}, {"values", AggregateOp::Count, "n"}});
    CHECK(r.column_count() == 2 && !r.contains_column("category") && !r.contains_column("group"));
    CHECK(r.has_multi_index() && (r.multi_index_names == std::vector<std::string>{"category", "group"}) && r.multi_index.size() == 4 && r.row_count() == 4);
    std::map<std::vector<std::string>, double> sums, counts;
    for (size_t i = 0; i < r.multi_index.size(); i++) {
        sums[r.multi_index[i]] = std::get<Float64Column>(r.column("sum")).data[i];
        counts[r.multi_index[i]] = std::get<Float64Column>(r.column("n")).data[i];
    }
    CHECK((sums[{"A", "X"}] == 60.0) && (sums[{"A", "Y"}] == 20.0) && (sums[{"B", "X"}] == 30.0) && (sums[{"B", "Y"}] == 40.0) && (counts[{"A", "X"}] == 2.0));
    CHECK(!df.group_by({"category"}).create_multi_index);                                  // one key: never (grouping.rs:107)
    auto flat = df.group_by_with_options({"category", "group"}, false).aggregate({{"values", AggregateOp::Sum, "sum"}});
    CHECK(flat.column_count() == 3 && !flat.has_multi_index() && (flat.column_names == std::vector<std::string>{"category", "group", "sum"}));
    auto lazy = LazyFrame(df).aggregate({"category", "group"}, {{"values", AggregateOp::Sum, "sum"}}).execute();
    CHECK(lazy.column_count() == 3 && !lazy.has_multi_index());
    OptimizedDataFrame none;                                  // no rows: from_tuples refuses an empty tuple list (multi_index.rs:160)
    none.add_column("a", Int64Column(std::vector<int64_t>{})); none.add_column("b", Int64Column(std::vector<int64_t>{})); none.add_column("v", Float64Column(std::vector<double>{}));
    bool threw = false;
    try { none.group_by({"a", "b"}).aggregate({{"v", AggregateOp::Sum, "s"}}); } catch (const Error &e) { threw = e.kind == Error::Index; }
    CHECK(threw);
}
// LazyFrame: Join(Inner) immediately followed by Aggregate([g], [(v, Sum, alias)]) (lazy.rs:405-425 then :186) runs as the fused
// device operator (BASELINE config 5, pandrs_hip_join_groupby_sum); same frame as the two arms one after the other
static void test_lazy_join_then_aggregate_is_fused() {
    const int64_t n_left = 200000, n_right = 5000;
    std::vector<int64_t> lid(n_left), rid(n_right), g(n_right), other(n_left);
    std::vector<double> v(n_left);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (int64_t i = 0; i < n_right; i++) { rid[i] = i * 7 + 3; g[i] = (int64_t)(rnd() % 37) - 5; }
    for (int64_t i = 0; i < n_left; i++) { lid[i] = (int64_t)(rnd() % (uint64_t)(n_right + 500)) * 7 + 3; v[i] = (double)(rnd() % 1000) / 8.0; other[i] = i; }
    OptimizedDataFrame left, right;
    left.add_column("id", Int64Column(lid)); left.add_column("v", Float64Column(v)); left.add_column("g", Int64Column(other));
    right.add_column("id", Int64Column(rid)); right.add_column("g", Int64Column(g)); right.add_column("w", Int64Column(g));
    // `g` exists on both sides: the joined frame calls the right one "g_right" (join.rs:478-482); "w" keeps its name
    for (const char *gname : {"g_right", "w"}) {
        auto fused = LazyFrame(left).join(right, "id", "id", JoinType::Inner).aggregate({gname}, {{"v", AggregateOp::Sum, "total"}}).execute();
        auto two_step = left.inner_join(right, "id", "id").group_by({gname}).aggregate({{"v", AggregateOp::Sum, "total"}});
        CHECK((fused.column_names == std::vector<std::string>{gname, "total"}) && fused.row_count() == two_step.row_count() && fused.row_count() == 37);
        auto a = by_key(fused, gname, "total"), b = by_key(two_step, gname, "total");
        CHECK(a.size() == b.size());
        for (auto &kv : b) CHECK(a.count(kv.first) && std::fabs(a[kv.first] - kv.second) <= 1e-9 * std::fabs(kv.second));
    }
    // not the fused shape (grouping by a LEFT column; a null in g): the two arms run, same answers as calling them by hand
    auto by_left = LazyFrame(left).join(right, "id", "id", JoinType::Inner).aggregate({"v"}, {{"w", AggregateOp::Sum, "s"}}).execute();
    CHECK(by_left.row_count() == left.inner_join(right, "id", "id").group_by({"v"}).aggregate({{"w", AggregateOp::Sum, "s"}}).row_count());
    OptimizedDataFrame rn;
    std::vector<bool> nulls(n_right, false); nulls[0] = nulls[17] = true;
    rn.add_column("id", Int64Column(rid)); rn.add_column("w", Int64Column::with_nulls(g, nulls));
    auto with_null = LazyFrame(left).join(rn, "id", "id", JoinType::Inner).aggregate({"w"}, {{"v", AggregateOp::Sum, "total"}}).execute();
    auto want_null = left.inner_join(rn, "id", "id").group_by({"w"}).aggregate({{"v", AggregateOp::Sum, "total"}});
    auto a = by_key(with_null, "w", "total"), b = by_key(want_null, "w", "total");
    CHECK(a.size() == b.size() && !a.count("NULL"));          // a null g is the join's fill value 0 (join.rs:304-307), not a NULL group
    for (auto &kv : b) CHECK(a.count(kv.first) && std::fabs(a[kv.first] - kv.second) <= 1e-9 * std::fabs(kv.second));
}

int main() {
    int32_t n_dev = 0;
    if (pandrs_hip_init(nullptr) != PANDRS_HIP_OK || pandrs_hip_device_count(&n_dev) != PANDRS_HIP_OK || n_dev == 0) {
        std::fprintf(stderr, "no HIP device available: %s\n", pandrs_hip_last_error());
        return 1;
    }
    RUN(test_optimized_groupby_creation);
    RUN(test_optimized_groupby_aggregation);
    RUN(test_optimized_groupby_multiple_aggregations);
    RUN(test_optimized_groupby_multiple_keys);
    RUN(test_groupby_example_and_shortcuts);
    RUN(test_pandas_compat_fixture_values);
    RUN(test_optimized_joins);
    RUN(test_string_key_merge_vectors);
    RUN(test_whole_column_reductions);
    RUN(test_key_strings_and_errors);
    RUN(test_concurrent_callers_share_one_frame);
    RUN(test_random_frame_matches_the_oracle);
    RUN(test_resident_frame_skips_the_staging);
    RUN(test_multi_key_group_by_builds_a_multi_index);
    RUN(test_lazy_join_then_aggregate_is_fused);
    std::printf("%d tests, %d failed checks\n", g_run, g_failed);
    return g_failed ? 2 : 0;
}
