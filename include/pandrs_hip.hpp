// pandrs_hip.hpp — C++17 host-side mirror of the reference's API for the accelerated path, over the
// C ABI of include/pandrs_hip.h (header only; link libpandrs_hip.so).
//
// The reference is a Rust crate and this image has no Rust toolchain, so this header plays the part of
// the crate-side shim in a compiled language: same type and method names, argument meaning and error
// behaviour as
//   OptimizedDataFrame        src/optimized/split_dataframe/core.rs, group/grouping.rs:22-115,
//                             join.rs:32-73, aggregate.rs:21-217
//   Column / *Column          src/column/{int64,float64,string,boolean}_column.rs, core/column.rs:163-177
//   GroupBy, AggregateOp      group/types.rs:11-55, group/aggregation.rs:763-871, group/operations.rs:438-547
//   LazyFrame                 src/optimized/lazy.rs:98-170, :186-425
//   JoinType                  join.rs:11-20
//   Error                     src/core/error.rs:6 (Result<T, Error> becomes: returns T, throws Error)
// Everything numeric happens in the library (host pointers, PANDRS_HIP_MEM_HOST — the way the Rust shim
// of INTEGRATION.md calls it); this header only stringifies keys, names columns and assembles frames,
// exactly the host-side work the reference keeps.  tests/cpp/reference_like_tests.cpp replays the
// reference's own tests through it.
#pragma once
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <variant>
#include <vector>

#include "pandrs_hip.h"

namespace pandrs {

// ---- errors (src/core/error.rs) ---------------------------------------------------------------------
struct Error : std::runtime_error {
    enum Kind { ColumnNotFound, ColumnTypeMismatch, OperationFailed, Computation, InvalidInput, DuplicateColumnName, InconsistentRowCount, Empty, Type, BelowThreshold, Index };
    Kind kind;
    Error(Kind k, const std::string &m) : std::runtime_error(m), kind(k) {}
};

namespace detail {
// device failures map like src/gpu/mod.rs:206-210
inline void check(int32_t st) {
    if (st == PANDRS_HIP_OK) return;
    const std::string msg = pandrs_hip_last_error();
    switch (st) {
    case PANDRS_HIP_ERR_INVALID_ARGUMENT: throw Error(Error::InvalidInput, msg);
    case PANDRS_HIP_ERR_TYPE_MISMATCH: throw Error(Error::ColumnTypeMismatch, msg);
    case PANDRS_HIP_ERR_OPERATION_FAILED: throw Error(Error::OperationFailed, msg);
    case PANDRS_HIP_ERR_BELOW_THRESHOLD: throw Error(Error::BelowThreshold, msg);     // the shim keeps its CPU path (gpu.rs:30-32)
    default: throw Error(Error::Computation, msg);
    }
}
// process-wide context, created on first use (cf. get_gpu_manager, src/gpu/mod.rs:249-282)
inline pandrs_hip_ctx *context() {
    static std::once_flag once;
    static pandrs_hip_ctx *ctx = nullptr;
    std::call_once(once, [] {
        check(pandrs_hip_init(nullptr));
        check(pandrs_hip_ctx_create(0, &ctx));
    });
    return ctx;
}
// create_bitmask (src/core/column.rs:163-177): LSB first, 1 = null; empty when nothing is null
inline std::vector<uint8_t> create_bitmask(const std::vector<bool> &nulls) {
    bool any = false;
    for (bool b : nulls) any = any || b;
    if (!any) return {};
    std::vector<uint8_t> m((nulls.size() + 7) / 8, 0);
    for (size_t i = 0; i < nulls.size(); i++)
        if (nulls[i]) m[i >> 3] |= (uint8_t)(1u << (i & 7));
    return m;
}
inline bool bit_at(const std::vector<uint8_t> &m, size_t i) { return !m.empty() && ((m[i >> 3] >> (i & 7)) & 1); }
// f64::to_string(): shortest round-trip digits, never an exponent ("1", "0.1", "NaN", "inf", "-0")
inline std::string rust_f64_to_string(double v) {
    if (v != v) return "NaN";
    if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
    char buf[400];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    return std::string(buf, r.ptr);
}
}  // namespace detail

// ---- GLOBAL_STRING_POOL (src/column/string_pool.rs:28-53): equal string <=> equal code ----------------
class StringPool {
public:
    static StringPool &global() { static StringPool p; return p; }
    uint32_t get_or_insert(const std::string &s) {
        std::lock_guard<std::mutex> lock(mu_);
        auto it = codes_.find(s);
        if (it != codes_.end()) return it->second;
        const uint32_t c = (uint32_t)strings_.size();
        strings_.push_back(s);
        codes_.emplace(s, c);
        return c;
    }
    std::string get(uint32_t code) const {
        std::lock_guard<std::mutex> lock(mu_);
        return strings_.at(code);
    }
private:
    mutable std::mutex mu_;
    std::vector<std::string> strings_;
    std::unordered_map<std::string, uint32_t> codes_;
};

// ---- columns ----------------------------------------------------------------------------------------------
struct Int64Column {          // src/column/int64_column.rs:52-66
    std::vector<int64_t> data;
    std::vector<uint8_t> null_mask;
    Int64Column() = default;
    explicit Int64Column(std::vector<int64_t> d) : data(std::move(d)) {}
    static Int64Column with_nulls(std::vector<int64_t> d, const std::vector<bool> &nulls) {
        Int64Column c(std::move(d)); c.null_mask = detail::create_bitmask(nulls); return c;
    }
    size_t len() const { return data.size(); }
};
struct Float64Column {        // src/column/float64_column.rs:9-13
    std::vector<double> data;
    std::vector<uint8_t> null_mask;
    Float64Column() = default;
    explicit Float64Column(std::vector<double> d) : data(std::move(d)) {}
    static Float64Column with_nulls(std::vector<double> d, const std::vector<bool> &nulls) {
        Float64Column c(std::move(d)); c.null_mask = detail::create_bitmask(nulls); return c;
    }
    size_t len() const { return data.size(); }
};
struct StringColumn {         // src/column/string_column.rs:26-72 (GlobalPool mode): pool codes
    std::vector<uint32_t> indices;
    std::vector<uint8_t> null_mask;
    StringColumn() = default;
    explicit StringColumn(const std::vector<std::string> &values) {
        indices.reserve(values.size());
        for (auto &s : values) indices.push_back(StringPool::global().get_or_insert(s));
    }
    static StringColumn with_nulls(const std::vector<std::string> &values, const std::vector<bool> &nulls) {
        StringColumn c(values); c.null_mask = detail::create_bitmask(nulls); return c;
    }
    size_t len() const { return indices.size(); }
    std::string get(size_t i) const { return StringPool::global().get(indices[i]); }
};
struct BooleanColumn {        // src/column/boolean_column.rs:10-15: LSB-first packed bits
    std::vector<uint8_t> bits;
    size_t length = 0;
    std::vector<uint8_t> null_mask;
    BooleanColumn() = default;
    explicit BooleanColumn(const std::vector<bool> &values) : bits((values.size() + 7) / 8, 0), length(values.size()) {
        for (size_t i = 0; i < values.size(); i++)
            if (values[i]) bits[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
    size_t len() const { return length; }
    bool get(size_t i) const { return (bits[i >> 3] >> (i & 7)) & 1; }
};
using Column = std::variant<Int64Column, Float64Column, StringColumn, BooleanColumn>;

namespace detail {
inline size_t col_len(const Column &c) { return std::visit([](auto &x) { return x.len(); }, c); }
inline int32_t col_dtype(const Column &c) { return (int32_t)c.index(); }   // variant order == pandrs_hip_dtype order
inline pandrs_hip_column view(const Column &c) {
    pandrs_hip_column v{};
    v.dtype = col_dtype(c);
    std::visit([&](auto &x) {
        using T = std::decay_t<decltype(x)>;
        if constexpr (std::is_same_v<T, StringColumn>) v.data = x.indices.data();
        else if constexpr (std::is_same_v<T, BooleanColumn>) v.data = x.bits.data();
        else v.data = x.data.data();
        v.null_mask = x.null_mask.empty() ? nullptr : x.null_mask.data();
    }, c);
    return v;
}
// Columns uploaded once (pandrs_hip_column_upload) and shared by every copy of the frame: the stand-in for the Rust
// shim's ResidentCache over the reference's immutable Arc<[T]> columns (src/column/int64_column.rs:10), which the
// public frame's operators Arc-clone on every call (src/optimized/dataframe/transformations.rs:524-577, :628-694).
struct ResidentSet {
    std::vector<pandrs_hip_column> cols;
    ResidentSet() = default;
    ResidentSet(const ResidentSet &) = delete;
    ResidentSet &operator=(const ResidentSet &) = delete;
    ~ResidentSet() { for (auto &c : cols) (void)pandrs_hip_column_release(context(), &c); }
};
// group-key cell -> the string the reference's result frame holds (grouping.rs:69-98)
inline std::string key_string(int32_t dtype, uint64_t cell, bool is_null, const char *null_string = "NULL") {
    if (is_null) return null_string;
    switch (dtype) {
    case PANDRS_HIP_I64: return std::to_string((int64_t)cell);
    case PANDRS_HIP_F64: { double d; std::memcpy(&d, &cell, 8); return rust_f64_to_string(d); }
    case PANDRS_HIP_U32CODE: return StringPool::global().get((uint32_t)cell);
    default: return cell ? "true" : "false";
    }
}
}  // namespace detail

enum class AggregateOp { Sum = 0, Mean, Min, Max, Count, Std, Var, Median, First, Last, Custom,   // types.rs:11-34
                         Nunique };   // + the legacy AggFunc::Nunique (src/dataframe/groupby.rs:41)
enum class JoinType { Inner = 0, Left, Right, Outer };                                              // join.rs:11-20

class GroupBy;

// ---- OptimizedDataFrame ------------------------------------------------------------------------------------
class OptimizedDataFrame {
public:
    std::vector<Column> columns;
    std::vector<std::string> column_names;
    std::unordered_map<std::string, size_t> column_indices;
    // the StringMultiIndex a multi-key group_by(..).aggregate(..) sets instead of key columns (aggregation.rs:812-853,
    // split_dataframe/index.rs:94-111): one tuple of key strings per row, the level names = the grouping columns
    std::vector<std::vector<std::string>> multi_index;
    std::vector<std::string> multi_index_names;
    bool has_multi_index() const { return !multi_index_names.empty(); }

    OptimizedDataFrame &add_column(const std::string &name, Column column) {
        if (column_indices.count(name)) throw Error(Error::DuplicateColumnName, name);
        if (!columns.empty() && detail::col_len(column) != row_count_)
            throw Error(Error::InconsistentRowCount, "expected " + std::to_string(row_count_) + " rows, found " + std::to_string(detail::col_len(column)));
        row_count_ = detail::col_len(column);
        column_indices[name] = columns.size();
        column_names.push_back(name);
        columns.push_back(std::move(column));
        resident_.reset();                       // the device copies describe the frame as it was
        return *this;
    }
    // Uploads every column to HBM once; group_by(..).aggregate(..), groups(), joins, gathers and the whole-column
    // reductions of this frame AND its copies then read the device copies (PANDRS_HIP_MEM_DEVICE) instead of staging
    // the host vectors on every call.  The columns must not be modified afterwards (the reference's are immutable).
    OptimizedDataFrame &make_resident() {
        auto rs = std::make_shared<detail::ResidentSet>();
        rs->cols.reserve(columns.size());
        for (auto &c : columns) {
            pandrs_hip_column host = detail::view(c), dev{};
            detail::check(pandrs_hip_column_upload(detail::context(), &host, (int64_t)detail::col_len(c), &dev));
            rs->cols.push_back(dev);
        }
        resident_ = std::move(rs);
        return *this;
    }
    bool is_resident() const { return resident_ != nullptr; }
    // the column's view for a library call, and the memory space it lives in
    pandrs_hip_column view_of(const std::string &name) const {
        auto it = column_indices.find(name);
        if (it == column_indices.end()) throw Error(Error::ColumnNotFound, name);
        return resident_ ? resident_->cols[it->second] : detail::view(columns[it->second]);
    }
    int32_t mem_space() const { return resident_ ? PANDRS_HIP_MEM_DEVICE : PANDRS_HIP_MEM_HOST; }
    const Column &column(const std::string &name) const {
        auto it = column_indices.find(name);
        if (it == column_indices.end()) throw Error(Error::ColumnNotFound, name);
        return columns[it->second];
    }
    bool contains_column(const std::string &name) const { return column_indices.count(name) != 0; }
    size_t row_count() const { return row_count_; }
    size_t column_count() const { return columns.size(); }

    GroupBy group_by(const std::vector<std::string> &cols) const;                      // grouping.rs:22-28: as_multi_index = true
    GroupBy group_by_with_options(const std::vector<std::string> &cols, bool as_multi_index) const;   // grouping.rs:38-115
    std::map<std::string, OptimizedDataFrame> par_groupby(const std::vector<std::string> &cols) const;   // grouping.rs:124-331

    OptimizedDataFrame inner_join(const OptimizedDataFrame &o, const std::string &l, const std::string &r) const { return join_impl(o, l, r, JoinType::Inner); }
    OptimizedDataFrame left_join(const OptimizedDataFrame &o, const std::string &l, const std::string &r) const { return join_impl(o, l, r, JoinType::Left); }
    OptimizedDataFrame right_join(const OptimizedDataFrame &o, const std::string &l, const std::string &r) const { return join_impl(o, l, r, JoinType::Right); }
    OptimizedDataFrame outer_join(const OptimizedDataFrame &o, const std::string &l, const std::string &r) const { return join_impl(o, l, r, JoinType::Outer); }

    // whole-column reductions (split_dataframe/aggregate.rs:21-215): the non-null values as f64, sum 0.0 when
    // there is none, mean / min / max Err(Error::Empty) then; min / max fold with f64::min / f64::max from +-inf
    double sum(const std::string &name) const { return stats(name).sum_f64; }
    double mean(const std::string &name) const { auto s = non_empty(name); return s.sum_f64 / (double)s.count; }
    double min(const std::string &name) const { return non_empty(name).min; }
    double max(const std::string &name) const { return non_empty(name).max; }

    // data_ops.rs:124-209: row gather of every column; nulls become 0 / 0.0 / "" / false
    OptimizedDataFrame filter_by_indices(const std::vector<int64_t> &indices) const {
        std::vector<int64_t> idx;
        for (int64_t i : indices) if (i >= 0 && (size_t)i < row_count_) idx.push_back(i);
        OptimizedDataFrame out;
        for (size_t c = 0; c < columns.size(); c++) out.add_column(column_names[c], gather(columns[c], idx));
        return out;
    }

private:
    size_t row_count_ = 0;
    std::shared_ptr<detail::ResidentSet> resident_;

    pandrs_hip_column_stats stats(const std::string &name) const {
        const Column &c = column(name);
        if (c.index() > 1) throw Error(Error::Type, "Column '" + name + "' is not a numeric type");      // aggregate.rs:57
        pandrs_hip_column v = view_of(name);
        pandrs_hip_column_stats st{};
        detail::check(pandrs_hip_reduce_stats(detail::context(), mem_space(), &v, (int64_t)detail::col_len(c), &st));
        return st;
    }
    pandrs_hip_column_stats non_empty(const std::string &name) const {
        auto st = stats(name);
        if (st.count == 0) throw Error(Error::Empty, "Column '" + name + "' is empty");                    // aggregate.rs:87
        return st;
    }
    // join_impl's / filter_by_indices' per-column gather on the device (join.rs:296-357, :475-552)
    static Column gather(const Column &src, const std::vector<int64_t> &idx) {
        const int64_t n = (int64_t)idx.size(), n_src = (int64_t)detail::col_len(src);
        pandrs_hip_column v = detail::view(src);
        auto call = [&](uint64_t fill, void *out) {
            detail::check(pandrs_hip_gather_column(detail::context(), PANDRS_HIP_MEM_HOST, &v, n_src, idx.data(), n, fill, out));
        };
        switch (src.index()) {
        case 0: { Int64Column o; o.data.resize(n); call(0, o.data.data()); return o; }
        case 1: { Float64Column o; o.data.resize(n); call(0, o.data.data()); return o; }
        case 2: { StringColumn o; o.indices.resize(n); call(StringPool::global().get_or_insert(""), o.indices.data()); return o; }
        default: {
            std::vector<uint8_t> bytes(n);
            call(0, bytes.data());
            std::vector<bool> b(n);
            for (int64_t i = 0; i < n; i++) b[i] = bytes[i] != 0;
            return BooleanColumn(b);
        }
        }
    }
    // join.rs:76-555
    OptimizedDataFrame join_impl(const OptimizedDataFrame &other, const std::string &left_on, const std::string &right_on, JoinType how) const {
        if (!contains_column(left_on)) throw Error(Error::ColumnNotFound, left_on);              // :84-87
        if (!other.contains_column(right_on)) throw Error(Error::ColumnNotFound, right_on);      // :89-92
        const Column &lc = column(left_on), &rc = other.column(right_on);
        const bool dev = is_resident() && other.is_resident();      // one memory space per call
        pandrs_hip_column lv = dev ? view_of(left_on) : detail::view(lc), rv = dev ? other.view_of(right_on) : detail::view(rc);
        int64_t n = 0;      // a key-type mismatch surfaces as ColumnTypeMismatch from the library (:98-104)
        detail::check(pandrs_hip_join_indices(detail::context(), dev ? PANDRS_HIP_MEM_DEVICE : PANDRS_HIP_MEM_HOST, &lv, (int64_t)detail::col_len(lc), &rv,
                                              (int64_t)detail::col_len(rc), (int32_t)how, &n));
        // the pairs stay in HBM (the context retains them); every output column is ONE gather through them and one
        // transfer of the finished column (pandrs_hip_join_gather) — not 16 bytes of indices per output row
        const int32_t space = dev ? PANDRS_HIP_MEM_DEVICE : PANDRS_HIP_MEM_HOST;
        OptimizedDataFrame result;
        if (n == 0) {       // empty result: NON-KEY columns only, suffix decided against the LEFT frame (:227-284)
            for (auto &name : column_names) if (name != left_on) result.add_column(name, gather(column(name), {}));
            for (auto &name : other.column_names)
                if (name != right_on) result.add_column(contains_column(name) ? name + "_right" : name, gather(other.column(name), {}));
            return result;
        }
        auto joined = [&](const OptimizedDataFrame &f, const std::string &name, int side, const pandrs_hip_column *key_right, int64_t n_right) -> Column {
            const Column &src = f.column(name);
            pandrs_hip_column v = dev ? f.view_of(name) : detail::view(src);
            const int64_t n_src = (int64_t)detail::col_len(src);
            auto call = [&](uint64_t fill, void *out) {
                if (key_right) detail::check(pandrs_hip_join_gather_key(detail::context(), space, &v, n_src, key_right, n_right, fill, PANDRS_HIP_MEM_HOST, out));
                else detail::check(pandrs_hip_join_gather(detail::context(), space, &v, n_src, side, fill, PANDRS_HIP_MEM_HOST, out));
            };
            switch (src.index()) {
            case 0: { Int64Column o; o.data.resize(n); call(0, o.data.data()); return o; }
            case 1: { Float64Column o; o.data.resize(n); call(0, o.data.data()); return o; }
            case 2: { StringColumn o; o.indices.resize(n); call(StringPool::global().get_or_insert(""), o.indices.data()); return o; }
            default: {
                std::vector<uint8_t> bytes(n);
                call(0, bytes.data());
                std::vector<bool> b(n);
                for (int64_t i = 0; i < n; i++) b[i] = bytes[i] != 0;
                return BooleanColumn(b);
            }
            }
        };
        for (auto &name : column_names) if (name != left_on) result.add_column(name, joined(*this, name, 0, nullptr, 0));     // :290-361
        result.add_column(left_on, joined(*this, left_on, 0, &rv, (int64_t)detail::col_len(rc)));       // key column: the left value, else the right one (:364-472)
        for (auto &name : other.column_names)                                                                       // :475-552
            if (name != right_on) result.add_column(result.contains_column(name) ? name + "_right" : name, joined(other, name, 1, nullptr, 0));
        return result;
    }
    friend class GroupBy;
};

// ---- GroupBy (group/types.rs:46-55) ------------------------------------------------------------------------
class GroupBy {
public:
    using Aggregation = std::tuple<std::string, AggregateOp, std::string>;      // (column, op, alias)
    const OptimizedDataFrame &df;
    std::vector<std::string> group_by_columns;
    bool create_multi_index = false;       // types.rs:54; set by group_by for >= 2 keys (grouping.rs:27, :107)

    GroupBy(const OptimizedDataFrame &d, std::vector<std::string> cols, bool multi_index = false)
        : df(d), group_by_columns(std::move(cols)), create_multi_index(multi_index) {}

    // aggregation.rs:763-871: key column(s) as strings, then one Float64 column per alias in request order
    OptimizedDataFrame aggregate(const std::vector<Aggregation> &aggregations) const {
        for (auto &a : aggregations)
            if (!df.contains_column(std::get<0>(a))) throw Error(Error::ColumnNotFound, std::get<0>(a));       // :770-774
        std::vector<pandrs_hip_column> keys, vals;
        std::vector<std::string> val_names;
        for (auto &k : group_by_columns) keys.push_back(df.view_of(k));
        std::vector<pandrs_hip_agg_spec> specs;
        for (auto &a : aggregations) {
            size_t vi = 0;
            while (vi < val_names.size() && val_names[vi] != std::get<0>(a)) vi++;
            if (vi == val_names.size()) { val_names.push_back(std::get<0>(a)); vals.push_back(df.view_of(std::get<0>(a))); }
            specs.push_back(pandrs_hip_agg_spec{(int32_t)vi, (int32_t)std::get<1>(a)});
        }
        int64_t g = 0;
        detail::check(pandrs_hip_groupby_agg(detail::context(), df.mem_space(), keys.data(), (int32_t)keys.size(), (int64_t)df.row_count(),
                                             vals.data(), (int32_t)vals.size(), specs.data(), (int32_t)specs.size(), &g));
        const size_t nk = keys.size(), na = specs.size();
        std::vector<std::vector<uint64_t>> kc(nk, std::vector<uint64_t>(g));
        std::vector<std::vector<uint8_t>> kn(nk, std::vector<uint8_t>(g));
        std::vector<std::vector<double>> oa(na, std::vector<double>(g));
        std::vector<uint64_t *> pk; std::vector<uint8_t *> pn; std::vector<double *> pa;
        for (auto &v : kc) pk.push_back(v.data());
        for (auto &v : kn) pn.push_back(v.data());
        for (auto &v : oa) pa.push_back(v.data());
        detail::check(pandrs_hip_groupby_fetch(detail::context(), PANDRS_HIP_MEM_HOST, pk.data(), pn.data(), pa.data()));
        OptimizedDataFrame result;
        std::vector<std::vector<std::string>> key_strings(nk, std::vector<std::string>(g));
        for (size_t k = 0; k < nk; k++)
            for (int64_t i = 0; i < g; i++) key_strings[k][i] = detail::key_string(keys[k].dtype, kc[k][i], kn[k][i] != 0);
        if (create_multi_index && nk > 1) {
            // :812-853: the key tuples become a StringMultiIndex (from_tuples refuses an empty list, multi_index.rs:160),
            // the result holds the aggregate columns only
            if (g == 0) throw Error(Error::Index, "Empty tuple list was passed");
            result.multi_index.assign((size_t)g, std::vector<std::string>(nk));
            for (int64_t i = 0; i < g; i++)
                for (size_t k = 0; k < nk; k++) result.multi_index[i][k] = key_strings[k][i];
            result.multi_index_names = group_by_columns;
        } else {
            for (size_t k = 0; k < nk; k++) result.add_column(group_by_columns[k], StringColumn(key_strings[k]));    // :856-860
        }
        for (size_t a = 0; a < na; a++) result.add_column(std::get<2>(aggregations[a]), Float64Column(oa[a]));  // :863-867
        return result;
    }
    // operations.rs:498-521: aliases "{col}_{op}"
    OptimizedDataFrame agg(const std::vector<std::pair<std::string, AggregateOp>> &aggs) const {
        std::vector<Aggregation> v;
        for (auto &a : aggs) v.emplace_back(a.first, a.second, a.first + "_" + op_name(a.second));
        return aggregate(v);
    }
    OptimizedDataFrame sum(const std::string &c) const { return agg({{c, AggregateOp::Sum}}); }
    OptimizedDataFrame mean(const std::string &c) const { return agg({{c, AggregateOp::Mean}}); }
    OptimizedDataFrame min(const std::string &c) const { return agg({{c, AggregateOp::Min}}); }
    OptimizedDataFrame max(const std::string &c) const { return agg({{c, AggregateOp::Max}}); }
    OptimizedDataFrame count(const std::string &c) const { return agg({{c, AggregateOp::Count}}); }
    OptimizedDataFrame std(const std::string &c) const { return agg({{c, AggregateOp::Std}}); }
    OptimizedDataFrame var(const std::string &c) const { return agg({{c, AggregateOp::Var}}); }
    OptimizedDataFrame median(const std::string &c) const { return agg({{c, AggregateOp::Median}}); }
    OptimizedDataFrame first(const std::string &c) const { return agg({{c, AggregateOp::First}}); }
    OptimizedDataFrame last(const std::string &c) const { return agg({{c, AggregateOp::Last}}); }
    OptimizedDataFrame nunique(const std::string &c) const { return agg({{c, AggregateOp::Nunique}}); }   // legacy GroupBy::nunique (src/dataframe/groupby.rs:386-393)

    // the pub field `groups` (types.rs:52): HashMap<Vec<String>, Vec<usize>>, every list ascending
    std::map<std::vector<std::string>, std::vector<size_t>> groups(const char *null_string = "NULL") const {
        std::vector<pandrs_hip_column> keys;
        for (auto &k : group_by_columns) keys.push_back(df.view_of(k));
        int64_t g = 0;
        const int64_t n = (int64_t)df.row_count();
        detail::check(pandrs_hip_groupby_indices(detail::context(), df.mem_space(), keys.data(), (int32_t)keys.size(), n, &g));
        const size_t nk = keys.size();
        std::vector<std::vector<uint64_t>> kc(nk, std::vector<uint64_t>(g));
        std::vector<std::vector<uint8_t>> kn(nk, std::vector<uint8_t>(g));
        std::vector<uint64_t *> pk; std::vector<uint8_t *> pn;
        for (auto &v : kc) pk.push_back(v.data());
        for (auto &v : kn) pn.push_back(v.data());
        std::vector<int64_t> off(g + 1), rows(n);
        detail::check(pandrs_hip_groupby_indices_fetch(detail::context(), PANDRS_HIP_MEM_HOST, pk.data(), pn.data(), off.data(), rows.data()));
        std::map<std::vector<std::string>, std::vector<size_t>> out;
        for (int64_t i = 0; i < g; i++) {
            std::vector<std::string> key;
            for (size_t k = 0; k < nk; k++) key.push_back(detail::key_string(keys[k].dtype, kc[k][i], kn[k][i] != 0, null_string));
            auto &v = out[key];
            v.insert(v.end(), rows.begin() + off[i], rows.begin() + off[i + 1]);
        }
        return out;
    }
    // CustomAggregation / aggregate_custom (types.rs:58-67, aggregation.rs:391-497): host closure over the
    // group's non-null values (Int64 cast to f64); the groups come from the device
    OptimizedDataFrame custom(const std::string &column, const std::string &result_name, const std::function<double(const std::vector<double> &)> &fn) const {
        if (!df.contains_column(column)) throw Error(Error::ColumnNotFound, column);
        const Column &c = df.column(column);
        if (c.index() > 1) throw Error(Error::OperationFailed, "column '" + column + "' is not numeric");
        auto gs = groups();
        OptimizedDataFrame result;
        for (size_t k = 0; k < group_by_columns.size(); k++) {
            std::vector<std::string> strs;
            for (auto &kv : gs) strs.push_back(kv.first[k]);
            result.add_column(group_by_columns[k], StringColumn(strs));
        }
        std::vector<double> vals;
        for (auto &kv : gs) {
            std::vector<double> v;
            for (size_t r : kv.second) {
                if (c.index() == 0) { auto &x = std::get<Int64Column>(c); if (!detail::bit_at(x.null_mask, r)) v.push_back((double)x.data[r]); }
                else { auto &x = std::get<Float64Column>(c); if (!detail::bit_at(x.null_mask, r)) v.push_back(x.data[r]); }
            }
            vals.push_back(fn(v));
        }
        result.add_column(result_name, Float64Column(vals));
        return result;
    }

    // operations.rs:51-74: keep the rows of the groups whose sub-frame passes filter_fn
    OptimizedDataFrame filter(const std::function<bool(const OptimizedDataFrame &)> &filter_fn) const {
        std::vector<int64_t> keep;
        for (auto &kv : groups()) {
            std::vector<int64_t> rows(kv.second.begin(), kv.second.end());
            if (filter_fn(df.filter_by_indices(rows))) keep.insert(keep.end(), rows.begin(), rows.end());
        }
        return df.filter_by_indices(keep);
    }
    // operations.rs:132-276: transform_fn(group sub-frame) -> frame for every group, results concatenated
    // column by column after the first result's schema (columns matched by position and type)
    OptimizedDataFrame transform(const std::function<OptimizedDataFrame(const OptimizedDataFrame &)> &transform_fn) const {
        std::vector<OptimizedDataFrame> outs;
        for (auto &kv : groups()) outs.push_back(transform_fn(df.filter_by_indices(std::vector<int64_t>(kv.second.begin(), kv.second.end()))));
        OptimizedDataFrame result;
        if (outs.empty()) return result;
        const OptimizedDataFrame &tmpl = outs[0];
        for (size_t ci = 0; ci < tmpl.columns.size(); ci++) {
            Column acc = tmpl.columns[ci];
            std::visit([&](auto &a) {
                using T = std::decay_t<decltype(a)>;
                std::vector<bool> nulls;
                auto push_nulls = [&](const T &x) { for (size_t i = 0; i < x.len(); i++) nulls.push_back(detail::bit_at(x.null_mask, i)); };
                push_nulls(a);
                for (size_t d = 1; d < outs.size(); d++) {
                    if (ci >= outs[d].columns.size() || !std::holds_alternative<T>(outs[d].columns[ci])) continue;
                    const T &x = std::get<T>(outs[d].columns[ci]);
                    if constexpr (std::is_same_v<T, StringColumn>) a.indices.insert(a.indices.end(), x.indices.begin(), x.indices.end());
                    else if constexpr (std::is_same_v<T, BooleanColumn>) {
                        std::vector<bool> v(a.length + x.length);
                        for (size_t i = 0; i < a.length; i++) v[i] = a.get(i);
                        for (size_t i = 0; i < x.length; i++) v[a.length + i] = x.get(i);
                        a = BooleanColumn(v);
                    } else a.data.insert(a.data.end(), x.data.begin(), x.data.end());
                    push_nulls(x);
                }
                a.null_mask = detail::create_bitmask(nulls);
            }, acc);
            result.add_column(tmpl.column_names[ci], std::move(acc));
        }
        return result;
    }

    static std::string op_name(AggregateOp op) {
        static const char *names[] = {"sum", "mean", "min", "max", "count", "std", "var", "median", "first", "last", "custom", "nunique"};
        return names[(int)op];
    }
};

inline GroupBy OptimizedDataFrame::group_by(const std::vector<std::string> &cols) const { return group_by_with_options(cols, true); }
inline GroupBy OptimizedDataFrame::group_by_with_options(const std::vector<std::string> &cols, bool as_multi_index) const {
    for (auto &c : cols) if (!contains_column(c)) throw Error(Error::ColumnNotFound, c);       // grouping.rs:53-57
    return GroupBy(*this, cols, as_multi_index && cols.size() > 1);                              // grouping.rs:107
}
inline std::map<std::string, OptimizedDataFrame> OptimizedDataFrame::par_groupby(const std::vector<std::string> &cols) const {
    for (auto &c : cols) if (!contains_column(c)) throw Error(Error::ColumnNotFound, c);
    std::map<std::string, std::vector<int64_t>> merged;           // parts joined with "_" (:186), a null part is "NA" (:158)
    for (auto &kv : GroupBy(*this, cols).groups("NA")) {
        std::string name;
        for (size_t i = 0; i < kv.first.size(); i++) name += (i ? "_" : "") + kv.first[i];
        auto &v = merged[name];
        v.insert(v.end(), kv.second.begin(), kv.second.end());
    }
    std::map<std::string, OptimizedDataFrame> out;
    for (auto &kv : merged) {
        std::vector<int64_t> rows = kv.second;
        std::sort(rows.begin(), rows.end());
        out.emplace(kv.first, filter_by_indices(rows));
    }
    return out;
}

// ---- LazyFrame (lazy.rs:98-170): the two operations on the accelerated path ---------------------------------
class LazyFrame {
public:
    explicit LazyFrame(OptimizedDataFrame df) : source_(std::move(df)) {}
    LazyFrame &aggregate(std::vector<std::string> group_by, std::vector<GroupBy::Aggregation> aggregations) {
        ops_.push_back(Op{true, std::move(group_by), std::move(aggregations), nullptr, "", "", JoinType::Inner});
        return *this;
    }
    LazyFrame &join(const OptimizedDataFrame &right, const std::string &left_on, const std::string &right_on, JoinType how) {
        ops_.push_back(Op{false, {}, {}, std::make_shared<OptimizedDataFrame>(right), left_on, right_on, how});
        return *this;
    }
    OptimizedDataFrame execute() const {
        OptimizedDataFrame df = source_;
        for (size_t oi = 0; oi < ops_.size(); oi++) {
            const Op &op = ops_[oi];
            if (op.is_aggregate) {
                for (auto &a : op.aggregations) {                   // lazy.rs:377-382: only these five ops
                    const AggregateOp o = std::get<1>(a);
                    if (o != AggregateOp::Sum && o != AggregateOp::Mean && o != AggregateOp::Min && o != AggregateOp::Max && o != AggregateOp::Count)
                        throw Error(Error::OperationFailed, "Aggregation operation " + GroupBy::op_name(o) + " is not supported in LazyFrame");
                }
                // the arm builds its frame inline and never a multi-index: key columns always (lazy.rs:390-394;
                // tests/optimized_groupby_test.rs:184 asserts 3 columns for two keys)
                df = df.group_by_with_options(op.group_by, false).aggregate(op.aggregations);
            } else {
                // Join(Inner) immediately followed by Aggregate([g], [(v, Sum, alias)]) with v a left column and g a right
                // column (lazy.rs:405-425 then :186) = BASELINE config 5: ONE fused device operator, no joined rows
                if (op.how == JoinType::Inner && oi + 1 < ops_.size() && ops_[oi + 1].is_aggregate) {
                    OptimizedDataFrame fused;
                    if (fused_join_groupby_sum(df, *op.right, op.left_on, op.right_on, ops_[oi + 1], fused)) {
                        df = std::move(fused);
                        oi++;
                        continue;
                    }
                }
                switch (op.how) {                                   // lazy.rs:405-425
                case JoinType::Inner: df = df.inner_join(*op.right, op.left_on, op.right_on); break;
                case JoinType::Left: df = df.left_join(*op.right, op.left_on, op.right_on); break;
                case JoinType::Right: df = df.right_join(*op.right, op.left_on, op.right_on); break;
                default: df = df.outer_join(*op.right, op.left_on, op.right_on);
                }
            }
        }
        return df;
    }
private:
    struct Op {
        bool is_aggregate;
        std::vector<std::string> group_by;
        std::vector<GroupBy::Aggregation> aggregations;
        std::shared_ptr<OptimizedDataFrame> right;
        std::string left_on, right_on;
        JoinType how;
    };
    // the shape test of hip_shim.rs `lazy_join_groupby_sum_hip` (same conditions, same result frame)
    static bool fused_join_groupby_sum(const OptimizedDataFrame &left, const OptimizedDataFrame &right, const std::string &left_on,
                                       const std::string &right_on, const Op &agg, OptimizedDataFrame &out) {
        if (agg.group_by.size() != 1 || agg.aggregations.size() != 1 || std::get<1>(agg.aggregations[0]) != AggregateOp::Sum) return false;
        const std::string &group_col = agg.group_by[0], &value_col = std::get<0>(agg.aggregations[0]), &alias = std::get<2>(agg.aggregations[0]);
        if (!left.contains_column(left_on) || !right.contains_column(right_on)) return false;     // the join arm throws ColumnNotFound
        if (value_col == left_on || !left.contains_column(value_col) || left.contains_column(group_col)) return false;
        std::string right_name;
        const std::string suffix = "_right";
        if (group_col.size() > suffix.size() && group_col.compare(group_col.size() - suffix.size(), suffix.size(), suffix) == 0 &&
            left.contains_column(group_col.substr(0, group_col.size() - suffix.size())) && right.contains_column(group_col.substr(0, group_col.size() - suffix.size())))
            right_name = group_col.substr(0, group_col.size() - suffix.size());               // join.rs:478-482
        else if (right.contains_column(group_col)) right_name = group_col;
        else return false;
        if (right_name == right_on) return false;
        const Column &lk = left.column(left_on), &lv = left.column(value_col), &rk = right.column(right_on), &rg = right.column(right_name);
        if (lk.index() != rk.index() || lv.index() > 1 || rg.index() == 3) return false;
        if (std::visit([](auto &x) { return !x.null_mask.empty(); }, rg)) return false;       // a null g would surface as the join's fill value (join.rs:304-307)
        const bool dev = left.is_resident() && right.is_resident();
        pandrs_hip_column a = dev ? left.view_of(left_on) : detail::view(lk), b = dev ? left.view_of(value_col) : detail::view(lv),
                          c = dev ? right.view_of(right_on) : detail::view(rk), d = dev ? right.view_of(right_name) : detail::view(rg);
        int64_t g = 0;
        detail::check(pandrs_hip_join_groupby_sum(detail::context(), dev ? PANDRS_HIP_MEM_DEVICE : PANDRS_HIP_MEM_HOST, &a, &b, (int64_t)left.row_count(), &c, &d,
                                                  (int64_t)right.row_count(), &g));
        std::vector<uint64_t> cells(g); std::vector<uint8_t> nulls(g); std::vector<double> sums(g);
        uint64_t *pk[1] = {cells.data()}; uint8_t *pn[1] = {nulls.data()}; double *pa[1] = {sums.data()};
        detail::check(pandrs_hip_groupby_fetch(detail::context(), PANDRS_HIP_MEM_HOST, pk, pn, pa));
        std::vector<std::string> strs(g);
        for (int64_t i = 0; i < g; i++) strs[i] = detail::key_string(d.dtype, cells[i], nulls[i] != 0);
        out.add_column(group_col, StringColumn(strs));
        out.add_column(alias, Float64Column(sums));
        return true;
    }
    OptimizedDataFrame source_;
    std::vector<Op> ops_;
};

}  // namespace pandrs
