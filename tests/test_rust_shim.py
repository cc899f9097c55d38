"""The Rust shim shipped as files (integration/rust/): hip_ffi.rs must declare EXACTLY what include/pandrs_hip.h
declares — every function (name, arity, parameter names and types, return type), every struct (field names, types,
order) and every constant — checked mechanically, because this image has no Rust toolchain to compile it (so Rust
SYNTAX beyond the generated declarations is not verified here).  The three seam patches must still apply to the
reference tree when it is present (it is not on the GPU box)."""
import importlib.util
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUST = os.path.join(ROOT, "integration", "rust")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_ffi", os.path.join(RUST, "gen_ffi.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_hip_ffi_rs_matches_the_header():
    g = _gen()
    h_funcs, h_structs, h_consts, h_enums = g.parse_header()
    r_funcs, r_structs, r_consts = g.parse_rust(os.path.join(RUST, "hip_ffi.rs"))
    assert len(h_funcs) >= 39
    assert [f[0] for f in r_funcs] == [f[0] for f in h_funcs], "functions differ or are out of order"
    for (hn, hp, hr), (rn, rp, rr) in zip(h_funcs, r_funcs):
        assert len(hp) == len(rp), "%s: arity %d in the header, %d in hip_ffi.rs" % (hn, len(hp), len(rp))
        assert hp == rp, "%s: parameters differ: %s vs %s" % (hn, hp, rp)
        assert hr == rr, "%s: return type" % hn
    for cname, fields in h_structs.items():
        rname = g.STRUCTS[cname]
        assert [(n, t) for n, t in r_structs[rname]] == fields, "struct %s" % cname
    for cname in g.OPAQUE:
        assert g.STRUCTS[cname] in r_structs or ("pub struct %s { _private: [u8; 0] }" % g.STRUCTS[cname]) in open(os.path.join(RUST, "hip_ffi.rs")).read()
    merged = dict(h_consts, **h_enums)
    assert r_consts == merged
    # the file on disk IS what the generator produces (no hand edits)
    assert open(os.path.join(RUST, "hip_ffi.rs")).read() == g.generate()


def test_ffi_types_match_the_ctypes_binding():
    """A second, independent reading of the same ABI: pandrs_amd/_lib.py's ctypes table (which the GPU tests
    exercise) and hip_ffi.rs agree on every function's arity and on pointer-vs-scalar at every position."""
    import ctypes as C
    from pandrs_amd import _lib as L
    g = _gen()
    r_funcs, _, _ = g.parse_rust(os.path.join(RUST, "hip_ffi.rs"))
    assert {f[0] for f in r_funcs} == set(L.SYMBOLS)
    scal = {"i32": C.c_int32, "i64": C.c_int64, "u64": C.c_uint64, "u32": C.c_uint32, "u8": C.c_uint8, "f64": C.c_double}
    for name, params, ret in r_funcs:
        res, args = L.SYMBOLS[name]
        assert len(args) == len(params), name
        for (pn, pt), a in zip(params, args):
            if pt.startswith("*"):
                assert a in (C.c_void_p, C.c_char_p) or hasattr(a, "contents") or issubclass(a, C._Pointer), (name, pn, pt, a)
            else:
                assert a is scal[pt], (name, pn, pt, a)


def test_shim_uses_only_declared_symbols():
    import re
    g = _gen()
    r_funcs, _, r_consts = g.parse_rust(os.path.join(RUST, "hip_ffi.rs"))
    declared = {f[0] for f in r_funcs}
    shim = open(os.path.join(RUST, "hip_shim.rs")).read()
    used = set(re.findall(r"\b(pandrs_hip_\w+)\s*\(", shim))
    assert used and used <= declared, used - declared
    for const in set(re.findall(r"\b(PANDRS_HIP_[A-Z0-9_]+)\b", shim)):
        assert const in r_consts, const
    # every call passes as many arguments as the declaration takes
    arity = {f[0]: len(f[1]) for f in r_funcs}
    for m in re.finditer(r"\b(pandrs_hip_\w+)\s*\(", shim):
        depth, i, n_args, any_arg = 1, m.end(), 0, False
        while depth:
            ch = shim[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 1:
                n_args += 1
            elif not ch.isspace():
                any_arg = True
            i += 1
        n_args = n_args + 1 if any_arg else 0
        assert n_args == arity[m.group(1)], "%s called with %d arguments, declared with %d" % (m.group(1), n_args, arity[m.group(1)])


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference tree is not on this box")
def test_seam_patches_apply_to_the_reference():
    patches = sorted(f for f in os.listdir(os.path.join(RUST, "patches")) if f.endswith(".patch"))
    assert len(patches) == 5
    with tempfile.TemporaryDirectory() as d:
        for rel in ("src/optimized/split_dataframe/group/aggregation.rs", "src/optimized/split_dataframe/join.rs", "src/optimized/lazy.rs",
                    "src/optimized/split_dataframe/group/types.rs", "src/optimized/split_dataframe/group/grouping.rs"):
            os.makedirs(os.path.join(d, os.path.dirname(rel)), exist_ok=True)
            shutil.copy(os.path.join("/root/reference", rel), os.path.join(d, rel))
        for p in patches:
            r = subprocess.run(["patch", "-p1", "-i", os.path.join(RUST, "patches", p)], cwd=d, capture_output=True, text=True)
            assert r.returncode == 0, p + "\n" + r.stdout + r.stderr
        # the patched sources call only what the shim defines, with the shim's arity
        import re
        shim = open(os.path.join(RUST, "hip_shim.rs")).read()
        defined = {}
        for m in re.finditer(r"pub fn (\w+)\s*\(", shim):
            depth, i = 1, m.end()
            while depth:
                depth += {"(": 1, ")": -1}.get(shim[i], 0)
                i += 1
            defined[m.group(1)] = shim[m.end():i - 1]
        for rel in ("group/aggregation.rs", "join.rs", "../lazy.rs", "group/grouping.rs"):
            text = open(os.path.join(d, "src/optimized/split_dataframe", rel)).read()
            for m in re.finditer(r"hip_shim::(\w+)\s*\(([^;{]*?)\)\s*[{)]", text, re.S):
                name = m.group(1)
                assert name in defined, name
                if name != "hip_wanted":
                    n_params = len(re.findall(r"\b\w+\s*:(?!:)", defined[name]))
                    assert len([x for x in m.group(2).split(",") if x.strip()]) == n_params, (name, m.group(2))
        # the braces of every patched file still balance
        for rel in ("group/aggregation.rs", "join.rs", "../lazy.rs", "group/grouping.rs", "group/types.rs"):
            text = open(os.path.join(d, "src/optimized/split_dataframe", rel)).read()
            text = re.sub(r'"(?:[^"\\]|\\.)*"', '""', re.sub(r"//[^\n]*", "", text))
            text = re.sub(r"'(?:[^'\\]|\\.)'", "' '", text)
            for a, b in ("{}", "()", "[]"):
                assert text.count(a) == text.count(b), (rel, a, text.count(a), text.count(b))


# names after a '.' in hip_shim.rs that belong to std / core (everything else must exist in the reference crate)
_STD_MEMBERS = {
    "as_ptr", "as_mut_ptr", "as_ref", "as_mut", "iter", "iter_mut", "into_iter", "map", "map_or", "collect", "len", "is_none", "is_null",
    "is_empty", "unwrap", "unwrap_or", "unwrap_or_default", "ok_or_else", "borrow_mut", "with", "to_string", "to_string_lossy", "into_owned",
    "push", "extend", "insert", "remove", "get", "get_mut", "entry", "or_default", "filter", "min_by_key", "position", "zip", "enumerate",
    "sort_unstable", "join", "clone", "strong_count", "keys", "values", "0", "1",
    "lock", "map_err", "get_or_init", "split_at", "chain", "strip_suffix", "unwrap_or_else", "is_some", "max", "as_str", "code", "contains_key",
}


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference tree is not on this box")
def test_every_reference_member_the_shim_touches_exists():
    """hip_shim.rs cannot be compiled here.  Every `.name` it uses that is not a std / core method, and every
    `Type::name` path into the crate, must be declared in the reference sources (pub or pub(crate): the shim lives in
    the crate) or be one of the shim's own items — round 2 shipped calls to StringColumn::pool_string and
    BitMask::as_bytes, which do not exist."""
    import re
    shim = open(os.path.join(RUST, "hip_shim.rs")).read()
    code = re.sub(r"//[^\n]*", "", shim)
    own = set(re.findall(r"\bfn (\w+)", code)) | set(re.findall(r"^\s*(?:pub )?(\w+):", code, re.M))     # the shim's fns and struct fields
    ffi = open(os.path.join(RUST, "hip_ffi.rs")).read()
    own |= set(re.findall(r"pub (\w+):", ffi))
    ref_files = ["src/column/int64_column.rs", "src/column/float64_column.rs", "src/column/string_column.rs", "src/column/boolean_column.rs",
                 "src/column/string_pool.rs", "src/core/column.rs", "src/core/error.rs", "src/gpu/mod.rs",
                 "src/optimized/split_dataframe/core.rs", "src/optimized/split_dataframe/column_ops.rs",
                 "src/optimized/split_dataframe/data_ops.rs", "src/optimized/split_dataframe/group/types.rs",
                 "src/optimized/split_dataframe/join.rs", "src/optimized/split_dataframe/index.rs", "src/index/multi_index.rs",
                 "src/optimized/dataframe/core.rs", "src/optimized/dataframe/operations.rs"]
    ref = "\n".join(open(os.path.join("/root/reference", f)).read() for f in ref_files)
    declared = set(re.findall(r"\bfn (\w+)", ref)) | set(re.findall(r"pub(?:\(crate\))? (\w+):", ref))
    used = set(re.findall(r"(?<![.\d])\.([A-Za-z_]\w*)\b", code)) - _STD_MEMBERS - own
    missing = sorted(n for n in used if n not in declared)
    assert not missing, "hip_shim.rs uses members the reference does not declare: %s" % missing
    for needed in ("indices", "optimization_mode", "columns", "column_indices", "filter_by_indices"):
        assert needed in used, needed
    # crate paths: modules re-export what the shim imports
    for path, src, pat in [
        ("crate::column::StringColumnOptimizationMode", "src/column/mod.rs", r"pub use string_column::\{[^}]*StringColumnOptimizationMode"),
        ("crate::column::string_pool::GLOBAL_STRING_POOL", "src/column/string_pool.rs", r"pub static ref GLOBAL_STRING_POOL"),
        ("crate::core::error::{Error, Result}", "src/core/error.rs", r"pub enum Error"),
        ("crate::gpu::get_gpu_manager", "src/gpu/mod.rs", r"pub fn get_gpu_manager"),
        ("StringColumnOptimizationMode::Legacy", "src/column/string_column.rs", r"\bLegacy,"),
        ("crate::index::StringMultiIndex", "src/index/mod.rs", r"pub use (self::)?multi_index::\{[^}]*StringMultiIndex|pub use (self::)?multi_index::StringMultiIndex"),
        ("crate::optimized::dataframe::OptimizedDataFrame", "src/optimized/mod.rs", r"pub use dataframe::\{[^}]*OptimizedDataFrame"),
        ("crate::optimized::operations::AggregateOp", "src/optimized/operations.rs", r"pub enum AggregateOp"),
    ]:
        assert re.search(pat, open(os.path.join("/root/reference", src)).read()), path
    for variant in re.findall(r"Error::(\w+)\(", code):
        assert re.search(r"\b%s\(String\)" % variant, open("/root/reference/src/core/error.rs").read()), variant



# ---- which TYPE does each patched call site hand to the shim? ---------------------------------------------------------
# The crate has two frame types and two AggregateOp enums with the same names; round 3 shipped a
# lazy.rs call that passed the PUBLIC frame and the PUBLIC enum to a shim function typed on the split ones.  Without a
# compiler, resolve the names: what does `OptimizedDataFrame` / `AggregateOp` / `JoinType` mean in the file a patch
# touches (its `use` lines, `super::` resolved against the file's module path, types the file itself declares), and
# what does the shim's signature say (its `use ... as Alias` lines)?  They must be the same path.
def _module_of(rel):
    parts = rel[len("src/"):-len(".rs")].split("/")
    if parts[-1] == "mod":
        parts.pop()
    return "crate::" + "::".join(parts)


def _resolve_path(path, module):
    segs = path.split("::")
    if segs[0] == "crate":
        return path
    base = module.split("::")[:-1]                 # `super` of a file module = its parent
    if segs[0] == "self":
        return "::".join(module.split("::") + segs[1:])
    while segs and segs[0] == "super":
        segs = segs[1:]
        if segs and segs[0] == "super":
            base = base[:-1]
    return "::".join(base + segs) if path.startswith("super") else path


def _names_in_scope(text, module):
    import re
    text = re.sub(r"//[^\n]*", "", text)
    scope = {}
    for m in re.finditer(r"^\s*(?:pub )?use ([\w:]+)(?:::\{([^}]*)\})?(?: as (\w+))?;", text, re.M):
        prefix, group, alias = m.group(1), m.group(2), m.group(3)
        if group is None:
            scope[alias or prefix.split("::")[-1]] = _resolve_path(prefix, module)
        else:
            for item in group.split(","):
                item = item.strip()
                if not item or item == "self":
                    continue
                name, _, al = item.partition(" as ")
                scope[(al or name.split("::")[-1]).strip()] = _resolve_path(prefix + "::" + name.strip(), module)
    for m in re.finditer(r"^pub (?:enum|struct) (\w+)", text, re.M):
        scope[m.group(1)] = module + "::" + m.group(1)
    return scope


_CRATE_TYPES = ("OptimizedDataFrame", "AggregateOp", "JoinType")


def _shim_signatures(shim):
    import re
    scope = _names_in_scope(shim, "crate::gpu::hip_shim")
    sigs = {}
    for m in re.finditer(r"pub fn (\w+)\s*(?:<[^>]*>)?\s*\(", shim):
        depth, i = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(shim[i], 0)
            i += 1
        params = shim[m.end():i - 1]
        ret = re.match(r"\s*->\s*([^{]+)\{", shim[i:])
        used = {}
        for ident in set(re.findall(r"\b[A-Z]\w+\b", params + (ret.group(1) if ret else ""))):
            full = scope.get(ident)
            if full and full.split("::")[-1] in _CRATE_TYPES and full.startswith("crate::optimized"):
                used.setdefault(full.split("::")[-1], set()).add(full)
        sigs[m.group(1)] = used
    return sigs


def test_import_resolver_reads_use_lines():
    scope = _names_in_scope("use super::super::core::OptimizedDataFrame;\nuse super::types::{AggregateFn, AggregateOp as Op, GroupBy};\n"
                            "use crate::error::{Error, Result};\npub enum JoinType {\n", "crate::optimized::split_dataframe::group::aggregation")
    assert scope["OptimizedDataFrame"] == "crate::optimized::split_dataframe::core::OptimizedDataFrame"
    assert scope["Op"] == "crate::optimized::split_dataframe::group::types::AggregateOp"
    assert scope["JoinType"] == "crate::optimized::split_dataframe::group::aggregation::JoinType"
    assert _module_of("src/optimized/lazy.rs") == "crate::optimized::lazy"


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference tree is not on this box")
def test_patched_call_sites_pass_the_types_the_shim_is_typed_on():
    import re
    shim = open(os.path.join(RUST, "hip_shim.rs")).read()
    sigs = _shim_signatures(shim)
    # the public entry points name their frame in their name, and are typed accordingly
    pub_frame, split_frame = "crate::optimized::dataframe::OptimizedDataFrame", "crate::optimized::split_dataframe::core::OptimizedDataFrame"
    assert sigs["lazy_aggregate_hip"] == {"OptimizedDataFrame": {pub_frame}, "AggregateOp": {"crate::optimized::operations::AggregateOp"}}
    assert sigs["lazy_join_groupby_sum_hip"] == {"OptimizedDataFrame": {pub_frame}}
    assert sigs["groupby_aggregate_split_hip"] == {"OptimizedDataFrame": {split_frame},
                                                   "AggregateOp": {"crate::optimized::split_dataframe::group::types::AggregateOp"}}
    assert sigs["join_frame_split_hip"] == {"OptimizedDataFrame": {split_frame}, "JoinType": {"crate::optimized::split_dataframe::join::JoinType"}}
    patches = sorted(f for f in os.listdir(os.path.join(RUST, "patches")) if f.endswith(".patch"))
    seen = set()
    with tempfile.TemporaryDirectory() as d:
        rels = ("src/optimized/split_dataframe/group/aggregation.rs", "src/optimized/split_dataframe/join.rs", "src/optimized/lazy.rs",
                "src/optimized/split_dataframe/group/types.rs", "src/optimized/split_dataframe/group/grouping.rs")
        for rel in rels:
            os.makedirs(os.path.join(d, os.path.dirname(rel)), exist_ok=True)
            shutil.copy(os.path.join("/root/reference", rel), os.path.join(d, rel))
        for p in patches:
            r = subprocess.run(["patch", "-p1", "-i", os.path.join(RUST, "patches", p)], cwd=d, capture_output=True, text=True)
            assert r.returncode == 0, p + "\n" + r.stdout + r.stderr
        for rel in rels:
            text = open(os.path.join(d, rel)).read()
            scope = _names_in_scope(text, _module_of(rel))
            for m in re.finditer(r"hip_shim::(\w+)\s*\(", text):
                name = m.group(1)
                assert name in sigs, "%s calls hip_shim::%s, which the shim does not define as pub fn" % (rel, name)
                seen.add(name)
                for base, fulls in sigs[name].items():
                    assert len(fulls) == 1, (name, base, fulls)
                    assert base in scope, "%s calls hip_shim::%s (typed on %s) but has no %s in scope" % (rel, name, next(iter(fulls)), base)
                    assert scope[base] in fulls, "%s: `%s` there is %s, but hip_shim::%s is typed on %s" % (rel, base, scope[base], name, next(iter(fulls)))
    assert {"lazy_aggregate_hip", "lazy_join_groupby_sum_hip", "groupby_aggregate_split_hip", "join_frame_split_hip", "group_indices_hip",
            "par_groupby_hip"} <= seen, seen


def test_the_resolver_catches_the_round3_seam_bug():
    """lazy.rs handing its (public) frame to a function typed on the split frame — what patch 0003 did in round 3."""
    lazy_scope = _names_in_scope("use crate::optimized::dataframe::OptimizedDataFrame;\nuse crate::optimized::operations::AggregateOp;\n", "crate::optimized::lazy")
    bad_shim = ("use crate::optimized::split_dataframe::core::OptimizedDataFrame;\nuse crate::optimized::split_dataframe::group::types::AggregateOp;\n"
                "pub fn groupby_aggregate_hip(df: &OptimizedDataFrame, aggregations: &[(String, AggregateOp, String)]) -> Result<OptimizedDataFrame> {\n}\n")
    sig = _shim_signatures(bad_shim)["groupby_aggregate_hip"]
    assert lazy_scope["OptimizedDataFrame"] not in sig["OptimizedDataFrame"] and lazy_scope["AggregateOp"] not in sig["AggregateOp"]


def test_shim_honours_create_multi_index_and_fuses_join_aggregate():
    """Source-level checks of the two seams VERDICT r3 found absent: the StringMultiIndex branch (aggregation.rs:812-853) and a route from
    LazyFrame's Join + Aggregate to the fused C5 operator; the join seam assembles columns on the device instead of fetching index pairs."""
    shim = open(os.path.join(RUST, "hip_shim.rs")).read()
    body = shim[shim.index("pub fn groupby_aggregate_split_hip"):shim.index("pub fn lazy_aggregate_hip")]
    assert "create_multi_index && group_by_columns.len() > 1" in body and "StringMultiIndex::from_tuples" in body and "set_index_from_multi_index" in body
    lazy = shim[shim.index("pub fn lazy_aggregate_hip"):shim.index("pub fn lazy_join_groupby_sum_hip")]
    assert "MultiIndex" not in lazy                                          # the lazy arm never builds one (lazy.rs:390-394)
    assert "pandrs_hip_join_groupby_sum(" in shim[shim.index("pub fn lazy_join_groupby_sum_hip"):]
    join = shim[shim.index("pub fn join_frame_split_hip"):]
    assert "pandrs_hip_join_gather" in shim and "pandrs_hip_join_fetch" not in shim, "the join seam must not move index pairs over PCIe"
    assert "pandrs_hip_join_indices(" in join
    p1 = open(os.path.join(RUST, "patches", "0001-groupby-aggregate-hip-callout.patch")).read()
    assert "self.create_multi_index" in p1
    p3 = open(os.path.join(RUST, "patches", "0003-lazyframe-aggregate-hip-callout.patch")).read()
    assert "lazy_join_groupby_sum_hip" in p3 and "lazy_aggregate_hip" in p3 and "peekable" in p3
