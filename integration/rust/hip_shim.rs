// hip_shim.rs — the crate-side shim over hip_ffi.rs: add as src/gpu/hip_shim.rs (cfg(hip_available)) together with
// hip_ffi.rs.  NOT compiled in this repository's build image (no Rust toolchain); the same ABI calls, in the same
// order, are what include/pandrs_hip.hpp (C++) and pandrs_amd/frame.py (Python) make and what their tests replay.
// tests/test_rust_shim.py checks every field and method this file touches on a reference type against
// /root/reference/src (all of them are `pub` or `pub(crate)` there: the shim lives inside the crate).
//
// Seams served (see patches/*.patch):
//   OptimizedDataFrame::group_by    src/optimized/split_dataframe/group/grouping.rs:38-115  (GroupBy.groups, filled lazily)
//   OptimizedDataFrame::par_groupby src/optimized/split_dataframe/group/grouping.rs:124-331
//   GroupBy::aggregate              src/optimized/split_dataframe/group/aggregation.rs:763
//   LazyFrame::execute, Aggregate   src/optimized/lazy.rs:186
//   OptimizedDataFrame::join_impl   src/optimized/split_dataframe/join.rs:106-224
// The PUBLIC frame (src/optimized/dataframe/transformations.rs:524-577 aggregate, :628-905 the four joins, :524
// par_groupby) needs no patch of its own: every wrapper Arc-clones its columns into a split frame and calls the seams
// above, and the clone keeps the Arc's data pointer — the key of the resident cache below — so a column uploaded
// for one call is found again by the next, whichever frame object carries it.
#![cfg(hip_available)]

use std::cell::RefCell;
use std::collections::HashMap;
use std::ffi::CStr;
use std::sync::Arc;

use super::hip_ffi::*;
use crate::column::string_pool::GLOBAL_STRING_POOL;
use crate::column::{Column, ColumnTrait, Float64Column, StringColumn, StringColumnOptimizationMode};
use crate::core::error::{Error, Result};
use crate::optimized::split_dataframe::core::OptimizedDataFrame;
use crate::optimized::split_dataframe::group::types::AggregateOp;
use crate::optimized::split_dataframe::join::JoinType;

/// One context (HIP stream + workspace + resident columns) per thread: contexts are independent, so rayon workers
/// never contend.
pub struct HipContext {
    ctx: *mut PandrsHipCtx,
    resident: ResidentCache,
}

impl HipContext {
    fn new() -> Result<Self> {
        let mut ctx: *mut PandrsHipCtx = std::ptr::null_mut();
        check(unsafe { pandrs_hip_ctx_create(-1, &mut ctx) })?;
        Ok(HipContext { ctx, resident: ResidentCache::default() })
    }
}

impl Drop for HipContext {
    fn drop(&mut self) {
        // pandrs_hip_ctx_destroy frees the columns still resident
        unsafe { pandrs_hip_ctx_destroy(self.ctx) };
    }
}

thread_local! {
    static CTX: RefCell<Option<HipContext>> = RefCell::new(None);
}

fn with_ctx<T>(f: impl FnOnce(&mut HipContext) -> Result<T>) -> Result<T> {
    CTX.with(|slot| {
        let mut slot = slot.borrow_mut();
        if slot.is_none() {
            *slot = Some(HipContext::new()?);
        }
        f(slot.as_mut().unwrap())
    })
}

/// Status -> pandrs::Error, the mapping the reference already uses for device failures (src/gpu/mod.rs:206-210).
fn check(status: i32) -> Result<()> {
    if status == PANDRS_HIP_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(pandrs_hip_last_error()) }.to_string_lossy().into_owned();
    Err(match status {
        PANDRS_HIP_ERR_INVALID_ARGUMENT => Error::InvalidInput(msg),
        PANDRS_HIP_ERR_TYPE_MISMATCH => Error::Type(msg),
        PANDRS_HIP_ERR_OPERATION_FAILED => Error::OperationFailed(msg),
        _ => Error::Computation(msg),
    })
}

/// GpuConfig gating, like src/optimized/split_dataframe/gpu.rs:30-32: below min_size_threshold, or with the device
/// path disabled, the caller keeps its CPU body.
pub fn hip_wanted(row_count: usize) -> bool {
    match crate::gpu::get_gpu_manager() {
        Ok(m) => m.context().config().enabled && row_count >= m.context().config().min_size_threshold,
        Err(_) => false,
    }
}

fn mask_ptr(mask: &Option<Arc<[u8]>>) -> *const u8 {
    mask.as_ref().map_or(std::ptr::null(), |m| m.as_ptr())
}

/// The reference's column layouts ARE the ABI's: no copy on the host side.  None = this column has
/// no device form and the caller keeps its CPU body: a Legacy-mode StringColumn's codes index its OWN pool
/// (string_column.rs:47-58), so equal codes of two columns do not mean equal strings; only GlobalPool / Categorical
/// codes (both built by new_with_global_pool, :61-84) come from GLOBAL_STRING_POOL (string_pool.rs:28-53).
fn view(col: &Column) -> Option<PandrsHipColumn> {
    Some(match col {
        Column::Int64(c) => PandrsHipColumn { data: c.data.as_ptr() as _, null_mask: mask_ptr(&c.null_mask), dtype: PANDRS_HIP_I64, reserved: 0 },
        Column::Float64(c) => PandrsHipColumn { data: c.data.as_ptr() as _, null_mask: mask_ptr(&c.null_mask), dtype: PANDRS_HIP_F64, reserved: 0 },
        Column::String(c) => {
            if c.optimization_mode == StringColumnOptimizationMode::Legacy {
                return None;
            }
            PandrsHipColumn { data: c.indices.as_ptr() as _, null_mask: mask_ptr(&c.null_mask), dtype: PANDRS_HIP_U32CODE, reserved: 0 }
        }
        // BooleanColumn.data is a BitMask whose `data: Arc<[u8]>` holds the LSB-first bits (src/core/column.rs:72-75)
        Column::Boolean(c) => PandrsHipColumn { data: c.data.data.as_ptr() as _, null_mask: mask_ptr(&c.null_mask), dtype: PANDRS_HIP_BOOLBITS, reserved: 0 },
    })
}

// ---- resident columns ----------------------------------------------------------------------------------------------
// Columns are immutable Arc<[T]> buffers (src/column/int64_column.rs:10): uploaded once, a column serves every later
// aggregate / join from HBM (through PANDRS_HIP_MEM_HOST every call would stage it over PCIe again: 75 ms instead of
// 3.3 ms for 100 M rows x 5 columns).  Key: the data Arc's pointer.  Each entry keeps Weak handles of the Arcs it
// was uploaded from: a Weak keeps the ALLOCATION alive after the last strong reference is gone, so the address
// cannot be reused while the entry exists, and `alive()` turning false is the signal to evict.
struct ResidentEntry {
    desc: PandrsHipColumn,
    bytes: usize,
    alive: Box<dyn Fn() -> bool>,
    last_use: u64,
}

#[derive(Default)]
struct ResidentCache {
    entries: HashMap<usize, ResidentEntry>,
    bytes: usize,
    clock: u64,
    pin_epoch: u64,     // entries used at or after this tick belong to the call in progress: never evicted
}

fn liveness<T: ?Sized + 'static>(data: &Arc<T>, mask: &Option<Arc<[u8]>>) -> Box<dyn Fn() -> bool> {
    let d = Arc::downgrade(data);
    let m = mask.as_ref().map(Arc::downgrade);
    Box::new(move || d.strong_count() > 0 && m.as_ref().map_or(true, |m| m.strong_count() > 0))
}

impl ResidentCache {
    fn release(ctx: *mut PandrsHipCtx, e: &ResidentEntry) {
        unsafe { pandrs_hip_column_release(ctx, &e.desc) };
    }

    /// Every public entry point of this file calls this first: descriptors handed out from here on stay valid
    /// until the library call that uses them has returned.
    fn begin_call(&mut self) {
        self.clock += 1;
        self.pin_epoch = self.clock;
    }

    /// Device descriptor of `col`, uploading it on first sight.  Dropped columns are evicted on every call; when
    /// the budget (half of GpuConfig.memory_limit, the other half is the library's workspace) would be exceeded,
    /// the least recently used live entries go first.
    fn get(&mut self, ctx: *mut PandrsHipCtx, col: &Column, n_rows: usize) -> Result<Option<PandrsHipColumn>> {
        let host = match view(col) {
            Some(v) => v,
            None => return Ok(None),
        };
        let dead: Vec<usize> = self.entries.iter().filter(|(_, e)| !(e.alive)()).map(|(k, _)| *k).collect();
        for k in dead {
            if let Some(e) = self.entries.remove(&k) {
                self.bytes -= e.bytes;
                Self::release(ctx, &e);
            }
        }
        let key = host.data as usize;
        if let Some(e) = self.entries.get_mut(&key) {
            if e.desc.dtype == host.dtype && e.desc.null_mask.is_null() == host.null_mask.is_null() {
                e.last_use = self.clock;
                return Ok(Some(e.desc));
            }
        }
        if let Some(e) = self.entries.remove(&key) {       // same buffer seen as another column type: start over
            self.bytes -= e.bytes;
            Self::release(ctx, &e);
        }
        let elem = match host.dtype { PANDRS_HIP_U32CODE => 4, PANDRS_HIP_BOOLBITS => 0, _ => 8 };
        let bytes = if elem == 0 { (n_rows + 7) / 8 } else { n_rows * elem } + if host.null_mask.is_null() { 0 } else { (n_rows + 7) / 8 };
        let budget = crate::gpu::get_gpu_manager().map(|m| m.context().config().memory_limit / 2).unwrap_or(usize::MAX);
        while self.bytes + bytes > budget {
            let pin = self.pin_epoch;
            let oldest = self.entries.iter().filter(|(_, e)| e.last_use < pin).min_by_key(|(_, e)| e.last_use).map(|(k, _)| *k);
            match oldest {
                Some(k) => {
                    let e = self.entries.remove(&k).unwrap();
                    self.bytes -= e.bytes;
                    Self::release(ctx, &e);
                }
                // the columns of THIS call alone exceed the budget: the caller keeps its CPU body
                None => return Err(Error::Computation("resident columns exceed GpuConfig.memory_limit / 2".to_string())),
            }
        }
        let mut desc = PandrsHipColumn { data: std::ptr::null(), null_mask: std::ptr::null(), dtype: host.dtype, reserved: 0 };
        check(unsafe { pandrs_hip_column_upload(ctx, &host, n_rows as i64, &mut desc) })?;
        let alive = match col {
            Column::Int64(c) => liveness(&c.data, &c.null_mask),
            Column::Float64(c) => liveness(&c.data, &c.null_mask),
            Column::String(c) => liveness(&c.indices, &c.null_mask),
            Column::Boolean(c) => liveness(&c.data.data, &c.null_mask),
        };
        self.entries.insert(key, ResidentEntry { desc, bytes, alive, last_use: self.clock });
        self.bytes += bytes;
        Ok(Some(desc))
    }
}

/// Device descriptors of several columns of one frame; None as soon as one of them has no device form.
fn resident_all(hc: &mut HipContext, df: &OptimizedDataFrame, names: &[&String]) -> Result<Option<Vec<PandrsHipColumn>>> {
    let mut out = Vec::with_capacity(names.len());
    for name in names {
        let col = &df.columns[df.column_indices[*name]];
        match hc.resident.get(hc.ctx, col, df.row_count())? {
            Some(d) => out.push(d),
            None => return Ok(None),
        }
    }
    Ok(Some(out))
}

fn no_device_form() -> Error {
    Error::OperationFailed("a Legacy-mode StringColumn has no device form".to_string())
}

/// Group-key cell -> the string the reference's `to_string()` produces (grouping.rs:79-96, lazy.rs:199-236).
fn key_string(col: &Column, cell: u64, is_null: bool, null_string: &str) -> String {
    if is_null {
        return null_string.to_string();
    }
    match col {
        Column::Int64(_) => (cell as i64).to_string(),
        Column::Float64(_) => f64::from_bits(cell).to_string(),
        // the codes of a non-Legacy StringColumn index GLOBAL_STRING_POOL (string_column.rs:62), not c.string_pool
        Column::String(_) => GLOBAL_STRING_POOL.get(cell as u32).unwrap_or_default(),
        Column::Boolean(_) => (cell != 0).to_string(),
    }
}

/// GroupBy::aggregate's body (aggregation.rs:763-871) on the device.  `null_string`: "NULL" for aggregate / the lazy
/// arm.  Needs neither GroupBy.groups nor any per-row host work: the seam sits BEFORE the lazily filled map is touched.
pub fn groupby_aggregate_hip(
    df: &OptimizedDataFrame,
    group_by_columns: &[String],
    aggregations: &[(String, AggregateOp, String)],
    null_string: &str,
) -> Result<OptimizedDataFrame> {
    // distinct value columns, in first-use order
    let mut val_names: Vec<&String> = Vec::new();
    let mut specs: Vec<PandrsHipAggSpec> = Vec::new();
    for (col, op, _) in aggregations {
        let idx = match val_names.iter().position(|n| *n == col) {
            Some(i) => i,
            None => { val_names.push(col); val_names.len() - 1 }
        };
        specs.push(PandrsHipAggSpec { col: idx as i32, op: *op as i32 });      // types.rs:11-34 order = pandrs_hip_agg_op
    }
    let key_names: Vec<&String> = group_by_columns.iter().collect();
    let n_rows = df.row_count() as i64;

    with_ctx(|hc| {
        hc.resident.begin_call();
        let keys = resident_all(hc, df, &key_names)?.ok_or_else(no_device_form)?;
        let vals = resident_all(hc, df, &val_names)?.ok_or_else(no_device_form)?;
        let ctx = hc.ctx;
        let mut n_groups: i64 = 0;
        check(unsafe {
            pandrs_hip_groupby_agg(ctx, PANDRS_HIP_MEM_DEVICE, keys.as_ptr(), keys.len() as i32, n_rows,
                                   vals.as_ptr(), vals.len() as i32, specs.as_ptr(), specs.len() as i32, &mut n_groups)
        })?;
        let g = n_groups as usize;
        let mut key_cells: Vec<Vec<u64>> = vec![vec![0u64; g]; keys.len()];
        let mut key_null: Vec<Vec<u8>> = vec![vec![0u8; g]; keys.len()];
        let mut aggs: Vec<Vec<f64>> = vec![vec![0f64; g]; specs.len()];
        let pk: Vec<*mut u64> = key_cells.iter_mut().map(|v| v.as_mut_ptr()).collect();
        let pn: Vec<*mut u8> = key_null.iter_mut().map(|v| v.as_mut_ptr()).collect();
        let pa: Vec<*mut f64> = aggs.iter_mut().map(|v| v.as_mut_ptr()).collect();
        check(unsafe { pandrs_hip_groupby_fetch(ctx, PANDRS_HIP_MEM_HOST, pk.as_ptr(), pn.as_ptr(), pa.as_ptr()) })?;

        // result frame exactly as aggregation.rs:812-867 builds it: one string column per key, one f64 column per alias
        let mut result = OptimizedDataFrame::new();
        for (i, name) in group_by_columns.iter().enumerate() {
            let col = &df.columns[df.column_indices[name]];
            let strings: Vec<String> = (0..g).map(|r| key_string(col, key_cells[i][r], key_null[i][r] != 0, null_string)).collect();
            result.add_column(name.clone(), Column::String(StringColumn::new(strings)))?;
        }
        for ((_, _, alias), values) in aggregations.iter().zip(aggs.into_iter()) {
            result.add_column(alias.clone(), Column::Float64(Float64Column::new(values)))?;
        }
        Ok(result)
    })
}

/// group_by's own result (grouping.rs:59-104): every group's key strings and its ascending row list, from the
/// device-built CSR (pandrs_hip_groupby_indices).  `null_string`: "NULL" for group_by, "NA" for par_groupby (:152).
pub fn group_indices_hip(
    df: &OptimizedDataFrame,
    group_by_columns: &[String],
    null_string: &str,
) -> Result<HashMap<Vec<String>, Vec<usize>>> {
    let key_names: Vec<&String> = group_by_columns.iter().collect();
    let n_rows = df.row_count();
    with_ctx(|hc| {
        hc.resident.begin_call();
        let keys = resident_all(hc, df, &key_names)?.ok_or_else(no_device_form)?;
        let ctx = hc.ctx;
        let mut n_groups: i64 = 0;
        check(unsafe { pandrs_hip_groupby_indices(ctx, PANDRS_HIP_MEM_DEVICE, keys.as_ptr(), keys.len() as i32, n_rows as i64, &mut n_groups) })?;
        let g = n_groups as usize;
        let mut key_cells: Vec<Vec<u64>> = vec![vec![0u64; g]; keys.len()];
        let mut key_null: Vec<Vec<u8>> = vec![vec![0u8; g]; keys.len()];
        let mut offsets: Vec<i64> = vec![0i64; g + 1];
        let mut rows: Vec<i64> = vec![0i64; n_rows];
        let pk: Vec<*mut u64> = key_cells.iter_mut().map(|v| v.as_mut_ptr()).collect();
        let pn: Vec<*mut u8> = key_null.iter_mut().map(|v| v.as_mut_ptr()).collect();
        check(unsafe { pandrs_hip_groupby_indices_fetch(ctx, PANDRS_HIP_MEM_HOST, pk.as_ptr(), pn.as_ptr(), offsets.as_mut_ptr(), rows.as_mut_ptr()) })?;
        let mut groups: HashMap<Vec<String>, Vec<usize>> = HashMap::with_capacity(g);
        for grp in 0..g {
            let key: Vec<String> = group_by_columns
                .iter()
                .enumerate()
                .map(|(i, name)| key_string(&df.columns[df.column_indices[name]], key_cells[i][grp], key_null[i][grp] != 0, null_string))
                .collect();
            let list = rows[offsets[grp] as usize..offsets[grp + 1] as usize].iter().map(|&r| r as usize);
            // two cells with one string (0.0 / -0.0 never collide: "0" vs "-0"; NaN payloads are collapsed by the library)
            groups.entry(key).or_default().extend(list);
        }
        Ok(groups)
    })
}

/// par_groupby's body (grouping.rs:124-331): key parts joined with "_" (:189), nulls as "NA" (:152), one sub-frame
/// per group through the reference's own filter_by_indices (data_ops.rs:124).
pub fn par_groupby_hip(df: &OptimizedDataFrame, group_by_columns: &[&str]) -> Result<HashMap<String, OptimizedDataFrame>> {
    let cols: Vec<String> = group_by_columns.iter().map(|s| s.to_string()).collect();
    let groups = group_indices_hip(df, &cols, "NA")?;
    let mut merged: HashMap<String, Vec<usize>> = HashMap::with_capacity(groups.len());
    for (key, rows) in groups {
        // different tuples can join to one string ("a_b" + "c" / "a" + "b_c"): the reference merges them too (:189-199)
        let slot = merged.entry(key.join("_")).or_default();
        slot.extend(rows);
        slot.sort_unstable();
    }
    let mut result = HashMap::with_capacity(merged.len());
    for (key, rows) in merged {
        result.insert(key, df.filter_by_indices(&rows)?);
    }
    Ok(result)
}

/// join_impl's index build (join.rs:106-224): (left row, right row) pairs in the reference's order; -1 <=> None.
pub fn join_indices_hip(left: &Column, right: &Column, join_type: JoinType) -> Result<(Vec<i64>, Vec<i64>)> {
    let how = match join_type {
        JoinType::Inner => PANDRS_HIP_JOIN_INNER,
        JoinType::Left => PANDRS_HIP_JOIN_LEFT,
        JoinType::Right => PANDRS_HIP_JOIN_RIGHT,
        JoinType::Outer => PANDRS_HIP_JOIN_OUTER,
    };
    with_ctx(|hc| {
        hc.resident.begin_call();
        let l = hc.resident.get(hc.ctx, left, left.len())?.ok_or_else(no_device_form)?;
        let r = hc.resident.get(hc.ctx, right, right.len())?.ok_or_else(no_device_form)?;
        let ctx = hc.ctx;
        let mut n: i64 = 0;
        check(unsafe { pandrs_hip_join_indices(ctx, PANDRS_HIP_MEM_DEVICE, &l, left.len() as i64, &r, right.len() as i64, how, &mut n) })?;
        let (mut li, mut ri) = (vec![0i64; n as usize], vec![0i64; n as usize]);
        check(unsafe { pandrs_hip_join_fetch(ctx, PANDRS_HIP_MEM_HOST, li.as_mut_ptr(), ri.as_mut_ptr()) })?;
        Ok((li, ri))
    })
}

/// K1 (split_dataframe/aggregate.rs:21-215, column/{int64,float64}_column.rs:100-199, jit/simd.rs:9-112): one pass.
pub fn column_stats_hip(col: &Column) -> Result<PandrsHipColumnStats> {
    with_ctx(|hc| {
        hc.resident.begin_call();
        let v = hc.resident.get(hc.ctx, col, col.len())?.ok_or_else(no_device_form)?;
        let mut st: PandrsHipColumnStats = unsafe { std::mem::zeroed() };
        check(unsafe { pandrs_hip_reduce_stats(hc.ctx, PANDRS_HIP_MEM_DEVICE, &v, col.len() as i64, &mut st) })?;
        Ok(st)
    })
}
