// hip_shim.rs — the crate-side shim over hip_ffi.rs: add as src/gpu/hip_shim.rs (cfg(hip_available)) together with
// hip_ffi.rs.  NOT compiled in this repository's build image (no Rust toolchain); the same ABI calls, in the same
// order, are what include/pandrs_hip.hpp (C++) and pandrs_amd/frame.py (Python) make and what their tests replay.
//
// Seams served (see patches/*.patch):
//   GroupBy::aggregate              src/optimized/split_dataframe/group/aggregation.rs:763
//   LazyFrame::execute, Aggregate   src/optimized/lazy.rs:186
//   OptimizedDataFrame::join_impl   src/optimized/split_dataframe/join.rs:106-224
#![cfg(hip_available)]

use std::cell::RefCell;
use std::ffi::CStr;

use super::hip_ffi::*;
use crate::column::{Column, ColumnTrait, Float64Column, StringColumn};
use crate::error::{Error, Result};
use crate::optimized::split_dataframe::core::OptimizedDataFrame;
use crate::optimized::split_dataframe::group::types::AggregateOp;
use crate::optimized::split_dataframe::join::JoinType;

/// One context (HIP stream + workspace) per thread: contexts are independent, so rayon workers never contend.
pub struct HipContext(*mut PandrsHipCtx);

impl HipContext {
    fn new() -> Result<Self> {
        let mut ctx: *mut PandrsHipCtx = std::ptr::null_mut();
        check(unsafe { pandrs_hip_ctx_create(-1, &mut ctx) })?;
        Ok(HipContext(ctx))
    }
}

impl Drop for HipContext {
    fn drop(&mut self) {
        unsafe { pandrs_hip_ctx_destroy(self.0) };
    }
}

thread_local! {
    static CTX: RefCell<Option<HipContext>> = RefCell::new(None);
}

fn with_ctx<T>(f: impl FnOnce(*mut PandrsHipCtx) -> Result<T>) -> Result<T> {
    CTX.with(|slot| {
        let mut slot = slot.borrow_mut();
        if slot.is_none() {
            *slot = Some(HipContext::new()?);
        }
        f(slot.as_ref().unwrap().0)
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

fn mask_ptr(mask: &Option<std::sync::Arc<[u8]>>) -> *const u8 {
    mask.as_ref().map_or(std::ptr::null(), |m| m.as_ptr())
}

/// The reference's column layouts ARE the ABI's: no copy on the host side.
fn view(col: &Column) -> PandrsHipColumn {
    match col {
        Column::Int64(c) => PandrsHipColumn { data: c.data.as_ptr() as _, null_mask: mask_ptr(&c.null_mask), dtype: PANDRS_HIP_I64, reserved: 0 },
        Column::Float64(c) => PandrsHipColumn { data: c.data.as_ptr() as _, null_mask: mask_ptr(&c.null_mask), dtype: PANDRS_HIP_F64, reserved: 0 },
        // GlobalPool mode: equal string <=> equal code (src/column/string_pool.rs:28-53)
        Column::String(c) => PandrsHipColumn { data: c.indices.as_ptr() as _, null_mask: mask_ptr(&c.null_mask), dtype: PANDRS_HIP_U32CODE, reserved: 0 },
        Column::Boolean(c) => PandrsHipColumn { data: c.data.as_bytes().as_ptr() as _, null_mask: mask_ptr(&c.null_mask), dtype: PANDRS_HIP_BOOLBITS, reserved: 0 },
    }
}

/// Group-key cell -> the string the reference's `to_string()` produces (grouping.rs:79-96, lazy.rs:199-236).
fn key_string(col: &Column, cell: u64, is_null: bool, null_string: &str) -> String {
    if is_null {
        return null_string.to_string();
    }
    match col {
        Column::Int64(_) => (cell as i64).to_string(),
        Column::Float64(_) => f64::from_bits(cell).to_string(),
        Column::String(c) => c.pool_string(cell as u32),
        Column::Boolean(_) => (cell != 0).to_string(),
    }
}

/// GroupBy::aggregate's body (aggregation.rs:763-871) on the device.  `null_string`: "NULL" for aggregate / the lazy
/// arm, "NA" for par_groupby.
pub fn groupby_aggregate_hip(
    df: &OptimizedDataFrame,
    group_by_columns: &[String],
    aggregations: &[(String, AggregateOp, String)],
    null_string: &str,
) -> Result<OptimizedDataFrame> {
    let keys: Vec<PandrsHipColumn> = group_by_columns.iter().map(|k| view(&df.columns[df.column_indices[k]])).collect();
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
    let vals: Vec<PandrsHipColumn> = val_names.iter().map(|n| view(&df.columns[df.column_indices[*n]])).collect();
    let n_rows = df.row_count() as i64;

    with_ctx(|ctx| {
        let mut n_groups: i64 = 0;
        check(unsafe {
            pandrs_hip_groupby_agg(ctx, PANDRS_HIP_MEM_HOST, keys.as_ptr(), keys.len() as i32, n_rows,
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

/// join_impl's index build (join.rs:106-224): (left row, right row) pairs in the reference's order; -1 <=> None.
pub fn join_indices_hip(left: &Column, right: &Column, join_type: JoinType) -> Result<(Vec<i64>, Vec<i64>)> {
    let (l, r) = (view(left), view(right));
    let how = match join_type {
        JoinType::Inner => PANDRS_HIP_JOIN_INNER,
        JoinType::Left => PANDRS_HIP_JOIN_LEFT,
        JoinType::Right => PANDRS_HIP_JOIN_RIGHT,
        JoinType::Outer => PANDRS_HIP_JOIN_OUTER,
    };
    with_ctx(|ctx| {
        let mut n: i64 = 0;
        check(unsafe { pandrs_hip_join_indices(ctx, PANDRS_HIP_MEM_HOST, &l, left.len() as i64, &r, right.len() as i64, how, &mut n) })?;
        let (mut li, mut ri) = (vec![0i64; n as usize], vec![0i64; n as usize]);
        check(unsafe { pandrs_hip_join_fetch(ctx, PANDRS_HIP_MEM_HOST, li.as_mut_ptr(), ri.as_mut_ptr()) })?;
        Ok((li, ri))
    })
}

/// K1 (split_dataframe/aggregate.rs:21-215, column/{int64,float64}_column.rs:100-199, jit/simd.rs:9-112): one pass.
pub fn column_stats_hip(col: &Column) -> Result<PandrsHipColumnStats> {
    let v = view(col);
    with_ctx(|ctx| {
        let mut st: PandrsHipColumnStats = unsafe { std::mem::zeroed() };
        check(unsafe { pandrs_hip_reduce_stats(ctx, PANDRS_HIP_MEM_HOST, &v, col.len() as i64, &mut st) })?;
        Ok(st)
    })
}
