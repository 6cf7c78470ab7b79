//! hip_mdd -- the binding a ddo maintainer adds to run `ParallelSolver` on the MI355X engine (`libddo_hip.so`, C ABI of
//! `include/ddo_hip.h`).  `HipMdd` implements `DecisionDiagram` (ddo/src/abstraction/mdd.rs:75-114) and is passed as the type
//! parameter `D` of `ParallelSolver<'a, State, D, C>` (ddo/src/implementation/solver/parallel.rs:287-306); `HipCache` implements
//! `Cache` (abstraction/cache.rs:27-57) over the device-side table.  INTEGRATION.md documents the mapping entry point by entry
//! point; `tests/test_integration_doc.py` checks this file mechanically against the header (structs, prototypes, constants).
//! NOT compiled in the build image (no Rust toolchain there): `cargo test` in this directory is the one-command check of the
//! binding and of the tie-break parity the C++ oracle cannot pin.
use bit_set::BitSet;
use ddo::*;
use std::os::raw::{c_int, c_void};
use std::sync::atomic::{AtomicI32, Ordering};
use std::sync::{Arc, RwLock};

// ---- include/ddo_hip.h, field for field --------------------------------------------------------------------------
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct DdoDecision { variable: i64, value: i64 }
#[repr(C)] pub struct DdoSubProblem { state: *const u64, state_words: usize, value: i64, ub: i64,
                                      depth: usize, path: *const DdoDecision, path_len: usize }
#[repr(C)] pub struct DdoCompileInput { comp_type: c_int, max_width: usize, best_lb: i64, residual: DdoSubProblem,
                                        cutoff: *const c_int, cache: *mut c_void, dominance: *mut c_void }
#[repr(C)] #[derive(Default)] pub struct DdoCompletion { is_exact: c_int, has_best_value: c_int, best_value: i64 }
#[repr(C)] pub struct DdoCutsetRows { count: usize, state_words: usize, path_stride: usize, states: *const u64, values: *const i64,
                                     ubs: *const i64, depths: *const usize, path_lens: *const usize, paths: *const DdoDecision }
pub const DDO_LAST_EXACT_LAYER: c_int = 1;
pub const DDO_FRONTIER: c_int = 2;
pub const DDO_MDD_CACHING: c_int = 0x10;
pub const DDO_MDD_POOLED: c_int = 0x20;             // the mdds are `Pooled` decision diagrams (mdd/pooled.rs): see install_pooled
pub const DDO_MDD_ENGINE_DENSE: c_int = 0x100;      // test / measurement hooks: bind an mdd to one kernel of the in-place engine
pub const DDO_MDD_ENGINE_TIER0: c_int = 0x200;
pub const DDO_MDD_ENGINE_TIER1: c_int = 0x300;

#[link(name = "ddo_hip")]
extern "C" {
    fn ddo_last_error() -> *const std::os::raw::c_char;
    fn ddo_model_create_misp(n: c_int, compl_adj_rows: *const u64, weights: *const i64) -> *mut c_void;
    fn ddo_model_destroy(model: *mut c_void);
    fn ddo_cache_create(model: *const c_void, device: c_int, capacity_entries: usize) -> *mut c_void;
    fn ddo_cache_destroy(cache: *mut c_void);
    fn ddo_cache_clear(cache: *mut c_void) -> c_int;
    fn ddo_cache_get_threshold(cache: *const c_void, state: *const u64, depth: usize, value: *mut i64, explored: *mut c_int) -> c_int;
    fn ddo_cache_update_threshold(cache: *mut c_void, state: *const u64, depth: usize, value: i64, explored: c_int) -> c_int;
    fn ddo_mdd_create(model: *const c_void, device: c_int, cutset_type: c_int, max_width: usize) -> *mut c_void;
    fn ddo_mdd_destroy(mdd: *mut c_void);
    fn ddo_mdd_compile(mdd: *mut c_void, input: *const DdoCompileInput, out: *mut DdoCompletion) -> c_int;
    fn ddo_mdd_is_exact(mdd: *const c_void) -> c_int;
    fn ddo_mdd_best_value(mdd: *const c_void, v: *mut i64) -> c_int;
    fn ddo_mdd_best_exact_value(mdd: *const c_void, v: *mut i64) -> c_int;
    fn ddo_mdd_best_solution(mdd: *const c_void, buf: *mut DdoDecision, len: *mut usize) -> c_int;
    fn ddo_mdd_best_exact_solution(mdd: *const c_void, buf: *mut DdoDecision, len: *mut usize) -> c_int;
    fn ddo_mdd_drain_cutset_rows(mdd: *mut c_void, ub_above: i64, rows: *mut DdoCutsetRows) -> c_int;
}

// ---- registry: what `D::default()` (parallel.rs:580) cannot be told through its signature ------------------------
pub struct Registry { model: usize, cache: usize, dominance: usize, device: c_int, cutset_type: c_int, max_width: usize, nb_vars: usize,
                      words: usize, stop: Arc<AtomicI32> }
impl Drop for Registry {             // the last HipMdd / HipCache user gone: the device objects go with it
    fn drop(&mut self) {
        unsafe {
            if self.cache != 0 { ddo_cache_destroy(self.cache as *mut c_void); }
            ddo_model_destroy(self.model as *mut c_void);
        }
    }
}
// One current registry per process; `install` replaces it (a test binary solves one instance after the other), every HipMdd keeps
// the registry it was created under alive.
static REGISTRY: RwLock<Option<Arc<Registry>>> = RwLock::new(None);

/// Call in `main` before the solver is built (again for another instance, once the previous solver is gone): describes the MISP instance to the device (the fields of
/// `Misp { nb_vars, neighbors, weight }`, examples/misp/main.rs:37-51) and fixes device, cut-set type and the largest
/// width any compile will ask for.  `cache_entries > 0` creates the device-side SimpleCache the `HipCache` below wraps.
pub fn install(nb_vars: usize, neighbors: &[BitSet], weight: &[isize], device: i32, frontier: bool, max_width: usize, cache_entries: usize) {
    let cutset_type = (if frontier { DDO_FRONTIER } else { DDO_LAST_EXACT_LAYER }) | (if cache_entries > 0 { DDO_MDD_CACHING } else { 0 });
    install_with(nb_vars, neighbors, weight, device, cutset_type, max_width, cache_entries)
}
/// The same for `Pooled` decision diagrams (implementation/mdd/pooled.rs; the four `*SolverPooled` aliases, solver/mod.rs:34, :38, :43,
/// :47): every `HipMdd` created afterwards compiles pooled DDs on the device -- use `ParallelSolver::<BitSet, HipMdd, EmptyCache<BitSet>>`
/// exactly as with `install`, or, with `cache_entries > 0`, `ParallelSolver::<BitSet, HipMdd, HipCache>` (Pooled behind the device-side
/// SimpleCache: `_filter_with_cache` / `_compute_thresholds` over the long arcs, pooled.rs:467-535, 662-680).
pub fn install_pooled(nb_vars: usize, neighbors: &[BitSet], weight: &[isize], device: i32, max_width: usize, cache_entries: usize) {
    let cutset_type = DDO_FRONTIER | DDO_MDD_POOLED | (if cache_entries > 0 { DDO_MDD_CACHING } else { 0 });
    install_with(nb_vars, neighbors, weight, device, cutset_type, max_width, cache_entries)
}
fn install_with(nb_vars: usize, neighbors: &[BitSet], weight: &[isize], device: i32, cutset_type: c_int, max_width: usize, cache_entries: usize) {
    let words = (nb_vars + 63) / 64;
    let mut rows = vec![0u64; nb_vars * words];
    for (i, nb) in neighbors.iter().enumerate() { for j in nb.iter() { rows[i * words + j / 64] |= 1u64 << (j % 64); } }
    let w: Vec<i64> = weight.iter().map(|x| *x as i64).collect();
    let model = unsafe { ddo_model_create_misp(nb_vars as c_int, rows.as_ptr(), w.as_ptr()) };
    assert!(!model.is_null(), "ddo_model_create_misp: {:?}", unsafe { std::ffi::CStr::from_ptr(ddo_last_error()) });
    let cache = if cache_entries > 0 { unsafe { ddo_cache_create(model, device, cache_entries) } } else { std::ptr::null_mut() };
    // (MISP has no dominance relation: EmptyDominanceChecker == null; a knapsack / TSPTW shim would call ddo_dominance_create here)
    *REGISTRY.write().unwrap() = Some(Arc::new(Registry { model: model as usize, cache: cache as usize, dominance: 0, device, cutset_type,
                                                          max_width, nb_vars, words, stop: Arc::new(AtomicI32::new(0)) }));
}

/// `TimeBudget` (cutoff.rs:302-323) whose flag the device engine can watch: `&dyn Cutoff` only answers `must_stop()`, it cannot
/// hand out an address, so the flag lives in the registry -- the timer thread raises it, `HipMdd::compile` passes its address as
/// `ddo_compile_input.cutoff`, and the engine polls it WHILE the compile runs (clean.rs:352: a budget that expires mid-compile
/// ends it with `Reason::CutoffOccurred`).  Any other `Cutoff` still works: `compile` also folds `must_stop()` into the flag.
pub struct HipTimeBudget { stop: Arc<AtomicI32> }
impl HipTimeBudget {
    pub fn new(budget: std::time::Duration) -> Self {            // after hip_mdd::install
        let stop = Arc::clone(&registry().stop);
        let t_flag = Arc::clone(&stop);
        std::thread::spawn(move || { std::thread::sleep(budget); t_flag.store(1, Ordering::Relaxed); });
        HipTimeBudget { stop }
    }
}
impl Cutoff for HipTimeBudget { fn must_stop(&self) -> bool { self.stop.load(Ordering::Relaxed) != 0 } }
fn registry() -> Arc<Registry> { REGISTRY.read().unwrap().clone().expect("hip_mdd::install(...) must run before the solver is created") }

fn words_of(s: &BitSet, words: usize) -> Vec<u64> {          // bit i of word i/64 <=> vertex i (include/ddo_hip.h)
    let mut w = vec![0u64; words];
    for i in s.iter() { w[i / 64] |= 1u64 << (i % 64); }
    w
}
fn bitset_of(words: &[u64], nb_vars: usize) -> BitSet {
    let mut s = BitSet::with_capacity(nb_vars);
    for (k, w) in words.iter().enumerate() { let mut x = *w; while x != 0 { s.insert(k * 64 + x.trailing_zeros() as usize); x &= x - 1; } }
    s
}
fn decisions_of(buf: &[DdoDecision]) -> Vec<Decision> {
    buf.iter().map(|d| Decision { variable: Variable(d.variable as usize), value: d.value as isize }).collect()
}

// ---- `D` of ParallelSolver<'a, BitSet, HipMdd, C> ------------------------------------------------------------------
// `ready`: the cut-set of the latest relaxed compile, ALREADY turned into SubProblems -- by compile(), in the worker's own thread.
// The reference drains a cut-set inside the solver's critical section (parallel.rs:456-469 holds the mutex across
// `mdd.drain_cutset(..)`): whatever drain_cutset does per node is serialised over all worker threads, so it only moves these out.
pub struct HipMdd { h: *mut c_void, r: Arc<Registry>, ready: Vec<SubProblem<BitSet>> }
unsafe impl Send for HipMdd {}

impl Default for HipMdd {            // parallel.rs:580  `let mut mdd = D::default();` -- one per worker thread
    fn default() -> Self {
        let r = registry();
        let h = unsafe { ddo_mdd_create(r.model as *const c_void, r.device, r.cutset_type, r.max_width) };
        assert!(!h.is_null(), "ddo_mdd_create: {:?}", unsafe { std::ffi::CStr::from_ptr(ddo_last_error()) });
        HipMdd { h, r, ready: vec![] }
    }
}
impl Drop for HipMdd { fn drop(&mut self) { unsafe { ddo_mdd_destroy(self.h) } } }

impl HipMdd {
    fn solution(&self, f: unsafe extern "C" fn(*const c_void, *mut DdoDecision, *mut usize) -> c_int) -> Option<Solution> {
        let mut buf = vec![DdoDecision::default(); 2 * self.r.nb_vars + 8];
        let mut len = buf.len();
        match unsafe { f(self.h, buf.as_mut_ptr(), &mut len) } {
            1 => Some(decisions_of(&buf[..len])),       // residual path first, then the DD's best-edge chain (clean.rs:329-343)
            0 => None,
            e => panic!("ddo_hip solution query failed: {e}"),
        }
    }
}

/// One FFI call for the whole cut-set (ddo_mdd_drain_cutset_rows) instead of a callback per node: the rows become SubProblems here,
/// outside the solver's lock.  Nodes whose bound does not exceed `best_lb` stay on the other side -- both solvers' closures drop
/// them (`if cutset_node.ub > best_lb`, parallel.rs:461-463, sequential.rs:372-376; the bound only rises until the drain).
fn fetch_cutset(h: *mut c_void, best_lb: isize, head: &[Decision], nb_vars: usize) -> Vec<SubProblem<BitSet>> {
    let mut rows = DdoCutsetRows { count: 0, state_words: 0, path_stride: 0, states: std::ptr::null(), values: std::ptr::null(),
                                   ubs: std::ptr::null(), depths: std::ptr::null(), path_lens: std::ptr::null(), paths: std::ptr::null() };
    let rc = unsafe { ddo_mdd_drain_cutset_rows(h, best_lb as i64, &mut rows) };
    assert!(rc == 0, "ddo_mdd_drain_cutset_rows failed: {rc}");
    (0..rows.count).map(|i| unsafe {
        let words = std::slice::from_raw_parts(rows.states.add(i * rows.state_words), rows.state_words);
        let part = std::slice::from_raw_parts(rows.paths.add(i * rows.path_stride), *rows.path_lens.add(i));
        let mut path = Vec::with_capacity(head.len() + part.len());
        path.extend_from_slice(head);                       // the residual's own path first (clean.rs:329-343) ...
        path.extend(part.iter().map(|d| Decision { variable: Variable(d.variable as usize), value: d.value as isize }));   // ... then the DD's part
        SubProblem { state: Arc::new(bitset_of(words, nb_vars)), value: *rows.values.add(i) as isize, path,
                     ub: *rows.ubs.add(i) as isize, depth: *rows.depths.add(i) }
    }).collect()
}

impl DecisionDiagram for HipMdd {
    type State = BitSet;
    fn compile(&mut self, input: &CompilationInput<BitSet>) -> Result<Completion, Reason> {
        let r = Arc::clone(&self.r);
        let st = words_of(&input.residual.state, r.words);
        let path: Vec<DdoDecision> = input.residual.path.iter()
            .map(|d| DdoDecision { variable: d.variable.id() as i64, value: d.value as i64 }).collect();
        if input.cutoff.must_stop() { r.stop.store(1, Ordering::Relaxed); }   // a Cutoff other than HipTimeBudget: its answer at call time
        let ci = DdoCompileInput {
            comp_type: match input.comp_type { CompilationType::Exact => 0, CompilationType::Relaxed => 1,
                                               CompilationType::Restricted => 2 },
            max_width: input.max_width, best_lb: input.best_lb as i64,
            residual: DdoSubProblem { state: st.as_ptr(), state_words: st.len(), value: input.residual.value as i64,
                                      ub: input.residual.ub as i64, depth: input.residual.depth,
                                      path: if path.is_empty() { std::ptr::null() } else { path.as_ptr() }, path_len: path.len() },
            cutoff: r.stop.as_ptr() as *const c_int,   // watched by the engine during the launch (heuristics.rs:100-105, clean.rs:352)
            cache: r.cache as *mut c_void,        // null == EmptyCache; else the table HipCache wraps (same thresholds on both sides)
            dominance: r.dominance as *mut c_void };   // null == EmptyDominanceChecker
        let mut out = DdoCompletion::default();
        self.ready.clear();
        match unsafe { ddo_mdd_compile(self.h, &ci, &mut out) } {
            0 => {
                if matches!(input.comp_type, CompilationType::Relaxed) && out.is_exact == 0 {
                    self.ready = fetch_cutset(self.h, input.best_lb, &input.residual.path, r.nb_vars);
                }
                Ok(Completion { is_exact: out.is_exact != 0,
                                best_value: if out.has_best_value != 0 { Some(out.best_value as isize) } else { None } })
            }
            2 => Err(Reason::CutoffOccurred),                            // DDO_CUTOFF
            // (3 == DDO_HANDED_UP is answered only by mdds bound to a capacity tier with DDO_MDD_ENGINE_*: the shim creates none)
            e => panic!("ddo_hip error {e}: {:?}", unsafe { std::ffi::CStr::from_ptr(ddo_last_error()) }),   // the reference aborts on internal errors too
        }
    }
    fn is_exact(&self) -> bool { unsafe { ddo_mdd_is_exact(self.h) != 0 } }
    fn best_value(&self) -> Option<isize> { let mut v = 0i64; (unsafe { ddo_mdd_best_value(self.h, &mut v) } == 1).then(|| v as isize) }
    fn best_exact_value(&self) -> Option<isize> { let mut v = 0i64; (unsafe { ddo_mdd_best_exact_value(self.h, &mut v) } == 1).then(|| v as isize) }
    fn best_solution(&self) -> Option<Solution> { self.solution(ddo_mdd_best_solution) }
    fn best_exact_solution(&self) -> Option<Solution> { self.solution(ddo_mdd_best_exact_solution) }
    fn drain_cutset<F: FnMut(SubProblem<BitSet>)>(&mut self, mut func: F) {      // mdd.rs:107-113: at most once per relaxed compile
        for sp in self.ready.drain(..) { func(sp) }                                // (called under the solver's lock: nothing is built here)
    }
}

// ---- `C` of ParallelSolver<'a, BitSet, HipMdd, HipCache>: the SimpleCache living in device memory -----------------
// The compiles read and write the table on the device (_filter_with_cache, _maybe_update_cache, clean.rs:534-545, 710-726);
// the solver's own calls (parallel.rs:537-549 must_explore / update_threshold) reach the same table through the host views.
#[derive(Default)] pub struct HipCache;
impl Cache for HipCache {
    type State = BitSet;
    fn initialize(&mut self, _problem: &dyn Problem<State = BitSet>) {}            // created by install()
    fn get_threshold(&self, state: &BitSet, depth: usize) -> Option<Threshold> {
        let r = registry();
        if r.cache == 0 { return None; }
        let (st, mut v, mut e) = (words_of(state, r.words), 0i64, 0 as c_int);
        (unsafe { ddo_cache_get_threshold(r.cache as *const c_void, st.as_ptr(), depth, &mut v, &mut e) } == 1)
            .then(|| Threshold { value: v as isize, explored: e != 0 })
    }
    fn update_threshold(&self, state: Arc<BitSet>, depth: usize, value: isize, explored: bool) {
        let r = registry();
        if r.cache != 0 {
            let st = words_of(&state, r.words);
            unsafe { ddo_cache_update_threshold(r.cache as *mut c_void, st.as_ptr(), depth, value as i64, explored as c_int) };
        }
    }
    fn clear_layer(&self, _depth: usize) {}   // memory management only in the reference (parallel.rs:506-511): no sub-problem of a
                                              // cleared depth can appear again, the device table keeps the entries until clear()
    fn clear(&self) { let r = registry(); if r.cache != 0 { unsafe { ddo_cache_clear(r.cache as *mut c_void) }; } }
}

// examples/misp/main.rs stays as it is, apart from one call and two type arguments:
//   hip_mdd::install(problem.nb_vars, &problem.neighbors, &problem.weight, /*device*/ 0, /*frontier*/ false, max_width, /*cache*/ 0);
//   let mut solver = ParallelSolver::<BitSet, HipMdd, EmptyCache<BitSet>>::custom(
//       &problem, &relaxation, &ranking, width.as_ref(), &dominance, cutoff.as_ref(), &mut fringe, nb_threads);
// and with the cache:  install(.., /*frontier*/ true, max_width, 1 << 22);  ParallelSolver::<BitSet, HipMdd, HipCache>::custom(..)
