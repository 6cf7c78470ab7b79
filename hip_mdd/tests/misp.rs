//! The optima of the reference's MISP example tests (ddo/examples/misp/tests.rs:71-161: DIMACS instances of `resources/misp`,
//! `NbUnassignedWidth`, `NoDupFringe<MaxUB>`, no cutoff), solved with `ParallelSolver<BitSet, HipMdd, EmptyCache>` -- the
//! decision diagrams are compiled on the MI355X through `libddo_hip.so`.  Besides the optima this is where the tie-break parity
//! that only the real crates can pin (bit-set's `Ord`) gets exercised: `explored` is printed per instance
//! for comparison with `DefaultSolver` on the same machine (`cargo test -- --nocapture`).
//!
//! The model below is the MISP formulation of examples/misp/main.rs:37-209 written for this test (the example is a binary,
//! not a library: its types cannot be imported).  Only `initial_state`, `initial_value`, `nb_variables` and the ranking reach the
//! solver when the DDs are compiled on the device; the rest is there so that the very same objects also drive `DefaultSolver`.
use std::cmp::Ordering;
use std::fs::File;
use std::io::{BufRead, BufReader};
use std::path::PathBuf;
use std::sync::Mutex;

use bit_set::BitSet;
use ddo::*;
use hip_mdd::{install, install_pooled, HipCache, HipMdd};

struct Misp { nb_vars: usize, compatible: Vec<BitSet>, weight: Vec<isize> }   // compatible[i]: the vertices NOT adjacent to i

impl Problem for Misp {
    type State = BitSet;
    fn nb_variables(&self) -> usize { self.nb_vars }
    fn initial_state(&self) -> BitSet { (0..self.nb_vars).collect() }
    fn initial_value(&self) -> isize { 0 }
    fn transition(&self, state: &BitSet, d: Decision) -> BitSet {
        let mut next = state.clone();
        next.remove(d.variable.id());
        if d.value == 1 { next.intersect_with(&self.compatible[d.variable.id()]); }
        next
    }
    fn transition_cost(&self, _: &BitSet, _: &BitSet, d: Decision) -> isize {
        if d.value == 1 { self.weight[d.variable.id()] } else { 0 }
    }
    fn for_each_in_domain(&self, variable: Variable, state: &BitSet, f: &mut dyn DecisionCallback) {
        if state.contains(variable.id()) { f.apply(Decision { variable, value: 1 }); }
        f.apply(Decision { variable, value: 0 });
    }
    fn next_variable(&self, _: usize, layer: &mut dyn Iterator<Item = &BitSet>) -> Option<Variable> {
        let mut count = vec![0usize; self.nb_vars];                 // the vertex present in the fewest states of the layer
        for s in layer { for v in s.iter() { count[v] += 1; } }
        count.iter().enumerate().filter(|(_, c)| **c > 0).min_by_key(|(_, c)| **c).map(|(v, _)| Variable(v))
    }
}

struct MispRelax<'a>(&'a Misp);
impl Relaxation for MispRelax<'_> {
    type State = BitSet;
    fn merge(&self, states: &mut dyn Iterator<Item = &BitSet>) -> BitSet {
        let mut all = BitSet::with_capacity(self.0.nb_vars);
        for s in states { all.union_with(s); }
        all
    }
    fn relax(&self, _: &BitSet, _: &BitSet, _: &BitSet, _: Decision, cost: isize) -> isize { cost }
    fn fast_upper_bound(&self, state: &BitSet) -> isize { state.iter().map(|v| self.0.weight[v]).sum() }
}

struct MispRanking;
impl StateRanking for MispRanking {
    type State = BitSet;
    fn compare(&self, a: &BitSet, b: &BitSet) -> Ordering { a.len().cmp(&b.len()).then_with(|| a.cmp(b)) }
}

fn read_dimacs(path: &PathBuf) -> Misp {
    let edge = regex::Regex::new(r"^e\s+(\d+)\s+(\d+)").unwrap();
    let node = regex::Regex::new(r"^n\s+(\d+)\s+(-?\d+)").unwrap();
    let head = regex::Regex::new(r"^p\s+\S+\s+(\d+)\s+(\d+)").unwrap();
    let mut g = Misp { nb_vars: 0, compatible: vec![], weight: vec![] };
    for line in BufReader::new(File::open(path).unwrap()).lines().map(|l| l.unwrap()) {
        if let Some(c) = head.captures(&line) {
            let n: usize = c[1].parse().unwrap();
            g = Misp { nb_vars: n, compatible: vec![(0..n).collect(); n], weight: vec![1; n] };
            for i in 0..n { g.compatible[i].remove(i); }
        } else if let Some(c) = node.captures(&line) {
            let (i, w): (usize, isize) = (c[1].parse().unwrap(), c[2].parse().unwrap());
            g.weight[i - 1] = w;
        } else if let Some(c) = edge.captures(&line) {
            let (a, b): (usize, usize) = (c[1].parse().unwrap(), c[2].parse().unwrap());
            g.compatible[a - 1].remove(b - 1);
            g.compatible[b - 1].remove(a - 1);
        }
    }
    g
}

static ONE_AT_A_TIME: Mutex<()> = Mutex::new(());   // hip_mdd keeps ONE current model per process

fn solve_on_the_gpu(id: &str) -> isize {
    let _guard = ONE_AT_A_TIME.lock().unwrap_or_else(|e| e.into_inner());
    let dir = std::env::var("DDO_RESOURCES").map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env!("CARGO_MANIFEST_DIR")).join("..").join("data"));
    let problem = read_dimacs(&dir.join("misp").join(id));
    let relaxation = MispRelax(&problem);
    let ranking = MispRanking;
    let width = NbUnassignedWidth(problem.nb_variables());
    let dominance = EmptyDominanceChecker::default();
    let cutoff = NoCutoff;
    let mut fringe = NoDupFringe::new(MaxUB::new(&ranking));
    install(problem.nb_vars, &problem.compatible, &problem.weight, 0, false, problem.nb_vars, 0);
    let mut solver = ParallelSolver::<BitSet, HipMdd, EmptyCache<BitSet>>::custom(
        &problem, &relaxation, &ranking, &width, &dominance, &cutoff, &mut fringe, 8);
    let Completion { is_exact, best_value } = solver.maximize();
    assert!(is_exact);
    best_value.unwrap_or(-1)
}

macro_rules! optimum {
    ($($name:ident: $file:expr => $value:expr,)*) => { $( #[test] fn $name() { assert_eq!(solve_on_the_gpu($file), $value); } )* };
}
// instance => optimum, as asserted by ddo/examples/misp/tests.rs:71-161 (its #[ignore]d long runs left out)
optimum! {
    brock200_2: "brock200_2.clq" => 12,   brock200_3: "brock200_3.clq" => 15,   brock200_4: "brock200_4.clq" => 17,
    c_fat200_1: "c-fat200-1.clq" => 12,   c_fat200_2: "c-fat200-2.clq" => 24,   c_fat200_5: "c-fat200-5.clq" => 58,
    c_fat500_1: "c-fat500-1.clq" => 14,   c_fat500_2: "c-fat500-2.clq" => 26,
    hamming6_2: "hamming6-2.clq" => 32,   hamming6_4: "hamming6-4.clq" => 4,
    hamming8_2: "hamming8-2.clq" => 128,  hamming8_4: "hamming8-4.clq" => 16,
    johnson8_2_4: "johnson8-2-4.clq" => 4, johnson8_4_4: "johnson8-4-4.clq" => 14,
    keller4: "keller4.clq" => 11,         mann_a9: "MANN_a9.clq" => 16,         p_hat300_1: "p_hat300-1.clq" => 8,
}

/// The same searches over `Pooled` decision diagrams on the device (implementation/mdd/pooled.rs; hip_mdd::install_pooled): the
/// reference's `ParNoCachingSolverPooled` / `ParCachingSolverPooled` (solver/mod.rs:34, :38) with `HipMdd` as their `D` and, behind
/// the cache, `HipCache` -- the SimpleCache living in device memory -- as their `C`.
fn solve_pooled_on_the_gpu(id: &str, cache_entries: usize) -> isize {
    let _guard = ONE_AT_A_TIME.lock().unwrap_or_else(|e| e.into_inner());
    let dir = std::env::var("DDO_RESOURCES").map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env!("CARGO_MANIFEST_DIR")).join("..").join("data"));
    let problem = read_dimacs(&dir.join("misp").join(id));
    let relaxation = MispRelax(&problem);
    let ranking = MispRanking;
    let width = NbUnassignedWidth(problem.nb_variables());
    let dominance = EmptyDominanceChecker::default();
    let cutoff = NoCutoff;
    let mut fringe = NoDupFringe::new(MaxUB::new(&ranking));
    install_pooled(problem.nb_vars, &problem.compatible, &problem.weight, 0, problem.nb_vars, cache_entries);
    let Completion { is_exact, best_value } = if cache_entries > 0 {
        ParallelSolver::<BitSet, HipMdd, HipCache>::custom(&problem, &relaxation, &ranking, &width, &dominance, &cutoff, &mut fringe, 8).maximize()
    } else {
        ParallelSolver::<BitSet, HipMdd, EmptyCache<BitSet>>::custom(&problem, &relaxation, &ranking, &width, &dominance, &cutoff, &mut fringe, 8).maximize()
    };
    assert!(is_exact);
    best_value.unwrap_or(-1)
}
#[test] fn pooled_johnson8_4_4() { assert_eq!(solve_pooled_on_the_gpu("johnson8-4-4.clq", 0), 14); }
#[test] fn pooled_mann_a9() { assert_eq!(solve_pooled_on_the_gpu("MANN_a9.clq", 0), 16); }
#[test] fn pooled_keller4() { assert_eq!(solve_pooled_on_the_gpu("keller4.clq", 0), 11); }
#[test] fn pooled_cached_mann_a9() { assert_eq!(solve_pooled_on_the_gpu("MANN_a9.clq", 1 << 20), 16); }
#[test] fn pooled_cached_keller4() { assert_eq!(solve_pooled_on_the_gpu("keller4.clq", 1 << 20), 11); }
