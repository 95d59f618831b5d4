//! GPU (MI355X / gfx950) implementations of the local-search solvers, behind the crate feature `gpu`.
//!
//! Each `solve` below has exactly the signature of the function it replaces and is selected by the same `match` in
//! `solve_with_context` (src/tsp/mod.rs) when the feature is on:
//!   `gpu::two_opt::solve`        <- `two_opt::solve`        (two_opt.rs:7-12)
//!   `gpu::three_opt::solve`      <- `three_opt::solve`      (three_opt.rs:16-21)
//!   `gpu::lin_kernighan::solve`  <- `lin_kernighan::solve`  (lin_kernighan.rs:35-40)
//!   `gpu::or_opt::solve`, `gpu::nearest_neighbor::solve` likewise.
//! The work is done by libteeline_gpu.so through the `teeline-gpu` crate; this file only translates
//! ids <-> positions ONCE per call (the CPU path does it with two HashMap lookups inside every distance call,
//! distance_matrix.rs:197-212) and builds the `Solution` with the crate's own `Solution::new`, so the printed total is
//! computed by `DistanceMatrix::tour_length` itself.
//!
//! Behaviour that differs from the CPU functions, all of it documented in INTEGRATION.md:
//!   * progress: every solver sends the reference's exact message sequence — CityChange per outer i and PathUpdate per move
//!     (2-opt), PathUpdate per applied move (3-opt, or-opt), PathUpdate per best tour (LK), CityChange + growing path per step (NN) —
//!     replayed from the records of `tl_two_opt_trace` / `tl_three_opt_trace` / `tl_or_opt_trace` (LK: sent live through `tl_lk_live`; NN: from the
//!     finished walk) AFTER the kernels return, not while they run — only the Qt front-end passes a sender;
//!   * a library error (no gfx950 device, HIP failure) panics with the library's message, like the `.expect(..)`s of the CPU
//!     code do on bad input: the solver functions are infallible by signature;
//!   * Lin-Kernighan kicks: the seed of the device-side splitmix64 stream is drawn from `rand::rng()` per call (the CPU
//!     code draws every kick from it), or taken from `TEELINE_GPU_LK_SEED` for reproducible runs.

use std::sync::mpsc;

use teeline_gpu as ffi;

use super::progress::ProgressMessage;
use super::route::Route;
use super::{HeuristicOptions, LKOptions, Solution, TspProblem};

/// City coordinates in matrix-position order plus the matrix the kernels should read (None = EUC_2D on the fly).
struct Boundary<'a> {
    problem: &'a TspProblem,
    xy: Vec<f32>,
}

impl<'a> Boundary<'a> {
    fn new(problem: &'a TspProblem) -> Self {
        // `problem.cities` is the slice the matrix was built from (tsplib.rs:84-98), so index == matrix position;
        // checked here because everything below relies on it.
        let mut xy = Vec::with_capacity(problem.cities.len() * 2);
        for (pos, c) in problem.cities.iter().enumerate() {
            assert_eq!(problem.distances.city_id2pos(c.id), Some(pos), "cities order differs from the distance matrix");
            xy.push(c.coords[0]);
            xy.push(c.coords[1]);
        }
        Boundary { problem, xy }
    }

    /// ids -> positions; an unknown id panics with the message of the CPU solver's `.expect(..)`.
    fn positions(&self, tour: &[usize], what: &str) -> Vec<u32> {
        tour.iter()
            .map(|&id| self.problem.distances.city_id2pos(id).unwrap_or_else(|| panic!("{what}")) as u32)
            .collect()
    }

    fn ids(&self, pos: &[u32]) -> Vec<usize> {
        pos.iter().map(|&p| self.problem.cities[p as usize].id).collect()
    }

    /// `DistanceMatrix` does not remember its DistanceType (distance_matrix.rs:86-93): ask the library whether the matrix
    /// is exactly the EUC_2D matrix of the coordinates.  If so the coordinate kernels apply (same tours, no 4n^2-byte
    /// expansion); otherwise (GEO, EXPLICIT) the matrix kernels read `problem.distances`.
    fn matrix<'b>(&'b self, ctx: &ffi::Context) -> Result<Option<&'b [f32]>, ffi::Error> {
        let dm = self.problem.distances.distances();
        Ok(if ctx.dm_is_euc2d(&self.xy, dm)? { None } else { Some(dm) })
    }
}

fn unwrap_gpu<T>(what: &str, r: Result<T, ffi::Error>) -> T {
    r.unwrap_or_else(|e| panic!("{what}: {e}"))
}

fn start_tour(problem: &TspProblem, init_tour: Option<&[usize]>) -> Vec<usize> {
    init_tour.map(|t| t.to_vec()).unwrap_or_else(|| problem.cities.iter().map(|c| c.id).collect())
}

fn finish(problem: &TspProblem, io: &Boundary, t: &ffi::Tour, tx: Option<&mpsc::Sender<ProgressMessage>>, done: bool) -> Solution {
    let route = io.ids(&t.pos);
    let sol = Solution::new(&route, problem);
    debug_assert_eq!(sol.total.to_bits(), t.cost.to_bits(), "device total differs from DistanceMatrix::tour_length");
    if let Some(tx) = tx {
        let _ = tx.send(ProgressMessage::PathUpdate(Route::new(&route), sol.total));
        if done {
            let _ = tx.send(ProgressMessage::Done);
        }
    }
    sol
}

pub mod two_opt {
    use super::*;

    pub fn solve(
        problem: &TspProblem,
        _opts: &HeuristicOptions,
        progress_tx: Option<&mpsc::Sender<ProgressMessage>>,
        init_tour: Option<&[usize]>,
    ) -> Solution {
        tracing::info!(cities = problem.cities.len(), "2-opt (gpu) starting");
        let io = Boundary::new(problem);
        let init = init_tour.map(|t| io.positions(t, "two_opt: invalid city pair"));
        if let Some(tx) = progress_tx {
            let _ = tx.send(ProgressMessage::PathUpdate(Route::new(&start_tour(problem, init_tour)), 0.0));
        }
        // With a progress channel (only teeline-qt passes one) the descent also lists its moves, and the reference's exact
        // message sequence is replayed from them once the kernel is back (the descent is ONE launch: nothing can be sent
        // while it runs).  Coordinate tours beyond the LDS-resident descent fall back to the final PathUpdate.
        if let Some(tx) = progress_tx {
            let traced = ffi::with_context(|ctx| {
                let dm = io.matrix(ctx)?;
                ctx.two_opt_trace(&io.xy, dm, init.as_deref())
            });
            if let Ok((t, moves)) = traced {
                replay_progress(problem, &io, init.as_deref(), &moves, t.stats.sweeps, tx);
                let route = io.ids(&t.pos);
                let sol = Solution::new(&route, problem);
                debug_assert_eq!(sol.total.to_bits(), t.cost.to_bits(), "device total differs from DistanceMatrix::tour_length");
                let _ = tx.send(ProgressMessage::Done);
                return sol;
            }
        }
        let t = unwrap_gpu("two_opt (gpu)", ffi::with_context(|ctx| {
            let dm = io.matrix(ctx)?;
            ctx.two_opt(&io.xy, dm, init.as_deref(), ffi::MODE_REF_ORDER)
        }));
        finish(problem, &io, &t, progress_tx, true)
    }

    /// two_opt.rs:26-61 as messages: per sweep and outer `i` a `CityChange(path[i])` (:30-32); per move `(i, j)` the reversal
    /// of `path[i+1..=j]` and `PathUpdate(path, new_distance)` with `new_distance = d(p[i],p[j]) + d(p[i+1],p[j+1])` taken on the
    /// path before the move through `problem.distances` (:42-56) — the f32 sum the reference has just compared.
    fn replay_progress(problem: &TspProblem, io: &Boundary, init: Option<&[u32]>, moves: &[Option<(u32, u32)>], sweeps: u64,
                       tx: &mpsc::Sender<ProgressMessage>) {
        let n = problem.cities.len();
        let mut pos: Vec<u32> = init.map(|p| p.to_vec()).unwrap_or_else(|| (0..n as u32).collect());
        let d = |p: u32, q: u32| problem.distances.distance_by_pos(p as usize, q as usize).expect("two_opt: invalid city pair");
        let mut k = 0usize;
        for _ in 0..sweeps {
            for i in 0..n.saturating_sub(3) {
                let _ = tx.send(ProgressMessage::CityChange(io.ids(&pos[i..=i])[0]));
                while let Some(Some((mi, mj))) = moves.get(k).copied() {
                    if mi as usize != i {
                        break;
                    }
                    let j = mj as usize;
                    let new_distance = d(pos[i], pos[j]) + d(pos[i + 1], pos[j + 1]);
                    pos[i + 1..=j].reverse();
                    let _ = tx.send(ProgressMessage::PathUpdate(Route::new(&io.ids(&pos)), new_distance));
                    k += 1;
                }
            }
            if let Some(None) = moves.get(k) {
                k += 1; // a new sweep begins
            }
        }
        debug_assert_eq!(k, moves.len());
    }
}

pub mod three_opt {
    use super::*;

    pub fn solve(
        problem: &TspProblem,
        _opts: &HeuristicOptions,
        progress_tx: Option<&mpsc::Sender<ProgressMessage>>,
        init_tour: Option<&[usize]>,
    ) -> Solution {
        tracing::info!(cities = problem.cities.len(), "3-opt (gpu) starting");
        let io = Boundary::new(problem);
        let init = init_tour.map(|t| io.positions(t, "three_opt: invalid city pair"));
        let n = problem.cities.len();
        // With a progress channel the solve also lists its moves, and the reference's message sequence — the start path, the path
        // after every apply_3opt (each with 0.0), Done (three_opt.rs:34,42,47-49; nothing when n < 4, :25-28) — is replayed from them.
        if let (Some(tx), true) = (progress_tx, n >= 4) {
            let (t, moves) = unwrap_gpu("three_opt (gpu)", ffi::with_context(|ctx| {
                let dm = io.matrix(ctx)?;
                ctx.three_opt_trace(&io.xy, dm, init.as_deref())
            }));
            let mut pos: Vec<u32> = init.clone().unwrap_or_else(|| (0..n as u32).collect());
            let _ = tx.send(ProgressMessage::PathUpdate(Route::new(&io.ids(&pos)), 0.0));
            for m in &moves {
                apply_3opt(&mut pos, m[0] as usize, m[1] as usize, m[2] as usize, m[3]);
                let _ = tx.send(ProgressMessage::PathUpdate(Route::new(&io.ids(&pos)), 0.0));
            }
            let _ = tx.send(ProgressMessage::Done);
            return finish(problem, &io, &t, None, false);
        }
        let t = unwrap_gpu("three_opt (gpu)", ffi::with_context(|ctx| {
            let dm = io.matrix(ctx)?;
            ctx.three_opt(&io.xy, dm, init.as_deref())
        }));
        finish(problem, &io, &t, None, false)
    }

    /// three_opt.rs:186-218 on positions: cases 1-3 reverse segments, 4-7 swap path[i+1..=j] and path[j+1..=k] with either reversed.
    fn apply_3opt(path: &mut [u32], i: usize, j: usize, k: usize, case: u32) {
        match case {
            1 => path[i + 1..=j].reverse(),
            2 => path[j + 1..=k].reverse(),
            3 => {
                path[i + 1..=j].reverse();
                path[j + 1..=k].reverse();
            }
            4..=7 => {
                let mut seg1 = path[i + 1..=j].to_vec();
                let mut seg2 = path[j + 1..=k].to_vec();
                if case == 5 || case == 7 {
                    seg1.reverse();
                }
                if case == 6 || case == 7 {
                    seg2.reverse();
                }
                seg2.extend_from_slice(&seg1);
                path[i + 1..=k].copy_from_slice(&seg2);
            }
            _ => unreachable!("apply_3opt: case must be 1-7, got {case}"),
        }
    }
}

pub mod or_opt {
    use super::*;

    pub fn solve(
        problem: &TspProblem,
        _opts: &HeuristicOptions,
        progress_tx: Option<&mpsc::Sender<ProgressMessage>>,
        init_tour: Option<&[usize]>,
    ) -> Solution {
        let io = Boundary::new(problem);
        let init = init_tour.map(|t| io.positions(t, "or_opt: invalid city pair"));
        let n = problem.cities.len();
        // With a progress channel the solve also lists its moves, and the reference's message sequence — the start path (0.0), the
        // path and its tour_length after every apply_relocation, Done (or_opt.rs:40-42,62-67,70-72; nothing when n < 4, :31-34) —
        // is replayed from them.
        if let (Some(tx), true) = (progress_tx, n >= 4) {
            let (t, moves) = unwrap_gpu("or_opt (gpu)", ffi::with_context(|ctx| {
                let dm = io.matrix(ctx)?;
                ctx.or_opt_trace(&io.xy, dm, init.as_deref())
            }));
            let mut pos: Vec<u32> = init.clone().unwrap_or_else(|| (0..n as u32).collect());
            let _ = tx.send(ProgressMessage::PathUpdate(Route::new(&io.ids(&pos)), 0.0));
            for m in &moves {
                apply_relocation(&mut pos, m[0] as usize, m[2] as usize, m[1] as usize, m[3] != 0);
                let route = io.ids(&pos);
                let len = problem.distances.tour_length(&route);
                let _ = tx.send(ProgressMessage::PathUpdate(Route::new(&route), len));
            }
            let _ = tx.send(ProgressMessage::Done);
            return finish(problem, &io, &t, None, false);
        }
        let t = unwrap_gpu("or_opt (gpu)", ffi::with_context(|ctx| {
            let dm = io.matrix(ctx)?;
            ctx.or_opt(&io.xy, dm, init.as_deref())
        }));
        finish(problem, &io, &t, None, false)
    }

    /// or_opt.rs:172-184 on positions.
    fn apply_relocation(tour: &mut Vec<u32>, i: usize, seg_len: usize, j: usize, reversed: bool) {
        let seg: Vec<u32> = tour.drain(i..i + seg_len).collect();
        let insert_at = if j >= i + seg_len { j - seg_len + 1 } else { j + 1 };
        if reversed {
            tour.splice(insert_at..insert_at, seg.into_iter().rev());
        } else {
            tour.splice(insert_at..insert_at, seg);
        }
    }
}

pub mod nearest_neighbor {
    use super::*;

    pub fn solve(
        problem: &TspProblem,
        opts: &HeuristicOptions,
        progress_tx: Option<&mpsc::Sender<ProgressMessage>>,
        _init_tour: Option<&[usize]>,
    ) -> Solution {
        let io = Boundary::new(problem);
        let t = unwrap_gpu("nearest_neighbor (gpu)", ffi::with_context(|ctx| {
            let dm = io.matrix(ctx)?;
            ctx.nearest_neighbor(&io.xy, dm, opts.n_nearest as u32)
        }));
        // the reference's messages follow from the finished walk (nearest_neighbor.rs:32-34,40-42,67-69,72-74): the start city,
        // then per step CityChange(current city) and the path so far, then Done
        if let Some(tx) = progress_tx {
            let route = io.ids(&t.pos);
            if !route.is_empty() {
                let _ = tx.send(ProgressMessage::PathUpdate(Route::new(&route[..1]), 0.0));
                for k in 1..route.len() {
                    let _ = tx.send(ProgressMessage::CityChange(route[k - 1]));
                    let _ = tx.send(ProgressMessage::PathUpdate(Route::new(&route[..=k]), 0.0));
                }
                let _ = tx.send(ProgressMessage::Done);
            }
        }
        finish(problem, &io, &t, None, false)
    }
}

pub mod lin_kernighan {
    use super::*;
    use rand::RngExt;

    pub fn solve(
        problem: &TspProblem,
        opts: &LKOptions,
        progress_tx: Option<&mpsc::Sender<ProgressMessage>>,
        init_tour: Option<&[usize]>,
    ) -> Solution {
        let io = Boundary::new(problem);
        let init = init_tour.map(|t| io.positions(t, "lin_kernighan: invalid city id in init_tour"));
        let lk = ffi::LkOpts {
            epochs: opts.heuristic.epochs as u32,
            platoo_epochs: opts.heuristic.platoo_epochs as u32,
            n_nearest: opts.heuristic.n_nearest as u32,
            max_depth: opts.max_depth as u32,
        };
        let seed = std::env::var("TEELINE_GPU_LK_SEED")
            .ok()
            .and_then(|v| v.parse::<u64>().ok())
            .unwrap_or_else(|| rand::rng().random::<u64>());
        // lin_kernighan.rs sends PathUpdate(best_tour, best_dist) after the first lk_pass and for every improving epoch, never Done
        // (:71,:90; nothing when n < 4, :57-59): with a channel the device-side state machine files exactly those and tl_lk_live hands
        // them over WHILE the search runs — teeline-qt sees a multi-second run progress as it does with the CPU solver.
        if let (Some(tx), true) = (progress_tx, problem.cities.len() >= 4) {
            let t = unwrap_gpu("lin_kernighan (gpu)", ffi::with_context(|ctx| {
                let dm = io.matrix(ctx)?;
                ctx.lin_kernighan_live(&io.xy, dm, init.as_deref(), lk, seed, |pos, dist| {
                    let _ = tx.send(ProgressMessage::PathUpdate(Route::new(&io.ids(pos)), dist));
                })
            }));
            return finish(problem, &io, &t, None, false);
        }
        let t = unwrap_gpu("lin_kernighan (gpu)", ffi::with_context(|ctx| {
            // problem.distances feeds the NN seed and the total only; the search is Euclidean (lin_kernighan.rs:41,47-55,99)
            let dm = io.matrix(ctx)?;
            ctx.lin_kernighan(&io.xy, dm, init.as_deref(), lk, seed)
        }));
        finish(problem, &io, &t, None, false)
    }
}
