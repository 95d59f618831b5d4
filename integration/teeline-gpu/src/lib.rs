//! teeline-gpu — safe Rust binding of `libteeline_gpu.so` (C ABI declared in `include/teeline_gpu.h`, ABI version 5).
//!
//! The library is the MI355X (gfx950) engine that replaces the bodies of
//!   `two_opt::solve`        (src/tsp/two_opt.rs:7-67)
//!   `three_opt::solve`      (src/tsp/three_opt.rs:16-51)
//!   `lin_kernighan::solve`  (src/tsp/lin_kernighan.rs:35-100)
//!   `or_opt::solve`, `nearest_neighbor::solve`, `DistanceMatrix::build`
//! of the `teeline` crate.  This crate knows nothing about `teeline`'s types (no dependency cycle): tours are
//! POSITIONS (indices into the city array), coordinates are `[x0, y0, x1, y1, ...]`, the optional matrix is the
//! reference's packed strict lower triangle (`DistanceMatrix::distances()`, distance_matrix.rs:171-173).
//! `src/tsp/gpu.rs` (added to the `teeline` crate by integration/patches/0001-gpu-feature.patch) maps
//! `TspProblem` / `Solution` onto these calls with the reference's exact `solve` signatures.
//!
//! Threading: a `tl_ctx` is single-threaded; `with_context` keeps one per thread, which is what the reference's callers
//! need (teeline-api runs solvers on `spawn_blocking` threads, tsp_service.rs:295,328).
//! There is no CPU fallback: without a gfx950 device every call returns `Err(Error { code: NoDevice, .. })`.

use std::cell::RefCell;
use std::ffi::{c_char, c_int, c_void, CStr};
use std::fmt;
use std::ptr;

pub const TL_ABI_VERSION: c_int = 5;

#[repr(C)]
pub struct TlCtx {
    _private: [u8; 0],
}

/// `tl_stats` (include/teeline_gpu.h)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct Stats {
    pub sweeps: u64,
    pub candidates: u64,
    pub moves: u64,
    pub reversed: u64,
    pub kernel_ms: f64,
    pub total_ms: f64,
}

/// `tl_lk_opts` = LKOptions (src/tsp/mod.rs:1249-1267)
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct LkOpts {
    pub epochs: u32,
    pub platoo_epochs: u32,
    pub n_nearest: u32,
    pub max_depth: u32,
}

/// `tl_lk_progress_fn` (include/teeline_gpu.h): called by `tl_lk_live` on the calling thread while the search runs.
pub type LkProgressFn = unsafe extern "C" fn(user: *mut c_void, best_pos: *const u32, n: u32, best_dist: f32);

pub const MODE_REF_ORDER: c_int = 0;
pub const MODE_BEST_SWEEP: c_int = 1;
pub const DM_PACKED_LOWER: c_int = 0;
pub const DIST_EUC2D: c_int = 0;
pub const DIST_GEO: c_int = 1;

unsafe extern "C" {
    fn tl_abi_version() -> c_int;
    fn tl_create(device: c_int, flags: u32, out: *mut *mut TlCtx) -> c_int;
    fn tl_destroy(ctx: *mut TlCtx);
    fn tl_last_error(ctx: *const TlCtx) -> *const c_char;
    fn tl_dm_build(ctx: *mut TlCtx, xy: *const f32, n: u32, dist: c_int, layout: c_int, out_host: *mut f32, kernel_ms: *mut f64) -> c_int;
    fn tl_dm_is_euc2d(ctx: *mut TlCtx, xy: *const f32, dm_packed: *const f32, n: u32, is_euc2d: *mut c_int) -> c_int;
    fn tl_tour_length(ctx: *mut TlCtx, xy: *const f32, dm_packed: *const f32, n: u32, perm: *const u32, out_cost: *mut f32) -> c_int;
    fn tl_two_opt(ctx: *mut TlCtx, xy: *const f32, n: u32, dm_packed: *const f32, init_pos: *const u32, mode: c_int,
                  out_pos: *mut u32, out_cost: *mut f32, stats: *mut Stats) -> c_int;
    fn tl_two_opt_trace(ctx: *mut TlCtx, xy: *const f32, n: u32, dm_packed: *const f32, init_pos: *const u32, out_pos: *mut u32,
                        out_cost: *mut f32, stats: *mut Stats, move_log: *mut u32, log_cap: u32, log_len: *mut u32) -> c_int;
    fn tl_three_opt_trace(ctx: *mut TlCtx, xy: *const f32, n: u32, dm_packed: *const f32, init_pos: *const u32, out_pos: *mut u32,
                          out_cost: *mut f32, stats: *mut Stats, move_log: *mut u32, log_cap: u32, log_len: *mut u32) -> c_int;
    fn tl_or_opt_trace(ctx: *mut TlCtx, xy: *const f32, n: u32, dm_packed: *const f32, init_pos: *const u32, out_pos: *mut u32,
                       out_cost: *mut f32, stats: *mut Stats, move_log: *mut u32, log_cap: u32, log_len: *mut u32) -> c_int;
    fn tl_lk_trace(ctx: *mut TlCtx, xy: *const f32, n: u32, dm_packed: *const f32, init_pos: *const u32, opts: *const LkOpts, seed: u64,
                   out_pos: *mut u32, out_cost: *mut f32, stats: *mut Stats, snap_pos: *mut u32, snap_dist: *mut f32, snap_cap: u32,
                   snap_len: *mut u32) -> c_int;
    fn tl_lk_live(ctx: *mut TlCtx, xy: *const f32, n: u32, dm_packed: *const f32, init_pos: *const u32, opts: *const LkOpts, seed: u64,
                  out_pos: *mut u32, out_cost: *mut f32, stats: *mut Stats, progress: LkProgressFn, user: *mut c_void) -> c_int;
    fn tl_three_opt(ctx: *mut TlCtx, xy: *const f32, n: u32, dm_packed: *const f32, init_pos: *const u32,
                    out_pos: *mut u32, out_cost: *mut f32, stats: *mut Stats) -> c_int;
    fn tl_or_opt(ctx: *mut TlCtx, xy: *const f32, n: u32, dm_packed: *const f32, init_pos: *const u32,
                 out_pos: *mut u32, out_cost: *mut f32, stats: *mut Stats) -> c_int;
    fn tl_lk(ctx: *mut TlCtx, xy: *const f32, n: u32, dm_packed: *const f32, init_pos: *const u32, opts: *const LkOpts, seed: u64,
             out_pos: *mut u32, out_cost: *mut f32, stats: *mut Stats) -> c_int;
    fn tl_nearest_neighbor(ctx: *mut TlCtx, xy: *const f32, dm_packed: *const f32, n: u32, n_nearest: u32,
                           out_pos: *mut u32, out_cost: *mut f32) -> c_int;
    fn tl_two_opt_multistart_devices(ctxs: *const *mut TlCtx, n_ctxs: c_int, xy: *const f32, n: u32, seed: u64, first: u32, count: u32,
                                     mode: c_int, out_best_pos: *mut u32, out_best_cost: *mut f32, out_best_restart: *mut u32,
                                     out_costs: *mut f32, stats: *mut Stats) -> c_int;
}

/// `tl_status` (include/teeline_gpu.h)
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Code {
    BadArg,
    /// Input on which the reference solver itself panics (e.g. `two_opt::solve` with fewer than 3 cities underflows
    /// `n_indices - 2`, two_opt.rs:17,29); the shim turns this into `panic!`.
    RefPanics,
    NoDevice,
    Hip,
    NoMem,
    Unsupported,
    NoConverge,
    /// Another thread is inside a call with the same context (`with_context` keeps one per thread, so this crate never sees it).
    Busy,
    Other(i32),
}

impl Code {
    fn from_raw(rc: c_int) -> Code {
        match rc {
            -1 => Code::BadArg,
            -2 => Code::RefPanics,
            -3 => Code::NoDevice,
            -4 => Code::Hip,
            -5 => Code::NoMem,
            -6 => Code::Unsupported,
            -7 => Code::NoConverge,
            -8 => Code::Busy,
            x => Code::Other(x),
        }
    }
}

#[derive(Clone, Debug)]
pub struct Error {
    pub code: Code,
    pub message: String,
}

impl fmt::Display for Error {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "libteeline_gpu {:?}: {}", self.code, self.message)
    }
}

impl std::error::Error for Error {}

/// Result of a solver call: the tour as positions, its cost as `DistanceMatrix::tour_length_by_pos` sums it
/// (distance_matrix.rs:235-245, bit for bit), and the counters.
#[derive(Clone, Debug)]
pub struct Tour {
    pub pos: Vec<u32>,
    pub cost: f32,
    pub stats: Stats,
}

/// One `tl_ctx`: a HIP stream and a device workspace on one GPU.  `!Sync`, `!Send`: use it from the thread that made it.
pub struct Context {
    raw: *mut TlCtx,
}

fn opt_ptr<T>(s: Option<&[T]>) -> *const T {
    s.map_or(ptr::null(), |v| v.as_ptr())
}

impl Context {
    pub fn new(device: i32, flags: u32) -> Result<Context, Error> {
        // SAFETY: plain C calls; `raw` is written by tl_create on success only.
        unsafe {
            if tl_abi_version() != TL_ABI_VERSION {
                return Err(Error { code: Code::Other(0), message: format!("libteeline_gpu ABI {} != {}", tl_abi_version(), TL_ABI_VERSION) });
            }
            let mut raw: *mut TlCtx = ptr::null_mut();
            let rc = tl_create(device, flags, &mut raw);
            if rc != 0 {
                let msg = CStr::from_ptr(tl_last_error(ptr::null())).to_string_lossy().into_owned();
                return Err(Error { code: Code::from_raw(rc), message: msg });
            }
            Ok(Context { raw })
        }
    }

    fn check(&self, rc: c_int) -> Result<(), Error> {
        if rc == 0 {
            return Ok(());
        }
        if rc == -8 {
            // TL_ERR_BUSY: the context's error string belongs to the thread that is inside the library; it is not read
            return Err(Error { code: Code::Busy, message: "the context is in use by another thread".to_string() });
        }
        // SAFETY: tl_last_error returns a NUL-terminated string owned by the context.
        let msg = unsafe { CStr::from_ptr(tl_last_error(self.raw)).to_string_lossy().into_owned() };
        Err(Error { code: Code::from_raw(rc), message: msg })
    }

    fn n_of(xy: &[f32]) -> u32 {
        assert!(xy.len() % 2 == 0, "xy holds (x, y) pairs");
        u32::try_from(xy.len() / 2).expect("more than u32::MAX cities")
    }

    fn check_inputs(n: u32, dm_packed: Option<&[f32]>, init_pos: Option<&[u32]>) {
        let n = n as usize;
        if let Some(d) = dm_packed {
            assert_eq!(d.len(), n * n.saturating_sub(1) / 2, "dm_packed length != n(n-1)/2");
        }
        if let Some(p) = init_pos {
            assert_eq!(p.len(), n, "init tour length != number of cities");
        }
    }

    /// `DistanceMatrix::build` (distance_matrix.rs:122-153): packed strict lower triangle, EUC_2D or GEO.
    pub fn dm_build(&self, xy: &[f32], geo: bool) -> Result<Vec<f32>, Error> {
        let n = Self::n_of(xy);
        let mut out = vec![0f32; (n as usize) * (n as usize).saturating_sub(1) / 2];
        // SAFETY: out has n(n-1)/2 elements, xy has 2n.
        let rc = unsafe { tl_dm_build(self.raw, xy.as_ptr(), n, if geo { DIST_GEO } else { DIST_EUC2D }, DM_PACKED_LOWER, out.as_mut_ptr(), ptr::null_mut()) };
        self.check(rc).map(|_| out)
    }

    /// Does `dm_packed` hold exactly the EUC_2D distances of `xy`?  (`DistanceMatrix` keeps no distance type,
    /// distance_matrix.rs:86-93.)  If so the coordinate kernels give the same tours without the matrix.
    pub fn dm_is_euc2d(&self, xy: &[f32], dm_packed: &[f32]) -> Result<bool, Error> {
        let n = Self::n_of(xy);
        Self::check_inputs(n, Some(dm_packed), None);
        let mut flag: c_int = 0;
        // SAFETY: lengths checked above.
        let rc = unsafe { tl_dm_is_euc2d(self.raw, xy.as_ptr(), dm_packed.as_ptr(), n, &mut flag) };
        self.check(rc).map(|_| flag != 0)
    }

    /// `DistanceMatrix::tour_length_by_pos` (distance_matrix.rs:235-245).
    pub fn tour_length(&self, xy: &[f32], dm_packed: Option<&[f32]>, perm: &[u32]) -> Result<f32, Error> {
        let n = Self::n_of(xy);
        Self::check_inputs(n, dm_packed, Some(perm));
        let mut cost = 0f32;
        // SAFETY: lengths checked above.
        let rc = unsafe { tl_tour_length(self.raw, if dm_packed.is_some() { ptr::null() } else { xy.as_ptr() }, opt_ptr(dm_packed), n, perm.as_ptr(), &mut cost) };
        self.check(rc).map(|_| cost)
    }

    /// `two_opt::solve` (two_opt.rs:7-67).  `mode` = MODE_REF_ORDER reproduces the reference's tours.
    pub fn two_opt(&self, xy: &[f32], dm_packed: Option<&[f32]>, init_pos: Option<&[u32]>, mode: c_int) -> Result<Tour, Error> {
        let n = Self::n_of(xy);
        Self::check_inputs(n, dm_packed, init_pos);
        let mut t = Tour { pos: vec![0u32; n as usize], cost: 0.0, stats: Stats::default() };
        // SAFETY: every buffer has the length the C ABI documents (n, 2n, n(n-1)/2), checked above.
        let rc = unsafe { tl_two_opt(self.raw, xy.as_ptr(), n, opt_ptr(dm_packed), opt_ptr(init_pos), mode, t.pos.as_mut_ptr(), &mut t.cost, &mut t.stats) };
        self.check(rc).map(|_| t)
    }

    /// `two_opt::solve` together with the moves it applied, `(i, j)` in the reference's order (`swap_2opt(path, i+1, j)`,
    /// two_opt.rs:50): what `gpu::two_opt::solve` replays the reference's per-move progress messages from.  Coordinates
    /// (`dm_packed` None; `Err(Unsupported)` beyond the LDS-resident descent) or the matrix form.
    /// The list holds `Some((i, j))` per move and `None` where a new sweep begins (`TL_TRACE_SWEEP`).
    pub fn two_opt_trace(&self, xy: &[f32], dm_packed: Option<&[f32]>, init_pos: Option<&[u32]>) -> Result<(Tour, Vec<Option<(u32, u32)>>), Error> {
        let n = Self::n_of(xy);
        Self::check_inputs(n, dm_packed, init_pos);
        let mut t = Tour { pos: vec![0u32; n as usize], cost: 0.0, stats: Stats::default() };
        let mut cap = (16 * n).max(64);
        loop {
            let mut log = vec![0u32; cap as usize];
            let mut len = 0u32;
            // SAFETY: every buffer has the length the C ABI documents (n, 2n, n(n-1)/2, log_cap), checked above.
            let rc = unsafe {
                tl_two_opt_trace(self.raw, xy.as_ptr(), n, opt_ptr(dm_packed), opt_ptr(init_pos), t.pos.as_mut_ptr(), &mut t.cost, &mut t.stats,
                                 log.as_mut_ptr(), cap, &mut len)
            };
            self.check(rc)?;
            if len <= cap {
                log.truncate(len as usize);
                return Ok((t, log.into_iter().map(|w| if w == 0xFFFF_FFFF { None } else { Some((w >> 16, w & 0xFFFF)) }).collect()));
            }
            cap = len; // the descent is deterministic: once more with room for every move
        }
    }

    /// `three_opt::solve` together with the moves it applied, `(i, j, k, case)` in order (three_opt.rs:36-45): what
    /// `gpu::three_opt::solve` replays the reference's per-move `PathUpdate`s from.
    pub fn three_opt_trace(&self, xy: &[f32], dm_packed: Option<&[f32]>, init_pos: Option<&[u32]>) -> Result<(Tour, Vec<[u32; 4]>), Error> {
        let n = Self::n_of(xy);
        Self::check_inputs(n, dm_packed, init_pos);
        let mut t = Tour { pos: vec![0u32; n as usize], cost: 0.0, stats: Stats::default() };
        let mut cap = (4 * n).max(64);
        loop {
            let mut log = vec![0u32; 4 * cap as usize];
            let mut len = 0u32;
            // SAFETY: as in two_opt_trace; the log holds 4 words per move.
            let rc = unsafe {
                tl_three_opt_trace(self.raw, xy.as_ptr(), n, opt_ptr(dm_packed), opt_ptr(init_pos), t.pos.as_mut_ptr(), &mut t.cost, &mut t.stats,
                                   log.as_mut_ptr(), cap, &mut len)
            };
            self.check(rc)?;
            if len <= cap {
                return Ok((t, log.chunks_exact(4).take(len as usize).map(|m| [m[0], m[1], m[2], m[3]]).collect()));
            }
            cap = len;
        }
    }

    /// `or_opt::solve` together with the moves it applied, `(i, j, seg_len, reversed)` in order (or_opt.rs:45-51): what
    /// `gpu::or_opt::solve` replays the reference's per-move `PathUpdate`s from.
    pub fn or_opt_trace(&self, xy: &[f32], dm_packed: Option<&[f32]>, init_pos: Option<&[u32]>) -> Result<(Tour, Vec<[u32; 4]>), Error> {
        let n = Self::n_of(xy);
        Self::check_inputs(n, dm_packed, init_pos);
        let mut t = Tour { pos: vec![0u32; n as usize], cost: 0.0, stats: Stats::default() };
        let mut cap = (4 * n).max(64);
        loop {
            let mut log = vec![0u32; 4 * cap as usize];
            let mut len = 0u32;
            // SAFETY: as in two_opt_trace; the log holds 4 words per move.
            let rc = unsafe {
                tl_or_opt_trace(self.raw, xy.as_ptr(), n, opt_ptr(dm_packed), opt_ptr(init_pos), t.pos.as_mut_ptr(), &mut t.cost, &mut t.stats,
                                log.as_mut_ptr(), cap, &mut len)
            };
            self.check(rc)?;
            if len <= cap {
                return Ok((t, log.chunks_exact(4).take(len as usize).map(|m| [m[0], m[1], m[2], m[3]]).collect()));
            }
            cap = len;
        }
    }

    /// `lin_kernighan::solve` together with every best tour it settles on and its `best_dist`, in order (lin_kernighan.rs:71,90):
    /// what `gpu::lin_kernighan::solve` sends as `PathUpdate`s.
    pub fn lin_kernighan_trace(&self, xy: &[f32], dm_packed: Option<&[f32]>, init_pos: Option<&[u32]>, opts: LkOpts, seed: u64)
                               -> Result<(Tour, Vec<(Vec<u32>, f32)>), Error> {
        let n = Self::n_of(xy);
        Self::check_inputs(n, dm_packed, init_pos);
        let mut t = Tour { pos: vec![0u32; n as usize], cost: 0.0, stats: Stats::default() };
        let mut cap = (opts.epochs + 1).min(64).max(1);
        loop {
            let mut snaps = vec![0u32; cap as usize * n as usize];
            let mut dists = vec![0f32; cap as usize];
            let mut len = 0u32;
            // SAFETY: as in two_opt_trace; snap_pos holds snap_cap x n positions, snap_dist snap_cap values.
            let rc = unsafe {
                tl_lk_trace(self.raw, xy.as_ptr(), n, opt_ptr(dm_packed), opt_ptr(init_pos), &opts, seed, t.pos.as_mut_ptr(), &mut t.cost, &mut t.stats,
                            snaps.as_mut_ptr(), dists.as_mut_ptr(), cap, &mut len)
            };
            self.check(rc)?;
            if len <= cap {
                let list = snaps.chunks_exact(n as usize).take(len as usize).zip(dists).map(|(p, d)| (p.to_vec(), d)).collect();
                return Ok((t, list));
            }
            cap = len; // deterministic for a seed: once more with room for every snapshot
        }
    }

    /// `lin_kernighan::solve` with `on_best(best_tour positions, best_dist)` called WHILE the search runs, once per best tour the ILS
    /// settles on, in order (lin_kernighan.rs:71,90) — `tl_lk_live`; what `gpu::lin_kernighan::solve` forwards to its channel.
    pub fn lin_kernighan_live<F: FnMut(&[u32], f32)>(&self, xy: &[f32], dm_packed: Option<&[f32]>, init_pos: Option<&[u32]>, opts: LkOpts, seed: u64,
                                                     mut on_best: F) -> Result<Tour, Error> {
        unsafe extern "C" fn trampoline<F: FnMut(&[u32], f32)>(user: *mut c_void, best_pos: *const u32, n: u32, best_dist: f32) {
            // SAFETY: `user` is the `&mut F` handed to tl_lk_live below, alive for the whole call; best_pos holds n positions for the
            // duration of this callback.  A panic must not unwind into C: it aborts the process, like the reference's panics do not.
            let f = unsafe { &mut *(user as *mut F) };
            let tour = unsafe { std::slice::from_raw_parts(best_pos, n as usize) };
            if std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| f(tour, best_dist))).is_err() {
                std::process::abort();
            }
        }
        let n = Self::n_of(xy);
        Self::check_inputs(n, dm_packed, init_pos);
        let mut t = Tour { pos: vec![0u32; n as usize], cost: 0.0, stats: Stats::default() };
        // SAFETY: as in two_opt; the callback and its state outlive the call.
        let rc = unsafe {
            tl_lk_live(self.raw, xy.as_ptr(), n, opt_ptr(dm_packed), opt_ptr(init_pos), &opts, seed, t.pos.as_mut_ptr(), &mut t.cost, &mut t.stats,
                       trampoline::<F>, &mut on_best as *mut F as *mut c_void)
        };
        self.check(rc).map(|_| t)
    }

    /// `three_opt::solve` (three_opt.rs:16-51).
    pub fn three_opt(&self, xy: &[f32], dm_packed: Option<&[f32]>, init_pos: Option<&[u32]>) -> Result<Tour, Error> {
        let n = Self::n_of(xy);
        Self::check_inputs(n, dm_packed, init_pos);
        let mut t = Tour { pos: vec![0u32; n as usize], cost: 0.0, stats: Stats::default() };
        // SAFETY: as in two_opt.
        let rc = unsafe { tl_three_opt(self.raw, xy.as_ptr(), n, opt_ptr(dm_packed), opt_ptr(init_pos), t.pos.as_mut_ptr(), &mut t.cost, &mut t.stats) };
        self.check(rc).map(|_| t)
    }

    /// `or_opt::solve` (or_opt.rs:18-74).
    pub fn or_opt(&self, xy: &[f32], dm_packed: Option<&[f32]>, init_pos: Option<&[u32]>) -> Result<Tour, Error> {
        let n = Self::n_of(xy);
        Self::check_inputs(n, dm_packed, init_pos);
        let mut t = Tour { pos: vec![0u32; n as usize], cost: 0.0, stats: Stats::default() };
        // SAFETY: as in two_opt.
        let rc = unsafe { tl_or_opt(self.raw, xy.as_ptr(), n, opt_ptr(dm_packed), opt_ptr(init_pos), t.pos.as_mut_ptr(), &mut t.cost, &mut t.stats) };
        self.check(rc).map(|_| t)
    }

    /// `lin_kernighan::solve` (lin_kernighan.rs:35-100).  The search is Euclidean over `xy` (the reference rebuilds its own
    /// matrix, :41); `dm_packed` (problem.distances of a GEO / EXPLICIT problem) feeds the NN seed (:47-55) and the total (:99).
    /// `seed` drives the double-bridge kicks (the reference draws them from an unseeded thread RNG, :73).
    pub fn lin_kernighan(&self, xy: &[f32], dm_packed: Option<&[f32]>, init_pos: Option<&[u32]>, opts: LkOpts, seed: u64) -> Result<Tour, Error> {
        let n = Self::n_of(xy);
        Self::check_inputs(n, dm_packed, init_pos);
        let mut t = Tour { pos: vec![0u32; n as usize], cost: 0.0, stats: Stats::default() };
        // SAFETY: as in two_opt; opts is a plain #[repr(C)] value.
        let rc = unsafe { tl_lk(self.raw, xy.as_ptr(), n, opt_ptr(dm_packed), opt_ptr(init_pos), &opts, seed, t.pos.as_mut_ptr(), &mut t.cost, &mut t.stats) };
        self.check(rc).map(|_| t)
    }

    /// `nearest_neighbor::solve` (nearest_neighbor.rs:8-76).
    pub fn nearest_neighbor(&self, xy: &[f32], dm_packed: Option<&[f32]>, n_nearest: u32) -> Result<Tour, Error> {
        let n = Self::n_of(xy);
        Self::check_inputs(n, dm_packed, None);
        let mut t = Tour { pos: vec![0u32; n as usize], cost: 0.0, stats: Stats::default() };
        // SAFETY: as in two_opt.
        let rc = unsafe { tl_nearest_neighbor(self.raw, xy.as_ptr(), opt_ptr(dm_packed), n, n_nearest, t.pos.as_mut_ptr(), &mut t.cost) };
        self.check(rc).map(|_| t)
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        // SAFETY: raw came from tl_create and is destroyed once.
        unsafe { tl_destroy(self.raw) }
    }
}

/// Multi-start 2-opt over several GPUs of one node from one process (north-star config 4): restarts
/// `[first, first + count)` from seeded Fisher-Yates permutations, dealt in contiguous blocks over `ctxs` (one context
/// per device).  Returns the best tour, its restart id and every restart's cost.
pub fn two_opt_multistart(ctxs: &[Context], xy: &[f32], seed: u64, first: u32, count: u32) -> Result<(Tour, u32, Vec<f32>), Error> {
    assert!(!ctxs.is_empty());
    let n = Context::n_of(xy);
    let raws: Vec<*mut TlCtx> = ctxs.iter().map(|c| c.raw).collect();
    let mut t = Tour { pos: vec![0u32; n as usize], cost: 0.0, stats: Stats::default() };
    let mut best = 0u32;
    let mut costs = vec![0f32; count as usize];
    // SAFETY: buffers sized as the C ABI documents; the contexts outlive the call.
    let rc = unsafe {
        tl_two_opt_multistart_devices(raws.as_ptr(), raws.len() as c_int, xy.as_ptr(), n, seed, first, count, MODE_REF_ORDER,
                                      t.pos.as_mut_ptr(), &mut t.cost, &mut best, costs.as_mut_ptr(), &mut t.stats)
    };
    ctxs[0].check(rc).map(|_| (t, best, costs))
}

thread_local! {
    static CTX: RefCell<Option<Context>> = const { RefCell::new(None) };
}

/// Runs `f` with this thread's context (device `TEELINE_GPU_DEVICE`, default 0), creating it on first use.
pub fn with_context<R>(f: impl FnOnce(&Context) -> Result<R, Error>) -> Result<R, Error> {
    CTX.with(|cell| {
        let mut slot = cell.borrow_mut();
        if slot.is_none() {
            let dev = std::env::var("TEELINE_GPU_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
            *slot = Some(Context::new(dev, 0)?);
        }
        f(slot.as_ref().expect("context was just created"))
    })
}

#[allow(dead_code)]
fn _assert_layout() {
    // the C structs are 48 and 16 bytes (include/teeline_gpu.h)
    const _: () = assert!(std::mem::size_of::<Stats>() == 48 && std::mem::size_of::<LkOpts>() == 16);
}
