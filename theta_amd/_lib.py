"""
ctypes binding of libtheta_hip.so (C ABI in include/theta_hip.h).  numpy in, numpy out, no torch.

The shared library IS the compute path.  If it is missing, or no MI355X is visible, every entry
point raises -- there is no CPU fallback anywhere in this package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("THETA_HIP_LIB") or os.path.join(_HERE, "libtheta_hip.so")   # THETA_HIP_LIB: A/B builds

THETA_OK, ERR_ARG, ERR_NO_CANDIDATES, ERR_HIP, ERR_OVERFLOW, ERR_CAPACITY = range(6)
MIX_PROPOSE, MIX_LINES, MIX_LINES_ONLY, MIX_DIVE = 1, 2, 4, 8          # theta_mix_search's mode bits (include/theta_hip.h)


class ThetaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libtheta_hip error %d: %s" % (code, msg))
        self.code = code


class ListOverflow(ThetaError):
    """A device list of one theta_search stayed too short although the library did what it could (three passes, a redo with the
    range's minimum as hint).  `kind`: "ties" (finalists within the window), "suspects" (rejected candidates near the minimum) or
    "degenerate" (rank-deficient candidates, see DegenerateOverflow).  All three shrink with the collection window: the driver
    narrows it before it gives up (search.collect_finalists)."""
    kind = "ties"

    def __init__(self, msg, kind=None):
        super().__init__(ERR_CAPACITY, msg)
        if kind:
            self.kind = kind


class DegenerateOverflow(ListOverflow):
    """More rank-deficient candidates in one theta_search than its device list holds: Problem.search halves the piece."""
    kind = "degenerate"

    def __init__(self, code, msg):
        ListOverflow.__init__(self, msg, "degenerate")


class NoCandidates(ThetaError):
    """The bounds admit no matrix (the reference prints an error and exits, RunTHetA.py:217-219)."""


class SearchStats(C.Structure):
    _fields_ = [("evaluated", C.c_uint64), ("accepted", C.c_uint64), ("degenerate", C.c_uint64),
                ("iterations", C.c_uint64), ("terms", C.c_uint64), ("list_overflow", C.c_uint64),
                ("flops", C.c_uint64), ("flops_f32", C.c_uint64), ("dismissed", C.c_uint64), ("best_nll", C.c_double), ("rejected_bound", C.c_double),
                ("rejected_rank", C.c_uint64 * 2), ("kernel_ms", C.c_double), ("setup_ms", C.c_double),
                ("phase_cycles", C.c_uint64 * 8), ("survivors", C.c_uint64), ("fallback_candidates", C.c_uint64),
                ("redo_flops", C.c_uint64), ("redo_flops_f32", C.c_uint64), ("redo_kernel_ms", C.c_double), ("kernel_launches", C.c_uint64), ("pruned", C.c_uint64)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("rejected_rank", "phase_cycles")}
        d["phase_cycles"] = [int(x) for x in self.phase_cycles]
        d["rejected_rank"] = int(self.rejected_rank[0]) | (int(self.rejected_rank[1]) << 64)
        return d


class BnbStats(C.Structure):
    _fields_ = [("nodes_expanded", C.c_uint64), ("children_bounded", C.c_uint64), ("newton_iterations", C.c_uint64),
                ("children_pruned", C.c_uint64), ("children_collinear", C.c_uint64), ("children_unbounded", C.c_uint64),
                ("ranges_raw", C.c_uint64), ("ranges", C.c_uint64), ("launches", C.c_uint64), ("max_frontier", C.c_uint64),
                ("chunk", C.c_uint64), ("leaves", C.c_double), ("kernel_ms", C.c_double), ("wall_ms", C.c_double),
                ("emit_depth", C.c_int), ("complete", C.c_int), ("frontier", C.c_uint64 * 257)]

    def as_dict(self, m):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "frontier"}
        d["frontier"] = [int(self.frontier[i]) for i in range(m + 1)]
        return d


class MixStats(C.Structure):
    _fields_ = [("boxes_tested", C.c_uint64), ("levels", C.c_uint64), ("max_boxes", C.c_uint64), ("leaves", C.c_uint64),
                ("listed", C.c_uint64), ("matrices", C.c_uint64), ("lines", C.c_uint64), ("line_leaves", C.c_uint64), ("syncs", C.c_uint64),
                ("kernel_ms", C.c_double), ("wall_ms", C.c_double), ("min_bound", C.c_double), ("min_bound_lines", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# theta_witness (include/theta_hip.h): what the n=3 sieve kernel left a sampled candidate at
WITNESS_DTYPE = np.dtype([("mu", np.float64, 3), ("nll", np.float64), ("l2_last", np.float32), ("l2_first", np.float32),
                          ("evaluations", np.uint16), ("status", np.uint16), ("mu_bound", np.float32)])

_lib = None

# every symbol include/theta_hip.h declares
EXPORTS = ["theta_create", "theta_device_count", "theta_destroy", "theta_last_error", "theta_device_info", "theta_problem_create",
           "theta_problem_destroy", "theta_problem_count", "theta_count_lower_bound", "theta_search", "theta_search_values", "theta_search_witness", "theta_bnb", "theta_search_ranges", "theta_mix_search", "theta_enumerate", "theta_enumerate_device",
           "theta_solve_batch", "theta_score_batch", "theta_score_masked", "theta_search_suspects", "theta_boundary_min", "theta_problem_hint",
           "theta_search_degenerate", "theta_problem_set_option", "theta_synchronize",
           "theta_score_batch_rows", "theta_device_alloc", "theta_device_free", "theta_device_copy", "theta_solve_batch_device",
           "theta_score_masked_device",
           "theta_refpow_check", "theta_comm_create", "theta_comm_destroy", "theta_comm_info", "theta_comm_barrier", "theta_comm_allreduce_min",
           "theta_comm_allreduce_max", "theta_comm_allreduce_sum", "theta_comm_allgather", "theta_exchange_finalists"]


def load():
    """dlopen the library (no GPU needed for this) and declare prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python -m theta_amd.build` (hipcc, gfx950). "
                          "theta_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, u64p, dp, u8p = C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
    i64p, i32p = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    lib.theta_last_error.restype = C.c_char_p
    lib.theta_create.argtypes = [i32, C.POINTER(vp)]
    lib.theta_device_count.argtypes = [C.POINTER(i32)]
    lib.theta_refpow_check.argtypes = [i32, C.POINTER(i32)]
    lib.theta_count_lower_bound.argtypes = [i32, i32, i32p, i32p, C.POINTER(C.c_double)]
    lib.theta_destroy.argtypes = [vp]
    lib.theta_destroy.restype = None
    lib.theta_device_info.argtypes = [vp, C.c_char_p, i32, C.POINTER(i32), u64p]
    lib.theta_problem_create.argtypes = [vp, i32, i32, i32, i64p, i64p, i32p, i32p, C.c_double, C.POINTER(vp)]
    lib.theta_problem_destroy.argtypes = [vp]
    lib.theta_problem_destroy.restype = None
    lib.theta_problem_count.argtypes = [vp, u64p]
    lib.theta_search.argtypes = [vp, u64p, u64p, C.c_double, i32, dp, dp, u64p, u8p, C.POINTER(i32), C.POINTER(SearchStats)]
    lib.theta_search_values.argtypes = [vp, u64p, C.c_uint64, dp, dp, C.POINTER(SearchStats)]
    lib.theta_search_witness.argtypes = [vp, u64p, u64p, C.c_double, i32, C.c_uint64, vp, u64p, C.POINTER(SearchStats)]
    lib.theta_search_ranges.argtypes = [vp, i32, u64p, C.c_double, i32, dp, dp, u64p, u8p, C.POINTER(i32), C.POINTER(SearchStats)]
    lib.theta_mix_search.argtypes = [vp, C.c_double, C.c_double, i32, C.c_uint64, u8p, u64p, C.POINTER(MixStats)]
    lib.theta_bnb.argtypes = [vp, C.c_double, C.c_uint64, i32, C.c_uint64, C.c_uint64, u64p, u64p, C.POINTER(BnbStats)]
    lib.theta_enumerate.argtypes = [vp, u64p, C.c_uint64, u8p]
    lib.theta_enumerate_device.argtypes = [vp, u64p, C.c_uint64, vp, dp]
    lib.theta_search_suspects.argtypes = [vp, i32, u64p, dp, u8p, C.POINTER(i32)]
    lib.theta_boundary_min.argtypes = [vp, i32, i32, i64p, i64p, i32, u8p, dp]
    lib.theta_problem_hint.argtypes = [vp, C.c_double]
    lib.theta_synchronize.argtypes = [vp]
    lib.theta_search_degenerate.argtypes = [vp, i32, u64p, u8p, C.POINTER(i32)]
    lib.theta_problem_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    lib.theta_comm_create.argtypes = [vp, i32, i32, C.c_char_p, i32, i32, C.POINTER(vp)]
    lib.theta_comm_destroy.argtypes = [vp]
    lib.theta_comm_destroy.restype = None
    lib.theta_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), u64p]
    lib.theta_comm_barrier.argtypes = [vp]
    lib.theta_comm_allreduce_min.argtypes = [vp, dp, i32]
    lib.theta_comm_allreduce_max.argtypes = [vp, dp, i32]
    lib.theta_comm_allreduce_sum.argtypes = [vp, dp, i32]
    lib.theta_comm_allgather.argtypes = [vp, vp, C.c_size_t, vp]
    lib.theta_exchange_finalists.argtypes = [vp, i32, i32, i32, dp, dp, u64p, u8p, dp, C.c_double, i32, dp, dp, u64p, u8p, dp,
                                             C.POINTER(i32), dp]
    lib.theta_solve_batch.argtypes = [vp, i32, i32, i32, i64p, i64p, C.c_double, i32, u8p, u8p, dp, dp, dp]
    lib.theta_score_batch.argtypes = [vp, i32, i32, i32, dp, dp, dp, dp, dp, u8p]
    lib.theta_score_masked.argtypes = [vp, i32, i32, i32, i32, i32, u8p, dp, dp, dp, u64p, dp, dp]
    lib.theta_score_batch_rows.argtypes = [vp, i32, i32, i32, dp, dp, dp, dp, dp, u8p]
    lib.theta_device_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.theta_device_free.argtypes = [vp, vp]
    lib.theta_device_copy.argtypes = [vp, vp, vp, C.c_size_t, i32]
    lib.theta_solve_batch_device.argtypes = [vp, i32, i32, i32, i64p, i64p, C.c_double, i32, vp, vp, vp, vp, vp, dp]
    lib.theta_score_masked_device.argtypes = [vp, i32, i32, i32, i32, i32, vp, dp, dp, vp, u64p, vp, dp]
    _lib = lib
    return lib


def _check(rc):
    if rc != THETA_OK:
        msg = load().theta_last_error().decode()
        raise (NoCandidates if rc == ERR_NO_CANDIDATES else ThetaError)(rc, msg)


def _p(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


def _u128(v):
    v = int(v)
    return (C.c_uint64 * 2)(v & 0xFFFFFFFFFFFFFFFF, v >> 64)


def count_lower_bound_log2(m, tau, lower_bounds, upper_bounds):
    """theta_count_lower_bound: log2 of a lower bound of the number of matrices of an n=3 space (host only, no GPU)."""
    lb = np.ascontiguousarray(lower_bounds, np.int32)
    ub = np.ascontiguousarray(upper_bounds, np.int32)
    out = C.c_double(0.0)
    _check(load().theta_count_lower_bound(int(m), int(tau), lb.ctypes.data_as(C.POINTER(C.c_int32)), ub.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(out)))
    return out.value


_pow_check = None


def libm_pow_matches():
    """True if this host's libm squares like the restatement the n=3 kernels carry (theta_refpow_check: glibc >= 2.28, x86-64 with
    FMA) -- the platform on which `best` equals the reference's entry by entry also for rank-deficient n=3 candidates."""
    global _pow_check
    if _pow_check is None:
        n = C.c_int()
        _check(load().theta_refpow_check(200000, C.byref(n)))
        _pow_check = n.value == 0
    return _pow_check


def device_count():
    """GPUs visible to this process (theta_device_count); 0 when there is none."""
    n = C.c_int()
    load().theta_device_count(C.byref(n))
    return n.value


class Context:
    """One per process, bound to one GPU (theta_create)."""

    def __init__(self, device=None):
        lib = load()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        h = C.c_void_p()
        _check(lib.theta_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        name = C.create_string_buffer(128)
        cu = C.c_int()
        hbm = C.c_uint64()
        _check(lib.theta_device_info(h, name, 128, C.byref(cu), C.byref(hbm)))
        self.name, self.cu_count, self.hbm_bytes = name.value.decode(), cu.value, hbm.value

    def close(self):
        if getattr(self, "_h", None):
            load().theta_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(load().theta_synchronize(self._h))

    # -- materialised operators ---------------------------------------------------------------
    def solve_batch(self, n, tau, r, rN, C_u8, max_normal=1.0, want_vals=True):
        """Optimizer.solve on a batch: C_u8 is (B, m) for n=2 or (B, m, 2) for n=3."""
        C_u8 = np.ascontiguousarray(C_u8, dtype=np.uint8)
        B = C_u8.shape[0]
        m = C_u8.shape[1]
        r = np.ascontiguousarray(r, dtype=np.int64)
        rN = np.ascontiguousarray(rN, dtype=np.int64)
        ok = np.zeros(B, np.uint8)
        mu = np.zeros((B, n))
        nll = np.zeros(B)
        vals = np.zeros((B, m)) if want_vals else None
        _check(load().theta_solve_batch(self._h, n, m, int(tau), _p(r, C.c_int64), _p(rN, C.c_int64), float(max_normal),
                                        B, _p(C_u8, C.c_uint8), _p(ok, C.c_uint8), _p(mu, C.c_double), _p(nll, C.c_double),
                                        _p(vals, C.c_double) if want_vals else None))
        self.last_solve_fallback = ok == 2     # n=3: entries that carry the reference's nu = (1/3,1/3,1/3) fallback
        return ok.astype(bool), mu, nll, vals

    def boundary_min(self, tau, r, rN, C_u8):
        """n=3: exact minimum of the NLL over the simplex boundary for each candidate C_u8 (B, m, 2)."""
        C_u8 = np.ascontiguousarray(C_u8, dtype=np.uint8)
        B, m = C_u8.shape[0], C_u8.shape[1]
        r = np.ascontiguousarray(r, dtype=np.int64)
        rN = np.ascontiguousarray(rN, dtype=np.int64)
        out = np.zeros(B)
        _check(load().theta_boundary_min(self._h, m, int(tau), _p(r, C.c_int64), _p(rN, C.c_int64), B,
                                         _p(C_u8, C.c_uint8), _p(out, C.c_double)))
        return out

    def score_batch(self, n, Cw, mu, r):
        """CalcAllC.L2/L3 on B literal matrices Cw (B, m, n); mu (B, n); r (m,) shared or (B, m) one per matrix;
        returns nll, vals, valid."""
        Cw = np.ascontiguousarray(Cw, dtype=np.float64)
        B, m, nn = Cw.shape
        assert nn == n
        mu = np.ascontiguousarray(mu, dtype=np.float64).reshape(B, n)
        r = np.ascontiguousarray(r, dtype=np.float64)
        nll = np.zeros(B)
        vals = np.zeros((B, m))
        valid = np.zeros((B, m), np.uint8)
        fn = load().theta_score_batch_rows if r.ndim == 2 else load().theta_score_batch
        assert r.shape == ((B, m) if r.ndim == 2 else (m,))
        _check(fn(self._h, n, m, B, _p(Cw, C.c_double), _p(mu, C.c_double), _p(r, C.c_double),
                  _p(nll, C.c_double), _p(vals, C.c_double), _p(valid, C.c_uint8)))
        return nll, vals, valid.astype(bool)

    # -- device-resident chains -----------------------------------------------------------------
    def device_array(self, shape, dtype):
        """Uninitialised array in this GPU's HBM (DeviceArray): for chains enumerate_device -> solve / score -> download."""
        return DeviceArray(self, shape, dtype)

    def solve_batch_device(self, n, tau, r, rN, d_C, B, m, max_normal=1.0, want_vals=False):
        """Optimizer.solve on B candidates already in HBM (d_C: DeviceArray u8).  Returns DeviceArrays ok, mu, nll, vals and
        the kernel's duration in ms."""
        r = np.ascontiguousarray(r, dtype=np.int64)
        rN = np.ascontiguousarray(rN, dtype=np.int64)
        ok, mu, nll = self.device_array((B,), np.uint8), self.device_array((B, n), np.float64), self.device_array((B,), np.float64)
        vals = self.device_array((B, m), np.float64) if want_vals else None
        ms = C.c_double()
        _check(load().theta_solve_batch_device(self._h, n, m, int(tau), _p(r, C.c_int64), _p(rN, C.c_int64), float(max_normal), B,
                                               d_C.ptr, ok.ptr, mu.ptr, nll.ptr, vals.ptr if vals is not None else None, C.byref(ms)))
        return ok, mu, nll, vals, ms.value

    def score_masked_device(self, n, tau, d_C, B, m, w, r, d_mu, masks=None):
        """theta_score_masked on device-resident candidates and mixtures; returns (DeviceArray nll (B, S), kernel ms)."""
        w = np.ascontiguousarray(w, dtype=np.float64)
        r = np.ascontiguousarray(r, dtype=np.float64)
        S = 1 if masks is None else masks.shape[0]
        if masks is not None:
            masks = np.ascontiguousarray(masks, dtype=np.uint64)
        nll = self.device_array((B, S), np.float64)
        ms = C.c_double()
        _check(load().theta_score_masked_device(self._h, n, m, int(tau), B, S, d_C.ptr, _p(w, C.c_double), _p(r, C.c_double), d_mu.ptr,
                                                _p(masks, C.c_uint64) if masks is not None else None, nll.ptr, C.byref(ms)))
        return nll, ms.value

    def score_masked(self, n, tau, C_u8, w, r, mu, masks=None):
        """Byte candidates x row masks; masks is (S, ceil(m/64)) uint64 or None. Returns (nll (B,S), kernel_ms)."""
        C_u8 = np.ascontiguousarray(C_u8, dtype=np.uint8)
        B, m = C_u8.shape[0], C_u8.shape[1]
        w = np.ascontiguousarray(w, dtype=np.float64)
        r = np.ascontiguousarray(r, dtype=np.float64)
        mu = np.ascontiguousarray(mu, dtype=np.float64).reshape(B, n)
        S = 1 if masks is None else masks.shape[0]
        if masks is not None:
            masks = np.ascontiguousarray(masks, dtype=np.uint64)
        nll = np.zeros((B, S))
        ms = C.c_double()
        _check(load().theta_score_masked(self._h, n, m, int(tau), B, S, _p(C_u8, C.c_uint8), _p(w, C.c_double),
                                         _p(r, C.c_double), _p(mu, C.c_double),
                                         _p(masks, C.c_uint64) if masks is not None else None, _p(nll, C.c_double),
                                         C.byref(ms)))
        return nll, ms.value


class DeviceArray:
    """A typed block of HBM owned by the caller (theta_device_alloc); numpy in / out through upload() / download()."""

    def __init__(self, ctx, shape, dtype):
        self.ctx, self.shape, self.dtype = ctx, tuple(int(x) for x in shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        h = C.c_void_p()
        _check(load().theta_device_alloc(ctx._h, self.nbytes, C.byref(h)))
        self.ptr = h

    def upload(self, arr):
        a = np.ascontiguousarray(arr, dtype=self.dtype)
        assert a.nbytes == self.nbytes
        _check(load().theta_device_copy(self.ctx._h, self.ptr, a.ctypes.data_as(C.c_void_p), self.nbytes, 1))
        return self

    def download(self):
        out = np.zeros(self.shape, self.dtype)
        _check(load().theta_device_copy(self.ctx._h, out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes, 0))
        return out

    def free(self):
        if getattr(self, "ptr", None):
            load().theta_device_free(self.ctx._h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_default_ctx = None


def _rss_bytes():
    """resident set size of this process (0 where /proc is not available)"""
    try:
        with open("/proc/self/statm") as f:
            return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")
    except Exception:
        return 0


def default_context():
    """Process-wide context on cuda:LOCAL_RANK (created on first use)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


class _Merged:
    """Running merge of the pieces of one Problem.search call (finalists, n=3 suspects and all-zero-column entries, stats).
    Every entry remembers the index of its piece: the lists come out in piece order whatever the order of the add() calls
    (a piece that had to be searched a second time is added last)."""
    SUMMED = ("evaluated", "accepted", "degenerate", "iterations", "terms", "list_overflow", "flops", "flops_f32", "dismissed",
              "survivors", "fallback_candidates", "kernel_ms", "setup_ms", "redo_flops", "redo_flops_f32", "redo_kernel_ms", "kernel_launches", "pruned")

    def __init__(self, n, m):
        self.n, self.m = n, m
        self.nll, self.mu, self.C, self.rank, self.pi = np.zeros(0), np.zeros((0, n)), None, [], np.zeros(0, np.int64)
        self.srk, self.slb, self.sC, self.spi = [], np.zeros(0), np.zeros((0, m, 2), np.uint8), np.zeros(0, np.int64)
        self.drk, self.dC_parts, self.dpi_parts = [], [], []      # (the rank-deficient list is never pruned: kept in parts, joined once)
        self.stats = None
        self.since_prune = 0

    def add(self, part, piece, running, window):
        res, sus, _dropped, deg = part
        if self.stats is None:
            self.stats = dict(res["stats"])
        else:
            for k, v in res["stats"].items():
                if k in self.SUMMED:
                    self.stats[k] += v
                elif k == "phase_cycles":
                    self.stats[k] = [a + b for a, b in zip(self.stats[k], v)]
                elif k == "best_nll":
                    self.stats[k] = min(self.stats[k], v)
                elif k == "rejected_bound" and v < self.stats[k]:
                    self.stats[k], self.stats["rejected_rank"] = v, res["stats"]["rejected_rank"]
        k = len(res["nll"])
        if k:
            self.nll = np.concatenate([self.nll, res["nll"]])
            self.mu = np.concatenate([self.mu, res["mu"]])
            self.C = res["C"] if self.C is None else np.concatenate([self.C, res["C"]])
            self.rank += list(res["rank"])
            self.pi = np.concatenate([self.pi, np.full(k, piece, np.int64)])
        if self.n == 3:
            if len(sus[0]):
                self.srk += list(sus[0])
                self.slb = np.concatenate([self.slb, sus[1]])
                self.sC = np.concatenate([self.sC, sus[2].reshape(-1, self.m, 2)])
                self.spi = np.concatenate([self.spi, np.full(len(sus[0]), piece, np.int64)])
            if len(deg[0]):
                self.drk += list(deg[0])
                self.dC_parts.append(np.asarray(deg[1]).reshape(-1, self.m, 2))
                self.dpi_parts.append(np.full(len(deg[0]), piece, np.int64))
        self.since_prune += 1
        if self.since_prune >= 64:                       # long walks: drop what the minimum so far has already ruled out
            self.prune(running, window)

    def prune(self, gmin, window):
        self.since_prune = 0
        if len(self.nll):
            keep = ~(self.nll > gmin + window)           # (NaN never arrives here: the device lists hold finite values)
            self.nll, self.mu, self.C, self.pi = self.nll[keep], self.mu[keep], self.C[keep], self.pi[keep]
            self.rank = [r for r, k in zip(self.rank, keep) if k]
        if len(self.slb):
            ks = np.nonzero(self.slb <= gmin + window)[0]
            self.srk, self.slb, self.sC, self.spi = [self.srk[i] for i in ks], self.slb[ks], self.sC[ks], self.spi[ks]

    def result(self, window):
        gmin = float(self.nll.min()) if len(self.nll) else float("inf")
        self.prune(gmin, window)
        o = np.argsort(self.pi, kind="stable")
        Cc = self.C[o] if self.C is not None else np.zeros((0, self.m) if self.n == 2 else (0, self.m, 2), np.uint8)
        out = {"nll": self.nll[o], "mu": self.mu[o], "rank": [self.rank[i] for i in o], "C": Cc, "stats": self.stats}
        if self.n != 3:
            return out, ([], np.zeros(0), None), ([], None)
        dC = np.concatenate(self.dC_parts) if self.dC_parts else np.zeros((0, self.m, 2), np.uint8)
        dpi = np.concatenate(self.dpi_parts) if self.dpi_parts else np.zeros(0, np.int64)
        so, do = np.argsort(self.spi, kind="stable"), np.argsort(dpi, kind="stable")
        return out, ([self.srk[i] for i in so], self.slb[so], self.sC[so]), ([self.drk[i] for i in do], dC[do])


class Problem:
    """One search instance resident in HBM (theta_problem_create)."""

    def __init__(self, ctx, n, m, tau, r, rN, lb, ub, max_normal=1.0):
        self.ctx, self.n, self.m, self.tau = ctx, int(n), int(m), int(tau)
        r = np.ascontiguousarray(r, dtype=np.int64)
        rN = np.ascontiguousarray(rN, dtype=np.int64)
        lb = np.ascontiguousarray(lb, dtype=np.int32)
        ub = np.ascontiguousarray(ub, dtype=np.int32)
        if not (len(r) == len(rN) == len(lb) == len(ub) == m):
            raise ValueError("r, rN, lb, ub must all have length m")
        h = C.c_void_p()
        _check(load().theta_problem_create(ctx._h, self.n, self.m, self.tau, _p(r, C.c_int64), _p(rN, C.c_int64),
                                           _p(lb, C.c_int32), _p(ub, C.c_int32), float(max_normal), C.byref(h)))
        self._h = h
        self.r, self.rN, self.max_normal = r, rN, float(max_normal)
        self._bounds = ([int(v) for v in lb], [int(v) for v in ub])
        cnt = (C.c_uint64 * 2)()
        _check(load().theta_problem_count(h, cnt))
        self.count = int(cnt[0]) | (int(cnt[1]) << 64)
        self.last_suspects = ([], np.zeros(0), None)
        self.last_degenerate = ([], None)
        self.suspects_dropped = 0
        self.suspect_reruns = 0

    def close(self):
        if getattr(self, "_h", None):
            load().theta_problem_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _shape_C(self, flat, k):
        return flat.reshape(k, self.m) if self.n == 2 else flat.reshape(k, self.m, 2)

    # one theta_search call takes at most this many candidates (n=3: 2^18 wave tasks of 2^13 candidates)
    MAX_PER_CALL = {2: 1 << 40, 3: 1 << 31}

    def set_option(self, name, value):
        """theta_problem_set_option: "n3_no_dismiss", "n3_force_f64", "n3_conv_l2", "n3_per_task", ..."""
        _check(load().theta_problem_set_option(self._h, name.encode(), float(value)))

    MAX_TASKS_PER_CALL = 1 << 17

    def _piece(self, b, e, window, cap, hint, group=None):
        """One theta_search call (group: one theta_search_ranges call over those ranges) with its side lists:
        (result, suspects, suspects_dropped, degenerate)."""
        if hint < float("inf"):
            _check(load().theta_problem_hint(self._h, hint))
        res = self._search_once(b, e, window, cap, hint, group)
        if res["stats"]["list_overflow"]:
            # the device tie list stayed full after the library's three passes: finalists were lost
            raise ListOverflow("%d finalists dropped by the device tie list in ranks [%d, %d): narrow the "
                               "window or split the range" % (res["stats"]["list_overflow"], b, e), "ties")
        if self.n != 3:
            return res, ([], np.zeros(0), None), 0, ([], None)
        sus = self.suspects()
        return res, sus, self.suspects_dropped, self.degenerate()

    def search(self, begin=0, end=None, window=0.5, cap=16384):       # (room for 16 384 finalists per piece up front: a retry for more runs the piece a second time)
        """
        Fused search over ranks [begin, end).  Returns dict(nll, mu, rank (python ints), C, stats).
        Ranges larger than one call can take are walked in pieces; the pieces' finalists are merged here
        (same rule as across GPUs: keep what lies within `window` of the overall minimum).

        n=3 side lists of the whole range end up in self.last_suspects (rejected candidates whose lower bound is within the
        window: the reference reports them at its nu = 1/3 fallback) and self.last_degenerate (all-zero tumour columns).
        A piece whose device suspect list overflowed is searched AGAIN with the minimum of the whole range as its hint
        (a piece whose own minimum is poor lists far too many); if it still overflows the call raises -- it never returns
        an incomplete list.
        """
        end = self.count if end is None else end
        return self.search_ranges([(begin, end)], window, cap)

    def search_ranges(self, ranges, window=0.5, cap=16384):
        """search() over several rank ranges [(begin, end), ...] in rank order (the survivors of a branch and bound, theta_bnb):
        the same walk over pieces, the same merge -- what lies within `window` of the minimum over ALL the ranges, and their side
        lists together."""
        ranges = [(int(b), int(e)) for b, e in ranges]
        begin, end = (ranges[0][0], ranges[-1][1]) if ranges else (0, 0)
        total = sum(e - b for b, e in ranges)
        step = self.MAX_PER_CALL[self.n]
        if total > step * self.MAX_PIECES:
            # e.g. 70 intervals with bounds [0, 2]: 2.5e34 matrices.  The reference would enumerate such a space for ever; here
            # the call says so (the pieces are walked one by one, nothing of this size is ever materialised on the host)
            raise ThetaError(ERR_OVERFLOW, "%d candidate matrices in one search: more than this library walks in one call "
                             "(%d); tighten the bounds, use fewer intervals, or search rank sub-ranges" % (total, step * self.MAX_PIECES))
        running = self._probe(begin, end) if len(ranges) == 1 else self._take_hint()
        if total <= 0:                                   # an empty range (or an empty space: theta_search says so)
            self.last_suspects, self.last_degenerate = ([], np.zeros(0), None), ([], None)
            return self._search_once(begin, end, window, cap)
        # The pieces are walked in order and merged as they come: what is kept between pieces is what can still matter
        # (finalists and suspects within `window` of the minimum so far, the all-zero-column list), not one entry per piece.
        acc = _Merged(self.n, self.m)
        redo = []                                        # pieces whose device suspect list overflowed: (index, b, e, hint used)
        spans = list(reversed(ranges))                   # (stack) what is still to be walked, in rank order
        piece = 0
        # several ranges: runs of them that fit one call (<= 2^31 candidates, <= MAX_TASKS_PER_CALL wave tasks) go through the kernels
        # together (theta_search_ranges) -- thousands of short ranges would each pay a call's fixed overhead otherwise
        batch_ok = self.n == 3 and len(ranges) > 1 and getattr(self, "_h", None) is not None and self.m >= 8
        max_group = 1 << 30                              # ranges per batch: halved whenever a batch overflows one of the device lists
        halves = []                                      # (stack) halves of a piece whose rank-deficient list overflowed, in rank order
        rss0 = _rss_bytes()
        while spans or halves:
            group = None
            if halves:
                b, e = halves.pop()
            else:
                b, e = spans.pop()
                if e - b > step:
                    spans.append((b + step, e))
                    e = b + step
                if e <= b:
                    continue
                if batch_ok and spans and e - b < step:
                    group, cands, tasks = [(b, e)], e - b, (e - b + 8191) // 8192
                    while spans and len(group) < max_group:
                        nb, ne = spans[-1]
                        nt = (ne - nb + 8191) // 8192
                        if cands + (ne - nb) > step or tasks + nt > self.MAX_TASKS_PER_CALL:
                            break
                        spans.pop()
                        if ne > nb:
                            group.append((nb, ne))
                            cands += ne - nb
                            tasks += nt
                    if len(group) == 1:
                        group = None
                    else:
                        e = group[-1][1]
            if piece and piece % 256 == 0 and _rss_bytes() - rss0 > self.MAX_HOST_GROWTH:
                # a guard, not a code path: the merge keeps what can still matter, so the host footprint of a search does not
                # grow with its length -- if it does (round 2 lost three GPU boxes to a list of 1e25 pieces), stop here
                raise ThetaError(ERR_CAPACITY, "host memory grew by %.1f GB while walking ranks [%d, %d): refusing to go on"
                                 % ((_rss_bytes() - rss0) / 2.0 ** 30, begin, b))
            try:
                if group is not None:
                    try:
                        part = self._piece(b, e, window, cap, running, group)
                    except (ListOverflow, DegenerateOverflow, ThetaError) as ex:
                        if isinstance(ex, ThetaError) and not isinstance(ex, (ListOverflow, DegenerateOverflow)) and ex.code != ERR_CAPACITY:
                            raise
                        # (a list of the batch overflowed: smaller batches, in the end the ranges one by one, where the
                        # single-range ladders apply)
                        spans.extend(reversed(group))
                        max_group = len(group) // 2
                        batch_ok = max_group >= 2
                        continue
                    if part[2] > 0:                   # (its suspect list overflowed: the re-search below takes contiguous ranges)
                        spans.extend(reversed(group))
                        max_group = len(group) // 2
                        batch_ok = max_group >= 2
                        continue
                else:
                    part = self._piece(b, e, window, cap, running)       # later pieces start from the minimum found so far
            except DegenerateOverflow:
                if e - b <= (1 << 20):                               # (cannot happen: the list holds 2^20)
                    raise
                mid = (b + e) // 2
                halves.append((mid, e))
                halves.append((b, mid))
                continue
            nl = part[0]["nll"]
            hint_used = running
            if len(nl):
                running = min(running, float(nl.min()))
            if part[2] > 0:
                redo.append((piece, b, e, hint_used, part[2]))       # (its lists are incomplete: searched again below)
            else:
                acc.add(part, piece, running, window)
            piece += 1
        self.suspect_reruns = 0
        for piece, b, e, hint_used, dropped in redo:
            if not running < hint_used:
                raise ListOverflow("n=3 suspect list overflowed in ranks [%d, %d) (%d entries dropped) although "
                                   "the search started from the range's own minimum" % (b, e, dropped), "suspects")
            # (the redone piece may itself hold more rank-deficient candidates than the device list: halved like in the walk above)
            todo = [(b, e)]
            while todo:
                hb, he = todo.pop()
                try:
                    part = self._piece(hb, he, window, cap, running)
                except DegenerateOverflow:
                    if he - hb <= (1 << 20):
                        raise
                    mid = (hb + he) // 2
                    todo += [(mid, he), (hb, mid)]
                    continue
                self.suspect_reruns += 1
                if part[2] > 0:
                    raise ListOverflow("n=3 suspect list overflowed in ranks [%d, %d) (%d entries dropped) with "
                                       "the minimum of the whole range as hint" % (hb, he, part[2]), "suspects")
                acc.add(part, piece, running, window)
        self.suspects_dropped = 0
        out, self.last_suspects, self.last_degenerate = acc.result(window)
        return out

    # pieces per search() call: 2^25 x 2^31 = 2^56 n=3 candidates (7e16: eleven days of this GPU at 7.5e10 candidates/s)
    MAX_PIECES = 1 << 25
    MAX_HOST_GROWTH = 8 << 30            # bytes the host process may grow by during one search (guard, see search())

    def hint(self, nll_upper_bound):
        """One-shot: an NLL already known to be attainable (keeps the next search's lists short)."""
        self._hint = min(getattr(self, "_hint", float("inf")), float(nll_upper_bound))
        _check(load().theta_problem_hint(self._h, self._hint))

    PROBE_MIN_RANGE = 1 << 26     # n=3 ranges at least this long are probed first
    PROBE_SAMPLES, PROBE_SIZE = 16, 1 << 16

    def _take_hint(self):
        running = getattr(self, "_hint", float("inf"))
        self._hint = float("inf")
        return running

    def _probe(self, begin, end):
        """
        n=3: the minimum over a few short sub-ranges spread over [begin, end) -- an attainable NLL the real search can
        start from (theta_problem_hint).  A range whose first stretch is poor (every optimum outside the simplex) would
        otherwise fill the suspect list and solve everything to convergence until it meets a decent candidate.
        Costs ~1e6 candidates; changes nothing in the result.  Includes a hint the caller set before.
        """
        running = getattr(self, "_hint", float("inf"))
        self._hint = float("inf")
        if self.n != 3 or end - begin < self.PROBE_MIN_RANGE or running < float("inf"):
            return running                 # (a caller that already knows an attainable NLL needs no probe)
        stride = (end - begin - self.PROBE_SIZE) // self.PROBE_SAMPLES
        for i in range(self.PROBE_SAMPLES):
            b = begin + i * stride
            if running < float("inf"):
                _check(load().theta_problem_hint(self._h, running))
            res = self._search_once(b, b + self.PROBE_SIZE, 0.0, 64)
            if len(res["nll"]):
                running = min(running, float(res["nll"].min()))
        return running

    def _search_once(self, begin, end, window, cap, hint=float("inf"), group=None):
        st = SearchStats()
        spec = None
        if group is not None:
            spec = np.zeros((len(group), 4), np.uint64)
            for i, (gb, ge) in enumerate(group):
                spec[i] = (gb & 0xFFFFFFFFFFFFFFFF, gb >> 64, (ge - gb) & 0xFFFFFFFFFFFFFFFF, (ge - gb) >> 64)
        while True:
            nll = np.zeros(cap)
            mu = np.zeros((cap, self.n))
            rank = np.zeros((cap, 2), np.uint64)
            Cb = np.zeros(cap * self.m * (self.n - 1), np.uint8)
            n_out = C.c_int()
            if spec is not None:
                rc = load().theta_search_ranges(self._h, len(group), _p(spec, C.c_uint64), float(window), cap, _p(nll, C.c_double),
                                                _p(mu, C.c_double), _p(rank, C.c_uint64), _p(Cb, C.c_uint8), C.byref(n_out), C.byref(st))
            else:
                rc = load().theta_search(self._h, _u128(begin), _u128(end), float(window), cap, _p(nll, C.c_double),
                                         _p(mu, C.c_double), _p(rank, C.c_uint64), _p(Cb, C.c_uint8), C.byref(n_out),
                                         C.byref(st))
            if rc == ERR_CAPACITY and n_out.value > cap:
                cap = n_out.value
                if hint < float("inf"):        # the device hint is one-shot: the retry starts from the same minimum (round-3 advice)
                    _check(load().theta_problem_hint(self._h, hint))
                continue
            _check(rc)
            break
        k = n_out.value
        ranks = [int(rank[i, 0]) | (int(rank[i, 1]) << 64) for i in range(k)]
        return {"nll": nll[:k].copy(), "mu": mu[:k].copy(), "rank": ranks,
                "C": self._shape_C(Cb[:k * self.m * (self.n - 1)], k).copy(), "stats": st.as_dict()}

    def suspects(self):
        """Rejected candidates of the last search whose lower bound lies within the window: (ranks, lbound, C)."""
        n_out = C.c_int()
        load().theta_search_suspects(self._h, -1, None, None, None, C.byref(n_out))
        self.suspects_dropped = n_out.value
        rc = load().theta_search_suspects(self._h, 0, None, None, None, C.byref(n_out))
        k = n_out.value
        if k == 0:
            return [], np.zeros(0), self._shape_C(np.zeros(0, np.uint8), 0)
        rank = np.zeros((k, 2), np.uint64)
        lb = np.zeros(k)
        Cb = np.zeros(k * self.m * (self.n - 1), np.uint8)
        _check(load().theta_search_suspects(self._h, k, _p(rank, C.c_uint64), _p(lb, C.c_double), _p(Cb, C.c_uint8),
                                            C.byref(n_out)))
        return [int(rank[i, 0]) | (int(rank[i, 1]) << 64) for i in range(k)], lb, self._shape_C(Cb, k)

    def degenerate(self):
        """n=3 rank-deficient candidates of the last theta_search (rows (x_i, y_i) on one line -- an all-zero tumour column among
        them --: the reference's outcome for those is not their optimum, csrc/n3_core.hpp: N3Line): (ranks, C)."""
        n_out = C.c_int()
        load().theta_search_degenerate(self._h, -1, None, None, C.byref(n_out))
        if n_out.value:
            raise DegenerateOverflow(ERR_CAPACITY, "%d rank-deficient candidates did not fit the device list" % n_out.value)
        load().theta_search_degenerate(self._h, 0, None, None, C.byref(n_out))
        k = n_out.value
        if k == 0:
            return [], self._shape_C(np.zeros(0, np.uint8), 0)
        rank = np.zeros((k, 2), np.uint64)
        Cb = np.zeros(k * self.m * (self.n - 1), np.uint8)
        _check(load().theta_search_degenerate(self._h, k, _p(rank, C.c_uint64), _p(Cb, C.c_uint8), C.byref(n_out)))
        return [int(rank[i, 0]) | (int(rank[i, 1]) << 64) for i in range(k)], self._shape_C(Cb, k)

    def values(self, begin, count):
        """Per-candidate (nll, mu), NaN = None: the --GET_VALUES dump (the fused kernel's own; n=3 with more than 64 intervals: what
        the reference reports for each candidate, through the generator and theta_solve_batch's kernel)."""
        nll = np.zeros(count)
        mu = np.zeros((count, self.n))
        st = SearchStats()
        _check(load().theta_search_values(self._h, _u128(begin), int(count), _p(nll, C.c_double), _p(mu, C.c_double),
                                          C.byref(st)))
        return nll, mu, st.as_dict()

    def mix_search(self, threshold, leaf_rel=2e-4, cap=1 << 16, propose=False, lines=False, lines_only=False, dive=False):
        """theta_mix_search: the matrices (k, m, 2) uint8 -- in enumeration order, a superset -- whose NLL can be <= threshold for
        some mixture, by branch and bound over the mixture space; and the walk's statistics.  lines: also the rank-deficient
        matrices the reference can report within the threshold at a mixture with negative entries (one more tree per line of the
        alphabet's grid); lines_only: those trees alone; dive: proposals from a beam search without a threshold."""
        mode = (MIX_PROPOSE if propose or dive else 0) | (MIX_LINES if lines else 0) | (MIX_LINES_ONLY if lines_only else 0) | (MIX_DIVE if dive else 0)
        for _attempt in range(2):
            st = MixStats()
            out = np.zeros((max(cap, 1), self.m, 2), np.uint8)
            n_out = C.c_uint64(0)
            rc = load().theta_mix_search(self._h, float(threshold), float(leaf_rel), mode, int(cap), _p(out, C.c_uint8), C.byref(n_out), C.byref(st))
            self.last_mix = st.as_dict()
            if rc == ERR_CAPACITY and n_out.value > cap:           # (the list is longer than the buffer: once more, with room)
                cap = int(n_out.value)
                continue
            break
        _check(rc)
        return out[:n_out.value].copy(), self.last_mix

    def constant_matrices(self):
        """The matrices of ONE repeated row (a, b) within every interval's bounds, (k, m, 2) uint8: rank 1 -- whatever the mixture,
        the model is p_i = rN_i / N --; the lines' trees of theta_mix_search leave them to the caller."""
        lb, ub = max(self._bounds[0]), min(self._bounds[1])
        rows = [(a, b) for b in range(lb, ub + 1) for a in range(lb, ub + 1) if (self.tau - a) * (self.tau - b) >= 0]
        out = np.zeros((len(rows), self.m, 2), np.uint8)
        for k, (a, b) in enumerate(rows):
            out[k, :, 0] = a
            out[k, :, 1] = b
        return out

    def bnb(self, threshold, beam=0, follow_collinear=False, max_nodes=0, cap=1 << 16):
        """theta_bnb: the rank ranges [(begin, end), ...] of the whole space that can hold a matrix whose optimum is <= threshold
        (beam > 0: the ranges of a dive that keeps the `beam` best nodes per level), and the walk's statistics."""
        st = BnbStats()
        while True:
            out = np.zeros((max(cap, 1), 4), np.uint64)
            n_out = C.c_uint64(0)
            rc = load().theta_bnb(self._h, float(threshold), int(beam), 1 if follow_collinear else 0, int(max_nodes), int(cap),
                                  _p(out, C.c_uint64), C.byref(n_out), C.byref(st))
            if rc == ERR_CAPACITY and n_out.value > cap and cap < (1 << 22):
                cap = max(int(n_out.value), 4 * cap)
                continue
            self.last_bnb = st.as_dict(self.m)
            _check(rc)
            break
        k = n_out.value
        rg = []
        for i in range(k):
            b = int(out[i, 0]) | (int(out[i, 1]) << 64)
            c = int(out[i, 2]) | (int(out[i, 3]) << 64)
            rg.append((b, b + c))
        return rg, self.last_bnb

    def witness(self, begin, end, every_log2=0, window=0.0):
        """theta_search_witness: one record (WITNESS_DTYPE) per 2^every_log2-th candidate of [begin, end) -- what the n=3 sieve
        kernel left it at under the instance's current options -- and the call's statistics.  Record i <-> rank begin + (i << every_log2)."""
        need = ((int(end) - int(begin)) + (1 << every_log2) - 1) >> every_log2
        out = np.zeros(max(need, 1), WITNESS_DTYPE)
        running = getattr(self, "_hint", float("inf"))           # (a hint is one-shot, like in search())
        self._hint = float("inf")
        if running < float("inf"):
            _check(load().theta_problem_hint(self._h, running))
        n_out = C.c_uint64(0)
        st = SearchStats()
        _check(load().theta_search_witness(self._h, _u128(begin), _u128(end), float(window), int(every_log2), int(need),
                                           out.ctypes.data_as(C.c_void_p), C.byref(n_out), C.byref(st)))
        return out[:n_out.value], st.as_dict()

    def enumerate(self, begin, count):
        """Candidates begin .. begin+count-1 in the reference's order, as uint8 (count, m[, 2])."""
        out = np.zeros(int(count) * self.m * (self.n - 1), np.uint8)
        _check(load().theta_enumerate(self._h, _u128(begin), int(count), _p(out, C.c_uint8)))
        return self._shape_C(out, int(count))

    def enumerate_device(self, begin, count, device_ptr):
        """Same candidates written to device memory (`device_ptr`: address of count*m*(n-1) bytes on this GPU,
        e.g. torch_tensor.data_ptr()).  Returns the kernels' duration in ms."""
        ms = C.c_double(0.0)
        if isinstance(device_ptr, DeviceArray):
            device_ptr = device_ptr.ptr.value
        _check(load().theta_enumerate_device(self._h, _u128(begin), int(count), C.c_void_p(int(device_ptr)), C.byref(ms)))
        return ms.value


class Comm:
    """
    The library's communicator for a sharded search (theta_comm_create): one process per GPU, RCCL over xGMI -- or, with
    transport="host", the library's TCP star (multi-process tests on machines without GPUs).  No torch anywhere.
    Rendezvous: rank 0 listens on addr:port.  Under `python -m torch.distributed.run` / torchrun the launcher's own store
    occupies MASTER_PORT, so the default port is MASTER_PORT + 1 (override with THETA_COMM_PORT).
    """
    RCCL, HOST = 0, 1

    def __init__(self, ctx=None, rank=None, world=None, addr=None, port=None, transport="rccl"):
        lib = load()
        rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        if port is None:
            port = int(os.environ.get("THETA_COMM_PORT", 0)) or int(os.environ.get("MASTER_PORT", "29400")) + 1
        tr = self.HOST if transport == "host" else self.RCCL
        h = C.c_void_p()
        _check(lib.theta_comm_create(ctx._h if ctx is not None else None, rank, world, addr.encode(), int(port), tr, C.byref(h)))
        self._h, self.ctx, self.rank, self.world, self.transport = h, ctx, rank, world, transport

    def close(self):
        if getattr(self, "_h", None):
            load().theta_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        r, w, t, v = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        n = C.c_uint64()
        _check(load().theta_comm_info(self._h, C.byref(r), C.byref(w), C.byref(t), C.byref(v), C.byref(n)))
        return {"rank": r.value, "world": w.value, "transport": "host" if t.value == self.HOST else "rccl",
                "rccl_version": v.value, "collectives": n.value}

    def barrier(self):
        _check(load().theta_comm_barrier(self._h))

    def _allreduce(self, fn, values):
        v = np.ascontiguousarray(np.atleast_1d(np.asarray(values, dtype=np.float64))).copy()
        _check(fn(self._h, _p(v, C.c_double), len(v)))
        return v

    def allreduce_min(self, values):
        return self._allreduce(load().theta_comm_allreduce_min, values)

    def allreduce_max(self, values):
        return self._allreduce(load().theta_comm_allreduce_max, values)

    def allreduce_sum(self, values):
        return self._allreduce(load().theta_comm_allreduce_sum, values)

    def allgather(self, arr):
        """Every rank's array (same shape and dtype everywhere) stacked along a new first axis."""
        a = np.ascontiguousarray(arr)
        out = np.zeros((self.world,) + a.shape, a.dtype)
        _check(load().theta_comm_allgather(self._h, a.ctypes.data_as(C.c_void_p), a.nbytes, out.ctypes.data_as(C.c_void_p)))
        return out

    def exchange_finalists(self, n, m, recs, window):
        """
        theta_exchange_finalists on a list of records dict(rank, c, mu, nll, vals): returns (merged records of all shards
        within `window` of the global minimum -- and every NaN-NLL record --, in rank order; the global minimum).
        """
        k = len(recs)
        nc = n - 1
        nll = np.array([t["nll"] for t in recs], dtype=np.float64).reshape(k)
        mu = np.array([t["mu"] for t in recs], dtype=np.float64).reshape(k, n)
        vals = np.array([t["vals"] for t in recs], dtype=np.float64).reshape(k, m)
        rk = np.zeros((k, 2), np.uint64)
        for i, t in enumerate(recs):
            rk[i, 0], rk[i, 1] = t["rank"] & 0xFFFFFFFFFFFFFFFF, t["rank"] >> 64
        Cb = np.array([np.asarray(t["c"], dtype=np.uint8).reshape(-1) for t in recs], dtype=np.uint8).reshape(k, m * nc)
        # The output capacity is agreed on BEFORE the exchange: every rank offers room for the records of ALL shards (one
        # all-reduce of the counts), so no rank can find its own room sufficient while another comes back for more -- with
        # uneven shards (all-zero-column records cluster at low ranks) the per-rank guess of round 2 left ranks in different
        # collectives (round-2 advice).  The retry below is collective too: ERR_CAPACITY is returned on every rank.
        cap = max(64, int(self.allreduce_sum(float(k))[0]))
        for attempt in range(8):
            o_nll, o_mu, o_vals = np.zeros(cap), np.zeros((cap, n)), np.zeros((cap, m))
            o_rk, o_C = np.zeros((cap, 2), np.uint64), np.zeros((cap, m * nc), np.uint8)
            n_out, gmin = C.c_int(), C.c_double()
            rc = load().theta_exchange_finalists(self._h, n, m, k, _p(nll, C.c_double), _p(mu, C.c_double), _p(rk, C.c_uint64),
                                                 _p(Cb, C.c_uint8), _p(vals, C.c_double), float(window), cap,
                                                 _p(o_nll, C.c_double), _p(o_mu, C.c_double), _p(o_rk, C.c_uint64),
                                                 _p(o_C, C.c_uint8), _p(o_vals, C.c_double), C.byref(n_out), C.byref(gmin))
            if rc == ERR_CAPACITY and attempt < 7:
                # (collective: every rank sees the same total, gets the same status and comes back with the same capacity)
                cap = max(2 * cap, n_out.value)
                continue
            _check(rc)
            break
        out = []
        for i in range(n_out.value):
            c = o_C[i].reshape(m) if n == 2 else o_C[i].reshape(m, 2)
            out.append({"rank": int(o_rk[i, 0]) | (int(o_rk[i, 1]) << 64), "c": c.copy(), "mu": o_mu[i].copy(),
                        "nll": float(o_nll[i]), "vals": o_vals[i].copy()})
        return out, gmin.value
