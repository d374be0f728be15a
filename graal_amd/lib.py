"""ctypes binding of ``libgraal_hip.so`` (C ABI: ``include/graal_hip.h``).

This is the whole host<->device boundary of the engine: plain pointers and sizes.  There is no CPU
fallback -- if the library or a HIP device is missing, :class:`Engine` raises.
"""
import ctypes
import os

import numpy as np

from . import build as _build

N_OPS = 13
ABI_VERSION = 7
MODE_REF_TRANS_ACCU, MODE_STRICT = 1, 2
MAX_NEIGHBOURS = 10
Q_SCALE = float(1 << 30)
Q_STEP_FAILED = 1 << 32  # graal_eval_candidates_q: added to the first not-finite flag word by a rank whose step failed (include/graal_hip.h)
Q_NAN = -(1 << 63)      # a candidate's Q sum EXACTLY this (INT64_MIN) stands for NaN: a term was not finite (graal_hip.hip: Q_NAN)


def q_to_float(q, c=None, flags=None):
    """A candidate's value from its two int64 sums (graal_hip.hip: q_value): coarse + fine / 2^30; NaN where the fine sum carries the
    not-finite marker (or `flags` is not zero).  `c` -- zero but for a rare candidate -- holds the finite terms of 2^31 log-likelihood units and more, rounded to
    whole units (kernels3.cu:191-210 adds such terms into its float64 sum like any other: the result must stay finite)."""
    q = np.asarray(q, dtype=np.int64)
    out = q.astype(np.float64) / Q_SCALE
    if c is not None:
        c = np.asarray(c, dtype=np.int64)
        if c.any():
            out = np.where(c != 0, c.astype(np.float64) + out, out)
    bad = q == Q_NAN
    if flags is not None:          # (the device buffer of the RCCL path: flags summed over the ranks)
        flags = np.asarray(flags, dtype=np.int64)
        if len(flags) and int(flags.flat[0]) >= Q_STEP_FAILED:
            raise GraalError("the step failed on some rank (a kernel of the step gave up or a list overflowed): the ranks' sums are not scores")
        bad = bad | (flags != 0)
    if bad.any():
        out = np.where(bad, np.nan, out)
    return out
Q_FULL_BAD = -(1 << 63)  # graal_eval_full_q: q[0] == INT64_MIN exactly flags a non-finite / out-of-range term
FIELDS = ("pos", "id_c", "start_bp", "len_bp", "circ", "id", "prev", "next", "l_cont", "l_cont_bp", "ori", "rep",
          "activ", "id_d")  # struct frag, kernels3.cu:9-24

_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)

SYMBOLS = ("graal_abi_version", "graal_create", "graal_destroy", "graal_last_error", "graal_set_params",
           "graal_upload_subfrags", "graal_upload_repeats", "graal_upload_contacts", "graal_upload_contacts_f32", "graal_upload_frags", "graal_download_frags",
           "graal_relabel_contigs", "graal_begin_step", "graal_begin_step_launch", "graal_layout_stats", "graal_eval_full_q", "graal_eval_full_params", "graal_eval_candidates_q",
           "graal_eval_candidates", "graal_exchange_bytes", "graal_attach_exchange", "graal_eval_candidates_x", "graal_exchange_selftest", "graal_detach_exchange", "graal_rccl_unique_id", "graal_attach_rccl", "graal_detach_rccl", "graal_upload_distance_ref", "graal_genome_distance", "graal_apply_move", "graal_set_finisher", "graal_set_mode", "graal_set_timing", "graal_last_timing", "graal_scan_times", "graal_strict_times", "graal_time_scan", "graal_last_counters", "graal_take_carry_correction", "graal_upload_own_obs", "graal_explode", "graal_run_counters",
           "graal_upload_proposal_tables", "graal_step", "graal_step_finish", "graal_steps", "graal_host_np_sum", "graal_host_select_move", "graal_host_neighbours", "graal_host_max_dist_intra")

STEP_DONE, STEP_PAUSED, STEP_FALLBACK, STEP_SELECT = 0, 1, 2, 3
STEPS_ROW = 10   # GRAAL_STEPS_ROW: doubles per step in graal_steps' rows


class StepOut(ctypes.Structure):
    """include/graal_hip.h: graal_step_out"""
    _fields_ = [("stats", ctypes.c_int64 * 8), ("max_id", ctypes.c_int32), ("n_neighbours", ctypes.c_int32),
                ("neighbours", ctypes.c_int32 * 128), ("sample_out", ctypes.c_int32), ("op_sampled", ctypes.c_int32),
                ("id_f_sampled", ctypes.c_int32), ("pad", ctypes.c_int32), ("o", ctypes.c_double),
                ("dist_half_units", ctypes.c_int64), ("scores", ctypes.c_double * (128 * N_OPS)), ("full_likelihood", ctypes.c_double)]


_lib = None


def lib_path():
    return _build.HIP_LIB


def load():
    """Load (never build implicitly on the GPU box: the .so travels with the snapshot) the C-ABI library."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError("libgraal_hip.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = ctypes.CDLL(path)
        L.graal_last_error.restype = ctypes.c_char_p
        L.graal_last_error.argtypes = [ctypes.c_void_p]
        L.graal_destroy.restype = None
        L.graal_destroy.argtypes = [ctypes.c_void_p]
        L.graal_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.graal_set_params.argtypes = [ctypes.c_void_p, _f32p]
        L.graal_upload_subfrags.argtypes = [ctypes.c_void_p, _i32p, _f32p, _i32p, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.c_float]
        L.graal_upload_repeats.argtypes = [ctypes.c_void_p, _i32p, ctypes.c_int32, _i32p, _i32p, ctypes.c_int32, _f32p]
        L.graal_upload_contacts.argtypes = [ctypes.c_void_p, _i32p, _i32p, _i32p, ctypes.c_int64]
        L.graal_upload_contacts_f32.argtypes = [ctypes.c_void_p, _i32p, _i32p, _f32p, ctypes.c_int64]
        L.graal_upload_frags.argtypes = [ctypes.c_void_p, ctypes.POINTER(_i32p), ctypes.c_int32]
        L.graal_download_frags.argtypes = [ctypes.c_void_p, ctypes.POINTER(_i32p)]
        L.graal_relabel_contigs.argtypes = [ctypes.c_void_p, _i32p]
        L.graal_layout_stats.argtypes = [ctypes.c_void_p, _i64p]
        L.graal_begin_step.argtypes = [ctypes.c_void_p, _i64p, _i32p]
        L.graal_begin_step_launch.argtypes = [ctypes.c_void_p]
        L.graal_eval_full_q.argtypes = [ctypes.c_void_p, _i64p]
        L.graal_eval_full_params.argtypes = [ctypes.c_void_p, _f32p, _i64p, _i32p, _i64p]
        L.graal_eval_candidates_q.argtypes = [ctypes.c_void_p, ctypes.c_int32, _i32p, ctypes.c_int32, ctypes.c_int32,
                                              ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        L.graal_eval_candidates.argtypes = [ctypes.c_void_p, ctypes.c_int32, _i32p, ctypes.c_int32, ctypes.c_int32, _f64p]
        L.graal_exchange_bytes.argtypes = [ctypes.c_int32, _i64p]
        L.graal_attach_exchange.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.c_int64, _i64p]
        L.graal_exchange_selftest.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32]
        L.graal_detach_exchange.argtypes = [ctypes.c_void_p]
        L.graal_rccl_unique_id.argtypes = [ctypes.c_void_p]
        L.graal_attach_rccl.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32]
        L.graal_detach_rccl.argtypes = [ctypes.c_void_p]
        L.graal_eval_candidates_x.argtypes = [ctypes.c_void_p, ctypes.c_int32, _i32p, ctypes.c_int32, ctypes.c_int32, _i64p, _i64p]
        L.graal_upload_distance_ref.argtypes = [ctypes.c_void_p, _i32p, _i32p, _i32p, _i32p, ctypes.POINTER(ctypes.c_uint8),
                                                ctypes.c_int32]
        L.graal_genome_distance.argtypes = [ctypes.c_void_p, _i64p]
        L.graal_apply_move.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _i32p]
        L.graal_last_timing.argtypes = [ctypes.c_void_p, _f32p]
        L.graal_last_counters.argtypes = [ctypes.c_void_p, _i64p]
        L.graal_run_counters.argtypes = [ctypes.c_void_p, _i64p]
        L.graal_take_carry_correction.argtypes = [ctypes.c_void_p, _i64p, _i32p]
        L.graal_upload_own_obs.argtypes = [ctypes.c_void_p, _f32p, ctypes.c_int32]
        L.graal_explode.argtypes = [ctypes.c_void_p, _i64p]
        L.graal_set_timing.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        L.graal_set_finisher.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        L.graal_set_mode.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        L.graal_scan_times.argtypes = [ctypes.c_void_p, ctypes.c_int32, _f32p]
        L.graal_strict_times.argtypes = [ctypes.c_void_p, ctypes.c_int32, _f32p]
        L.graal_time_scan.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, _f32p]
        _u8p = ctypes.POINTER(ctypes.c_uint8)
        L.graal_upload_proposal_tables.argtypes = [ctypes.c_void_p, _i32p, _f32p, ctypes.c_int32, ctypes.c_int32, _i32p, ctypes.c_int32,
                                                   _i32p, _i32p, ctypes.c_int32, _u8p, _u8p]
        L.graal_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_int32,
                                 ctypes.c_int32, ctypes.POINTER(StepOut)]
        L.graal_step_finish.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_int32, ctypes.POINTER(StepOut)]
        L.graal_steps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, _i32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_int32,
                                  ctypes.c_int32, _f64p, _i32p, ctypes.POINTER(StepOut)]
        L.graal_host_np_sum.restype = ctypes.c_double
        L.graal_host_np_sum.argtypes = [_f64p, ctypes.c_int64]
        L.graal_host_select_move.argtypes = [ctypes.c_void_p, _f64p, ctypes.c_int32, ctypes.c_int32]
        L.graal_host_neighbours.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, _i32p, ctypes.c_int32]
        L.graal_host_max_dist_intra.argtypes = [_f64p, ctypes.c_double, ctypes.c_int32, _f64p, _i32p]
        _lib = L
    return _lib


class GraalError(RuntimeError):
    pass


_mdi_p = (ctypes.c_double * 5)()
_mdi_x = ctypes.c_double(0.0)
_mdi_info = ctypes.c_int32(0)


def host_max_dist_intra(p5, val_inter, f32):
    """include/graal_hip.h: graal_host_max_dist_intra -- (x, MINPACK info).  No device needed."""
    L = load()
    _mdi_p[0], _mdi_p[1], _mdi_p[2], _mdi_p[3], _mdi_p[4] = (float(v) for v in p5)
    rc = L.graal_host_max_dist_intra(_mdi_p, float(val_inter), 1 if f32 else 0, ctypes.byref(_mdi_x), ctypes.byref(_mdi_info))
    if rc != 0:
        raise GraalError("graal_host_max_dist_intra failed (%d)" % rc)
    return _mdi_x.value, int(_mdi_info.value)


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


class Engine:
    """One handle = one GPU.  Thin, typed wrapper; every failure raises :class:`GraalError`."""

    def __init__(self, device=0):
        self._L = load()
        self._h = ctypes.c_void_p()
        rc = self._L.graal_create(int(device), ctypes.byref(self._h))
        if rc != 0:
            msg = self._L.graal_last_error(self._h).decode() if self._h else "graal_create failed"
            if self._h:
                self._L.graal_destroy(self._h)
                self._h = ctypes.c_void_p()
            raise GraalError("graal_create(device=%d): %s" % (device, msg))
        self.device = int(device)
        self.n = 0
        self._x_map = self._x_view = None
        # buffers of the per-step calls, allocated once (numpy's .ctypes accessor costs more than the call itself)
        self._fb_buf = np.zeros(MAX_NEIGHBOURS, dtype=np.int32)
        self._fb_ptr = self._fb_buf.ctypes.data_as(_i32p)
        self._delta_buf = np.zeros(MAX_NEIGHBOURS * N_OPS, dtype=np.float64)
        self._delta_ptr = self._delta_buf.ctypes.data_as(_f64p)
        self._q_buf = np.zeros(MAX_NEIGHBOURS * N_OPS, dtype=np.int64)
        self._q_ptr = self._q_buf.ctypes.data_as(_i64p)
        self._c_buf = np.zeros(MAX_NEIGHBOURS * N_OPS, dtype=np.int64)     # the coarse sums (graal_eval_candidates_x)
        self._c_ptr = self._c_buf.ctypes.data_as(_i64p)
        self._st_buf = np.zeros(8, dtype=np.int64)
        self._st_ptr = self._st_buf.ctypes.data_as(_i64p)
        self._max_id = ctypes.c_int32(0)
        self._max_id_ref = ctypes.byref(self._max_id)

    def close(self):
        if getattr(self, "_h", None):
            self._L.graal_destroy(self._h)   # (unregisters the exchange segment before its mapping goes away)
            self._h = ctypes.c_void_p()
        if getattr(self, "_x_view", None) is not None:
            self._x_view = None
            try:
                self._x_map.close()
            except (BufferError, ValueError):
                pass
            self._x_map = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise GraalError("%s: %s (code %d)" % (what, self._L.graal_last_error(self._h).decode(), rc))

    # -- uploads ------------------------------------------------------------------------------
    def set_params(self, param8):
        p = _c(np.asarray(param8, dtype=np.float32).reshape(8), np.float32)
        self._ck(self._L.graal_set_params(self._h, p.ctypes.data_as(_f32p)), "graal_set_params")

    def upload_subfrags(self, sub_id, sub_len_kb, sub_accu, n_sub_total, n_frags_per_bins):
        sid = _c(np.asarray(sub_id).reshape(-1, 4), np.int32)
        sl = _c(np.asarray(sub_len_kb).reshape(-1, 3), np.float32)
        sa = _c(np.asarray(sub_accu).reshape(-1, 3), np.int32)
        assert len(sid) == len(sl) == len(sa)
        self._ck(self._L.graal_upload_subfrags(self._h, sid.ctypes.data_as(_i32p), sl.ctypes.data_as(_f32p),
                                               sa.ctypes.data_as(_i32p), len(sid), int(n_sub_total),
                                               ctypes.c_float(float(n_frags_per_bins))), "graal_upload_subfrags")

    def upload_repeats(self, dup_bins, dispatcher, collector, obs_rows):
        """Repeated bins: their ids, the copies of every bin, and their rows of the observation matrix
        ([n_dup, 3, n_sub_total] float32).  Call between upload_subfrags and upload_contacts / upload_frags."""
        d = _c(dup_bins, np.int32)
        disp = _c(np.asarray(dispatcher).reshape(-1, 2), np.int32)
        coll = _c(collector, np.int32)
        obs = _c(np.asarray(obs_rows, dtype=np.float32).reshape(len(d), 3, -1), np.float32)
        self._ck(self._L.graal_upload_repeats(self._h, d.ctypes.data_as(_i32p), len(d), disp.ctypes.data_as(_i32p),
                                              coll.ctypes.data_as(_i32p), len(coll), obs.ctypes.data_as(_f32p)),
                 "graal_upload_repeats")

    def upload_contacts(self, row, col, count):
        """Counts may be integers or float32 (blacklist fill); integer arrays go through the int32 entry point."""
        r, c = _c(row, np.int32), _c(col, np.int32)
        count = np.asarray(count)
        assert len(r) == len(c) == len(count)
        if np.issubdtype(count.dtype, np.integer):
            v = _c(count, np.int32)
            self._ck(self._L.graal_upload_contacts(self._h, r.ctypes.data_as(_i32p), c.ctypes.data_as(_i32p),
                                                   v.ctypes.data_as(_i32p), len(r)), "graal_upload_contacts")
        else:
            v = _c(count, np.float32)
            self._ck(self._L.graal_upload_contacts_f32(self._h, r.ctypes.data_as(_i32p), c.ctypes.data_as(_i32p),
                                                       v.ctypes.data_as(_f32p), len(r)), "graal_upload_contacts_f32")
        self.nnz = len(r)

    def upload_frags(self, soa):
        arrs = [_c(soa[k], np.int32) for k in FIELDS]
        n = len(arrs[0])
        assert all(len(a) == n for a in arrs)
        ptrs = (_i32p * 14)(*[a.ctypes.data_as(_i32p) for a in arrs])
        self._ck(self._L.graal_upload_frags(self._h, ptrs, n), "graal_upload_frags")
        self.n = n

    def download_frags(self, out=None):
        if out is None:
            out = {k: np.empty(self.n, dtype=np.int32) for k in FIELDS}
        ptrs = (_i32p * 14)(*[out[k].ctypes.data_as(_i32p) for k in FIELDS])
        self._ck(self._L.graal_download_frags(self._h, ptrs), "graal_download_frags")
        return out

    # -- layout maintenance ------------------------------------------------------------------
    def relabel_contigs(self):
        m = ctypes.c_int32(0)
        self._ck(self._L.graal_relabel_contigs(self._h, ctypes.byref(m)), "graal_relabel_contigs")
        return int(m.value)

    def begin_step_launch(self):
        """Launch what begin_step launches and return at once (optional; begin_step then only waits)."""
        self._ck(self._L.graal_begin_step_launch(self._h), "graal_begin_step_launch")

    def begin_step(self):
        """(stats[8], max_id): layout statistics + contig relabel + index rebuild with one synchronisation."""
        rc = self._L.graal_begin_step(self._h, self._st_ptr, self._max_id_ref)
        if rc != 0:
            self._ck(rc, "graal_begin_step")
        return self._st_buf.copy(), int(self._max_id.value)

    def layout_stats(self):
        out = np.zeros(8, dtype=np.int64)
        self._ck(self._L.graal_layout_stats(self._h, out.ctypes.data_as(_i64p)), "graal_layout_stats")
        return out

    def apply_move(self, fA, fB, op, max_id, wait=True):
        """Commit a candidate.  wait=False does not synchronise (the stale-paste count then comes with begin_step)."""
        st = ctypes.c_int32(0)
        self._ck(self._L.graal_apply_move(self._h, int(fA), int(fB), int(op), int(max_id),
                                          ctypes.byref(st) if wait else None), "graal_apply_move")
        return int(st.value)

    # -- likelihood ---------------------------------------------------------------------------
    def eval_full_q(self):
        q = np.zeros(2, dtype=np.int64)
        self._ck(self._L.graal_eval_full_q(self._h, q.ctypes.data_as(_i64p)), "graal_eval_full_q")
        return q

    def eval_full_params(self, param8=None):
        """graal_eval_full_params: (q[2], stats[8], max_id) -- new parameters (kept in force), relabel, full evaluation, one wait."""
        q = np.zeros(2, dtype=np.int64)
        p = None if param8 is None else _c(np.asarray(param8, dtype=np.float32).reshape(8), np.float32)
        rc = self._L.graal_eval_full_params(self._h, None if p is None else p.ctypes.data_as(_f32p), self._st_ptr, self._max_id_ref,
                                            q.ctypes.data_as(_i64p))
        if rc != 0:
            self._ck(rc, "graal_eval_full_params")
        return q, self._st_buf.copy(), int(self._max_id.value)

    def eval_full(self):
        q = self.eval_full_q()
        if int(q[0]) == Q_FULL_BAD:         # a term was not finite / out of range (the reference would return -inf / NaN)
            return float("nan")
        return float(int(q[0]) + int(q[1])) / Q_SCALE

    def eval_candidates(self, fA, fB, max_id):
        """Delta logL of the 13 candidates of every neighbour: float64 [K, 13].  Single GPU, synchronous."""
        K = len(fB)
        if 1 <= K <= MAX_NEIGHBOURS:   # the per-step path: preallocated buffers, cached pointers (this call is ~10 % of a step)
            self._fb_buf[:K] = fB
            rc = self._L.graal_eval_candidates(self._h, int(fA), self._fb_ptr, K, int(max_id), self._delta_ptr)
            if rc != 0:
                self._ck(rc, "graal_eval_candidates")
            return self._delta_buf[:K * N_OPS].reshape(K, N_OPS).copy()
        fb = _c(fB, np.int32)
        out = np.zeros((len(fb), N_OPS), dtype=np.float64)
        for k0 in range(0, len(fb), MAX_NEIGHBOURS):  # > 10 neighbours (copies of repeated bins): one scan pass per group of 10
            out[k0:k0 + MAX_NEIGHBOURS] = self.eval_candidates(fA, fb[k0:k0 + MAX_NEIGHBOURS], max_id)
        return out

    # -- genome distance ------------------------------------------------------------------------
    def upload_distance_ref(self, init_prev, init_next, init_ori, orientable, counted):
        a = [_c(x, np.int32) for x in (init_prev, init_next, init_ori, orientable)]
        c = _c(counted, np.uint8)
        assert all(len(x) == len(c) for x in a)
        self._ck(self._L.graal_upload_distance_ref(self._h, *[x.ctypes.data_as(_i32p) for x in a],
                                                   c.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), len(c)),
                 "graal_upload_distance_ref")

    def genome_distance_half_units(self):
        v = ctypes.c_int64(0)
        self._ck(self._L.graal_genome_distance(self._h, ctypes.byref(v)), "graal_genome_distance")
        return int(v.value)

    # -- node-local exchange through pinned host memory (one process per GPU) ------------------
    def exchange_bytes(self, world):
        b = ctypes.c_int64(0)
        self._ck(self._L.graal_exchange_bytes(int(world), ctypes.byref(b)), "graal_exchange_bytes")
        return int(b.value)

    def step_seq(self):
        """Sequence number of the last candidate evaluation of this handle."""
        s = ctypes.c_int64(0)
        self._ck(self._L.graal_attach_exchange(self._h, None, 0, 0, 1, 0, ctypes.byref(s)), "graal_attach_exchange")
        return int(s.value)

    def attach_exchange(self, shared_map, rank, world, seq_floor):
        """`shared_map`: an ``mmap`` of the segment all ranks of the node share (>= exchange_bytes(world), zero filled).
        The engine keeps it alive; from now on :meth:`eval_candidates_x` returns sums over all ranks."""
        view = ctypes.c_char.from_buffer(shared_map)
        s = ctypes.c_int64(0)
        self._ck(self._L.graal_attach_exchange(self._h, ctypes.c_void_p(ctypes.addressof(view)), len(shared_map), int(rank),
                                               int(world), int(seq_floor), ctypes.byref(s)), "graal_attach_exchange")
        self._x_map, self._x_view = shared_map, view
        return int(s.value)

    def exchange_selftest(self, tag, phase):
        """phase 0: this rank's GPU tags its slots; (barrier); phase 1: True if this host sees every rank's tag."""
        rc = self._L.graal_exchange_selftest(self._h, int(tag), int(phase))
        if phase == 0:
            self._ck(rc, "graal_exchange_selftest")
        return rc == 0

    @staticmethod
    def rccl_available():
        """True if librccl can be loaded in this process (what graal_attach_rccl needs; no communicator is made)."""
        buf = ctypes.create_string_buffer(128)
        return load().graal_rccl_unique_id(buf) == 0

    @staticmethod
    def rccl_unique_id():
        """The 128-byte id of a new RCCL communicator (rank 0 makes it, every rank attaches with it)."""
        buf = ctypes.create_string_buffer(128)
        rc = load().graal_rccl_unique_id(buf)
        if rc != 0:
            raise GraalError("graal_rccl_unique_id failed (%d): RCCL not found?" % rc)
        return bytes(buf.raw)

    def attach_rccl(self, id128, rank, world):
        """From now on the ranks' Q vectors are summed by one ncclAllReduce on the engine's stream (include/graal_hip.h)."""
        self._ck(self._L.graal_attach_rccl(self._h, ctypes.c_char_p(bytes(id128)), int(rank), int(world)), "graal_attach_rccl")

    def detach_rccl(self):
        self._ck(self._L.graal_detach_rccl(self._h), "graal_detach_rccl")

    def detach_exchange(self):
        self._ck(self._L.graal_detach_exchange(self._h), "graal_detach_exchange")
        if self._x_view is not None:
            self._x_view = None
            try:
                self._x_map.close()
            except (BufferError, ValueError):
                pass
            self._x_map = None

    def eval_candidates_x(self, fA, fB, max_id):
        """Sharded, synchronous: float64 [K, 13] = (sum over ALL ranks of the int64 sums) / 2^30; every rank calls it."""
        K = len(fB)
        if 1 <= K <= MAX_NEIGHBOURS:
            self._fb_buf[:K] = fB
            rc = self._L.graal_eval_candidates_x(self._h, int(fA), self._fb_ptr, K, int(max_id), self._q_ptr, self._c_ptr)
            if rc != 0:
                self._ck(rc, "graal_eval_candidates_x")
            return q_to_float(self._q_buf[:K * N_OPS], self._c_buf[:K * N_OPS]).reshape(K, N_OPS)
        fb = _c(fB, np.int32)
        out = np.zeros((len(fb), N_OPS), dtype=np.float64)
        for k0 in range(0, len(fb), MAX_NEIGHBOURS):
            out[k0:k0 + MAX_NEIGHBOURS] = self.eval_candidates_x(fA, fb[k0:k0 + MAX_NEIGHBOURS], max_id)
        return out

    def eval_candidates_q_async(self, fA, fB, max_id, d_out_ptr, stream_ptr, rank=0, world=1):
        """Asynchronous Q-valued form for the multi-GPU path: K*13 int64 into a DEVICE buffer on `stream`."""
        fb = _c(fB, np.int32)
        assert 1 <= len(fb) <= MAX_NEIGHBOURS
        self._ck(self._L.graal_eval_candidates_q(self._h, int(fA), fb.ctypes.data_as(_i32p), len(fb), int(max_id),
                                                 int(rank), int(world), ctypes.c_void_p(int(d_out_ptr)),
                                                 ctypes.c_void_p(int(stream_ptr) if stream_ptr else 0)),
                 "graal_eval_candidates_q")

    # -- the per-step host logic in C (graal_step) ---------------------------------------------------
    def upload_proposal_tables(self, xk, pk, id_d, dispatcher, collector, dup_bin_flags, black_frag_flags):
        xk = _c(xk, np.int32); pk = _c(pk, np.float32)
        idd = _c(id_d, np.int32); disp = _c(np.asarray(dispatcher).reshape(-1, 2), np.int32); coll = _c(collector, np.int32)
        dup = _c(dup_bin_flags, np.uint8); blk = _c(black_frag_flags, np.uint8)
        assert xk.shape == pk.shape and len(dup) == xk.shape[0] == len(disp) and len(blk) == len(idd)
        u8 = ctypes.POINTER(ctypes.c_uint8)
        self._ck(self._L.graal_upload_proposal_tables(self._h, xk.ctypes.data_as(_i32p), pk.ctypes.data_as(_f32p), xk.shape[0], xk.shape[1],
                                                      idd.ctypes.data_as(_i32p), len(idd), disp.ctypes.data_as(_i32p),
                                                      coll.ctypes.data_as(_i32p), len(coll), dup.ctypes.data_as(u8), blk.ctypes.data_as(u8)),
                 "graal_upload_proposal_tables")
        self.step_out = StepOut()
        self._step_ref = ctypes.byref(self.step_out)
        self.step_scores = np.frombuffer(self.step_out, dtype=np.float64, count=128 * N_OPS, offset=StepOut.scores.offset)

    def step(self, mt_addr, fA, delta, likelihood_t, flags, prev_circ):
        """graal_step: STEP_DONE / STEP_PAUSED / STEP_FALLBACK / STEP_SELECT; results in self.step_out"""
        rc = self._L.graal_step(self._h, mt_addr, fA, delta, likelihood_t, flags, prev_circ, self._step_ref)
        if rc >= 16:
            self._ck(rc - 16, "graal_step")
        return rc

    def steps(self, mt_addr, ids, delta, likelihood_t, flags, prev_circ):
        """graal_steps: a run of steps in one call.  Returns (rc of the last step started, rows[n_done, STEPS_ROW]); a step that
        did not end STEP_DONE has left its state in self.step_out, like step()."""
        ids = _c(ids, np.int32)
        rows = np.empty((len(ids), STEPS_ROW), dtype=np.float64)
        n_done = ctypes.c_int32(0)
        rc = self._L.graal_steps(self._h, mt_addr, ids.ctypes.data_as(_i32p), len(ids), delta, likelihood_t, flags, prev_circ,
                                 rows.ctypes.data_as(_f64p), ctypes.byref(n_done), self._step_ref)
        if rc >= 16:
            self._ck(rc - 16, "graal_steps")
        return rc, rows[:n_done.value]

    def step_finish(self, mt_addr, likelihood_t, flags):
        rc = self._L.graal_step_finish(self._h, mt_addr, likelihood_t, flags, self._step_ref)
        if rc >= 16:
            self._ck(rc - 16, "graal_step_finish")
        return rc

    def set_finisher(self, enabled):
        """Let the table kernel's last block finish short-contig steps (default) or always use the finishing kernel."""
        self._ck(self._L.graal_set_finisher(self._h, 1 if enabled else 0), "graal_set_finisher")

    def set_mode(self, ref_trans_accu=False, strict=False):
        """Reference-arithmetic switches (include/graal_hip.h: GRAAL_MODE_REF_TRANS_ACCU = 1, GRAAL_MODE_STRICT = 2)."""
        self._ck(self._L.graal_set_mode(self._h, (MODE_REF_TRANS_ACCU if ref_trans_accu else 0) | (MODE_STRICT if strict else 0)),
                 "graal_set_mode")

    def set_timing(self, enabled):
        self._ck(self._L.graal_set_timing(self._h, int(enabled)), "graal_set_timing")

    def last_timing(self):
        t = np.zeros(4, dtype=np.float32)
        self._ck(self._L.graal_last_timing(self._h, t.ctypes.data_as(_f32p)), "graal_last_timing")
        return t

    def scan_times(self, n):
        """k_scan durations (ms) of the last n candidate evaluations (HIP event pairs recorded around each launch)."""
        t = np.zeros(int(n), dtype=np.float32)
        self._ck(self._L.graal_scan_times(self._h, int(n), t.ctypes.data_as(_f32p)), "graal_scan_times")
        return t

    def strict_times(self, n):
        """Durations (ms) of the tiled reference-arithmetic kernel (k_strict2) in the last n evaluations that carried an event pair."""
        t = np.zeros(int(n), dtype=np.float32)
        self._ck(self._L.graal_strict_times(self._h, int(n), t.ctypes.data_as(_f32p)), "graal_strict_times")
        return t

    def time_scan(self, K, reps=50):
        """Average k_scan duration (ms) over `reps` back-to-back replays between two HIP events."""
        ms = ctypes.c_float(0.0)
        self._ck(self._L.graal_time_scan(self._h, int(K), int(reps), ctypes.byref(ms)), "graal_time_scan")
        return float(ms.value)

    def take_carry_correction(self):
        """graal_take_carry_correction: (correction in log-likelihood units, valid) of the commits since the last take."""
        q, v = ctypes.c_int64(0), ctypes.c_int32(0)
        self._ck(self._L.graal_take_carry_correction(self._h, ctypes.byref(q), ctypes.byref(v)), "graal_take_carry_correction")
        return float(q.value) / Q_SCALE, bool(v.value)

    def upload_own_obs(self, own):
        """graal_upload_own_obs: float32 [n_bins, 3], the observed counts of every bin's own sub-fragment pairs over the WHOLE contact list."""
        own = _c(own, np.float32)
        assert own.ndim == 2 and own.shape[1] == 3
        self._ck(self._L.graal_upload_own_obs(self._h, own.ctypes.data_as(_f32p), own.shape[0]), "graal_upload_own_obs")

    def explode(self):
        """graal_explode: explode_genome's loop (relabel + eject, fragment by fragment) as one call; returns the stale-paste count."""
        v = ctypes.c_int64(0)
        self._ck(self._L.graal_explode(self._h, ctypes.byref(v)), "graal_explode")
        return int(v.value)

    def discard_carry_correction(self):
        """The caller holds a full evaluation of the current layout: the corrections of the commits up to it are void."""
        self._ck(self._L.graal_take_carry_correction(self._h, None, None), "graal_take_carry_correction")

    def run_counters(self):
        """include/graal_hip.h: graal_run_counters -- evaluations, repeats behind events (`fallbacks`), in-kernel waits in use,
        k_strict2 launches behind k_gprep's word / behind the event, k_strict_flat launches, hand-overs to a finishing kernel, finisher give-ups."""
        c = np.zeros(12, dtype=np.int64)
        self._ck(self._L.graal_run_counters(self._h, c.ctypes.data_as(_i64p)), "graal_run_counters")
        return dict(zip(("evaluations", "fallbacks", "in_kernel_waits_in_use", "strict2_behind_the_word", "strict2_behind_the_event",
                         "flat_launches", "handed_to_a_finishing_kernel", "finisher_gave_up", "unit_list_grown", "unit_list_capacity",
                         "carried_totals_repaired"),
                        (int(x) for x in c[:11])))

    def last_counters(self):
        c = np.zeros(4, dtype=np.int64)
        self._ck(self._L.graal_last_counters(self._h, c.ctypes.data_as(_i64p)), "graal_last_counters")
        return c
