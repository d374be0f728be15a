"""Shared helpers of the test-suite (CPU side)."""
import ctypes

import numpy as np

from graal_amd import build as gbuild
from oracle import oracle as O

FIELDS = O.FIELDS
_i32p = ctypes.POINTER(ctypes.c_int32)


def random_layout(rng, n, n_contigs=None, p_circ=0.3, p_rev=0.4, max_len_bp=5000, len_grid=1):
    """A random VALID fragment layout: random partition into contigs, random order, orientations and a few
    circular contigs (as the reference's paste_contigs produces them, kernels3.cu:1977-2033)."""
    if n_contigs is None:
        n_contigs = int(rng.randint(1, max(2, n // 2)))
    n_contigs = max(1, min(n_contigs, n))
    perm = rng.permutation(n)
    cuts = np.sort(rng.choice(np.arange(1, n), size=n_contigs - 1, replace=False)) if n_contigs > 1 else np.array([], int)
    groups = np.split(perm, cuts)
    s = O.new_state(n)
    s["len_bp"][:] = (1 + rng.randint(0, max_len_bp, size=n)) * len_grid
    s["ori"][:] = np.where(rng.random_sample(n) < p_rev, -1, 1)
    labels = rng.permutation(n_contigs)  # dense labels 0..n_contigs-1 as after a relabel
    for g, lab in zip(groups, labels):
        L = int(s["len_bp"][g].sum())
        circ = int(len(g) >= 2 and rng.random_sample() < p_circ)
        start = 0
        for k, f in enumerate(g):
            s["pos"][f] = k
            s["id_c"][f] = lab
            s["start_bp"][f] = start
            s["circ"][f] = circ
            s["l_cont"][f] = len(g)
            s["l_cont_bp"][f] = L
            s["prev"][f] = g[k - 1] if k > 0 else (g[-1] if circ else -1)
            s["next"][f] = g[k + 1] if k + 1 < len(g) else (g[0] if circ else -1)
            start += int(s["len_bp"][f])
    s["id"][:] = np.arange(n)
    s["id_d"][:] = np.arange(n)
    return s


def check_invariants(c):
    """The reference's own run-time checks (cuda_lib_gl.py:1530-1537) + link consistency (diagnosis, :1016-1042)."""
    n = len(c["pos"])
    assert not np.any(c["pos"] < 0) and not np.any(c["l_cont"] <= 0) and not np.any(c["l_cont_bp"] <= 0)
    assert not np.any(c["start_bp"] < 0) and not np.any(c["l_cont_bp"] - c["start_bp"] <= 0)
    assert not np.any((c["start_bp"] != 0) & (c["pos"] == 0)) and not np.any((c["start_bp"] == 0) & (c["pos"] != 0))
    assert np.all(c["id"] == np.arange(n))
    for lab in np.unique(c["id_c"]):
        m = np.nonzero(c["id_c"] == lab)[0]
        order = m[np.argsort(c["pos"][m], kind="stable")]
        assert list(c["pos"][order]) == list(range(len(order))), "positions must be dense"
        assert np.all(c["l_cont"][order] == len(order))
        assert np.all(c["l_cont_bp"][order] == c["len_bp"][order].sum())
        assert np.all(np.cumsum(c["len_bp"][order]) - c["len_bp"][order] == c["start_bp"][order])
        circ = c["circ"][order[0]]
        assert np.all(c["circ"][order] == circ)
        for k, f in enumerate(order):
            want_prev = order[k - 1] if k > 0 else (order[-1] if circ else -1)
            want_next = order[k + 1] if k + 1 < len(order) else (order[0] if circ else -1)
            if len(order) == 1 and circ == 0:
                want_prev = want_next = -1
            assert c["prev"][f] == want_prev and c["next"][f] == want_next, (lab, k, f)


def oracle_candidate(state, fA, fB, op, max_id):
    """The reference's kernel sequence for one candidate (cuda_lib_gl.py:841-954) on fresh slots.
    Returns (candidate state, stale flag)."""
    n = len(state["pos"])
    D = O.DenseOracle
    out, ids = O.new_state(n), np.zeros(n, np.int32)
    if op <= 8:
        pop = O.new_state(n)
        D.pop_out(pop, state, ids, fA, max_id)
        m2 = ids.max()
        if op == 0:
            D.copy(out, pop)
        elif op == 1:
            D.flip(out, state, fA)
        elif op in (2, 3):
            D.pop_in(1, out, pop, fA, fB, m2, 1 if op == 2 else -1)
        elif op in (4, 5):
            D.pop_in(2, out, pop, fA, fB, m2, 1 if op == 4 else -1)
        elif op in (6, 7):
            D.pop_in(3, out, pop, fA, fB, m2, 1 if op == 6 else -1)
        else:
            D.swap_activity(out, pop, fA, m2)
        return out, False
    upA, upB = (op - 9) >> 1, (op - 9) & 1
    t1, t2 = O.new_state(n), O.new_state(n)
    D.split(t1, state, ids, fA, upA, max_id)
    m1 = ids.max()
    D.split(t2, t1, ids, fB, upB, m1)
    m2 = ids.max()
    stale = D.paste(out, t2, fA, fB, m2)
    return out, stale > 0


# ---- ctypes access to the TEST-ONLY host build of frag_ops.h ---------------------------------------------
_hc = None


def hostcheck():
    global _hc
    if _hc is None:
        _hc = ctypes.CDLL(gbuild.build_hostcheck())
        _hc.hc_apply_move.restype = ctypes.c_int
    return _hc


def _ptrs(s):
    arr = (_i32p * 14)()
    for i, k in enumerate(FIELDS):
        arr[i] = s[k].ctypes.data_as(_i32p)
    return arr


def hc_apply_move(state, fA, fB, op, max_id):
    n = len(state["pos"])
    out = O.new_state(n)
    stale = hostcheck().hc_apply_move(op, int(fA), int(fB), int(max_id), _ptrs(state), _ptrs(out), n)
    return out, stale


def hc_piece_tables(state, fA, fB, max_id):
    n = len(state["pos"])
    piece = np.zeros(n, np.int32)
    xf = np.zeros((13, 7, 5), np.int32)
    changed = np.zeros(13, np.uint64)
    rep = np.zeros(7, np.int32)
    hostcheck().hc_piece_tables(int(fA), int(fB), int(max_id), _ptrs(state), n, piece.ctypes.data_as(_i32p),
                                xf.ctypes.data_as(_i32p), changed.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                rep.ctypes.data_as(_i32p))
    return piece, xf, changed, rep
