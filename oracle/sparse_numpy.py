"""TEST / BASELINE INFRASTRUCTURE -- vectorised numpy re-score of the SAME likelihood in sparse (COO) form.

This is the "numpy re-score on the box's own host cores" that BASELINE.json's north_star asks to be reported next
to the GPU numbers (bench.py ``cpu_baseline`` leg, kind "port"), and a second, independent check of the sparse
reformulation against the dense oracle (tests/test_sparse_reformulation.py).  Not product code.

    logL = sum_{contacts} [ob*log(ex) - lf(ob)]  -  sum_{all sub-fragment pairs} ex

with ex computed in float32 exactly as the reference does (kernels3.cu:120-166, 2997-3078, 3184-3195) and the
all-pairs term split into a layout independent all-trans part plus a windowed cis correction.  The trans-branch
accu quirk of the reference (kernels3.cu:3155) is NOT applied (== oracle with fix_trans_accu=True; identical to the
reference whenever every bin has uniform accu).
"""
import numpy as np

f32 = np.float32


def lf_term(ob):
    """log-factorial term of evaluate_likelihood_double (kernels3.cu:191-210) for an array of counts."""
    ob = np.asarray(ob, dtype=np.float64)
    out = np.zeros_like(ob)
    big = ob >= 15
    out[big] = ob[big] * np.log(ob[big]) - ob[big] + np.log(np.sqrt(ob[big] * 2.0 * np.pi))
    small = (ob > 0) & ~big
    n = np.floor(ob[small]).astype(np.float32)
    fact = np.ones_like(n, dtype=np.float32)
    for c in range(1, 10):  # float32 running product for n < 10
        fact = np.where(n >= c, fact * f32(c), fact).astype(np.float32)
    stir = (np.power(n, n, dtype=np.float32) * np.exp(-n, dtype=np.float32)
            * np.sqrt((2 * np.pi * n.astype(np.float64)).astype(np.float32), dtype=np.float32)).astype(np.float32)
    out[small] = np.log(np.where(n < 10, fact, stir).astype(np.float64))
    return out


def rippe_f32(s, p):
    s = np.asarray(s, dtype=np.float32)
    kuhn, lm, c1, slope, d, d_max, fact, v = [f32(x) for x in p]
    ok = (s > 0) & (s < d_max)
    ss = np.where(ok, s, f32(1.0)).astype(np.float32)
    inner = (np.power(ss * lm / kuhn, f32(2.0), dtype=np.float32) + d).astype(np.float32)
    r = (c1 * np.power(ss, slope, dtype=np.float32) * np.exp((d - f32(2)) / inner, dtype=np.float32)) * fact
    r = np.where(ok, r, f32(0)).astype(np.float32)
    return np.maximum(r, v)


def rippe_circ_f32(s, s_tot, p):
    s = np.asarray(s, dtype=np.float32)
    s_tot = np.asarray(s_tot, dtype=np.float32)
    kuhn, lm, c1, slope, d, d_max, fact, v = [f32(x) for x in p]
    ok = (s > 0) & (s < d_max)
    ss = np.where(ok, s, f32(1.0)).astype(np.float32)
    K = lm / kuhn
    nmax = K * f32(1)
    n = (K * ss * (s_tot - ss) / s_tot).astype(np.float32)
    n = np.where(ok, n, f32(1.0)).astype(np.float32)
    norm_lin = rippe_f32(ss, p)
    k3 = np.power(kuhn, f32(-3.0), dtype=np.float32)
    norm_circ = (k3 * np.power(nmax, slope, dtype=np.float32)
                 * np.exp((d - f32(2.0)) / (np.power(nmax, f32(2.0), dtype=np.float32) + d), dtype=np.float32)) * fact
    with np.errstate(invalid="ignore", divide="ignore"):
        val = (k3 * np.power(n, slope, dtype=np.float32)
               * np.exp((d - f32(2.0)) / (np.power(n, f32(2.0), dtype=np.float32) + d), dtype=np.float32)) * fact
        r = (val * norm_lin / norm_circ).astype(np.float32)
    r = np.where(ok, r, f32(0)).astype(np.float32)
    return np.maximum(r, v)


class SparseScorer:
    """Static data of one problem; ``full(state)`` re-scores a layout from scratch."""

    def __init__(self, row, col, count, sub_id, sub_len_kb, sub_accu, nfpb, param):
        self.row = np.ascontiguousarray(row, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int64)
        self.count = np.ascontiguousarray(count, dtype=np.float64)
        sub_id = np.asarray(sub_id, dtype=np.int64).reshape(-1, 4)
        self.n_bins = len(sub_id)
        self.n_sub = sub_id[:, 3].copy()
        self.sub_len = np.asarray(sub_len_kb, dtype=np.float32).reshape(-1, 3)
        self.sub_accu = np.asarray(sub_accu, dtype=np.int64).reshape(-1, 3)
        S = int(self.n_sub.sum())
        self.S = S
        self.bin_of = np.zeros(S, np.int64)
        self.slot_of = np.zeros(S, np.int64)
        for k in range(3):
            m = self.n_sub > k
            self.bin_of[sub_id[m, k]] = np.nonzero(m)[0]
            self.slot_of[sub_id[m, k]] = k
        self.accu_of = self.sub_accu[self.bin_of, self.slot_of]
        self.nfpb = f32(nfpb)
        self.param = np.asarray(param, dtype=np.float32).reshape(8)
        self.c_lf = float(lf_term(self.count).sum())
        self._t_all = None

    def set_param(self, param):
        self.param = np.asarray(param, dtype=np.float32).reshape(8)
        self._t_all = None

    def _c_trans(self, prod):
        return (self.param[7] * (np.asarray(prod, dtype=np.int64).astype(np.float32) / self.nfpb)).astype(np.float32)

    def t_all(self):
        if self._t_all is None:
            vals, cnt = np.unique(self.accu_of, return_counts=True)
            tot = 0.0
            for u, cu in zip(vals, cnt):
                tot += float((cu * cnt.astype(np.float64) * self._c_trans(u * vals).astype(np.float64)).sum())
            self_pairs = 0.0
            for a in range(3):
                for b in range(3):
                    m = (self.n_sub > a) & (self.n_sub > b)
                    self_pairs += float(self._c_trans(self.sub_accu[m, a] * self.sub_accu[m, b]).astype(np.float64).sum())
            self._t_all = 0.5 * (tot - self_pairs)
        return self._t_all

    def centres(self, state):
        """float32 centre (kb) of every sub-fragment, walking each bin in its orientation (kernels3.cu:2997-3060)."""
        start = (state["start_bp"].astype(np.float32) / f32(1000.0)).astype(np.float32)
        fwd = state["ori"] == 1
        lim = self.n_sub - 1
        idx = np.arange(self.n_bins)
        c = np.zeros((self.n_bins, 3), np.float32)
        run = start.copy()
        for w in range(3):  # w = walk index
            slot = np.where(fwd, w, lim - w)
            ok = w <= lim
            sl = np.clip(slot, 0, 2)
            ln = np.where(ok, self.sub_len[idx, sl], f32(0)).astype(np.float32)
            cw = (run + ln / f32(2.0)).astype(np.float32)
            c[idx[ok], sl[ok]] = cw[ok]
            run = (run + ln).astype(np.float32)
        return c

    def _ex(self, state, centres, sx, sy):
        bx, by = self.bin_of[sx], self.bin_of[sy]
        norm = ((self.accu_of[sx] * self.accu_of[sy]).astype(np.float32) / self.nfpb).astype(np.float32)
        cis = state["id_c"][bx] == state["id_c"][by]
        s = np.abs(centres[by, self.slot_of[sy]] - centres[bx, self.slot_of[sx]]).astype(np.float32)
        circ = cis & (state["circ"][bx] == 1)
        r = rippe_f32(np.where(cis, s, f32(0)), self.param)
        if np.any(circ):
            s_tot = (state["l_cont_bp"][bx].astype(np.float32) / f32(1000.0)).astype(np.float32)
            rc = rippe_circ_f32(np.where(circ, s, f32(0)), np.where(circ, s_tot, f32(1.0)), self.param)
            r = np.where(circ, rc, r)
        return np.where(cis, (r * norm).astype(np.float32), (self.param[7] * norm).astype(np.float32)), norm

    def nnz_part(self, state, centres=None, lo=0, hi=None, same_bin=True):
        centres = self.centres(state) if centres is None else centres
        hi = len(self.row) if hi is None else hi
        ex, _ = self._ex(state, centres, self.row[lo:hi], self.col[lo:hi])
        w = self.count[lo:hi]
        if not same_bin:
            w = np.where(self.bin_of[self.row[lo:hi]] == self.bin_of[self.col[lo:hi]], 0.0, w)
        return float((w * np.log(ex.astype(np.float64))).sum())

    def mass_cis(self, state, centres=None, same_bin=True):
        """sum over cis sub-fragment pairs of (ex - ex_trans) [different bins] + ex [same bin]."""
        centres = self.centres(state) if centres is None else centres
        sub = np.arange(self.S)
        b = self.bin_of
        key = state["id_c"][b].astype(np.int64)
        # order sub-fragments along their contig: by (contig, bin position, walk index)
        walk = np.where(state["ori"][b] == 1, self.slot_of, (self.n_sub[b] - 1) - self.slot_of)
        order = np.lexsort((walk, state["pos"][b], key))
        so = sub[order]
        ko = key[order]
        co = centres[b[order], self.slot_of[order]]
        d_max = float(self.param[5])
        tot = 0.0
        k = 1
        while k < self.S:
            x, y = so[:-k], so[k:]
            same = ko[:-k] == ko[k:]
            # coordinates are monotone along a contig: once no pair at offset k is in reach, none is further out
            near = same & (((co[k:] - co[:-k]) < d_max + 1.0) | (self.bin_of[x] == self.bin_of[y]))
            if not near.any():
                break
            x, y = x[near], y[near]
            ex, norm = self._ex(state, centres, x, y)
            samebin = self.bin_of[x] == self.bin_of[y]  # a bin's own pairs are never priced as trans
            ctr = (self.param[7] * norm).astype(np.float32)
            term = ex.astype(np.float64) - np.where(samebin, 0.0, ctr.astype(np.float64))
            if not same_bin:
                term = np.where(samebin, 0.0, term)
            tot += float(term.sum())
            k += 1
        return tot

    def mass_cis_windowed(self, state, centres=None, same_bin=True, in_set=None, block=1 << 22):
        """The sum of ``mass_cis`` enumerated differently -- every sub-fragment's partners inside its reach found by ONE binary
        search on a globally monotone key (contig rank x span + centre), the pairs priced in blocks of ``block`` -- so that long
        contigs (thousands of fragments inside the window: 1e8 pairs at C5's 7 contigs) cost large vector operations instead of
        thousands of short ones.  ``in_set`` (bool per bin) restricts the sum to pairs whose two bins are both in the set.
        tests/test_sparse_reformulation.py holds it to ``mass_cis``."""
        centres = self.centres(state) if centres is None else centres
        b = self.bin_of
        sub = np.arange(self.S) if in_set is None else np.flatnonzero(np.asarray(in_set, dtype=bool)[b])
        if len(sub) < 2:
            return 0.0
        bs = b[sub]
        key = state["id_c"][bs].astype(np.int64)
        walk = np.where(state["ori"][bs] == 1, self.slot_of[sub], (self.n_sub[bs] - 1) - self.slot_of[sub])
        order = np.lexsort((walk, state["pos"][bs], key))
        so, ko, wo = sub[order], key[order], walk[order]
        bo = b[so]
        co = centres[bo, self.slot_of[so]].astype(np.float64)
        d_max = float(self.param[5])
        span = float(co.max() - min(co.min(), 0.0)) + d_max + 4.0
        rank = np.cumsum(np.concatenate([[0], ko[1:] != ko[:-1]]))
        g = rank * span + co                      # monotone along a contig, and a contig's reach never enters the next one
        m = len(so)
        hi = np.searchsorted(g, g + (d_max + 1.0), side="left")
        if same_bin:                              # a bin's own pairs are priced whatever their distance (they follow each other)
            hi = np.maximum(hi, np.arange(m) + (self.n_sub[bo] - 1 - wo) + 1)
        cnt = np.maximum(hi - (np.arange(m) + 1), 0)
        cum = np.concatenate([[0], np.cumsum(cnt)])
        tot = 0.0
        i0 = 0
        while i0 < m:
            i1 = int(np.searchsorted(cum, cum[i0] + block, side="right")) - 1
            i1 = min(max(i1, i0 + 1), m)
            c = cnt[i0:i1]
            n_pairs = int(cum[i1] - cum[i0])
            if n_pairs:
                xi = np.repeat(np.arange(i0, i1), c)
                yi = np.arange(n_pairs) - np.repeat(cum[i0:i1] - cum[i0], c) + xi + 1
                x, y = so[xi], so[yi]
                ex, norm = self._ex(state, centres, x, y)
                samebin = b[x] == b[y]
                ctr = (self.param[7] * norm).astype(np.float32)
                term = ex.astype(np.float64) - np.where(samebin, 0.0, ctr.astype(np.float64))
                if not same_bin:
                    term = np.where(samebin, 0.0, term)
                tot += float(term.sum())
            i0 = i1
        return tot

    def set_contacts(self, in_set):
        """Indices of the contacts that join two DIFFERENT bins of the set (bool per bin)."""
        in_set = np.asarray(in_set, dtype=bool)
        br, bc = self.bin_of[self.row], self.bin_of[self.col]
        return np.flatnonzero(in_set[br] & in_set[bc] & (br != bc))

    def restricted(self, state, in_set, contacts=None):
        """The part of ``full(state, same_bin=False)`` that a move inside the set can change: the contacts between two different
        bins of the set, minus the windowed cis correction of the set's sub-fragment pairs.  What is left out -- pixels with a bin
        outside the set, the all-trans mass, the log-factorial constant -- is the same for every layout that differs from another
        only inside the set (bins of different contigs are priced as trans whatever their labels), so

            restricted(candidate) - restricted(current) == full(candidate, same_bin=False) - full(current, same_bin=False)

        with the set = contig(fA) u contig(fB) of the current layout: the pixel set sub_compute_likelihood revisits
        (kernels3.cu:3356-3380).  The large-size checker of the GPU suite (tests/test_independent_checker_gpu.py)."""
        centres = self.centres(state)
        idx = self.set_contacts(in_set) if contacts is None else contacts
        ex, _ = self._ex(state, centres, self.row[idx], self.col[idx])
        nnz = float((self.count[idx] * np.log(ex.astype(np.float64))).sum())
        return nnz - self.mass_cis_windowed(state, centres, same_bin=False, in_set=in_set)

    def full(self, state, same_bin=True, windowed=False):
        """same_bin=False leaves out every bin's own (diagonal) pixel: the pixel set sub_compute_likelihood
        revisits for a candidate never contains them (kernels3.cu:3356-3380, repeats aside), so candidate deltas
        are differences of full(..., same_bin=False)."""
        centres = self.centres(state)
        mass = self.mass_cis_windowed if windowed else self.mass_cis
        return (self.nnz_part(state, centres, same_bin=same_bin) - self.c_lf
                - (self.t_all() + mass(state, centres, same_bin=same_bin)))
