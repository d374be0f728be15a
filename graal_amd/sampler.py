"""``sampler`` -- drop-in for ``cuda_lib_gl.sampler`` on the ``start_EM`` path, over the MI355X engine.

Same constructor (the 30 positional arguments of ``cuda_lib_gl.py:33-42``, as ``simulation_loader.py:92-107``
passes them), same method names, argument orders and return tuples for everything ``simulation_loader.py`` and
``main_gl.py`` use.  What changed underneath:

* no PyCUDA / OpenGL: a ctypes handle on ``libgraal_hip.so`` (``graal_amd/lib.py``); ``gl_window`` may be ``None``
  and the display-only methods are no-ops;
* the two contact matrices may be scipy sparse matrices or COO triples; nothing is densified
  (``simulation_loader.py:81-82`` is what is NOT reproduced);
* one MCMC step = relabel + ONE fused scan of the contact list for all 13*K candidates + one commit, instead of
  ~42 launches and ~58 blocking syncs per neighbour;
* the step's full likelihood is carried over from the accepted candidate instead of being recomputed by a full
  dense pass every step (``cuda_lib_gl.py:1828-1848``); ``init_likelihood()`` and circular-contig events resync it;
* the numpy RNG is an explicit ``RandomState`` (``rng=``); ``None`` keeps the reference's global ``np.random``;
  every ``argsort`` is stable.

Blacklisted bins (``cuda_lib_gl.py:161-172``) are supported: their rows of the observation matrix become explicit
contacts with the float32 fill value.  Repeated fragments (``allow_repeats``): the observations of the repeated bins leave
the contact list and go to the engine as dense rows (``split_repeat_observations``); every pixel of a repeated bin is
priced over the active copies like ``kernels3.cu:2915-2930``.
"""
import bisect

import numpy as np

from . import dist as gdist
from .gpustruct import GPUStruct
from .lib import FIELDS, MAX_NEIGHBOURS, N_OPS, Q_FULL_BAD, Q_SCALE, STEP_DONE, STEP_FALLBACK, STEP_PAUSED, STEP_SELECT, Engine, q_to_float

N_TMP_STRUCT = N_OPS  # cuda_lib_gl.py:112
MODIFICATION_STR = ['eject frag', 'flip frag',
                    'pop out split insert @ left or 1', 'pop out split insert @ left or -1',
                    'pop out split insert @ right or 1', 'pop out split insert @ right or -1',
                    'pop out insert @ right or 1', 'pop out insert @ right or -1', 'swap activity',
                    'transloc_1', 'transloc_2', 'transloc_3', 'transloc_4']  # cuda_lib_gl.py:403-411


# --------------------------------------------------------------------------------- host-side pure functions
def as_coo_upper(m):
    """(row, col, val) with row < col, sorted by (row, col), from a dense array, a scipy sparse matrix (either the
    upper triangle or the symmetrised ``csr + csr.T`` of ``simulation_loader.py:81-82``) or a COO triple."""
    if isinstance(m, (tuple, list)) and len(m) == 3:
        r, c, v = (np.asarray(x) for x in m)
    elif hasattr(m, "tocoo"):
        coo = m.tocoo()
        r, c, v = coo.row, coo.col, coo.data
    else:
        a = np.asarray(m)
        r, c = np.nonzero(np.triu(a, k=1))
        v = a[r, c]
    r = np.asarray(r, dtype=np.int64)
    c = np.asarray(c, dtype=np.int64)
    keep = (r < c) & (np.asarray(v) != 0)  # the diagonal is zeroed by the reference (cuda_lib_gl.py:157-160)
    r, c, v = r[keep], c[keep], np.asarray(v)[keep]
    order = np.lexsort((c, r))
    return r[order].astype(np.int32), c[order].astype(np.int32), v[order]


def own_pair_counts(sub_coo, np_sub_frags_id, n_sub_total):
    """float32 [n_bins, 3]: the observed contacts between the sub-fragments of ONE bin, by data-slot pair (0,1), (0,2), (1,2) -- the
    diagonal pixels of the bin-level matrix the reference's evaluate_likelihood sums (kernels3.cu:3213), which no candidate delta contains
    (include/graal_hip.h: graal_upload_own_obs)."""
    ids = np.asarray(np_sub_frags_id)
    n_bins = ids.shape[0]
    bin_of = np.full(n_sub_total, -1, dtype=np.int64)
    slot_of = np.zeros(n_sub_total, dtype=np.int64)
    for k in range(3):
        has = ids[:, 3] > k
        bin_of[ids[has, k]] = np.nonzero(has)[0]
        slot_of[ids[has, k]] = k
    row, col, val = (np.asarray(a) for a in sub_coo[:3])
    same = (bin_of[row] == bin_of[col]) & (row != col) & (bin_of[row] >= 0)
    a, b = np.minimum(slot_of[row[same]], slot_of[col[same]]), np.maximum(slot_of[row[same]], slot_of[col[same]])
    own = np.zeros((n_bins, 3), dtype=np.float32)
    own[bin_of[row[same]], np.where(a == 0, b - 1, 2)] = val[same].astype(np.float32)
    return own


def split_repeat_observations(sub_coo, sub_ids_of_bins, dup_bins, n_sub_total):
    """Contacts without the repeated bins' sub-fragments + the rows of the (symmetric, zero-diagonal) observation matrix
    of those sub-fragments, [n_dup, 3, n_sub_total] float32 -- what ``graal_upload_repeats`` takes."""
    r, c, v = (np.asarray(x) for x in sub_coo)
    dup_bins = [int(b) for b in dup_bins]
    S = int(n_sub_total)
    slot_of = {}
    for i, b in enumerate(dup_bins):
        ids = sub_ids_of_bins[b]
        for a in range(int(ids[3])):
            slot_of[int(ids[a])] = (i, a)
    dup_sub = np.zeros(S, dtype=bool)
    dup_sub[list(slot_of)] = True
    obs = np.zeros((len(dup_bins), 3, S), dtype=np.float32)
    touch = np.nonzero(dup_sub[r] | dup_sub[c])[0]
    for k in touch:
        rk, ck, vk = int(r[k]), int(c[k]), np.float32(v[k])
        if rk in slot_of:
            i, a = slot_of[rk]; obs[i, a, ck] = vk
        if ck in slot_of:
            i, a = slot_of[ck]; obs[i, a, rk] = vk
    keep = ~(dup_sub[r] | dup_sub[c])
    return (r[keep], c[keep], np.asarray(v)[keep]), obs


def blacklist_fill(sub_coo, bin_coo, sub_ids_of_bins, black_bins, fill_value, n_sub_total):
    """The blacklist fill of ``cuda_lib_gl.py:161-172`` on COO lists instead of dense matrices.

    Bin level (neighbour proposal): rows and columns of blacklisted bins are zeroed -> their entries are dropped.
    Sub level (observations): every row and column of every sub-fragment of a blacklisted bin is overwritten with
    ``fill_value`` (= mean_value_trans, float32) -- a dense, non-integer observation for all pairs with that
    sub-fragment.  They become explicit contacts: (#blacklisted sub-fragments) x n_sub_total entries.
    Returns (sub_coo, bin_coo) with float32 counts at the sub level, both sorted by (row, col)."""
    black_bins = np.unique(np.asarray(black_bins, dtype=np.int64))
    r, c, v = (np.asarray(x) for x in sub_coo)
    br, bc, bv = (np.asarray(x) for x in bin_coo)
    if len(black_bins) == 0:
        return (r, c, v), (br, bc, bv)
    is_black_bin = np.zeros(int(max(br.max(initial=0), bc.max(initial=0), black_bins.max()) + 1), dtype=bool)
    is_black_bin[black_bins] = True
    keep = ~(is_black_bin[br] | is_black_bin[bc])
    br, bc, bv = br[keep], bc[keep], bv[keep]
    black_sub = np.zeros(int(n_sub_total), dtype=bool)
    for b in black_bins:
        ids = sub_ids_of_bins[b]
        black_sub[ids[:ids[3]]] = True
    keep = ~(black_sub[r] | black_sub[c])
    r, c, v = r[keep].astype(np.int64), c[keep].astype(np.int64), np.asarray(v)[keep].astype(np.float32)
    subs = np.nonzero(black_sub)[0]
    all_ids = np.arange(int(n_sub_total), dtype=np.int64)
    rr, cc = [], []
    for s_ in subs:   # pairs (s_, y): every y != s_; a pair of two blacklisted sub-fragments is listed once
        y = all_ids[(all_ids != s_) & ~(black_sub & (all_ids < s_))]
        rr.append(np.minimum(s_, y)); cc.append(np.maximum(s_, y))
    rr, cc = np.concatenate(rr), np.concatenate(cc)
    r = np.concatenate([r, rr]); c = np.concatenate([c, cc])
    v = np.concatenate([v, np.full(len(rr), np.float32(fill_value), dtype=np.float32)])
    order = np.lexsort((c, r))
    return (r[order].astype(np.int32), c[order].astype(np.int32), v[order]), (br, bc, bv)


def neighbour_distributions(bin_row, bin_col, bin_val, n_frags, n_neighbors=10, fact=3):
    """``setup_distri_frags`` (``cuda_lib_gl.py:2363-2390``) from the bin-level COO list instead of dense rows:
    for every bin the ``n_neighbors`` columns that come last in a stable ascending sort of its row (i.e. value
    descending, ties by index descending; zero columns, the bin itself included, fill up short rows) and
    ``pk`` proportional to count**3 in float32 (uniform if the row is empty)."""
    n = int(n_frags)
    r = np.concatenate([bin_row, bin_col]).astype(np.int64)
    c = np.concatenate([bin_col, bin_row]).astype(np.int64)
    v = np.concatenate([bin_val, bin_val]).astype(np.float32)
    order = np.lexsort((-c, -v.astype(np.float64), r))
    r, c, v = r[order], c[order], v[order]
    start = np.searchsorted(r, np.arange(n), side="left")
    end = np.searchsorted(r, np.arange(n), side="right")
    k = min(n_neighbors, n)
    xk = np.zeros((n, k), dtype=np.int32)
    pk = np.zeros((n, k), dtype=np.float32)
    for i in range(n):
        cols = c[start[i]:min(end[i], start[i] + k)]
        vals = v[start[i]:min(end[i], start[i] + k)]
        if len(cols) < k:  # pad with zero-valued columns, largest index first
            have = set(int(x) for x in c[start[i]:end[i]])
            pad = []
            j = n - 1
            while len(cols) + len(pad) < k and j >= 0:
                if j not in have:
                    pad.append(j)
                j -= 1
            cols = np.concatenate([cols, np.asarray(pad, dtype=np.int64)])
            vals = np.concatenate([vals, np.zeros(len(pad), dtype=np.float32)])
        dat = vals.astype(np.float32) ** fact
        if dat.sum() > 0:
            p = dat / dat.sum()
        else:
            tmp = np.ones_like(dat, dtype=np.float32)
            p = tmp / tmp.sum()
        xk[i] = cols
        pk[i] = p
    return xk, pk


def select_move(score, n_tmp_struct, rng, F_t=1.0):
    """Score post-processing and sampling of ``step_max_likelihood`` (``cuda_lib_gl.py:1898-1947``): duplicate
    eject / flip entries of neighbours >= 1 are zeroed, scores are shifted to ``max - 30`` and the move is drawn
    with probability LINEAR in that shifted score.  Returns (sample_out, or_score[sample_out]).

    Same float64 operations in the same order as the reference's lines (tests/test_host_logic.py holds the literal
    restatement, results AND the generator's state afterwards are compared), with fewer temporaries; the final
    ``choice(ids, 1, p=sub_score)`` is spelled out (``legacy_choice_one``)."""
    score = np.asarray(score, dtype=np.float64)
    id_max = int(score.argmax())
    filtered_score = score - score.min()
    filtered_score[n_tmp_struct::n_tmp_struct] = 0          # remove extra pop
    filtered_score[n_tmp_struct + 1::n_tmp_struct] = 0      # remove extra flip
    thresh_overflow = 30
    filtered_score -= filtered_score.max() - thresh_overflow
    filtered_score[filtered_score < 0] = 0
    ids = np.flatnonzero(filtered_score > 0)
    if len(ids) <= 1:                                       # (also when a NaN score leaves nothing to sample from)
        sample_out = id_max
    else:
        sub_score = filtered_score[ids]
        sub_score /= sub_score.sum()
        pos = sub_score > 0
        if pos.all():
            sub_score = np.power(sub_score, 1. / F_t)
        else:
            sub_score[pos] = np.power(sub_score[pos], 1. / F_t)
        sub_score /= sub_score.sum()
        sample_out = int(ids[legacy_choice_one(rng, sub_score)])
    return int(sample_out), float(score[sample_out])


def legacy_choice_one(rng, p):
    """Index drawn by ``RandomState.choice(len(p), 1, p=p)`` (replace=True): one uniform, inverse CDF
    (``cdf = p.cumsum(); cdf /= cdf[-1]; cdf.searchsorted(u, side='right')``) -- numpy's legacy algorithm, whose
    stream numpy guarantees stable; without its argument validation (p here is a freshly normalised float64 vector;
    anything else goes to numpy itself, so errors stay numpy's)."""
    cdf = p.cumsum()
    tot = cdf[-1]
    if not (abs(tot - 1.0) < 1e-9):          # not normalised / NaN: let numpy judge (and raise)
        return int(rng.choice(len(p), 1, p=p)[0])
    cdf /= tot
    return int(cdf.searchsorted(rng.random_sample(1), side='right')[0])


def legacy_choice_without_replacement(rng, a, size, p, p_list=None):
    """``RandomState.choice(a, size, replace=False, p=p)`` for a handful of entries (the neighbour proposal draws <= 10 of
    10): numpy's legacy algorithm -- draw ``size - found`` uniforms, inverse CDF of p with the found entries zeroed, keep
    the first occurrences, repeat -- on Python floats (the float32 -> float64 conversion, the running sum and the division
    are the same IEEE operations as numpy's ``cumsum`` / ``/=``).  Anything unusual (size 0, p not summing to 1) is left
    to numpy, so its errors and corner cases stay its own."""
    pl = [float(v) for v in p] if p_list is None else list(p_list)
    if size < 1 or size > len(pl) or not (abs(sum(pl) - 1.0) < 1e-4) or min(pl) < 0.0 or sum(1 for v in pl if v > 0) < size:
        return rng.choice(a, size, p=p, replace=False)
    found = []
    while len(found) < size:
        x = rng.random_sample(size - len(found)).tolist()
        for f in found:
            pl[f] = 0.0
        acc, cdf = 0.0, []
        for v in pl:
            acc += v
            cdf.append(acc)
        tot = cdf[-1]
        cdf = [c / tot for c in cdf]
        for u in x:
            i = bisect.bisect_right(cdf, u)
            if i not in found:
                found.append(i)
    return a[found]


def dist_inter_genome(prev, nxt, ori, id_d, init_prev, init_next, init_ori, orientable, counted, n_frags_4_dist):
    """``dist_inter_genome`` (``cuda_lib_gl.py:475-541``) vectorised; every term is a multiple of 0.5, so the
    float64 result is identical to the reference's sequential loop."""
    n = len(prev)
    d = 3.0 * (n - n_frags_4_dist)
    norm_distance = 3.0 * (n - n_frags_4_dist)
    prev_t1 = np.where(prev != -1, id_d[np.maximum(prev, 0)], -1)
    next_t1 = np.where(nxt != -1, id_d[np.maximum(nxt, 0)], -1)
    p0, n0 = init_prev, init_next
    both = ((prev_t1 == p0) & (next_t1 == n0)) | ((prev_t1 == n0) & (next_t1 == p0))
    d -= float(np.count_nonzero(both & counted))
    ob = counted & (orientable != 0)
    flipped = ob & (init_ori != ori)
    p1 = np.where(flipped, next_t1, prev_t1)
    n1 = np.where(flipped, prev_t1, next_t1)
    swap = np.where(flipped, -1, 1)
    for t0, t1 in ((p0, p1), (n0, n1)):
        m = ob & (t0 == t1)
        is_end = m & (t0 == -1)
        idx = np.maximum(t1, 0)
        not_or = m & ~is_end & (orientable[idx] == 0)
        half = m & ~is_end & ~not_or
        same_ori = half & (init_ori[np.maximum(t0, 0)] == swap * ori[idx])
        d -= float(np.count_nonzero(is_end)) + float(np.count_nonzero(not_or))
        d -= 0.5 * float(np.count_nonzero(half)) + 0.5 * float(np.count_nonzero(same_ori))
    nb = counted & (orientable == 0)
    d -= float(np.count_nonzero(nb & ((prev_t1 == p0) | (prev_t1 == n0))))
    d -= float(np.count_nonzero(nb & ((next_t1 == n0) | (next_t1 == p0))))
    return d / norm_distance


def _as_table(a, width, dtype):
    """int4 / float3 / int3 structured arrays of simulation_loader.py:60-63 -> plain [n, width] arrays."""
    a = np.asarray(a)
    if a.dtype.fields is not None:
        names = list(a.dtype.names)[:width]
        a = np.stack([a[k] for k in names], axis=1)
    return np.ascontiguousarray(a.reshape(-1, width), dtype=dtype)


# --------------------------------------------------------------------------------- the sampler
class sampler(object):
    def __init__(self, use_rippe, S_o_A_frags, collector_id_repeats, frag_dispatcher,
                 id_frag_duplicated, id_frags_blacklisted,
                 n_frags, n_new_frags, init_n_sub_frags, n_new_sub_frags, np_rep_sub_frags_id,
                 hic_matrix_sub_sampled,
                 np_sub_frags_len_bp, np_sub_frags_id, np_sub_frags_accu,
                 mean_squared_frags_per_bin, norm_vect_accu,
                 S_o_A_sub_frags,
                 hic_matrix, mean_value_trans, n_iterations, is_simu, gl_window=None, pos_vbo=None, col_vbo=None,
                 vel=None, pos=None, raw_im_init=None, pbo_im_buffer=None, sub_sample_factor=0,
                 device=None, rng=None, group=None, param_simu=None, compute_dist=True, exchange=None,
                 reference_arithmetic="strict"):
        self.o = 0
        self.use_rippe = use_rippe
        self.gl_window = gl_window
        self.sub_sample_factor = sub_sample_factor
        self.n_iterations = n_iterations
        self.is_simu = is_simu
        self.rng = np.random if rng is None else rng
        self._rng_given = rng is not None
        self.compute_dist = compute_dist
        self.id_frags_blacklisted = list(id_frags_blacklisted) if id_frags_blacklisted is not None else []
        self.id_frag_duplicated = list(id_frag_duplicated) if id_frag_duplicated is not None else []
        self.id_frag_duplicated = [int(x) for x in self.id_frag_duplicated]
        if (int(n_new_frags) != int(n_frags)) != bool(len(self.id_frag_duplicated)):
            raise ValueError("n_new_frags != n_frags needs the list of duplicated bins (and vice versa)")
        self.np_id_frag_duplicated = np.int32(self.id_frag_duplicated)
        self.n_frags = np.int32(n_frags)
        self.n_new_frags = np.int32(n_new_frags)
        self.init_n_sub_frags = np.int32(init_n_sub_frags)
        self.n_new_sub_frags = np.int32(n_new_sub_frags)
        self.uniq_frags = np.int32(np.setdiff1d(np.arange(0, self.n_frags, dtype=np.int32), self.np_id_frag_duplicated))
        self.n_frags_uniq = np.int32(len(self.uniq_frags))
        self.n_tmp_struct = N_TMP_STRUCT
        self.n_modif_metropolis = N_TMP_STRUCT
        self.modification_str = list(MODIFICATION_STR)
        self.norm_vect_accu = norm_vect_accu
        self.mean_squared_frags_per_bin = np.float32(mean_squared_frags_per_bin)
        self.mean_value_trans = mean_value_trans
        self.S_o_A_frags = S_o_A_frags
        self.S_o_A_sub_frags = S_o_A_sub_frags
        self.collector_id_repeats = np.ascontiguousarray(collector_id_repeats, dtype=np.int32)
        self.frag_dispatcher = _as_table(frag_dispatcher, 2, np.int32)
        self.np_sub_frags_id = _as_table(np_sub_frags_id, 4, np.int32)
        self.np_sub_frags_len_bp = _as_table(np_sub_frags_len_bp, 3, np.float32)
        self.np_sub_frags_accu = _as_table(np_sub_frags_accu, 3, np.int32)
        self.param_simu_rippe = np.dtype([('kuhn', np.float32), ('lm', np.float32), ('c1', np.float32),
                                          ('slope', np.float32), ('d', np.float32), ('l_max', np.float32),
                                          ('fact', np.float32), ('v_inter', np.float32)], align=True)
        self.param_simu_T = self.param_simu_rippe
        # ---- contacts: COO, never dense ------------------------------------------------------------------
        self.sub_coo = as_coo_upper(hic_matrix)                 # sub-level (observed data of the likelihood)
        self.bin_coo = as_coo_upper(hic_matrix_sub_sampled)     # bin level (neighbour proposal only)
        if len(self.id_frags_blacklisted):                      # cuda_lib_gl.py:161-172
            if mean_value_trans is None or not float(mean_value_trans) > 0:
                raise ValueError("blacklisted fragments need mean_value_trans > 0 (the fill value of their observations)")
            black_bins = np.asarray(S_o_A_frags["id_d"])[np.asarray(self.id_frags_blacklisted, dtype=np.int64)]
            self.sub_coo, self.bin_coo = blacklist_fill(self.sub_coo, self.bin_coo, self.np_sub_frags_id, black_bins,
                                                        np.float32(mean_value_trans), int(self.init_n_sub_frags))
        self.sub_n_frags = self.init_n_sub_frags
        # ---- device ------------------------------------------------------------------------------------------
        if group is None:
            rank, world, local = gdist.env_world()
            group = gdist.Group(rank, world) if world > 1 else gdist.Group(0, 1)
            if device is None:
                device = local if world > 1 else 0
        self.group = group
        if group.world > 1 and not self._rng_given:
            # every rank draws the proposal and the move itself: without identically seeded generators the ranks would
            # sum Q vectors of different proposals and their layouts would diverge silently
            raise ValueError("a sharded sampler (world > 1) needs an explicit, identically seeded rng= on every rank")
        self.engine = Engine(0 if device is None else int(device))  # raises if the HIP library / GPU is missing
        # Which arithmetic the candidate scores are computed in (include/graal_hip.h):
        #   "strict" (default) = the reference's: every pixel of contig(A) u contig(B) re-priced from the float32 kb coordinates of
        #       the candidate layout like sub_compute_likelihood (kernels3.cu:3259-3718), the reference's RF-count indexing in the
        #       trans branch (kernels3.cu:3155) -- accepted-move traces are the reference's on any coordinates;
        #   "trans_accu" = that indexing in the full evaluation only, exact deltas;
        #   "exact" (or None) = mathematically exact deltas: pairs whose geometry a move leaves unchanged contribute exactly
        #       nothing (the reference re-rounds their float32 coordinates and lets the noise into its scores).  Equal to the
        #       reference when every kb coordinate is exact in float32, 2-4x faster once contigs hold thousands of fragments.
        if reference_arithmetic == "exact":
            reference_arithmetic = None
        if reference_arithmetic not in (None, "trans_accu", "strict"):
            raise ValueError("reference_arithmetic must be 'strict', 'trans_accu', 'exact' or None")
        self.reference_arithmetic = reference_arithmetic
        if reference_arithmetic:
            self.engine.set_mode(ref_trans_accu=True, strict=reference_arithmetic == "strict")
        self.engine.upload_subfrags(self.np_sub_frags_id, self.np_sub_frags_len_bp, self.np_sub_frags_accu,
                                    int(self.init_n_sub_frags), float(self.mean_squared_frags_per_bin))
        self.sub_coo_full = self.sub_coo   # (what estimate_parameters fits: the whole observation matrix, cuda_lib_gl.py:1244)
        if len(self.id_frag_duplicated):   # repeated bins: observation rows to the engine, their contacts out of the list
            self.sub_coo, obs_rows = split_repeat_observations(self.sub_coo, self.np_sub_frags_id, self.id_frag_duplicated,
                                                               int(self.init_n_sub_frags))
            self.engine.upload_repeats(self.id_frag_duplicated, self.frag_dispatcher, self.collector_id_repeats, obs_rows)
        take = gdist.shard_take(len(self.sub_coo[0]), group.rank, group.world)
        counts = np.asarray(self.sub_coo[2])
        if np.all(counts == np.round(counts)) and (len(counts) == 0 or counts.max() < 2 ** 24):
            counts = counts.astype(np.int32)
        else:
            counts = counts.astype(np.float32)   # the reference's observation type (blacklist fill: non-integer)
        self.engine.upload_contacts(self.sub_coo[0][take], self.sub_coo[1][take], counts[take])
        n = int(self.n_new_frags)
        soa = {k: np.array(S_o_A_frags[k], dtype=np.int32, copy=True) for k in FIELDS if k != "ori"}
        soa["ori"] = np.ones((n,), dtype=np.int32)  # cuda_lib_gl.py:244,259
        self.gpu_vect_frags = GPUStruct(self.engine, soa)
        self.gpu_vect_frags.copy_to_gpu()
        self.np_init_prev = np.copy(soa["prev"])
        self.np_init_next = np.copy(soa["next"])
        self.np_init_ori = np.ones((n,), dtype=np.int32)
        self.np_init_orientable = (self.np_sub_frags_id[soa["id_d"], 3] > 1).astype(np.int32)
        self.id_d = np.copy(soa["id_d"])
        self._single_sub = bool(np.all(self.np_sub_frags_id[:, 3] == 1))
        self._n_circ_prev = int((soa["circ"] == 1).sum())
        self._d_q = None
        self._dist_ref_uploaded = False
        self._dist_counted_mask = None
        self.exchange = self._setup_exchange(exchange)
        if group.world > 1 and self.exchange == "host" and not len(self.id_frag_duplicated) and not self._single_sub:
            # every rank holds a shard of the list but prices the commits' own-pixel corrections in full (graal_upload_own_obs): the observed
            # counts of each bin's own sub-fragment pairs, from the WHOLE list
            self.engine.upload_own_obs(own_pair_counts(self.sub_coo, self.np_sub_frags_id, int(self.init_n_sub_frags)))
        import os as _os
        # MCMC steps between full re-evaluations of the carried-over likelihood.  With repeats every step: the reference's
        # candidate pixel ranges miss some pixels an activity swap changes (kernels3.cu:3368-3373), so its per-step total
        # (always a full evaluation, cuda_lib_gl.py:1828-1848) is not the previous score
        # Reference arithmetic: a strict delta IS full(after) - full(before) pixel by pixel, so that total is carried like the
        # default one (tests/test_strict_windowed_gpu.py::test_strict_total_carried_over_equals_a_full_evaluation) -- except with
        # the trans-branch RF-count indexing (kernels3.cu:3155) when a bin's sub-fragments have different counts: mirroring a bin
        # then changes its trans pixels with bins OUTSIDE contig(A) u contig(B) too, which no candidate delta of the reference
        # contains; its per-step full evaluation does.
        # And with several sub-fragments per bin, a bin's OWN pixel (its sub-fragment pairs) is in no candidate delta either
        # (kernels3.cu:3356-3380) while its float32 value moves with the bin's coordinates: 1e-7 relative per 100 steps, measured.
        # So the strict total is carried as it is only at one sub-fragment per bin (level 0: the C4 / C5 shapes).  With sub-fragments
        # the commit kernel computes what it does to the own pixels of the bins it moves and the next step adds that to the carried
        # total (include/graal_hip.h: graal_take_carry_correction, graal_step flag 16) -- which makes it the full likelihood of the
        # committed layout again, without the 35-50 us evaluation per step; one rank, no repeats.  A commit that mirrors a bin whose
        # sub-fragments carry DIFFERENT RF counts reports its correction as unknown (the trans-branch indexing above moves that bin's
        # trans pixels too) and the next step evaluates in full: in a pyramid those are the ragged last bins of the initial contigs;
        # where they are more than a tenth of the bins the per-step evaluation stays (it overlaps the scoring kernels, a repair does not).
        accu = self.np_sub_frags_accu
        nsub = self.np_sub_frags_id[:, 3]
        mixed = ((nsub >= 2) & (accu[:, 1] != accu[:, 0])) | ((nsub >= 3) & (accu[:, 2] != accu[:, 0]))
        # (several ranks: over the host exchange every rank computes the same correction from the whole list's own-pair table; behind an RCCL
        # all-reduce a repair could not be summed inside the step: per-step evaluation there)
        self._own_corr = bool(self.reference_arithmetic == "strict" and not self._single_sub and not len(self.id_frag_duplicated)
                              and (self.group.world == 1 or self.exchange == "host") and mixed.mean() <= 0.1
                              and not _os.environ.get("GRAAL_NO_OWN_PIXEL_CARRY"))
        self.resync_every = 512
        if len(self.id_frag_duplicated) or (self.reference_arithmetic and not self._single_sub and not self._own_corr):
            self.resync_every = 1
        if _os.environ.get("GRAAL_RESYNC_EVERY"):     # (measurement switch: what the reference's per-step full evaluation costs a run)
            self.resync_every = max(1, int(_os.environ["GRAAL_RESYNC_EVERY"]))
        self._steps_since_full = 0
        self._force_full = False   # the carried-over total is not a likelihood of the current layout / parameters
        # ---- proposal ----------------------------------------------------------------------------------------
        self.n_neighbors = 10  # cuda_lib_gl.py:444
        self.setup_distri_frags()
        self.define_repeats()
        self._setup_c_step()
        self.param_simu = None
        self.bins = np.zeros(0)
        self.likelihood_t = None
        self.n_stale_paste = 0
        if param_simu is not None:
            self.set_param_simu(param_simu)

    # ------------------------------------------------------------------ parameters
    def set_param_simu(self, p):
        """param_simu as the structured array of ``cuda_lib_gl.py:1213`` (or 8 plain floats)."""
        a = np.asarray(p)
        if a.dtype.fields is not None:
            flat = np.array([a[k][0] if a.shape else a[k] for k in a.dtype.names], dtype=np.float32)
        else:
            flat = a.astype(np.float32).reshape(8)
        self.param_simu = np.array([tuple(flat)], dtype=self.param_simu_rippe)
        self._param_flat = flat
        self.engine.set_params(flat)
        self._force_full = True    # (the carried total was a likelihood under the old parameters)

    def setup_rippe_parameters(self, param, d_max):
        kuhn, lm, slope, d, fact = param
        kuhn = np.float32(kuhn)
        lm = np.float32(lm)
        c1 = np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))
        return np.array([(kuhn, lm, c1, np.float32(slope), np.float32(d), np.float32(d_max), np.float32(fact),
                          self.mean_value_trans)], dtype=self.param_simu_rippe)

    def estimate_parameters(self, max_dist_kb, size_bin_kb):
        """``cuda_lib_gl.py:1229-1294``: histogram of sub-level cis contacts vs genomic distance (a pass over the COO
        list plus the count of zero pairs per distance bin, instead of the reference's O(S^2) Python double loop),
        log-space least squares, fsolve for d_max."""
        from . import rippe_fit
        self.bins = np.arange(size_bin_kb, max_dist_kb + size_bin_kb, size_bin_kb)
        self.mean_contacts = rippe_fit.mean_contacts_per_bin(self.S_o_A_sub_frags, self.sub_coo_full, self.bins, max_dist_kb,
                                                             size_bin_kb)
        p, self.y_estim = rippe_fit.estimate_param_rippe(self.mean_contacts, self.bins)
        estim_max_dist = rippe_fit.estimate_max_dist_intra(p, self.mean_value_trans)
        self.set_param_simu(self.setup_rippe_parameters(p, estim_max_dist))

    def return_rippe_vals(self, p0):
        from . import rippe_fit
        return rippe_fit.peval(self.bins, p0)  # cuda_lib_gl.py:1982-1984 (5-list: p0[3] = d is the amplitude, kept)

    def compute_likelihood_4_nuisance(self, test_param, restore=True):
        """Full likelihood of the current layout under TEST parameters (``cuda_lib_gl.py:1986-2017``).  ``restore=False`` leaves
        the test parameters in force on the device (the caller accepts them or puts ``_param_flat`` back itself)."""
        keep = np.copy(self._param_flat)
        try:
            if self.group.world == 1:
                # parameters + relabel + evaluation behind ONE wait (include/graal_hip.h: graal_eval_full_params)
                q, st, _ = self.engine.eval_full_params(test_param)
                self.n_stale_paste += int(st[7])
                return float("nan") if int(q[0]) == Q_FULL_BAD else float(int(q[0]) + int(q[1])) / Q_SCALE
            self.engine.set_params(test_param)
            return self.eval_likelihood()
        except BaseException:
            restore = True
            raise
        finally:
            if restore:
                self.engine.set_params(keep)

    def step_nuisance_parameters(self, dt=0, t=0, n_step=1):
        """Random-walk Metropolis step on (fact, slope, d_max, v_inter): ``cuda_lib_gl.py:2022-2107`` with its quirks kept
        (``np.random.choice(4)`` never picks the ``d`` branch; ``peval`` gets a 5-list so ``d`` acts as the amplitude)."""
        from . import rippe_fit as opti
        # the walk as a table: (perturbed field of param_simu, the attribute that holds its step width).  The reference sets the five widths
        # on every call and draws the branch with choice(4): the fifth row is never picked.  After the perturbation EITHER the trans level
        # follows the new cut-off (l_max perturbed: v_inter = the curve at l_max) OR the cut-off is re-solved for the trans level
        # (everything else: fsolve for the distance where the curve reaches v_inter); c1 is recomputed from the slope in force.
        # numpy's float32 scalars meet Python floats here (NEP 50: the sums stay float32) exactly as in the reference's expressions.
        walk = (("fact", "sigma_fact"), ("slope", "sigma_slope"), ("l_max", "sigma_d_max"), ("v_inter", "sigma_d_nuc"), ("d", "sigma_d"))
        names = self.param_simu.dtype.names
        p = dict(zip(names, np.copy(self.param_simu)[0]))
        self.sigma_fact = 10 ** (np.log10(p["fact"]) - 2)
        self.sigma_slope, self.sigma_d_max, self.sigma_d_nuc, self.sigma_d = 0.05, 100, 0.5, 10
        field, width = walk[self.rng.choice(4)]
        p[field] = p[field] + self.rng.normal(loc=0.0, scale=getattr(self, width))
        curve = [p["kuhn"], p["lm"], p["slope"], p["d"], p["fact"]]
        if field == "l_max":
            p["v_inter"] = opti.peval(p["l_max"], curve)
        else:
            p["l_max"] = opti.estimate_max_dist_intra_step(curve, p["v_inter"])
        p["c1"] = np.float32((0.53 * np.power(p["lm"] / p["kuhn"], p["slope"])) * np.power(p["kuhn"], -3))
        out_test_param = [tuple(p[k] for k in names)]
        out_test_param = np.array(out_test_param, dtype=self.param_simu_T)
        flat = np.array([out_test_param[0][k] for k in out_test_param.dtype.names], dtype=np.float32)
        if self.likelihood_t is None:
            self._set_total_from_full(self.eval_likelihood())
        valid = bool(np.isfinite(flat).all() and flat[7] > 0 and 0 < flat[5] < 2.0e6)
        test_likelihood = self.compute_likelihood_4_nuisance(flat, restore=False) if valid else -np.inf
        F_t = self.temperature(t, n_step)
        with np.errstate(over="ignore"):
            ratio = np.exp((test_likelihood - self.likelihood_t) / F_t)
        u = self.rng.rand()
        success = 0
        if ratio >= u:
            success = 1
            self.set_param_simu(out_test_param)
            self._set_total_from_full(test_likelihood)
            self._force_full = False   # a full evaluation of the current layout under the parameters now in force
        elif valid:
            self.engine.set_params(self._param_flat)   # rejected: back to the parameters in force
        kuhn, lm, c1, slope, d, d_max, fact, d_nuc = self.param_simu[0]
        y_rippe = self.return_rippe_vals([kuhn, lm, slope, d, fact])
        return fact, d, d_max, d_nuc, slope, self.likelihood_t, success, y_rippe

    # ------------------------------------------------------------------ display-only surface (no-ops)
    def setup_texture(self):
        self.data = None

    def load_gl_cuda_vbo(self):
        pass

    def load_gl_cuda_tex_buffer(self, im_init):
        pass

    def display_current_matrix(self, file=None, max_px=2048):
        """Fragment order by contig (``cuda_lib_gl.py:1581-1624``) and, when ``file`` is given, the sub-level contact matrix in that order
        as a 32-bit float TIFF like the reference's ``Image.fromarray(hic_matrix[np.ix_(full_order_high, full_order_high)]).save(file)`` --
        from the COO list; binned beyond ``max_px`` sub-fragments (graal_amd/image.py)."""
        self.gpu_vect_frags.copy_from_gpu()
        c = self.gpu_vect_frags
        dict_contig, full_order, full_order_high = dict(), [], []
        for k in np.unique(c.id_c):
            id_pos = np.nonzero(c.id_c == k)[0]
            dict_contig[k] = []
            if np.all(c.activ[id_pos] == 1):
                ordered_frag = c.id_d[id_pos[np.argsort(c.pos[id_pos], kind="stable")]]
                dict_contig[k].extend(ordered_frag)
                full_order.extend(ordered_frag)
                for i in ordered_frag:
                    v = list(self.np_sub_frags_id[i])
                    id_2_push = v[:v[3]]
                    if c.ori[i] == -1:
                        id_2_push.reverse()
                    full_order_high.extend(id_2_push)
        if file:
            from . import image
            image.write_tiff_f32(file, image.matrix_image(self.sub_coo_full, full_order_high, max_px))
        return full_order, dict_contig, full_order_high

    def free_gpu(self):
        self.engine.close()

    # ------------------------------------------------------------------ likelihood
    def _full_likelihood(self):
        q = self.engine.eval_full_q()
        bad = int(q[0]) == Q_FULL_BAD          # some term was not finite (the reference's double sum would be -inf / NaN)
        if self.group.world > 1:
            bad = self.group.all_reduce_max_int(1 if bad else 0) != 0
        q0 = self.group.all_reduce_sum_int(0 if bad else int(q[0]))
        if bad:
            return float("nan")
        return float(q0 + int(q[1])) / Q_SCALE

    def eval_likelihood(self):
        self.modify_gl_cuda_buffer(0)
        return self._full_likelihood()

    def init_likelihood(self):
        self._set_total_from_full(self.eval_likelihood())
        self._force_full = False

    def _setup_exchange(self, mode):
        """How the ranks' per-step Q vectors (13*K int64) are summed.  ``"host"``: through pinned host memory shared by the
        ranks of one node -- no collective launch, no device->host copy, the step costs what a single-rank step costs;
        ``"rccl"``: one RCCL all-reduce of a device buffer per step (the only choice across nodes).  Both are bit
        identical (int64 sums).  Default: ``GRAAL_EXCHANGE`` or "host" when every rank is on this node."""
        import os
        self._rccl_c = False
        if self.group.world == 1:
            if os.environ.get("GRAAL_RCCL_FORCE"):   # (test hook: a one-rank communicator, so that ONE GPU exercises the all-reduce flow)
                self.engine.attach_rccl(self.engine.rccl_unique_id(), 0, 1)
                self._rccl_c = True
            return "none"
        mode = mode or os.environ.get("GRAAL_EXCHANGE", "auto")
        if mode not in ("auto", "host", "rccl"):
            raise ValueError("exchange must be 'host', 'rccl' or 'auto'")
        same_node = self.group.single_node()          # (collective: every rank asks)
        if mode == "host" and not same_node:
            raise RuntimeError("exchange='host' needs every rank on one node")
        if mode == "rccl" or not same_node:
            self._attach_rccl_c()
            return "rccl"
        seg = self.group.shared_host_segment(self.engine.exchange_bytes(self.group.world))
        floor = self.group.all_reduce_max_int(self.engine.step_seq())
        self.engine.attach_exchange(seg, self.group.rank, self.group.world, floor)
        # collective self-test before relying on it: every rank's GPU tags its slots, every host must see every tag
        tag = 0x47524141 + (floor << 8)
        self.engine.exchange_selftest(tag, 0)
        self.group.barrier()
        ok = self.engine.exchange_selftest(tag, 1)
        if self.group.all_reduce_max_int(0 if ok else 1) != 0:
            self.engine.detach_exchange()
            if mode == "host":
                raise RuntimeError("exchange='host': the shared-segment self-test failed on some rank")
            import warnings
            warnings.warn("graal_amd: the shared-host-memory exchange failed its self-test; using the RCCL all-reduce")
            self.group.barrier()
            self._attach_rccl_c()
            return "rccl"
        self.group.barrier()
        return "host"

    def _attach_rccl_c(self, opt_in=None):
        """exchange="rccl" driven by the library (include/graal_hip.h: graal_attach_rccl): one ncclAllReduce per step on the engine's
        stream, the step's host logic in C.  Only when torch.distributed itself runs on RCCL (one process per GPU); with the gloo
        rehearsal -- several ranks on ONE GPU, which RCCL refuses -- the all-reduce stays torch's (the Python path below).

        OPT-IN (``GRAAL_RCCL_C=1``, or ``opt_in=True`` as bench.py's `exchange_alt` passes): no box this was built on holds two GPUs,
        so a communicator of several ranks has never run; until it has, exchange="rccl" means torch's all-reduce unless asked otherwise.

        Collective by construction: every rank reaches every collective below whatever happened to it locally -- failures travel as
        VALUES (an all-reduce before ncclCommInitRank, which is itself a collective: a rank that cannot load librccl must not leave
        its peers inside it; rank 0's id or None through the broadcast; an all-reduce behind the attach), never as a rank that
        skipped a collective; the collectives themselves are not wrapped -- if one of them raises, it raises on every rank."""
        import os
        self._rccl_c = False
        import torch.distributed as td
        if opt_in is None:
            opt_in = bool(os.environ.get("GRAAL_RCCL_C"))
        if not (td.is_initialized() and td.get_backend() == "nccl") or os.environ.get("GRAAL_RCCL_TORCH") or not opt_in:
            return          # (the same on every rank: the backend and the environment are the launcher's)
        try:
            can = Engine.rccl_available()
        except Exception:
            can = False
        if self.group.all_reduce_max_int(0 if can else 1) != 0:     # all or none, BEFORE anybody enters ncclCommInitRank
            return
        uid = None
        if self.group.rank == 0:
            try:
                uid = self.engine.rccl_unique_id()
            except Exception:
                uid = None                                          # (travels: every rank returns below)
        box = [uid]
        td.broadcast_object_list(box, src=0)
        if box[0] is None:
            return
        ok = 1
        try:
            self.engine.attach_rccl(box[0], self.group.rank, self.group.world)
        except Exception:
            ok = 0
        if self.group.all_reduce_max_int(1 - ok) != 0:              # all or none
            if ok:
                self.engine.detach_rccl()
            return
        self._rccl_c = True

    def _candidate_deltas(self, id_fA, id_neighbours, max_id):
        """float64 [K, 13]; one fused scan per group of <= 8 neighbours, one exchange per scan when sharded."""
        if self.group.world == 1:
            return self.engine.eval_candidates(id_fA, id_neighbours, max_id)
        if self.exchange == "host" or self._rccl_c:
            return self.engine.eval_candidates_x(id_fA, id_neighbours, max_id)
        import torch
        out = np.zeros((len(id_neighbours), N_OPS), dtype=np.float64)
        dev = torch.device("cuda", self.engine.device)
        if self._d_q is None:
            import torch.distributed as td
            self._torch_nccl = td.is_initialized() and td.get_backend() == "nccl"
            self._d_q = torch.zeros(3 * MAX_NEIGHBOURS * N_OPS, dtype=torch.int64, device=dev)   # (fine sums, coarse sums, not-finite flags: include/graal_hip.h)
            # a dedicated, non-null stream: the C ABI reads a null handle as "the engine's own stream", and the
            # collective must be ordered after the kernels that fill the buffer
            self._torch_stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(self._torch_stream):
            for k0 in range(0, len(id_neighbours), MAX_NEIGHBOURS):
                part = id_neighbours[k0:k0 + MAX_NEIGHBOURS]
                self.engine.eval_candidates_q_async(id_fA, part, max_id, self._d_q.data_ptr(),
                                                    self._torch_stream.cuda_stream, self.group.rank, self.group.world)
                if self._torch_nccl:
                    self.group.all_reduce_sum_(self._d_q)                    # one RCCL all-reduce of the device buffer over xGMI
                    qc = self._d_q.cpu().numpy()
                else:
                    # (the gloo rehearsal -- several ranks on ONE GPU: gloo is a CPU backend, the buffer goes through the host explicitly
                    # instead of through gloo's own handling of device tensors)
                    q_host = self._d_q.cpu()
                    self.group.all_reduce_sum_(q_host)
                    qc = q_host.numpy()
                nq = len(part) * N_OPS
                m = MAX_NEIGHBOURS * N_OPS
                out[k0:k0 + len(part)] = q_to_float(qc[:nq], qc[m:m + nq], qc[2 * m:2 * m + nq]).reshape(len(part), N_OPS)
        return out

    # ------------------------------------------------------------------ layout maintenance
    def modify_gl_cuda_buffer(self, id_fi, dt=0):
        """Contig relabel half of ``cuda_lib_gl.py:1695-1788`` (+ ``kernels3.cu:3848-3851``); returns max_id."""
        st, max_id = self.engine.begin_step()
        self.n_stale_paste += int(st[7])
        return np.int32(max_id)

    def test_copy_struct(self, id_fA, id_f_sampled, mode, max_id):
        self.engine.apply_move(id_fA, id_f_sampled, mode, max_id, wait=False)  # stale-paste count: next begin_step
        self.likelihood_t = None  # a layout change outside step_max_likelihood: re-evaluate before the next step

    def explode_genome(self, dt=0):
        """``cuda_lib_gl.py:1539-1556``: eject every fragment in index order (relabel before each)."""
        import os
        if os.environ.get("GRAAL_PY_STEP"):          # (the loop as written, for comparison: tests/test_sampler_gpu.py)
            for i in range(0, int(self.n_new_frags)):
                max_id = self.modify_gl_cuda_buffer(i, dt)
                self.test_copy_struct(i, 0, 0, max_id)
            return
        self.n_stale_paste += self.engine.explode()  # (the same loop behind the C ABI: include/graal_hip.h, graal_explode)
        self.likelihood_t = None                     # (a layout change outside step_max_likelihood, as test_copy_struct notes)

    def define_repeats(self):
        """``cuda_lib_gl.py:452-473``: every copy of a duplicated bin (the original included) is a "repeat"."""
        self._black_set = set(int(x) for x in self.id_frags_blacklisted)
        dup = set(self.id_frag_duplicated)
        self.is_repeat = [int(d) in dup for d in self.id_d]
        self.n_frags_duplicated = int(sum(self.is_repeat))
        self.n_frags_4_dist = len(set(self._black_set) | set(i for i, r in enumerate(self.is_repeat) if r))
        self._dup_set = dup

    def dist_inter_genome(self, tmp_gpu_vect_frags=None):
        """``cuda_lib_gl.py:475-541``.  The current layout is evaluated on the device (``k_dist``: the per-fragment terms
        are multiples of 0.5, summed there as integers, so the float64 value is the reference loop's exactly); a layout
        passed explicitly goes through the host function of the same name."""
        counted = self._dist_counted()
        if tmp_gpu_vect_frags is None:
            if not self._dist_ref_uploaded:
                self.engine.upload_distance_ref(self.np_init_prev, self.np_init_next, self.np_init_ori,
                                                self.np_init_orientable, counted)
                self._dist_ref_uploaded = True
            n = len(self.np_init_prev)
            norm_distance = 3.0 * (n - self.n_frags_4_dist)
            return (norm_distance - 0.5 * self.engine.genome_distance_half_units()) / norm_distance
        g = tmp_gpu_vect_frags
        g.copy_from_gpu()
        return dist_inter_genome(g.prev, g.next, g.ori, g.id_d, self.np_init_prev, self.np_init_next, self.np_init_ori,
                                 self.np_init_orientable, counted, self.n_frags_4_dist)

    def _dist_counted(self):
        if self._dist_counted_mask is None:
            counted = ~np.asarray(self.is_repeat, dtype=bool)       # cuda_lib_gl.py:485
            if len(self.id_frags_blacklisted):
                counted[np.asarray(self.id_frags_blacklisted, dtype=np.int64)] = False
            self._dist_counted_mask = counted
        return self._dist_counted_mask

    def temperature(self, t, n_step):
        return 1.0  # cuda_lib_gl.py:2602

    # ------------------------------------------------------------------ proposal
    def setup_distri_frags(self):
        xk, pk = neighbour_distributions(self.bin_coo[0], self.bin_coo[1], self.bin_coo[2], int(self.n_frags),
                                         self.n_neighbors)
        self.distri_frags = {"xk": xk, "pk": pk}
        self._row_cache, self._copies_cache = {}, {}

    def return_neighbours(self, id_fA, delta0):
        """``cuda_lib_gl.py:2295-2331`` (no repeats, no blacklist: the expansion loops are identities)."""
        ori_id = int(self.id_d[id_fA])
        delta = min(self.n_neighbors, delta0)
        distri = self.distri_frags["pk"][ori_id]
        row = self._row_cache.get(ori_id)     # per bin, built on first use: (#non-zero entries, the row as Python floats)
        if row is None:
            row = self._row_cache[ori_id] = (int(np.nonzero(distri != 0)[0].shape[0]), [float(v) for v in distri])
        n_max_candidates = min(delta, row[0])
        init_id = legacy_choice_without_replacement(self.rng, self.distri_frags["xk"][ori_id], n_max_candidates, distri, row[1])
        out = []
        if ori_id in self._dup_set:           # the other copies of a repeated fragment are candidates too (:2314-2318)
            d = self.frag_dispatcher[ori_id]
            out.extend(int(x) for x in np.setdiff1d(self.collector_id_repeats[d[0]:d[1]], id_fA))
        copies = self._copies_cache
        for id_fB in init_id.tolist():        # every copy of a proposed bin (:2320-2322)
            c = copies.get(id_fB)
            if c is None:
                d = self.frag_dispatcher[id_fB]
                c = copies[id_fB] = tuple(int(x) for x in self.collector_id_repeats[d[0]:d[1]])
            out.extend(c)
        black = self._black_set
        return [x for x in out if x not in black]   # cuda_lib_gl.py:2326-2329

    # ------------------------------------------------------------------ the per-step host logic in C
    def _mt_state_address(self):
        """Address of the generator's MT19937 state (uint32 key[624]; int pos) if `rng` is numpy's legacy RandomState (or the
        module-level np.random), else None: graal_step draws from that very state, so Python and C stay one stream."""
        rng = self.rng
        try:
            if rng is np.random:
                rng = np.random.mtrand._rand
            if not isinstance(rng, np.random.RandomState):
                return None
            bg = rng._bit_generator
            if not isinstance(bg, np.random.MT19937):
                return None
            self._bitgen_keepalive = bg
            return int(bg.ctypes.state_address)
        except Exception:
            return None

    def _setup_c_step(self):
        import os
        self._c_step = False
        if os.environ.get("GRAAL_PY_STEP"):        # (the Python path, for comparison)
            return
        if self.exchange == "rccl" and not self._rccl_c:   # the all-reduce is torch's: the Python path drives it
            return
        self._mt_addr = self._mt_state_address()
        if self._mt_addr is None:
            return
        n, nb = int(self.n_new_frags), int(self.n_frags)
        dup = np.zeros(nb, dtype=np.uint8)
        dup[self.id_frag_duplicated] = 1
        black = np.zeros(n, dtype=np.uint8)
        if len(self.id_frags_blacklisted):
            black[np.asarray(self.id_frags_blacklisted, dtype=np.int64)] = 1
        self.engine.upload_proposal_tables(self.distri_frags["xk"], self.distri_frags["pk"], self.id_d, self.frag_dispatcher,
                                           self.collector_id_repeats, dup, black)
        self._c_step = True

    def _step_flags(self):
        """graal_step's flags (include/graal_hip.h): 16 = the carried total + the last commit's own-pixel correction; else 1 = with
        sub-fragments, re-evaluate whenever circular contigs are around (their own pixels follow the circular model); 4 = distance."""
        return (16 if self._own_corr else (0 if self._single_sub else 1)) | (4 if self.compute_dist else 0)

    def _set_total_from_full(self, value):
        """The carried total := a full evaluation of the CURRENT layout: the own-pixel corrections of the commits up to it are void."""
        self.likelihood_t = value
        if self._own_corr:
            self.engine.discard_carry_correction()

    def _step_c(self, id_fA, delta, t, n_step):
        """step_max_likelihood through graal_step (include/graal_hip.h): the proposal, the score post-processing, the sampling and
        the commit run behind the C ABI, on this sampler's own generator state; the rare full re-evaluations stay here.
        Returns the 9-tuple, or None when the C side hands the step back before drawing anything."""
        e = self.engine
        F_t = self.temperature(t, n_step)
        if F_t != 1.0:
            return None
        so = e.step_out
        flags = self._step_flags()
        resync = self.likelihood_t is None or self._force_full or self._steps_since_full + 1 >= self.resync_every
        if resync:
            flags |= 2
        if self.group.world == 1 or self.exchange == "host":
            flags |= 8    # a full re-evaluation that is due (resync, circular contigs with sub-fragments) runs INSIDE the step, next to the scoring
                          # kernels (several ranks: their contact parts are summed through the exchange segment, host_step.h / full_exchange)
        if self.compute_dist and not self._dist_ref_uploaded:
            self.dist_inter_genome()                               # (uploads the reference layout of the distance once)
        rc = e.step(self._mt_addr, id_fA, int(delta), 0.0 if self.likelihood_t is None else float(self.likelihood_t), flags,
                    self._n_circ_prev)
        return self._step_c_tail(rc, id_fA, flags, resync, F_t)

    def _step_c_tail(self, rc, id_fA, flags, resync, F_t):
        """What follows graal_step's return (also for the step graal_steps stopped at): the full re-evaluations, the rare fallback
        to numpy's own selection, the 9-tuple."""
        e = self.engine
        so = e.step_out
        if rc == STEP_FALLBACK:
            return None
        st = so.stats
        n_circ = int(st[6])
        self.n_stale_paste += int(st[7])
        self._steps_since_full += 1
        if (flags & 8) and rc != STEP_PAUSED and resync:   # (re-evaluated inside the step: graal_step_out.full_likelihood)
            self._steps_since_full = 0
            self._force_full = False
        if rc == STEP_PAUSED:
            if resync:
                self.likelihood_t = self._full_likelihood()
                self._steps_since_full = 0
                self._force_full = False
            elif (n_circ or self._n_circ_prev) and not self._single_sub and not self._own_corr:
                self.likelihood_t = self._full_likelihood()
            rc = e.step_finish(self._mt_addr, float(self.likelihood_t), flags)
        self._n_circ_prev = n_circ
        K = int(so.n_neighbours)
        self.last_neighbours = list(so.neighbours[:K])
        self.score = e.step_scores[:K * self.n_tmp_struct]
        max_id = np.int32(so.max_id)
        if rc == STEP_SELECT:                                      # an unusual score vector: numpy judges (and raises) itself -- the
                                                                   # neighbours are drawn and scored, only the selection is left
            sample_out, o = select_move(self.score, self.n_tmp_struct, self.rng, F_t)
            id_f_sampled = self.last_neighbours[sample_out // self.n_tmp_struct]
            op_sampled = sample_out % self.n_tmp_struct
            self.test_copy_struct(id_fA, id_f_sampled, op_sampled, max_id)
            dist = self.dist_inter_genome() if self.compute_dist else 0.0
        else:
            o, op_sampled, id_f_sampled = float(so.o), int(so.op_sampled), int(so.id_f_sampled)
            if self.compute_dist:
                norm_distance = 3.0 * (len(self.np_init_prev) - self.n_frags_4_dist)
                dist = (norm_distance - 0.5 * int(so.dist_half_units)) / norm_distance
            else:
                dist = 0.0
        self.o = o
        self.likelihood_t = o
        if not np.isfinite(o):
            self._force_full = True   # (a flagged candidate won: the carried total is not a likelihood; re-evaluate next step)
        return (o, int(st[0]), np.int32(st[5]), float(st[3]) / float(st[2]), np.int32(st[4]), op_sampled, id_f_sampled, dist, F_t)

    # ------------------------------------------------------------------ a run of MCMC steps (the inner loop of start_EM)
    def steps_max_likelihood(self, ids, delta, size_block=512, dt=0, t=0, n_step=1):
        """step_max_likelihood for every fragment of `ids`, in order (the inner loop of start_EM, ``cuda_lib_gl.py:2196-2220``);
        returns the list of their 9-tuples.  Runs of steps that need nothing from Python -- no blacklisted fragment, no full
        re-evaluation due, temperature 1 -- go through graal_steps in ONE call each (the per-step Python around graal_step is
        ~25 us of a 55-90 us step); everything else takes step_max_likelihood, and so does the step a run stopped at."""
        ids = [int(i) for i in ids]
        out = []
        i = 0
        n = len(ids)
        F_t = self.temperature(t, n_step)
        while i < n:
            room = self.resync_every - self._steps_since_full - 1 if self.likelihood_t is not None and not self._force_full else 0
            if not (self._c_step and F_t == 1.0 and room >= 2 and (not self.compute_dist or self._dist_ref_uploaded)):
                out.append(self.step_max_likelihood(ids[i], delta, size_block, dt, t, n_step))
                i += 1
                continue
            j = i
            while j < n and j - i < room and ids[j] not in self._black_set:
                j += 1
            if j - i < 2:
                out.append(self.step_max_likelihood(ids[i], delta, size_block, dt, t, n_step))
                i += 1
                continue
            e = self.engine
            flags = self._step_flags()
            rc, rows = e.steps(self._mt_addr, ids[i:j], int(delta), float(self.likelihood_t), flags, self._n_circ_prev)
            for r in rows:
                o = float(r[0])
                if self.compute_dist:
                    norm_distance = 3.0 * (len(self.np_init_prev) - self.n_frags_4_dist)
                    dist = (norm_distance - 0.5 * int(r[7])) / norm_distance
                else:
                    dist = 0.0
                out.append((o, int(r[1]), np.int32(r[2]), float(r[3]), np.int32(r[4]), int(r[5]), int(r[6]), dist, F_t))
            k = len(rows)
            if k:
                last = rows[-1]
                self.n_stale_paste += int(rows[:, 9].sum())
                self._steps_since_full += k
                self._n_circ_prev = int(last[8])
                self.o = self.likelihood_t = float(last[0])
                if rc == STEP_DONE:              # (what the last step left, as after step_max_likelihood)
                    so = e.step_out
                    K = int(so.n_neighbours)
                    self.last_neighbours = list(so.neighbours[:K])
                    self.score = e.step_scores[:K * self.n_tmp_struct]
                if not np.isfinite(self.likelihood_t):
                    self._force_full = True
            i += k
            if rc != STEP_DONE and i < n:        # the step the run stopped at: its state is in step_out, as after graal_step
                res = self._step_c_tail(rc, ids[i], flags, False, F_t)
                if res is None:                  # (handed back before anything was drawn: the Python path takes it)
                    res = self._step_py(ids[i], delta, size_block, dt, t, n_step)
                out.append(res)
                i += 1
        return out

    # ------------------------------------------------------------------ one MCMC step
    def step_max_likelihood(self, id_fA, delta, size_block=512, dt=0, t=0, n_step=1):
        """``cuda_lib_gl.py:1793-1980``.  Returns (o, n_contigs, min_len, mean_len_bp, max_len, op_sampled,
        id_f_sampled, dist, F_t)."""
        id_fA = int(id_fA)
        if self._c_step and id_fA not in self._black_set:
            res = self._step_c(id_fA, delta, t, n_step)
            if res is not None:
                return res
        return self._step_py(id_fA, delta, size_block, dt, t, n_step)

    def _step_py(self, id_fA, delta, size_block=512, dt=0, t=0, n_step=1):
        """The step with its host logic in Python (blacklisted fragments, temperatures other than 1, whatever graal_step hands back)."""
        # relabel + index are launched; the proposal is drawn while they run (nothing between here and the reference's
        # return_neighbours call draws from the generator, so the stream is the reference's); then the statistics are read
        self.engine.begin_step_launch()
        id_neighbours = None if id_fA in self._black_set else self.return_neighbours(id_fA, delta)
        st, max_id = self.engine.begin_step()   # statistics (published by the commit kernel), one wait
        max_id = np.int32(max_id)
        n_circ = int(st[6])
        self.n_stale_paste += int(st[7])
        n_contigs = int(st[0])
        mean_len_bp = float(st[3]) / float(st[2])
        max_len = np.int32(st[4])
        min_len = np.int32(st[5])
        if id_fA in self._black_set:            # cuda_lib_gl.py:1962-1978: nothing is proposed for a blacklisted fragment
            o = self.o
            dist = self.dist_inter_genome() if self.compute_dist else 0.0
            # The reference sets likelihood_t = o here (the score of the last sampled move, possibly under parameters a
            # nuisance step has replaced since) and loses at most this one step, because its next step re-evaluates the full
            # likelihood.  Here the total is CARRIED from step to step: keep the reference's value for a nuisance step that
            # may follow, but make the next scored step start from a full evaluation instead of this stale number.
            self.likelihood_t = o
            self._force_full = True
            return o, n_contigs, min_len, mean_len_bp, max_len, -1, id_fA, dist, self.temperature(t, n_step)
        self._steps_since_full += 1
        corr, corr_ok = self.engine.take_carry_correction() if self._own_corr else (0.0, True)
        if self.likelihood_t is None or self._force_full or self._steps_since_full >= self.resync_every or not corr_ok:
            # the carried-over total drifts by ~1e-10 |logL| per step against a full evaluation (deviation 1):
            # re-evaluate now and then (the reference does it every step, cuda_lib_gl.py:1828-1848)
            self.likelihood_t = self._full_likelihood()
            self._steps_since_full = 0
            self._force_full = False
        elif self._own_corr:
            self.likelihood_t += corr          # (what the last commit did to its bins' own pixels: graal_take_carry_correction)
        elif (n_circ or self._n_circ_prev) and not self._single_sub:
            # candidate deltas never include a bin's own pixel (as in the reference); those pixels only change with
            # the circular model, so resynchronise the carried-over total whenever circular contigs are around
            self.likelihood_t = self._full_likelihood()
        self._n_circ_prev = n_circ
        likelihood_t = self.likelihood_t
        n_neighbours = len(id_neighbours)
        id_neighbours.sort()
        self.last_neighbours = list(id_neighbours)
        deltas = self._candidate_deltas(id_fA, id_neighbours, max_id)
        self.score = (deltas + likelihood_t).reshape(n_neighbours * self.n_tmp_struct)
        F_t = self.temperature(t, n_step)
        sample_out, o = select_move(self.score, self.n_tmp_struct, self.rng, F_t)
        id_f_sampled = id_neighbours[sample_out // self.n_tmp_struct]
        op_sampled = sample_out % self.n_tmp_struct
        self.test_copy_struct(id_fA, id_f_sampled, op_sampled, max_id)
        self.o = o
        dist = self.dist_inter_genome() if self.compute_dist else 0.0
        self.likelihood_t = o
        if not np.isfinite(o):
            self._force_full = True   # (a flagged candidate won: the carried total is not a likelihood; re-evaluate next step)
        return o, n_contigs, min_len, mean_len_bp, max_len, op_sampled, id_f_sampled, dist, F_t
