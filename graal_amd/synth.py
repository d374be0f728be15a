"""Synthetic GRAAL problems: the inputs ``simulation_loader.simulation`` hands to
the sampler constructor (``simulation_loader.py:92-107``), generated instead of read from a pyramid.

The real S1 / T. reesei tarballs are not in the tree (``README.md:99-101`` of the reference), so every
BASELINE config runs on a stand-in produced here; the generator is seeded and committed so fixtures are
reproducible.  Contacts are produced directly in COO form (three int32 arrays sorted by (i, j), the
``(3, nnz)`` layout of ``pyramid_sparse.py:314-322``); nothing here densifies.
"""
import numpy as np

FIELDS = ("pos", "id_c", "start_bp", "len_bp", "circ", "id", "prev", "next", "l_cont", "l_cont_bp", "ori",
          "rep", "activ", "id_d")

C5_CONTIG_WEIGHTS = (6.8, 6.2, 5.3, 4.6, 4.0, 3.4, 2.7)


def rippe_np(s, p):
    """Rippe contact model in float64 numpy (``kernels3.cu:120-133`` without the float32 rounding)."""
    kuhn, lm, c1, slope, d, d_max, fact, v_inter = [float(x) for x in p]
    s = np.asarray(s, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = c1 * np.power(s, slope) * np.exp((d - 2.0) / ((s * lm / kuhn) ** 2 + d)) * fact
    r = np.where((s > 0) & (s < d_max), r, 0.0)
    return np.maximum(r, v_inter)


def make_param_simu(kuhn=1.0, lm=9.6, slope=-1.5, d=3.0, fact=1e4, v_inter=1e-3, d_max=None):
    """param_simu = (kuhn, lm, c1, slope, d, d_max, fact, v_inter) float32; c1 as ``cuda_lib_gl.py:1208``;
    d_max defaults to the smallest s with rippe(s) <= v_inter (bisection)."""
    kuhn = np.float32(kuhn)
    lm = np.float32(lm)
    c1 = np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))
    if d_max is None:
        f = lambda s: float(c1) * s ** slope * np.exp((d - 2.0) / ((s * float(lm) / float(kuhn)) ** 2 + d)) * fact
        lo, hi = 1e-3, 1e-3
        while f(hi) > v_inter:
            hi *= 2.0
        lo = hi / 2.0
        for _ in range(100):
            mid = 0.5 * (lo + hi)
            if f(mid) > v_inter:
                lo = mid
            else:
                hi = mid
        d_max = hi
    return np.array([kuhn, lm, c1, slope, d, d_max, fact, v_inter], dtype=np.float32)


def _layout(rng, n_bins, n_sub, contig_weights, mean_len_bp, ragged):
    """Sub-level fragments grouped into bins of <= n_sub consecutive sub-frags inside each contig."""
    w = np.asarray(contig_weights, dtype=np.float64)
    bins_per_contig = np.maximum(1, np.floor(w / w.sum() * n_bins).astype(np.int64))
    bins_per_contig[0] += n_bins - bins_per_contig.sum()
    assert bins_per_contig.min() >= 1 and bins_per_contig.sum() == n_bins
    sub_per_bin = np.full(n_bins, n_sub, dtype=np.int32)
    if ragged and n_sub > 1:
        ends = np.cumsum(bins_per_contig) - 1  # last bin of each contig holds 1..n_sub sub-frags
        sub_per_bin[ends] = rng.randint(1, n_sub + 1, size=len(ends))
    n_sub_total = int(sub_per_bin.sum())
    sub_len_bp = (1 + np.round(rng.exponential(mean_len_bp, size=n_sub_total))).astype(np.int32)
    return bins_per_contig, sub_per_bin, sub_len_bp


def make_problem(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217, contig_weights=C5_CONTIG_WEIGHTS,
                 mean_len_bp=660.0, accu=1, ragged=True, param=None, frac_cis=0.85, max_count_mean=50.0,
                 grid_bp=None):
    """Build one synthetic problem.

    n_sub = 1 is the "level 0" semantics (bins = restriction fragments, accu = 1); n_sub = 3 mimics a
    pyramid level >= 1 (bins of 3 sub-level fragments, ``simulation_loader.py:673-704``).
    ``grid_bp``: if set, every sub-frag length is rounded up to a multiple of it (e.g. 2000 makes every
    float32 kb coordinate exact, so dense float32 geometry is shift-invariant -- used by trace fixtures).
    Returns a dict with the sampler-constructor inputs plus the COO lists.
    """
    rng = np.random.RandomState(seed)
    bins_per_contig, sub_per_bin, sub_len_bp = _layout(rng, n_bins, n_sub, contig_weights, mean_len_bp, ragged)
    if grid_bp:
        sub_len_bp = (((sub_len_bp + grid_bp - 1) // grid_bp) * grid_bp).astype(np.int32)
    n_sub_total = int(sub_per_bin.sum())
    sub_first = np.concatenate([[0], np.cumsum(sub_per_bin)[:-1]]).astype(np.int32)
    bin_of_sub = np.repeat(np.arange(n_bins, dtype=np.int32), sub_per_bin)
    # bin-level fragment SoA (pyramid_sparse.py:1316-1327)
    len_bp = np.add.reduceat(sub_len_bp.astype(np.int64), sub_first).astype(np.int32)
    contig_of_bin = np.repeat(np.arange(len(bins_per_contig), dtype=np.int32), bins_per_contig)
    first_bin = np.concatenate([[0], np.cumsum(bins_per_contig)[:-1]])
    pos = (np.arange(n_bins) - first_bin[contig_of_bin]).astype(np.int32)
    cum = np.cumsum(len_bp.astype(np.int64))
    contig_start = np.concatenate([[0], cum])[first_bin]
    start_bp = (cum - len_bp - contig_start[contig_of_bin]).astype(np.int32)
    l_cont = bins_per_contig[contig_of_bin].astype(np.int32)
    l_cont_bp = np.add.reduceat(len_bp.astype(np.int64), first_bin)[contig_of_bin].astype(np.int32)
    idx = np.arange(n_bins, dtype=np.int32)
    prev = np.where(pos == 0, -1, idx - 1).astype(np.int32)
    nxt = np.where(pos == l_cont - 1, -1, idx + 1).astype(np.int32)
    soa = dict(pos=pos, id_c=(contig_of_bin + 1).astype(np.int32), start_bp=start_bp, len_bp=len_bp,
               circ=np.zeros(n_bins, np.int32), id=idx.copy(), prev=prev, next=nxt, l_cont=l_cont,
               l_cont_bp=l_cont_bp, ori=np.ones(n_bins, np.int32), rep=np.zeros(n_bins, np.int32),
               activ=np.ones(n_bins, np.int32), id_d=idx.copy())
    # sub-frag tables (simulation_loader.py:673-704): int4 ids, float3 len in kb, int3 accu
    sub_id = np.zeros((n_bins, 4), dtype=np.int32)
    sub_len = np.zeros((n_bins, 3), dtype=np.float32)
    sub_accu = np.zeros((n_bins, 3), dtype=np.int32)
    sub_id[:, 3] = sub_per_bin
    # accu = an int (uniform RF counts) or ("random", lo, hi): every sub-fragment draws its own count in [lo, hi]
    if isinstance(accu, (tuple, list)):
        accu_of_sub = rng.randint(int(accu[1]), int(accu[2]) + 1, size=n_sub_total).astype(np.int32)
        accu = float(accu_of_sub.mean())
    else:
        accu_of_sub = np.full(n_sub_total, accu, dtype=np.int32)
    for k in range(3):
        m = sub_per_bin > k
        sub_id[m, k] = sub_first[m] + k
        sub_len[m, k] = np.float32(sub_len_bp[sub_first[m] + k]) / np.float32(1000.0)
        sub_accu[m, k] = accu_of_sub[sub_first[m] + k]
    accu_all = accu_of_sub.astype(np.float32)
    nfpb = np.float32(accu_all.mean() ** 2)  # simulation_loader.py:73
    if param is None:
        param = make_param_simu()
    param = np.asarray(param, dtype=np.float32)
    # sub-level centre coordinates in kb (within contig) for drawing contacts
    sub_contig = contig_of_bin[bin_of_sub]
    scum = np.cumsum(sub_len_bp.astype(np.int64))
    sub_first_of_contig = sub_first[first_bin]
    sub_cstart = np.concatenate([[0], scum])[sub_first_of_contig]
    centre_kb = (scum - sub_len_bp / 2.0 - sub_cstart[sub_contig]) / 1000.0
    # ---- contacts: frac_cis cis with offset ~ rippe, rest uniform trans ---------------------
    S = n_sub_total
    nnz = int(min(nnz, S * (S - 1) // 2 // 2))
    mean_kb = float(sub_len_bp.mean()) / 1000.0
    kmax = int(min(S - 1, max(2, np.ceil(float(param[5]) / mean_kb))))
    off_p = rippe_np(np.arange(1, kmax + 1) * mean_kb, param)
    off_p = off_p / off_p.sum()
    off_cdf = np.cumsum(off_p)
    # never ask for more distinct pairs than exist (tiny test problems): cap at half of each population
    n_sub_contig = np.bincount(sub_contig).astype(np.int64)
    max_cis = int((n_sub_contig * (n_sub_contig - 1) // 2).sum())
    max_trans = S * (S - 1) // 2 - max_cis
    ks = np.arange(1, kmax + 1)
    reach_cis = int(sum(np.maximum(0, nc - ks).sum() for nc in n_sub_contig))  # pairs with offset <= kmax
    target_cis = int(min(round(nnz * frac_cis), max_cis // 2, reach_cis // 2))
    nnz = int(min(nnz, target_cis + max_trans // 2))
    contig_first = np.concatenate([[0], np.cumsum(n_sub_contig)[:-1]])  # first sub-fragment of every contig

    def draw_cis(target):
        """`target` distinct cis pairs (i, i + k): the count per offset k follows off_p, capped by the number of
        pairs that exist at that offset (water filling), positions drawn without replacement per offset."""
        n_k = np.array([np.maximum(0, n_sub_contig - k).sum() for k in ks], dtype=np.int64)
        lo_s, hi_s = 0.0, float(target) / max(off_p[n_k > 0].min(), 1e-300) + 1.0
        for _ in range(200):
            mid = 0.5 * (lo_s + hi_s)
            if np.minimum(n_k, np.floor(mid * off_p)).sum() < target:
                lo_s = mid
            else:
                hi_s = mid
        m_k = np.minimum(n_k, np.floor(hi_s * off_p)).astype(np.int64)
        excess = int(m_k.sum() - target)
        for k in np.argsort(-m_k, kind="stable"):  # trim the rounding excess from the fullest offsets
            if excess <= 0:
                break
            take = min(excess, int(m_k[k]))
            m_k[k] -= take
            excess -= take
        out = []
        for idx in np.nonzero(m_k)[0]:
            k = int(ks[idx])
            per = np.maximum(0, n_sub_contig - k)           # valid start positions per contig
            cum = np.concatenate([[0], np.cumsum(per)])
            t = rng.choice(int(cum[-1]), int(m_k[idx]), replace=False) if m_k[idx] < cum[-1] else np.arange(cum[-1])
            c = np.searchsorted(cum, t, side="right") - 1
            i = contig_first[c] + (t - cum[c])
            out.append(i * S + (i + k))
        return np.concatenate(out) if out else np.zeros(0, dtype=np.int64)

    def draw_trans(target):
        got = np.zeros(0, dtype=np.int64)
        guard = 0
        while len(got) < target:
            guard += 1
            if guard > 200:
                raise RuntimeError("synthetic contact generation does not converge")
            m = int((target - len(got)) * 1.15) + 4096
            i = rng.randint(0, S, size=m).astype(np.int64)
            j = rng.randint(0, S, size=m).astype(np.int64)
            i, j = np.minimum(i, j), np.maximum(i, j)
            ok = (i != j) & (sub_contig[i] != sub_contig[j])
            new = i[ok] * S + j[ok]
            got = np.unique(np.concatenate([got, new])) if len(got) else np.unique(new)
        if len(got) > target:
            keep = np.ones(len(got), dtype=bool)
            keep[rng.choice(len(got), len(got) - target, replace=False)] = False
            got = got[keep]
        return got

    keys = np.concatenate([draw_cis(target_cis), draw_trans(nnz - target_cis)])
    keys.sort()
    row = (keys // S).astype(np.int32)
    col = (keys % S).astype(np.int32)
    same = sub_contig[row] == sub_contig[col]
    lam = np.where(same, rippe_np(np.abs(centre_kb[col] - centre_kb[row]), param), float(param[7]))
    lam = np.clip(lam * accu * accu / float(nfpb), 0.0, max_count_mean)
    val = (1 + rng.poisson(lam)).astype(np.int32)
    # bin-level COO (upper, i<j) for the neighbour proposal (cuda_lib_gl.py:2363-2390)
    bi, bj = bin_of_sub[row].astype(np.int64), bin_of_sub[col].astype(np.int64)
    m = bi != bj
    bkeys = np.minimum(bi[m], bj[m]) * n_bins + np.maximum(bi[m], bj[m])
    ub, inv = np.unique(bkeys, return_inverse=True)
    bval = np.bincount(inv, weights=val[m].astype(np.float64)).astype(np.float32)
    # mean_value_trans: contacts between different contigs / number of such ordered sub-frag pairs
    n_per_contig = np.bincount(sub_contig).astype(np.float64)
    n_trans_pairs = float(S) * S - float((n_per_contig ** 2).sum())
    mean_value_trans = np.float32(2.0 * val[~same].sum() / max(n_trans_pairs, 1.0))
    return dict(
        S_o_A_frags=soa, n_frags=n_bins, n_new_frags=n_bins, init_n_sub_frags=S, n_new_sub_frags=S,
        collector_id_repeats=np.arange(n_bins, dtype=np.int32),
        frag_dispatcher=np.stack([np.arange(n_bins), np.arange(n_bins) + 1], axis=1).astype(np.int32),
        id_frag_duplicated=[], id_frags_blacklisted=[],
        np_sub_frags_id=sub_id, np_sub_frags_len_bp=sub_len, np_sub_frags_accu=sub_accu,
        mean_squared_frags_per_bin=nfpb, mean_value_trans=mean_value_trans, param_simu=param,
        coo_row=row, coo_col=col, coo_val=val,
        bin_coo_row=(ub // n_bins).astype(np.int32), bin_coo_col=(ub % n_bins).astype(np.int32), bin_coo_val=bval,
        sub_len_bp=sub_len_bp, bin_of_sub=bin_of_sub)


def add_repeats(problem, dup_bins, n_copies):
    """Repeated fragments as ``simulation_loader.modify_vect_frags`` appends them (``simulation_loader.py:182-280``): every bin
    of `dup_bins` gets `n_copies` extra fragments, each a singleton contig of its own (fresh label, ``rep = 1``,
    ``activ = 1``, ``id_d`` = the bin); ``collector_id_repeats`` / ``frag_dispatcher`` list the copies of every bin."""
    p = dict(problem)
    S = {k: np.array(v, dtype=np.int32, copy=True) for k, v in p["S_o_A_frags"].items()}
    n0 = int(p["n_frags"])
    S.setdefault("rep", np.zeros(n0, np.int32)); S.setdefault("activ", np.ones(n0, np.int32))
    S.setdefault("id_d", np.arange(n0, dtype=np.int32)); S.setdefault("ori", np.ones(n0, np.int32))
    add = {k: [] for k in S}
    max_f, max_c = n0, int(S["id_c"].max()) + 1
    dup_bins = [int(b) for b in dup_bins]
    for b in dup_bins:
        for _ in range(int(n_copies)):
            row = dict(pos=0, id_c=max_c, start_bp=0, len_bp=int(S["len_bp"][b]), circ=int(S["circ"][b]), id=max_f, prev=-1,
                       next=-1, l_cont=1, l_cont_bp=int(S["len_bp"][b]), ori=1, rep=1, activ=1, id_d=b)
            for k in add:
                add[k].append(row[k])
            max_f += 1; max_c += 1
    S = {k: np.concatenate([S[k], np.asarray(add[k], dtype=np.int32)]) for k in S}
    collector, dispatcher, x = [], [], 0
    for b in range(n0):
        ids = np.nonzero(S["id_d"] == b)[0] if b in dup_bins else np.array([b])
        collector.extend(int(i) for i in ids)
        dispatcher.append((x, x + len(ids)))
        x += len(ids)
    p.update(S_o_A_frags=S, n_new_frags=max_f, collector_id_repeats=np.asarray(collector, np.int32),
             frag_dispatcher=np.asarray(dispatcher, np.int32), id_frag_duplicated=dup_bins)
    return p


def write_dataset(problem, folder, max_reads=None):
    """A synthetic problem (n_sub = 1: bins = restriction fragments) as the reference's 3-file TEXT dataset (``README.md:111-113``):
    ``info_contigs.txt``, ``fragments_list.txt`` and ``abs_fragments_contacts_weighted.txt`` -- ONE line per read pair, so a contact of
    count c is written c times (``abs_contact_2_coo_file``, ``pyramid_sparse.py:222-264``, counts lines).  What ``python -m graal_amd.run
    --dataset folder --size-pyramid 1 --level 0`` starts from: BASELINE config 4's data path on a stand-in.
    ``max_reads``: cap the number of read lines (contacts keep at least one read each).  Returns the number of read lines."""
    import os
    S = problem["S_o_A_frags"]
    n = int(problem["n_frags"])
    if int(problem["init_n_sub_frags"]) != n:
        raise ValueError("write_dataset needs a problem with one sub-fragment per bin")
    os.makedirs(folder, exist_ok=True)
    id_c = np.asarray(S["id_c"])
    labels = np.unique(id_c)
    with open(os.path.join(folder, "info_contigs.txt"), "w") as f:
        f.write("contig\tlength_kb\tn_frags\tcumul_length\n")
        cum = 0
        for c in labels:
            m = id_c == c
            f.write("contig%d\t%d\t%d\t%d\n" % (int(c), int(np.asarray(S["len_bp"])[m].sum()), int(m.sum()), cum))
            cum += int(m.sum())
    start, length, pos = (np.asarray(S[k]).astype(np.int64) for k in ("start_bp", "len_bp", "pos"))
    with open(os.path.join(folder, "fragments_list.txt"), "w") as f:
        f.write("id\tchrom\tstart_pos\tend_pos\tsize\tgc_content\n")
        f.write("".join("%d\tcontig%d\t%d\t%d\t%d\t0.5\n" % (pos[i] + 1, id_c[i], start[i], start[i] + length[i], length[i]) for i in range(n)))
    row, col, val = (np.asarray(problem[k]).astype(np.int64) for k in ("coo_row", "coo_col", "coo_val"))
    if max_reads is not None and val.sum() > max_reads:
        extra = np.maximum(val - 1, 0)
        keep = max(0, int(max_reads) - len(val))
        val = 1 + np.floor(extra * (keep / max(1, int(extra.sum())))).astype(np.int64)
    a, b = np.repeat(row + 1, val), np.repeat(col + 1, val)
    path = os.path.join(folder, "abs_fragments_contacts_weighted.txt")
    try:
        import pandas as pd
        pd.DataFrame({"id_read_a": a, "id_read_b": b, "w": np.ones(len(a), np.int8)}).to_csv(path, sep="\t", index=False)
    except ImportError:
        with open(path, "w") as f:
            f.write("id_read_a\tid_read_b\tw\n")
            for i in range(0, len(a), 1 << 20):
                f.write("".join("%d\t%d\t1\n" % t for t in zip(a[i:i + (1 << 20)].tolist(), b[i:i + (1 << 20)].tolist())))
    return int(len(a))


def dense_from_coo(row, col, val, n, dtype=np.float32):
    """Symmetric dense matrix with a zero diagonal, as ``simulation_loader.py:81-82`` +
    ``cuda_lib_gl.py:155-160`` build it.  Test / oracle helper for SMALL problems only."""
    m = np.zeros((n, n), dtype=dtype)
    m[row, col] = val
    m[col, row] = val
    return m


def with_dense(problem):
    """Add the dense matrices the reference's sampler (and the oracle) take.  SMALL problems only."""
    p = dict(problem)
    p["hic_matrix"] = dense_from_coo(p["coo_row"], p["coo_col"], p["coo_val"], p["init_n_sub_frags"])
    p["hic_matrix_sub_sampled"] = dense_from_coo(p["bin_coo_row"], p["bin_coo_col"], p["bin_coo_val"], p["n_frags"])
    return p
