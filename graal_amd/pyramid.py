"""Pyramid builder and loader for GRAAL's on-disk dataset -- SURVEY.md section 8 row f3 (and f4: FASTA export).

What it mirrors in the reference (``pyramid_sparse.py``), function by function:

* ``abs_contacts_to_coo``        <- ``abs_contact_2_coo_file``   (:222-264)  one line per read pair -> COO counts
* ``init_frag_list``             <- ``init_frag_list``           (:328-355)
* ``remove_problematic_fragments`` <- same name                   (:573-846)  sparsity filter ``mean - 1.01 sigma``
* ``subsample_data_set``         <- same name                    (:358-569)  bins of ``factor`` fragments per contig
* ``build_and_filter``           <- ``build`` + ``build_and_filter`` (:25-218)
* ``Pyramid`` / ``Level``        <- ``pyramid`` / ``level``      (:896-1380) fragment dictionaries, S_o_A_frags, mean trans
* ``simulation_inputs``          <- ``simulation.__init__`` + ``create_sub_frags`` (``simulation_loader.py:41-127, 673-704``)
* ``load_reference_sequence`` / ``generate_new_fasta`` <- (:1148-1174, 1430-1488)

The same text files are written in the same folder layout (``pyramids/pyramid_<n>_thresh_auto/level_<k>/<k>_*.txt``), so a
pyramid built here can be read by the reference and vice versa -- except for the contact matrices, which the reference
keeps in ``pyramid.hdf5`` (h5py is not available here): they go to ``pyramid.npz`` with the same logical layout, one
``(3, nnz)`` int32 array ``<level>/data`` (rows: id_a, id_b, count) plus ``<level>/nfrags``.  READING goes both ways: a folder
that holds the reference's ``pyramid.hdf5`` instead is loaded through h5py when it can be imported and through
``graal_amd/hdf5_min.py`` (a pure-Python reader of that file's HDF5 subset) when not; ``write_hdf5`` exports with h5py.  Nothing is ever densified
(``simulation_loader.py:81-82`` is what is NOT reproduced): the COO arrays go straight to the engine.

Reference quirks kept on purpose (switchable), because they change the numbers a reference run would produce:

* ``subsample_data_set`` drops the FIRST contact line of every level it sub-samples (header consumed by ``readline()``,
  then ``range(1, len(all_lines))``, ``pyramid_sparse.py:527-531``) -- ``drop_first_contact=True``;
* ``load_reference_sequence`` drops the last line of the FASTA file and leaves the newlines inside the LAST record
  (``:1169-1173``) -- ``strict_reference=True``.
"""
import os

import numpy as np


# ----------------------------------------------------------------------------------------- small text-table helpers
def _read_table(path):
    with open(path) as f:
        header = f.readline().rstrip("\n").split("\t")
        rows = [line.rstrip("\n").split("\t") for line in f if line.strip()]
    return header, rows


def _write_table(path, header, rows):
    with open(path, "w") as f:
        f.write("\t".join(header) + "\n")
        for r in rows:
            f.write("\t".join(str(x) for x in r) + "\n")


def _coo_sum(fa, fb, nc):
    """Sum duplicates of (min, max) pairs; sorted by (a, b) -- what the reference's dict-of-dicts + sorted keys produce."""
    fa, fb, nc = np.asarray(fa, np.int64), np.asarray(fb, np.int64), np.asarray(nc, np.int64)
    a, b = np.minimum(fa, fb), np.maximum(fa, fb)
    if len(a) == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32)
    key = a * (int(b.max()) + 1) + b
    order = np.argsort(key, kind="stable")
    key, a, b, nc = key[order], a[order], b[order], nc[order]
    first = np.concatenate([[True], key[1:] != key[:-1]])
    idx = np.nonzero(first)[0]
    return a[idx].astype(np.int32), b[idx].astype(np.int32), np.add.reduceat(nc, idx).astype(np.int32)


def abs_contacts_to_coo(abs_contact_file):
    """``abs_contact_2_coo_file``: one line per contact with two 1-based fragment ids -> 0-based COO counts.
    (pandas' C reader when it is importable -- a level-0 dataset holds 1e7-1e8 read lines --, plain Python otherwise: same result)"""
    ab = None
    try:
        import pandas as pd
        df = pd.read_csv(abs_contact_file, sep=r"\s+", header=0, usecols=[0, 1], dtype=np.int64, engine="c")
        ab = df.to_numpy(dtype=np.int64) - 1
    except ImportError:
        pass
    except Exception:      # (an unusual file: the plain reader judges it)
        ab = None
    if ab is None:
        with open(abs_contact_file) as f:
            f.readline()
            rows = [line.split()[:2] for line in f if line.strip()]
        if not rows:
            return _coo_sum([], [], [])
        ab = np.array(rows, dtype=np.int64) - 1
    if len(ab) == 0:
        return _coo_sum([], [], [])
    return _coo_sum(ab[:, 0], ab[:, 1], np.ones(len(ab), np.int64))


def write_coo(path, coo):
    _write_table(path, ["id_frag_a", "id_frag_b", "n_contact"], zip(*[x.tolist() for x in coo]))


def read_coo(path):
    _, rows = _read_table(path)
    if not rows:
        return _coo_sum([], [], [])
    a = np.array([[int(x) for x in r[:3]] for r in rows], dtype=np.int64)
    return a[:, 0].astype(np.int32), a[:, 1].astype(np.int32), a[:, 2].astype(np.int32)


def init_frag_list(fragment_list, new_frag_list):
    """``init_frag_list``: the raw fragments list + accu_frag = 1, frag_start = frag_end = id."""
    _, rows = _read_table(fragment_list)
    out = [[r[0], r[1], r[2], r[3], r[4], str(float(r[5])), "1", r[0], r[0]] for r in rows]
    _write_table(new_frag_list, ["id", "chrom", "start_pos", "end_pos", "size", "gc_content", "accu_frag", "frag_start",
                                 "frag_end"], out)
    return len(out)


# ----------------------------------------------------------------------------------------- the sparsity filter
def remove_problematic_fragments(contig_info, fragments_list, coo, nfrags, new_contig_list_file, new_fragments_list_file):
    """``remove_problematic_fragments``: fragments whose row of the (symmetrised) contact matrix has a fraction of non-zero
    entries <= mean - 1.01 sigma, or whose size is <= 1, are "locked": a locked fragment is merged into the following
    fragments of its contig until an unlocked one closes the bin (a locked run at the end of a contig is destroyed).
    Returns (thresh, old_2_new) with old_2_new[i] = new 0-based id of level-0 fragment i, or -1 if destroyed."""
    a, b, v = coo
    deg = np.zeros(int(nfrags), dtype=np.int64)
    off = a != b
    # CSR + CSR^T: a diagonal entry appears once in the sum, an off-diagonal pair once in each row
    np.add.at(deg, a[off], 1); np.add.at(deg, b[off], 1); np.add.at(deg, a[~off], 1)
    sparsity = np.float32(deg) / np.float32(nfrags)
    mean_s, std_s = sparsity.mean(), sparsity.std()
    thresh = mean_s - 1.01 * std_s
    problem = set(np.nonzero(sparsity <= thresh)[0].tolist())
    _, frows = _read_table(fragments_list)
    _, crows = _read_table(contig_info)
    info = {c[0]: {"n_new_frags": 0, "length_kb": 0} for c in crows}
    prob_names = set(frows[i][0] + "-" + frows[i][1] for i in problem)
    old_2_new = np.full(len(frows), -1, dtype=np.int64)
    out = []
    new_abs, new_rel = 1, 0
    cum = {"start_pos": 0, "size": 0, "accu": 0, "lock": False, "ids": [], "gc": []}
    for init_abs, r in enumerate(frows, start=1):
        fid, chrom, start_pos, end_pos = int(r[0]), r[1], r[2], r[3]
        size, gc, accu = int(r[4]), float(r[5]), int(r[6])
        if fid == 1:  # first fragment of a contig: an open (locked) run of the previous contig is destroyed
            new_rel = 1
            if cum["lock"]:
                for e in cum["ids"]:
                    old_2_new[e - 1] = -1
            cum.update(gc=[], start_pos=0, ids=[], size=0)
        lock = (str(fid) + "-" + chrom) in prob_names
        cum["size"] += size
        cum["accu"] += accu
        cum["lock"] = lock or size <= 1
        cum["gc"].append(gc)
        cum["ids"].append(init_abs)
        old_2_new[init_abs - 1] = new_abs
        if not lock:
            info[chrom]["n_new_frags"] += 1
            info[chrom]["length_kb"] += cum["size"]
            out.append([new_rel, chrom, cum["start_pos"], end_pos, cum["size"], float(np.array(cum["gc"]).mean()), cum["accu"],
                        new_rel, new_rel])
            cum.update(start_pos=end_pos, size=0, accu=0, lock=lock, gc=[], ids=[])
            new_rel += 1
            new_abs += 1
    if cum["lock"]:
        for e in cum["ids"]:
            old_2_new[e - 1] = -1
    _write_table(new_fragments_list_file, ["id", "chrom", "start_pos", "end_pos", "size", "gc_content", "accu_frag",
                                           "frag_start", "frag_end"], out)
    cumul, crows_out = 0, []
    for c in crows:
        n = info[c[0]]["n_new_frags"]
        if n > 0:
            crows_out.append([c[0], info[c[0]]["length_kb"], n, cumul])
            cumul += n
    _write_table(new_contig_list_file, ["contig", "length_kb", "n_frags", "cumul_length"], crows_out)
    o2n0 = np.where(old_2_new > 0, old_2_new - 1, -1)
    return float(thresh), o2n0


def remap_coo(coo, old_2_new, skip_first=False):
    """Contacts of the next level: ids through old_2_new (0-based, -1 = destroyed), duplicates summed.  skip_first
    reproduces the reference's off-by-one that drops the first contact line (see the module docstring)."""
    a, b, v = (np.asarray(x) for x in coo)
    if skip_first:
        a, b, v = a[1:], b[1:], v[1:]
    na, nb = old_2_new[a], old_2_new[b]
    keep = (na >= 0) & (nb >= 0)
    return _coo_sum(na[keep], nb[keep], v[keep])


# ----------------------------------------------------------------------------------------- sub-sampling
def subsample_data_set(contig_info, fragments_list, factor, new_contig_list_file, new_fragments_list_file, old_2_new_file,
                       min_bin_per_contig=1):
    """``subsample_data_set`` without the contact part (see ``remap_coo``): bins of `factor` consecutive fragments per
    contig.  Returns (nfrags, old_2_new 0-based)."""
    _, crows = _read_table(contig_info)
    _, frows = _read_table(fragments_list)
    old_2_new = np.zeros(len(frows), dtype=np.int64)
    spec, new_contigs = {}, []
    new_abs = id_abs = 0
    for c in crows:
        n_frag = int(c[2])
        cond = (n_frag / np.float32(factor)) >= min_bin_per_contig and factor != 1
        new_rel = 0
        for rel in range(1, n_frag + 1):
            id_abs += 1
            if (not cond) or rel % factor == 1:
                new_abs += 1
                new_rel += 1
                spec[new_abs] = {"frag_start": id_abs, "id_rel": new_rel, "init_contig": c[0], "gc": []}
            spec[new_abs]["frag_end"] = id_abs
            old_2_new[id_abs - 1] = new_abs
        new_contigs.append([c[0], c[1], new_rel, new_abs - new_rel])
    for id_abs, r in enumerate(frows, start=1):
        s = spec[old_2_new[id_abs - 1]]
        s["gc"].append(float(r[5]))
        if id_abs == s["frag_start"]:
            s["start_pos"], s["init_frag_start"] = int(r[2]), int(r[7])
        if id_abs == s["frag_end"]:
            s["end_pos"], s["init_frag_end"] = int(r[3]), int(r[8])
    out = []
    for k in sorted(spec):
        s = spec[k]
        out.append([s["id_rel"], s["init_contig"], s["start_pos"], s["end_pos"], s["end_pos"] - s["start_pos"],
                    float(np.array(s["gc"]).mean()), s["init_frag_end"] - s["init_frag_start"] + 1, s["init_frag_start"],
                    s["init_frag_end"], s["frag_start"], s["frag_end"]])
    _write_table(new_contig_list_file, ["contig", "length_kb", "n_frags", "cumul_length"], new_contigs)
    _write_table(new_fragments_list_file, ["id", "chrom", "start_pos", "end_pos", "size", "gc_content", "accu_frag",
                                           "init_frag_start", "init_frag_end", "sub_frag_start", "sub_frag_end"], out)
    _write_table(old_2_new_file, ["current_id", "super_id"], [[i + 1, int(old_2_new[i])] for i in range(len(old_2_new))])
    return len(out), old_2_new - 1


# ----------------------------------------------------------------------------------------- the whole build
def build_and_filter(base_folder, size_pyramid, factor, drop_first_contact=True):
    """``build_and_filter``: 3-file text dataset (``info_contigs.txt``, ``fragments_list.txt``,
    ``abs_fragments_contacts_weighted.txt``, ``README.md:111-113``) -> filtered level 0 -> `size_pyramid` levels of
    bins of `factor`.  Returns the :class:`Pyramid`."""
    root = os.path.join(base_folder, "pyramids", "pyramid_%d_thresh_auto" % size_pyramid)
    os.makedirs(root, exist_ok=True)
    raw = os.path.join(base_folder, "pyramids", "pyramid_1_no_thresh", "level_0")
    os.makedirs(raw, exist_ok=True)
    coo0 = abs_contacts_to_coo(os.path.join(base_folder, "abs_fragments_contacts_weighted.txt"))
    raw_frags = os.path.join(raw, "0_fragments_list.txt")
    n0 = init_frag_list(os.path.join(base_folder, "fragments_list.txt"), raw_frags)
    raw_contigs = os.path.join(raw, "0_contig_info.txt")
    _write_table(raw_contigs, *_read_table(os.path.join(base_folder, "info_contigs.txt")))
    write_coo(os.path.join(raw, "0_abs_frag_contacts.txt"), coo0)
    lev = os.path.join(root, "level_0")
    os.makedirs(lev, exist_ok=True)
    cur_contigs, cur_frags = os.path.join(lev, "0_contig_info.txt"), os.path.join(lev, "0_fragments_list.txt")
    thresh, o2n = remove_problematic_fragments(raw_contigs, raw_frags, coo0, n0, cur_contigs, cur_frags)
    # (remove_problematic_fragments reads the contacts with readlines(): no line is dropped there, pyramid_sparse.py:806-812)
    coo = remap_coo(coo0, o2n)
    write_coo(os.path.join(lev, "0_abs_frag_contacts.txt"), coo)
    store = {"0/data": np.stack(coo), "0/nfrags": np.int32(len(_read_table(cur_frags)[1])), "thresh": np.float64(thresh)}
    for level in range(1, size_pyramid):
        prev_dir, lev = lev, os.path.join(root, "level_%d" % level)
        os.makedirs(lev, exist_ok=True)
        new_contigs, new_frags = os.path.join(lev, "%d_contig_info.txt" % level), os.path.join(lev, "%d_fragments_list.txt" % level)
        nfr, o2n = subsample_data_set(cur_contigs, cur_frags, factor, new_contigs, new_frags,
                                      os.path.join(prev_dir, "%d_sub_2_super_index_frag.txt" % (level - 1)))
        coo = remap_coo(coo, o2n, skip_first=drop_first_contact)
        write_coo(os.path.join(lev, "%d_abs_frag_contacts.txt" % level), coo)
        store["%d/data" % level], store["%d/nfrags" % level] = np.stack(coo), np.int32(nfr)
        cur_contigs, cur_frags = new_contigs, new_frags
    np.savez(os.path.join(root, "pyramid.npz"), **store)
    return Pyramid(root, size_pyramid)


def write_hdf5(pyramid_folder, n_levels):
    """Export ``pyramid.npz`` as the reference's ``pyramid.hdf5`` (``pyramid_sparse.py:83-126, 313-322``); needs h5py."""
    import h5py
    data = dict(np.load(os.path.join(pyramid_folder, "pyramid.npz")))
    with h5py.File(os.path.join(pyramid_folder, "pyramid.hdf5"), "w") as f:
        for level in range(n_levels):
            d = np.asarray(data["%d/data" % level], dtype=np.int32)
            g = f.create_group(str(level))
            ds = g.create_dataset("data", d.shape, "i")
            ds[...] = d
            nf = g.create_dataset("nfrags", (1, 1), "i")
            nf[:] = int(data["%d/nfrags" % level])
            f.attrs[str(level)] = "done"


# ----------------------------------------------------------------------------------------- reading a pyramid
class Pyramid:
    """``pyramid`` (``pyramid_sparse.py:896-1034``): per level the fragment table and the sub <-> super bin indices."""

    def __init__(self, pyramid_folder, n_levels):
        self.pyramid_folder, self.n_levels = pyramid_folder, n_levels
        npz, h5 = os.path.join(pyramid_folder, "pyramid.npz"), os.path.join(pyramid_folder, "pyramid.hdf5")
        if os.path.exists(npz):
            self.data = dict(np.load(npz))
        elif os.path.exists(h5):   # a pyramid built by the reference (pyramid_sparse.py:904): h5py if present, else graal_amd/hdf5_min.py
            from . import hdf5_min
            self.data = {}
            for lev, (d, nf) in hdf5_min.read_pyramid_levels(h5).items():
                self.data["%d/data" % lev], self.data["%d/nfrags" % lev] = d, np.int32(nf)
        else:
            raise FileNotFoundError("neither pyramid.npz nor pyramid.hdf5 in %s" % pyramid_folder)
        self.resol_F_s_kb = 3
        self.dist_max_kb = 30 * 2 * self.resol_F_s_kb
        self.spec_level = {}
        for i in range(n_levels):
            folder = os.path.join(pyramid_folder, "level_%d" % i)
            _, rows = _read_table(os.path.join(folder, "%d_fragments_list.txt" % i))
            frags, contigs = [], {}
            for k, r in enumerate(rows, start=1):
                f = {"np_id_abs": k, "curr_id": int(r[0]), "init_contig": r[1], "start_pos": int(r[2]), "end_pos": int(r[3]),
                     "size": int(r[4]), "gc_content": float(r[5]), "n_accu_frags": int(r[6]),
                     "sub_low_index": int(r[9]) if i > 0 else int(r[0]), "sub_high_index": int(r[10]) if i > 0 else int(r[0]),
                     "super_index": int(r[0])}
                if r[1] not in contigs:
                    contigs[r[1]] = len(contigs) + 1
                f["contig_id"] = contigs[r[1]]
                frags.append(f)
            self.spec_level[str(i)] = {"fragments": frags, "contig_ids": contigs, "level_folder": folder}
            if i == 0:
                self.list_contigs_name = list(contigs)
            sup = os.path.join(folder, "%d_sub_2_super_index_frag.txt" % i)
            if i < n_levels - 1 and os.path.exists(sup):
                for r in _read_table(sup)[1]:
                    frags[int(r[0]) - 1]["super_index"] = int(r[1])

    def level_coo(self, level):
        d = self.data["%d/data" % level]
        return d[0].astype(np.int32), d[1].astype(np.int32), d[2].astype(np.int32)

    def get_level(self, level):
        return Level(self, level)

    # ---- sequences (f4)
    def load_reference_sequence(self, genome_fasta, strict_reference=True):
        """``load_reference_sequence``.  strict_reference keeps the reference's handling of the last record (its last line
        is dropped and its newlines stay inside the sequence); False reads the FASTA file properly."""
        with open(genome_fasta) as f:
            lines = f.readlines()
        seqs, names, start, name = {}, [], 1, lines[0][1:-1]
        for i in range(1, len(lines)):
            if lines[i][0] == ">":
                names.append(name)
                seqs[name] = "".join(lines[start:i])
                start, name = i + 1, lines[i][1:-1]
        seqs[name] = "".join(lines[start:-1] if strict_reference else lines[start:])
        for n in names if strict_reference else names + [name]:
            seqs[n] = seqs[n].replace("\n", "")
        self.dict_sequence_contigs = seqs
        return seqs


class Level:
    """``level.load_data`` (``pyramid_sparse.py:1206-1380``): the initial S_o_A_frags of a level (one contig per input
    contig, fragments in file order) and ``mean_value_trans``."""

    def __init__(self, pyramid, level):
        self.pyramid, self.level = pyramid, level
        frags = pyramid.spec_level[str(level)]["fragments"]
        self.n_frags = int(pyramid.data["%d/nfrags" % level])
        n = len(frags)
        keys = ("pos", "id_c", "start_bp", "len_bp", "circ", "id", "prev", "next", "l_cont", "sub_l_cont", "l_cont_bp", "n_accu")
        S = {k: np.zeros(n, dtype=np.int32) for k in keys}
        self.frags_init_contigs = [f["init_contig"] for f in frags]
        by_contig = {}
        for f in frags:
            by_contig.setdefault(f["contig_id"], []).append(f)
        sub = pyramid.spec_level[str(level - 1)]["fragments"] if level > 0 else frags
        sub_count = {}
        for f in sub:
            sub_count[f["contig_id"]] = sub_count.get(f["contig_id"], 0) + 1
        self.coord_cont = {}
        for cid in sorted(by_contig):
            cf = by_contig[cid]
            total = sum(f["size"] for f in cf)
            prev = -1
            for j, f in enumerate(cf):
                i = f["np_id_abs"] - 1
                S["pos"][i] = f["curr_id"] - 1
                S["id_c"][i] = f["contig_id"]
                S["start_bp"][i] = f["start_pos"]
                S["len_bp"][i] = f["size"]
                S["id"][i] = i
                S["prev"][i] = prev
                S["next"][i] = cf[j + 1]["np_id_abs"] - 1 if j + 1 < len(cf) else -1
                S["l_cont"][i] = len(cf)
                S["sub_l_cont"][i] = sub_count.get(cid, len(cf))
                S["l_cont_bp"][i] = total
                S["n_accu"][i] = f["n_accu_frags"]
                prev = i
            self.coord_cont[cid] = [f["np_id_abs"] - 1 for f in cf]
        self.S_o_A_frags = S
        self.n_contigs = len(by_contig)
        self.coo = pyramid.level_coo(level)
        # mean_value_trans (:1345-1366): the stored (upper) matrix, rows of a contig vs. columns outside it
        a, b, v = self.coo
        ca, cb = S["id_c"][a], S["id_c"][b]
        total_trans = int(v[ca != cb].sum())
        sizes = np.array([len(x) for x in self.coord_cont.values()], dtype=np.int64)
        n_tot = int((sizes * n).sum() - (sizes * sizes).sum())
        # (numpy int64 / float32 -> float64 in the reference's numpy; the divisor is rounded to float32 first)
        self.mean_value_trans = np.float64(total_trans) / np.float64(np.float32(n_tot)) if n_tot else np.float64(0)

    # ---- f4
    def generate_new_fasta(self, vect_frags, new_fasta, info_frags):
        """``generate_new_fasta`` (:1430-1488): the contigs of a fragment layout as sequences (61 columns per line)."""
        comp = str.maketrans("TAGCtagc", "ATCGATCG")
        frags = self.pyramid.spec_level[str(self.level)]["fragments"]
        seqs = self.pyramid.dict_sequence_contigs
        id_c, pos, ori, activ, id_d = (np.asarray(getattr(vect_frags, k)) for k in ("id_c", "pos", "ori", "activ", "id_d"))
        with open(new_fasta, "w") as hf, open(info_frags, "w") as hi:
            done = []
            for c in np.unique(id_c):
                lf = np.nonzero(id_c == c)[0]
                if not np.all(activ[lf] == 1):
                    continue
                header = ">3C-assembly|contig_" + str(c)
                hi.write(header + "\n")
                hi.write("init_contig\tid_frag\torientation\tstart\tend\n")
                seq = ""
                for f in lf[np.argsort(pos[lf], kind="stable")]:
                    fr = frags[int(id_d[f])]
                    s = seqs[fr["init_contig"]][fr["start_pos"]:fr["end_pos"]]
                    if ori[f] == -1:
                        s = s[::-1].translate(comp)
                    hi.write("%s\t%s\t%s\t%s\t%s\n" % (fr["init_contig"], int(id_d[f]), int(ori[f]), fr["start_pos"], fr["end_pos"]))
                    seq += s
                done.append((header, seq))
            for header, seq in done:
                hf.write(header + "\n")
                cuts = list(range(0, len(seq), 61))
                for k in range(1, len(cuts)):
                    hf.write(seq[cuts[k - 1]:cuts[k]] + "\n")
                if cuts and cuts[-1] != len(seq) - 1:
                    hf.write(seq[cuts[-1]:] + "\n")


# ----------------------------------------------------------------------------------------- sampler inputs
def select_repeated_frags(level_coo, n_frags, allow_repeats):
    """``select_repeated_frags`` (``simulation_loader.py:369-394``): bins whose coverage (row + column sum of the stored
    matrix) exceeds mean + 3 sigma, with ``max(1, round(cov / (mean + 3 sigma)) - 1)`` extra copies each."""
    a, b, v = level_coo
    cov = np.zeros(int(n_frags), dtype=np.float64)
    np.add.at(cov, a, v); np.add.at(cov, b, v)
    ext = cov.mean() + 3 * cov.std()
    cand = np.nonzero(cov > ext)[0] if allow_repeats else np.zeros(0, dtype=np.int64)
    return [(int(e), int(max(1, np.round(cov[e] / ext) - 1))) for e in cand]


def simulation_inputs(pyramid, level, candidates_blacklist=(0,), allow_repeats=False, level0_accu_from_file=False):
    """What ``simulation.__init__`` hands to the sampler constructor (``simulation_loader.py:41-107``), from level `level`
    (bins) and level - 1 (sub-fragments = observations), with the contact matrices as COO triples; with ``allow_repeats``
    the high-coverage bins get extra copies (``modify_vect_frags``, ``simulation_loader.py:182-280``).  Returns a dict whose
    keys are the constructor's argument names.

    ``level == 0`` -- BASELINE.json's "pyramid level 0 (full restriction-fragment resolution)", SURVEY.md section 8d "C4" -- is what the
    reference cannot run (``simulation_loader.py:45,68`` needs level - 1, the GUI offers levels 1.. ``main_window.py:452``, and the dense
    pixel index stops at ~4,500 bins): the loader of ``pyramid_sparse.py:1206-1380`` generalised so that the bins ARE the level-0
    fragments.  Every bin is its own single sub-fragment (``np_sub_frags_id[i] = (i, 0, 0, 1)``, length ``len_bp / 1000`` kb), counts
    as ONE restriction fragment (``accu = 1``, ``mean_squared_frags_per_bin = 1.0``: the pixel of two bins is their contact count, no
    area normalisation; ``level0_accu_from_file=True`` takes the filter's ``accu_frag`` column instead -- a merged bin then weighs as the
    fragments it swallowed, ``simulation_loader.py:695``), both contact matrices are the level-0 COO list, ``mean_value_trans`` and the
    Rippe fit's histogram come from level 0 too."""
    lev = pyramid.get_level(level)
    sub = pyramid.get_level(level - 1) if level > 0 else lev
    frags = pyramid.spec_level[str(level)]["fragments"]
    n = lev.n_frags
    ids, lens, accu, collect = np.zeros((n, 4), np.int32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.int32), []
    n_sub_total = 0
    for i, f in enumerate(frags):               # create_sub_frags, simulation_loader.py:673-704
        lo, hi = (f["sub_low_index"] - 1, f["sub_high_index"] - 1) if level > 0 else (i, i)
        ns = hi - lo + 1
        ids[i, 3] = ns
        n_sub_total += ns
        for j in range(ns):
            lens[i, j] = np.float32(sub.S_o_A_frags["len_bp"][lo + j]) / np.float32(1000.0)
            ids[i, j] = lo + j
            accu[i, j] = sub.S_o_A_frags["n_accu"][lo + j] if (level > 0 or level0_accu_from_file) else 1
            collect.append(accu[i, j])
    S = {k: np.array(lev.S_o_A_frags[k], dtype=np.int32) for k in ("pos", "id_c", "start_bp", "len_bp", "circ", "id", "prev",
                                                                     "next", "l_cont", "l_cont_bp")}
    S["ori"] = np.ones(n, np.int32); S["rep"] = np.zeros(n, np.int32); S["activ"] = np.ones(n, np.int32)
    S["id_d"] = np.arange(n, dtype=np.int32)
    # modify_vect_frags: every copy a singleton contig of its own with a fresh label, rep = 1, activ = 1, id_d = the bin
    dups = select_repeated_frags(lev.coo, n, allow_repeats)
    max_f, max_c = n, int(S["id_c"].max()) + 1
    add = {k: [] for k in S}
    for b, n_dup in dups:
        for _ in range(n_dup):
            row = dict(pos=0, id_c=max_c, start_bp=0, len_bp=int(S["len_bp"][b]), circ=int(S["circ"][b]), id=max_f, prev=-1,
                       next=-1, l_cont=1, l_cont_bp=int(S["len_bp"][b]), ori=1, rep=1, activ=1, id_d=b)
            for k in add:
                add[k].append(row[k])
            max_f += 1; max_c += 1
    if dups:
        S = {k: np.concatenate([S[k], np.asarray(add[k], dtype=np.int32)]) for k in S}
    dup_bins = [b for b, _ in dups]
    collector, dispatcher = [], []
    for b in range(n):
        copies = np.nonzero(S["id_d"] == b)[0] if b in dup_bins else [b]
        dispatcher.append((len(collector), len(collector) + len(copies)))
        collector.extend(int(i) for i in copies)
    black = [] if list(candidates_blacklist) == [0] else list(candidates_blacklist)
    frag_blacklisted = []
    for c in black:                                                   # blacklist_contig, :129-159 (all copies of its bins)
        for f in np.nonzero(lev.S_o_A_frags["id_c"] == c)[0]:
            frag_blacklisted.extend(collector[dispatcher[f][0]:dispatcher[f][1]])
    return dict(
        S_o_A_frags=S, collector_id_repeats=np.asarray(collector, dtype=np.int32),
        frag_dispatcher=np.asarray(dispatcher, dtype=np.int32),
        id_frag_duplicated=dup_bins, id_frags_blacklisted=frag_blacklisted, n_frags=n, n_new_frags=max_f,
        init_n_sub_frags=n_sub_total, n_new_sub_frags=n_sub_total, np_rep_sub_frags_id=None,
        hic_matrix_sub_sampled=lev.coo, np_sub_frags_len_bp=lens, np_sub_frags_id=ids, np_sub_frags_accu=accu,
        mean_squared_frags_per_bin=np.float32(np.float32(collect).mean() ** 2), norm_vect_accu=accu.sum(axis=1),
        S_o_A_sub_frags=sub.S_o_A_frags, hic_matrix=sub.coo, mean_value_trans=sub.mean_value_trans)
