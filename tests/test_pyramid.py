"""CPU: the pyramid builder / loader (graal_amd/pyramid.py) and the FASTA export (f4) on a small
hand-made dataset in the reference's 3-file text format.  The expectations are computed here independently, from the
definitions in pyramid_sparse.py (cited in graal_amd/pyramid.py), not by calling the module under test."""
import os

import numpy as np
import pytest

from graal_amd import pyramid as pyr


def make_dataset(tmp, rng, contig_sizes=(11, 7, 5), n_pairs=6000, empty=(4, 15), polymer_like=False):
    """info_contigs.txt / fragments_list.txt / abs_fragments_contacts_weighted.txt (README.md:111-113).
    polymer_like: index offsets of the contact pairs follow a power law (~ k^-1.3) plus a uniform background instead of a
    geometric law, so that the Rippe fit of the dataset lands on polymer-like parameters."""
    frag_rows, contig_rows, seqs = [], [], {}
    cum = 0
    for ci, nf in enumerate(contig_sizes):
        name = "chr%d" % (ci + 1)
        pos = 0
        for k in range(1, nf + 1):
            size = int(rng.randint(300, 3000))
            frag_rows.append([k, name, pos, pos + size, size, round(float(rng.rand()), 3)])
            pos += size
        contig_rows.append([name, pos, nf, cum])
        cum += nf
        seqs[name] = "".join(rng.choice(list("ACGT"), size=pos))
    n = cum
    os.makedirs(tmp, exist_ok=True)
    pyr._write_table(os.path.join(tmp, "info_contigs.txt"), ["contig", "length_kb", "n_frags", "cumul_length"], contig_rows)
    pyr._write_table(os.path.join(tmp, "fragments_list.txt"), ["id", "chrom", "start_pos", "end_pos", "size", "gc_content"], frag_rows)
    ok = np.setdiff1d(np.arange(n), empty)
    a = rng.choice(ok, size=n_pairs)
    if polymer_like:
        ks = np.arange(1, max(2, n // 2))
        pk = ks ** -1.3
        off = rng.choice(ks, size=n_pairs, p=pk / pk.sum())
        off = np.where(rng.random_sample(n_pairs) < 0.12, rng.randint(1, n, size=n_pairs), off)   # uniform background
    else:
        off = rng.geometric(0.35, size=n_pairs)
    b = np.clip(a + off * rng.choice([-1, 1], size=n_pairs), 0, n - 1)
    b = np.where(np.isin(b, empty), a, b)
    extra = [(e, e + 1) for e in empty]           # the "empty" fragments get one contact each: very sparse rows
    pairs = np.concatenate([np.stack([a, b], 1), np.array(extra)]) + 1
    with open(os.path.join(tmp, "abs_fragments_contacts_weighted.txt"), "w") as f:
        f.write("id_read_a\tid_read_b\tw\n")
        for x, y in pairs:
            f.write("%d\t%d\t1.0\n" % (x, y))
    with open(os.path.join(tmp, "genome.fa"), "w") as f:
        for name in seqs:
            f.write(">%s\n" % name)
            for i in range(0, len(seqs[name]), 70):
                f.write(seqs[name][i:i + 70] + "\n")
    return n, pairs - 1, frag_rows, seqs


def test_build_and_filter_and_levels(tmp_path):
    rng = np.random.RandomState(3)
    base = str(tmp_path / "ds")
    n, pairs, frag_rows, _ = make_dataset(base, rng)
    P = pyr.build_and_filter(base, 3, 3)
    # ---- level 0 before filtering: symmetrised counts
    dense = np.zeros((n, n), np.int64)
    for x, y in pairs:
        dense[min(x, y), max(x, y)] += 1
    a0, b0, v0 = pyr.read_coo(os.path.join(base, "pyramids", "pyramid_1_no_thresh", "level_0", "0_abs_frag_contacts.txt"))
    got = np.zeros_like(dense); got[a0, b0] = v0
    assert np.array_equal(got, dense)
    # ---- the filter: fraction of non-zero entries per row of M + M^T, threshold mean - 1.01 sigma (float32)
    full = dense + dense.T
    spars = np.float32((full != 0).sum(axis=1)) / np.float32(n)
    thresh = spars.mean() - 1.01 * spars.std()
    locked = set(np.nonzero(spars <= thresh)[0].tolist())
    assert locked == {4, 15}                       # the two starved fragments
    assert float(P.data["thresh"]) == pytest.approx(float(thresh))
    # a locked fragment is merged into the next unlocked one of its contig: fragment 4 -> bin of fragment 5, 15 -> 16
    o2n, new = {}, 0
    contig_of = [r[1] for r in frag_rows]
    run = []
    for i in range(n):
        run.append(i)
        if i not in locked:
            for j in run:
                o2n[j] = new
            new += 1; run = []
        if i + 1 == n or contig_of[i + 1] != contig_of[i]:
            run = []                               # (an open run at the end of a contig would be destroyed; none here)
    lv0 = P.get_level(0)
    assert lv0.n_frags == new == n - 2
    want = np.zeros((new, new), np.int64)
    for x, y in pairs:
        p, q = o2n[min(x, y)], o2n[max(x, y)]
        want[min(p, q), max(p, q)] += 1
    a, b, v = P.level_coo(0)
    got = np.zeros_like(want); got[a, b] = v
    assert np.array_equal(got, want)
    # merged bin: size and accu add up, start of the first, end of the last
    S = lv0.S_o_A_frags
    assert S["len_bp"][o2n[4]] == frag_rows[4][4] + frag_rows[5][4] and S["n_accu"][o2n[4]] == 2
    assert S["start_bp"][o2n[4]] == frag_rows[4][2]
    assert np.array_equal(S["l_cont"][[0, 10, 17]], [10, 6, 5])     # 11-1, 7-1, 5 fragments per contig
    assert np.array_equal(np.unique(S["id_c"]), [1, 2, 3]) and S["prev"][0] == -1 and S["next"][9] == -1 and S["next"][0] == 1
    # ---- level 1: bins of 3 consecutive level-0 fragments per contig; the reference drops the first contact line
    sizes0 = [10, 6, 5]
    sup, k, base0 = {}, 0, 0
    for s in sizes0:
        for r in range(s):
            sup[base0 + r] = k + r // 3
        k += (s + 2) // 3; base0 += s
    lv1 = P.get_level(1)
    assert lv1.n_frags == k == 4 + 2 + 2
    want1 = np.zeros((k, k), np.int64)
    for (x, y, c) in list(zip(a, b, v))[1:]:
        p, q = sup[x], sup[y]
        want1[min(p, q), max(p, q)] += c
    a1, b1, v1 = P.level_coo(1)
    got1 = np.zeros_like(want1); got1[a1, b1] = v1
    assert np.array_equal(got1, want1)
    frs = P.spec_level["1"]["fragments"]
    assert [f["sub_low_index"] for f in frs[:4]] == [1, 4, 7, 10] and frs[3]["sub_high_index"] == 10
    assert lv1.S_o_A_frags["n_accu"][0] == 3 and lv1.S_o_A_frags["l_cont"][0] == 4 and lv1.S_o_A_frags["sub_l_cont"][0] == 10
    assert P.spec_level["0"]["fragments"][4]["super_index"] == 2      # sub -> super index file
    # mean trans value: stored (upper) entries whose row and column lie in different contigs / #such ordered pairs
    cid = lv1.S_o_A_frags["id_c"]
    tot = sum(int(c) for x, y, c in zip(a1, b1, v1) if cid[x] != cid[y])
    n_tot = sum(s * k - s * s for s in (4, 2, 2))
    assert lv1.mean_value_trans == pytest.approx(tot / n_tot, rel=1e-6)


def test_simulation_inputs_feed_the_sampler_constructor(tmp_path):
    rng = np.random.RandomState(4)
    base = str(tmp_path / "ds")
    make_dataset(base, rng)
    P = pyr.build_and_filter(base, 3, 3)
    inp = pyr.simulation_inputs(P, 1, candidates_blacklist=[3])
    n = inp["n_frags"]
    assert n == 8 and inp["init_n_sub_frags"] == 21
    ids = inp["np_sub_frags_id"]
    assert np.array_equal(ids[0], [0, 1, 2, 3]) and np.array_equal(ids[3], [9, 0, 0, 1]) and np.array_equal(ids[4, :3], [10, 11, 12])
    sub = P.get_level(0)
    assert inp["np_sub_frags_len_bp"][0, 1] == np.float32(sub.S_o_A_frags["len_bp"][1]) / np.float32(1000.0)
    assert inp["np_sub_frags_accu"][1].tolist() == [1, 2, 1]          # the merged level-0 bin carries 2 restriction fragments
    assert inp["id_frags_blacklisted"] == [6, 7]                      # the bins of contig 3
    assert inp["mean_squared_frags_per_bin"] == np.float32(np.float32(sub.S_o_A_frags["n_accu"]).mean() ** 2)
    assert inp["hic_matrix"][0].max() < 21 and inp["hic_matrix_sub_sampled"][0].max() < 8
    # the host side of the sampler accepts them as they are (COO triples, nothing densified)
    from graal_amd.sampler import as_coo_upper, blacklist_fill, neighbour_distributions
    r, c, v = as_coo_upper(inp["hic_matrix_sub_sampled"])
    xk, pk = neighbour_distributions(r, c, v, n)
    assert xk.shape == (n, 8) and np.allclose(pk.sum(axis=1), 1.0)
    (sr, sc, sv), _ = blacklist_fill(as_coo_upper(inp["hic_matrix"]), (r, c, v), ids, [6, 7], inp["mean_value_trans"], 21)
    assert sv.dtype == np.float32 and len(sr) >= 5 * 20


def test_fasta_export(tmp_path):
    rng = np.random.RandomState(5)
    base = str(tmp_path / "ds")
    n, _, frag_rows, seqs = make_dataset(base, rng)
    P = pyr.build_and_filter(base, 2, 3)
    P.load_reference_sequence(os.path.join(base, "genome.fa"), strict_reference=False)
    assert P.dict_sequence_contigs == seqs
    lv = P.get_level(1)
    frs = P.spec_level["1"]["fragments"]

    class V:   # a layout: contig 7 = bins 1 (reversed) then 0; everything else a singleton
        pass
    m = lv.n_frags
    v = V()
    v.id_c = np.arange(m) + 10; v.pos = np.zeros(m, int); v.ori = np.ones(m, int); v.activ = np.ones(m, int); v.id_d = np.arange(m)
    v.id_c[0] = v.id_c[1] = 7; v.pos[0] = 1; v.ori[1] = -1
    out, info = str(tmp_path / "g.fa"), str(tmp_path / "info.txt")
    lv.generate_new_fasta(v, out, info)
    txt = open(out).read().split(">")[1:]
    assert txt[0].startswith("3C-assembly|contig_7\n")
    body = "".join(txt[0].split("\n")[1:])
    s0 = seqs["chr1"][frs[0]["start_pos"]:frs[0]["end_pos"]]
    s1 = seqs["chr1"][frs[1]["start_pos"]:frs[1]["end_pos"]]
    rc = s1[::-1].translate(str.maketrans("ACGT", "TGCA"))
    assert body == rc + s0
    assert all(len(line) == 61 for line in txt[0].split("\n")[1:-2])
    lines = open(info).read().split("\n")
    assert lines[0] == ">3C-assembly|contig_7" and lines[2].split("\t") == ["chr1", "1", "-1", str(frs[1]["start_pos"]), str(frs[1]["end_pos"])]
    # the reference's reader: last record keeps its newlines and loses its last line
    strict = P.load_reference_sequence(os.path.join(base, "genome.fa"), strict_reference=True)
    assert strict["chr1"] == seqs["chr1"] and "\n" in strict["chr3"] and len(strict["chr3"].replace("\n", "")) < len(seqs["chr3"])


def test_repeated_fragments_are_selected_by_coverage(tmp_path):
    rng = np.random.RandomState(6)
    base = str(tmp_path / "ds")
    n, pairs, _, _ = make_dataset(base, rng, contig_sizes=(40, 30, 20), n_pairs=20000, empty=(4, 45))
    # one hot fragment: many extra contacts with everybody (a repeated sequence)
    with open(os.path.join(base, "abs_fragments_contacts_weighted.txt"), "a") as f:
        for y in rng.randint(0, n, size=12000):
            if y != 33:
                f.write("%d\t%d\t1.0\n" % (34, y + 1))
    P = pyr.build_and_filter(base, 2, 3)
    lev = P.get_level(1)
    a, b, v = lev.coo
    cov = np.zeros(lev.n_frags); np.add.at(cov, a, v); np.add.at(cov, b, v)
    ext = cov.mean() + 3 * cov.std()
    hot = np.nonzero(cov > ext)[0]
    assert len(hot) == 1
    inp = pyr.simulation_inputs(P, 1, allow_repeats=True)
    k = int(max(1, np.round(cov[hot[0]] / ext) - 1))
    assert inp["id_frag_duplicated"] == [int(hot[0])] and inp["n_new_frags"] == inp["n_frags"] + k
    S = inp["S_o_A_frags"]
    new = np.arange(inp["n_frags"], inp["n_new_frags"])
    assert (S["rep"][new] == 1).all() and (S["id_d"][new] == hot[0]).all() and (S["l_cont"][new] == 1).all()
    assert len(np.unique(S["id_c"][new])) == k and S["id_c"][new].min() > lev.S_o_A_frags["id_c"].max()
    d = inp["frag_dispatcher"][hot[0]]
    assert sorted(inp["collector_id_repeats"][d[0]:d[1]].tolist()) == [int(hot[0])] + new.tolist()
    assert pyr.simulation_inputs(P, 1, allow_repeats=False)["id_frag_duplicated"] == []


# ---- the reference's pyramid.hdf5 (pyramid_sparse.py:83-126, 313-322, 904) without h5py -----------------------------------
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONDA_H5PY = "/opt/conda/bin/python3.9"     # this image carries a second interpreter that has h5py (the main one does not)


def test_pure_python_reader_reads_a_file_written_by_h5py():
    """tests/golden/pyramid_fixture.hdf5 was written by real h5py 3.3 / libhdf5 1.10.6 with the reference's own calls
    (generator: tests/golden/make_hdf5_fixture.py); graal_amd/hdf5_min.py must return the numbers stored beside it."""
    import json
    from graal_amd import hdf5_min as H
    f = H.File(os.path.join(GOLDEN, "pyramid_fixture.hdf5"))
    e = json.load(open(os.path.join(GOLDEN, "pyramid_fixture.json")))
    assert [k for k in f.keys() if k.isdigit()] == ["0", "1", "2"]
    assert all(f.attrs[str(k)] == "done" for k in range(3)) and f.attrs["thresh"] == 0.125
    for lev, d in e["levels"].items():
        got = f[lev]["data"].read()
        assert got.dtype == np.int32 and np.array_equal(got, np.array(d["data"]))
        assert int(f[lev + "/nfrags"].read()[0, 0]) == d["nfrags"]
    many = f["many"]                              # 40 links: more than one symbol-table node under the group's B-tree
    assert len(many.keys()) == 40
    for i in range(40):
        assert np.array_equal(many["d%02d" % i].read(), np.arange(i + 1) * (i + 1))
    assert np.array_equal(f["chunked_gzip_shuffle"].read(), np.array(e["chunked_gzip_shuffle"]))   # chunk B-tree, deflate, shuffle
    assert np.array_equal(f["big_endian"].read(), np.array(e["big_endian"]))
    assert np.allclose(f["float64"].read(), e["float64"]) and np.array_equal(f["never_written"].read(), np.zeros((4, 2)))
    lev = H.read_pyramid_levels(os.path.join(GOLDEN, "pyramid_fixture.hdf5"))
    assert sorted(lev) == [0, 1, 2] and lev[1][1] == e["levels"]["1"]["nfrags"]
    with pytest.raises(H.Hdf5Error):
        H.File(os.path.join(GOLDEN, "appendix_e.json"))      # not an HDF5 file: a loud error, not garbage


def test_pyramid_folder_with_the_references_hdf5_file(tmp_path):
    """A pyramid folder that holds pyramid.hdf5 (what the reference writes) instead of pyramid.npz loads to the same levels."""
    import shutil
    import subprocess
    if not os.path.exists(CONDA_H5PY) or subprocess.run([CONDA_H5PY, "-c", "import h5py"], capture_output=True).returncode != 0:
        pytest.skip("no interpreter with h5py on this box to write the file with")
    base = str(tmp_path / "ds")
    make_dataset(base, np.random.RandomState(5))
    P = pyr.build_and_filter(base, 3, 3)
    root = P.pyramid_folder
    script = ("import sys, numpy as np, h5py\\n"
              "root = sys.argv[1]; d = dict(np.load(root + '/pyramid.npz'))\\n"
              "f = h5py.File(root + '/pyramid.hdf5', 'w')\\n"
              "for level in range(3):\\n"
              "    g = f.create_group(str(level)); a = d['%d/data' % level]\\n"
              "    ds = g.create_dataset('data', a.shape, 'i'); ds[0, :] = a[0]; ds[1, :] = a[1]; ds[2, :] = a[2]\\n"
              "    nf = g.create_dataset('nfrags', (1, 1), 'i'); nf[:] = int(d['%d/nfrags' % level]); f.attrs[str(level)] = 'done'\\n"
              "f.close()\\n")
    subprocess.check_call([CONDA_H5PY, "-c", script.replace("\\n", "\n"), root], env={"PATH": "/usr/bin:/bin"})
    want = [P.level_coo(k) for k in range(3)]
    os.remove(os.path.join(root, "pyramid.npz"))
    P2 = pyr.Pyramid(root, 3)
    for k in range(3):
        for a, b in zip(P2.level_coo(k), want[k]):
            assert np.array_equal(a, b)
        assert int(P2.data["%d/nfrags" % k]) == int(P.data["%d/nfrags" % k])
    inp = pyr.simulation_inputs(P2, 1)
    assert inp["n_frags"] == int(P.data["1/nfrags"])


def test_pyramid_matches_the_reference_run(tmp_path):
    """tests/golden/pyramid_ref.json holds what the REFERENCE's pyramid_sparse.py produced for the toy dataset below (its functions
    run by tests/golden/make_pyramid_fixture.py under an interpreter with h5py: filtered level 0, two sub-sampled levels, the HDF5
    contact lists, every text file, and the level loader's S_o_A_frags / mean_value_trans).  graal_amd/pyramid.py must reproduce
    all of it from the same three files: contact lists and tables byte for byte."""
    import json
    e = json.load(open(os.path.join(GOLDEN, "pyramid_ref.json")))
    ds = e["dataset"]
    base = str(tmp_path / "ds")
    make_dataset(base, np.random.RandomState(ds["seed"]), contig_sizes=tuple(ds["contig_sizes"]), n_pairs=ds["n_pairs"],
                 empty=tuple(ds["empty"]))
    P = pyr.build_and_filter(base, e["size_pyramid"], e["factor"])
    assert float(P.data["thresh"]) == pytest.approx(e["thresh"], rel=1e-6)
    for lev in range(e["size_pyramid"]):
        r = e["levels"][str(lev)]
        assert int(P.data["%d/nfrags" % lev]) == r["nfrags"]
        assert np.array_equal(np.stack(P.level_coo(lev)), np.array(r["data"]))          # the (3, nnz) dataset of pyramid.hdf5
        folder = os.path.join(P.pyramid_folder, "level_%d" % lev)
        for key, name in (("fragments_list", "%d_fragments_list.txt"), ("contig_info", "%d_contig_info.txt"),
                          ("sub_2_super", "%d_sub_2_super_index_frag.txt")):
            if key in r:
                assert open(os.path.join(folder, name % lev)).read() == r[key], (lev, key)
        if "S_o_A_frags" in r:                                                            # level loader, pyramid_sparse.py:1206-1380
            L = P.get_level(lev)
            assert sorted(L.S_o_A_frags) == sorted(r["S_o_A_frags"])
            for k, v in r["S_o_A_frags"].items():
                assert np.array_equal(np.asarray(L.S_o_A_frags[k]).astype(int), np.array(v)), (lev, k)
            assert float(L.mean_value_trans) == pytest.approx(r["mean_value_trans"], rel=1e-12)


def test_level0_simulation_inputs(tmp_path):
    """simulation_inputs(pyramid, 0) -- BASELINE.json's "pyramid level 0 (full restriction-fragment resolution)", SURVEY 8d C4, which the
    reference cannot run (simulation_loader.py:45,68 needs level - 1): bins = the filtered level-0 fragments, one sub-fragment each,
    accu = 1, n_frags_per_bins = 1, both matrices the level-0 COO list, mean_value_trans and the fit's layout from level 0."""
    rng = np.random.RandomState(4)
    base = str(tmp_path / "ds")
    make_dataset(base, rng)
    P = pyr.build_and_filter(base, 1, 3)                  # a pyramid of ONE level is enough
    lev = P.get_level(0)
    inp = pyr.simulation_inputs(P, 0, candidates_blacklist=[3])
    n = inp["n_frags"]
    assert n == lev.n_frags == 21 == inp["n_new_frags"] == inp["init_n_sub_frags"] == inp["n_new_sub_frags"]
    assert np.array_equal(inp["np_sub_frags_id"], np.stack([np.arange(n), np.zeros(n), np.zeros(n), np.ones(n)], 1).astype(np.int32))
    assert np.array_equal(inp["np_sub_frags_len_bp"][:, 0], np.float32(lev.S_o_A_frags["len_bp"]) / np.float32(1000.0))
    assert not inp["np_sub_frags_len_bp"][:, 1:].any()
    assert np.array_equal(inp["np_sub_frags_accu"], np.stack([np.ones(n), np.zeros(n), np.zeros(n)], 1).astype(np.int32))
    assert inp["mean_squared_frags_per_bin"] == np.float32(1.0)
    for k in range(3):
        assert np.array_equal(inp["hic_matrix"][k], lev.coo[k]) and np.array_equal(inp["hic_matrix_sub_sampled"][k], lev.coo[k])
    assert inp["mean_value_trans"] == lev.mean_value_trans
    for k in lev.S_o_A_frags:
        assert np.array_equal(inp["S_o_A_sub_frags"][k], lev.S_o_A_frags[k]), k
    assert np.array_equal(inp["S_o_A_frags"]["id_d"], np.arange(n)) and (inp["S_o_A_frags"]["ori"] == 1).all()
    assert inp["id_frags_blacklisted"] == list(np.nonzero(lev.S_o_A_frags["id_c"] == 3)[0])
    assert np.array_equal(inp["frag_dispatcher"], np.stack([np.arange(n), np.arange(n) + 1], 1))
    # the merged level-0 bin (two restriction fragments swallowed by the sparsity filter) weighs as one bin by default, as the
    # fragments it holds with level0_accu_from_file (simulation_loader.py:695 at the levels above)
    inp2 = pyr.simulation_inputs(P, 0, level0_accu_from_file=True)
    assert np.array_equal(inp2["np_sub_frags_accu"][:, 0], lev.S_o_A_frags["n_accu"]) and inp2["np_sub_frags_accu"][:, 0].max() == 2
    assert inp2["mean_squared_frags_per_bin"] == np.float32(np.float32(lev.S_o_A_frags["n_accu"]).mean() ** 2)
    # the Rippe histogram comes from the same COO list / layout (cuda_lib_gl.py:1236-1270)
    from graal_amd import rippe_fit
    bins = np.arange(1.0, 12.0, 1.0)
    mc = rippe_fit.mean_contacts_per_bin(inp["S_o_A_sub_frags"], inp["hic_matrix"], bins, 11.0, 1.0)
    assert mc.shape == bins.shape and (mc > 0).all() and mc[0] > mc[-1]


def test_synthetic_problem_as_a_three_file_dataset(tmp_path):
    """synth.write_dataset: a synthetic level-0 problem written as the reference's text dataset and read back through the pyramid
    builder (one line per read pair: counts come back as the numbers of lines) -- tools/run_configs.py feeds C4 this way."""
    from graal_amd import synth
    P = synth.make_problem(n_bins=300, nnz=12000, n_sub=1, seed=5, contig_weights=(5, 3, 2), mean_len_bp=1500.0)
    base = str(tmp_path / "ds")
    n_reads = synth.write_dataset(P, base)
    assert n_reads == int(P["coo_val"].sum())
    r, c, v = pyr.abs_contacts_to_coo(os.path.join(base, "abs_fragments_contacts_weighted.txt"))
    assert np.array_equal(r, P["coo_row"]) and np.array_equal(c, P["coo_col"]) and np.array_equal(v, P["coo_val"])
    py = pyr.build_and_filter(base, 1, 3)
    inp = pyr.simulation_inputs(py, 0)
    kept = inp["n_frags"]
    assert 0.7 * 300 <= kept <= 300                      # the sparsity filter merges / drops the sparsest rows
    assert inp["S_o_A_frags"]["len_bp"].sum() <= P["S_o_A_frags"]["len_bp"].sum()
    assert int(inp["hic_matrix"][2].sum()) <= n_reads
