#!/usr/bin/env python3
"""Run the REFERENCE's pyramid builder (pyramid_sparse.build_and_filter + pyramid / level loaders) on a toy dataset and store what
it produced: tests/golden/pyramid_ref.json.  Pins SURVEY.md section 8 row f3 with the reference itself.

Needs an interpreter with h5py (the reference keeps its contact matrices in HDF5); this image carries one:

    /opt/conda/bin/python3.9 tests/golden/make_pyramid_fixture.py

A temporary copy of pyramid_sparse.py, progressbar.py and fragment.py is converted from Python 2 with lib2to3 and imported from
the temporary directory (build container only; the reference's source never enters the repository -- the JSON of inputs and
outputs does).  The toy dataset is written by tests/test_pyramid.py:make_dataset with a fixed seed; graal_amd/pyramid.py must
reproduce every number from the same three text files (tests/test_pyramid.py::test_pyramid_matches_the_reference_run)."""
import importlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def main():
    import matplotlib
    matplotlib.use("Agg")
    tmp = tempfile.mkdtemp(prefix="graal_ref_pyr_")
    for name in ("pyramid_sparse.py", "progressbar.py", "fragment.py", "terminal_progress.py"):
        if os.path.exists(os.path.join(REF, name)):
            shutil.copy(os.path.join(REF, name), os.path.join(tmp, name))
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", tmp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # h5py 2.x (the reference's, requirements.txt) opened h5py.File(name) in mode 'a' (create if missing); h5py 3 defaults to 'r'.
    # Give the module the default it was written for -- the library itself is the real one.
    import h5py
    _File = h5py.File
    h5py.File = lambda name, mode="a", **kw: _File(name, mode, **kw)
    sys.path.insert(0, tmp)
    ref = importlib.import_module("pyramid_sparse")
    sys.path.remove(tmp)
    # scipy 1.0 (the reference's) accepted one-element arrays as matrix dimensions (the reference passes `nfrags[0]` of a (1, 1)
    # dataset, i.e. array([n])); scipy >= 1.3 insists on integers.  Coerce the shape, nothing else.
    import scipy.sparse as _sp

    def _int_shape(cls):
        class _C(cls):
            def __init__(self, *a, **kw):
                if kw.get("shape") is not None:
                    kw["shape"] = tuple(int(np.asarray(x).ravel()[0]) for x in kw["shape"])
                cls.__init__(self, *a, **kw)
        _C.__name__ = cls.__name__
        return _C
    ref.sp = type(_sp)("sp_shim")
    ref.sp.__dict__.update(_sp.__dict__)
    for name in ("csr_matrix", "csc_matrix", "coo_matrix", "lil_matrix"):
        setattr(ref.sp, name, _int_shape(getattr(_sp, name)))
    # numpy 1.13 (the reference's) let a one-element integer array stand for an integer (range(), shapes); numpy >= 1.2x does
    # not.  The level loader keeps `nfrags` as such an array (np.copy(data['nfrags'][0]), pyramid_sparse.py:1216) and uses it
    # both ways (range(0, n_frags) and n_frags[0]): give the module's np.copy a result that still answers __index__.
    class _I1(np.ndarray):
        def __index__(self):
            return int(np.asarray(self).ravel()[0])

    _np = type(np)("np_shim")
    _np.__dict__.update(np.__dict__)

    def _copy(a, *args, **kw):
        r = np.copy(a, *args, **kw)
        return r.view(_I1) if isinstance(r, np.ndarray) and r.size == 1 and r.dtype.kind in "iu" else r
    _np.copy = _copy
    ref.np = _np

    class _NoBar(object):       # (terminal progress bar: display only, and its integer arithmetic is Python 2's)
        def __init__(self, *a, **kw): pass
        def render(self, *a, **kw): pass
    ref.ProgressBar = _NoBar
    # ---- the toy dataset (same generator and seed as the test)
    from tests.test_pyramid import make_dataset
    work = tempfile.mkdtemp(prefix="graal_ref_ds_")
    base = os.path.join(work, "ds")
    make_dataset(base, np.random.RandomState(2014), contig_sizes=(23, 16, 12), n_pairs=30000, empty=(4, 30))
    size_pyramid, factor = 3, 3
    stdout = sys.stdout
    sys.stdout = open(os.devnull, "w")          # (the reference prints progress bars)
    try:
        # The reference's build_and_filter (pyramid_sparse.py:25-136) converts the level-0 contact file a SECOND time with
        # abs_contact_2_coo_file (:44-45) -- on a file that build() (:147-148) has already converted -- which shifts every id by
        # one, resets every count to 1 and raises KeyError as soon as fragment 0 has a contact (it does here): run from scratch
        # on a raw dataset it does not finish.  Its FUNCTIONS are run here in build_and_filter's order, each on the input its
        # own code expects (0-based COO with counts), i.e. without that second conversion.
        ref.build(base, 1, factor, 1)                                  # -> pyramids/pyramid_1_no_thresh (level 0 + pyramid.hdf5)
        init = os.path.join(base, "pyramids", "pyramid_1_no_thresh")
        l0 = os.path.join(init, "level_0")
        root = os.path.join(base, "pyramids", "pyramid_%d_thresh_auto" % size_pyramid)
        os.makedirs(os.path.join(root, "level_0"))
        cur_contigs = os.path.join(root, "level_0", "0_contig_info.txt")
        cur_frags = os.path.join(root, "level_0", "0_fragments_list.txt")
        cur_coo = os.path.join(root, "level_0", "0_abs_frag_contacts.txt")
        p0 = h5py.File(os.path.join(init, "pyramid.hdf5"))
        thresh = ref.remove_problematic_fragments(os.path.join(l0, "0_contig_info.txt"), os.path.join(l0, "0_fragments_list.txt"),
                                                  os.path.join(l0, "0_abs_frag_contacts.txt"), cur_contigs, cur_frags, cur_coo, p0)
        p0.close()
        handle = h5py.File(os.path.join(root, "pyramid.hdf5"))
        nfrags = ref.file_len(cur_frags) - 1
        ref.fill_sparse_pyramid_level(handle, 0, cur_coo, nfrags)
        handle.attrs["0"] = "done"
        sup = os.path.join(root, "level_0", "0_sub_2_super_index_frag.txt")
        for lev in range(1, size_pyramid):
            folder = os.path.join(root, "level_%d" % lev)
            os.makedirs(folder)
            new_contigs = os.path.join(folder, "%d_contig_info.txt" % lev)
            new_frags = os.path.join(folder, "%d_fragments_list.txt" % lev)
            new_coo = os.path.join(folder, "%d_abs_frag_contacts.txt" % lev)
            nfrags = ref.subsample_data_set(cur_contigs, cur_frags, factor, cur_coo, new_coo, 1, new_contigs, new_frags, sup)
            ref.fill_sparse_pyramid_level(handle, lev, new_coo, nfrags)
            handle.attrs[str(lev)] = "done"
            cur_contigs, cur_frags, cur_coo = new_contigs, new_frags, new_coo
            sup = os.path.join(folder, "%d_sub_2_super_index_frag.txt" % lev)
        handle.close()
        P = ref.pyramid(root, size_pyramid)
        out = {"thresh": float(thresh), "generated_by": "tests/golden/make_pyramid_fixture.py from /root/reference/pyramid_sparse.py (lib2to3 copy)",
               "dataset": {"seed": 2014, "contig_sizes": [23, 16, 12], "n_pairs": 30000, "empty": [4, 30]},
               "size_pyramid": size_pyramid, "factor": factor, "levels": {}}
        for lev in range(size_pyramid):
            d = np.array(P.data[str(lev)]["data"])
            folder = os.path.join(root, "level_%d" % lev)
            frag_file = os.path.join(folder, "%d_fragments_list.txt" % lev)
            contig_file = os.path.join(folder, "%d_contig_info.txt" % lev)
            entry = {"nfrags": int(np.array(P.data[str(lev)]["nfrags"]).ravel()[0]), "data": d.astype(int).tolist(),
                     "fragments_list": open(frag_file).read(), "contig_info": open(contig_file).read()}
            sup = os.path.join(folder, "%d_sub_2_super_index_frag.txt" % lev)
            if os.path.exists(sup):
                entry["sub_2_super"] = open(sup).read()
            out["levels"][str(lev)] = entry
        # ---- the level object simulation_loader builds its inputs from (pyramid_sparse.py:1176-1380): S_o_A_frags, mean trans
        for lev in (1, 2):
            L = P.get_level(lev)
            # (level.__init__ has run load_data: S_o_A_frags, sparse matrices, mean_value_trans)
            soa = {k: np.asarray(v).astype(int).tolist() for k, v in L.S_o_A_frags.items()}
            out["levels"][str(lev)]["S_o_A_frags"] = soa
            out["levels"][str(lev)]["mean_value_trans"] = float(L.mean_value_trans)
    finally:
        sys.stdout = stdout
    with open(os.path.join(HERE, "pyramid_ref.json"), "w") as f:
        json.dump(out, f)
    shutil.rmtree(tmp, ignore_errors=True)
    shutil.rmtree(work, ignore_errors=True)
    print("wrote pyramid_ref.json", {k: (v["nfrags"], len(v["data"][0])) for k, v in out["levels"].items()})


if __name__ == "__main__":
    main()
