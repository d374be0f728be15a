"""CPU: the contact matrix in the genome's order as an image (graal_amd/image.py; display_current_matrix, cuda_lib_gl.py:1581-1624)."""
import numpy as np

from graal_amd import image, synth


def test_matrix_image_is_the_reordered_dense_matrix(tmp_path):
    P = synth.with_dense(synth.make_problem(n_bins=30, nnz=300, n_sub=3, seed=3, contig_weights=(5, 3, 2), mean_len_bp=1500.0, accu=9))
    S = P["init_n_sub_frags"]
    rng = np.random.RandomState(1)
    order = rng.permutation(S)[:S - 5]          # a few sub-fragments left out (an inactive copy's, in the reference's loop)
    want = P["hic_matrix"][np.ix_(order, order)].astype(np.float32)
    got = image.matrix_image((P["coo_row"], P["coo_col"], P["coo_val"]), order)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    path = str(tmp_path / "m.tiff")
    image.write_tiff_f32(path, got)
    assert np.array_equal(image.read_tiff_f32(path), want)
    try:
        from PIL import Image
    except ImportError:
        Image = None
    if Image is not None:                       # the reference's reader / writer, when present: the same pixels, mode F
        im = Image.open(path)
        assert im.mode == "F" and np.array_equal(np.asarray(im), want)


def test_matrix_image_bins_large_maps():
    P = synth.make_problem(n_bins=500, nnz=20000, n_sub=1, seed=4, contig_weights=(5, 3, 2), mean_len_bp=1500.0)
    order = np.arange(500)[::-1]
    img = image.matrix_image((P["coo_row"], P["coo_col"], P["coo_val"]), order, max_px=64)
    assert img.shape == (63, 63)                # blocks of ceil(500 / 64) = 8 fragments
    assert np.array_equal(img, img.T)
    assert float(img.sum()) == 2.0 * float(P["coo_val"].sum())
    full = image.matrix_image((P["coo_row"], P["coo_col"], P["coo_val"]), order, max_px=4096)
    assert full.shape == (500, 500) and float(full[:8, :8].sum()) == float(img[0, 0])
