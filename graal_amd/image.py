"""The contact matrix in a given fragment order as an image file -- what ``sampler.display_current_matrix(file)`` leaves behind in the
reference (``cuda_lib_gl.py:1581-1624``: ``Image.fromarray(hic_matrix[np.ix_(full_order_high, full_order_high)]).save(file)``: PIL's
32-bit float grayscale TIFF of the sub-level matrix, rows and columns in the current genome order; ``main_gl.py:213, 283`` write
``pre_simu.tiff`` / ``post_em.tiff`` with it).

From the COO list instead of a dense matrix: the order is applied to the contacts, never to S x S floats.  Up to ``max_px`` sub-fragments
the image is the reference's pixel for pixel; beyond that (level 0: tens of thousands of fragments, a 10 GB image in the reference's form)
rows and columns are binned by ``ceil(S / max_px)`` consecutive sub-fragments of the current order and the counts summed.  The TIFF writer
is self-contained (baseline TIFF 6.0, one uncompressed strip, SampleFormat = IEEE float): no PIL needed, any TIFF reader opens it."""
import struct

import numpy as np


def matrix_image(coo, order, max_px=2048):
    """float32 [n, n] image of the symmetric, zero-diagonal contact matrix with rows / columns ``order`` (sub-fragment ids, each at most
    once; ids not listed are left out), binned to at most max_px x max_px."""
    r, c, v = (np.asarray(x) for x in coo)
    order = np.asarray(order, dtype=np.int64)
    n = len(order)
    if n == 0:
        return np.zeros((0, 0), dtype=np.float32)
    size = int(max(order.max() + 1, r.max(initial=-1) + 1, c.max(initial=-1) + 1))
    rank = np.full(size, -1, dtype=np.int64)
    rank[order] = np.arange(n)
    b = max(1, -(-n // int(max_px)))
    m = -(-n // b)
    rr, cc = rank[r], rank[c]
    keep = (rr >= 0) & (cc >= 0) & (r != c)
    rr, cc, vv = rr[keep] // b, cc[keep] // b, np.asarray(v, dtype=np.float64)[keep]
    img = np.zeros(m * m, dtype=np.float64)
    np.add.at(img, rr * m + cc, vv)
    np.add.at(img, cc * m + rr, vv)      # (the list holds every pair once, row < col: the matrix is its symmetrisation, simulation_loader.py:81-82)
    return img.reshape(m, m).astype(np.float32)


def write_tiff_f32(path, img):
    """A float32 2-D array as an uncompressed grayscale TIFF (little endian, one strip)."""
    a = np.ascontiguousarray(img, dtype="<f4")
    h, w = a.shape
    data = a.tobytes()
    tags = [  # (tag, type, count, value): SHORT = 3, LONG = 4
        (256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, 32), (259, 3, 1, 1), (262, 3, 1, 1), (273, 4, 1, 8), (277, 3, 1, 1), (278, 4, 1, h),
        (279, 4, 1, len(data)), (339, 3, 1, 3)]
    ifd_off = 8 + len(data) + (len(data) & 1)
    with open(path, "wb") as f:
        f.write(b"II" + struct.pack("<HI", 42, ifd_off))
        f.write(data)
        if len(data) & 1:
            f.write(b"\0")
        f.write(struct.pack("<H", len(tags)))
        for tag, typ, cnt, val in tags:
            f.write(struct.pack("<HHI", tag, typ, cnt) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val)))
        f.write(struct.pack("<I", 0))


def read_tiff_f32(path):
    """Read back what write_tiff_f32 wrote (tests; any TIFF library reads the same)."""
    raw = open(path, "rb").read()
    assert raw[:2] == b"II" and struct.unpack("<H", raw[2:4])[0] == 42
    off = struct.unpack("<I", raw[4:8])[0]
    n = struct.unpack("<H", raw[off:off + 2])[0]
    t = {}
    for i in range(n):
        tag, typ, cnt = struct.unpack("<HHI", raw[off + 2 + 12 * i: off + 10 + 12 * i])
        val = raw[off + 10 + 12 * i: off + 14 + 12 * i]
        t[tag] = struct.unpack("<H", val[:2])[0] if typ == 3 else struct.unpack("<I", val)[0]
    assert t[258] == 32 and t[339] == 3 and t[259] == 1
    return np.frombuffer(raw, dtype="<f4", count=t[256] * t[257], offset=t[273]).reshape(t[257], t[256]).copy()
