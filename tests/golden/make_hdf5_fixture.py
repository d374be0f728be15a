#!/usr/bin/env python3
"""Write tests/golden/pyramid_fixture.hdf5 (+ pyramid_fixture.json with the expected numbers) with REAL h5py, using the calls
the reference uses for its pyramid file (pyramid_sparse.py:83-126, 313-322): h5py.File(path), create_group(str(level)),
create_dataset('data', (3, n), 'i'), create_dataset('nfrags', (1, 1), 'i'), attrs[str(level)] = "done".

h5py is not installed for the interpreter of this repository; this image happens to carry a second one that has it:

    /opt/conda/bin/python3.9 tests/golden/make_hdf5_fixture.py

The file is data (a fixture), kept small; graal_amd/hdf5_min.py must read it without h5py (tests/test_pyramid.py).  Besides the
three levels it holds a group of 40 datasets (several symbol-table nodes), a chunked + gzip + shuffle dataset, a big-endian one, a
compact one and a float64 one, so that the reader's other branches meet genuine libhdf5 output too."""
import json
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.RandomState(20141217)
path = os.path.join(HERE, "pyramid_fixture.hdf5")
if os.path.exists(path):
    os.remove(path)
f = h5py.File(path, "w")
expect = {"h5py": h5py.__version__, "hdf5": h5py.version.hdf5_version, "levels": {}}
n_frags = [90, 30, 10]
for level, nf in enumerate(n_frags):
    n = [400, 150, 40][level]
    keys = np.sort(rng.choice(nf * nf, size=n, replace=False))
    a, b = keys // nf, keys % nf
    keep = a < b
    a, b = a[keep], b[keep]
    v = 1 + rng.poisson(3.0, size=len(a))
    g = f.create_group(str(level))
    d = g.create_dataset("data", (3, len(a)), "i")
    nfd = g.create_dataset("nfrags", (1, 1), "i")
    d[0, :] = a
    d[1, :] = b
    d[2, :] = v
    nfd[:] = nf
    f.attrs[str(level)] = np.string_("done")      # (what Python 2's h5py stored for the reference's str "done")
    expect["levels"][str(level)] = {"nfrags": nf, "data": np.stack([a, b, v]).astype(int).tolist()}
f.attrs["thresh"] = np.float64(0.125)
many = f.create_group("many")
for i in range(40):
    many.create_dataset("d%02d" % i, data=np.arange(i + 1, dtype=np.int32) * (i + 1))
x = (rng.randint(0, 1000, size=(37, 53))).astype(np.int32)
f.create_dataset("chunked_gzip_shuffle", data=x, chunks=(8, 16), compression="gzip", shuffle=True)
f.create_dataset("big_endian", data=x[:5, :7].astype(">i4"))
f.create_dataset("float64", data=np.linspace(0.0, 1.0, 11))
f.create_dataset("compact_like", data=np.arange(6, dtype=np.int16))
f.create_dataset("never_written", (4, 2), "i")
expect["chunked_gzip_shuffle"] = x.tolist()
expect["big_endian"] = x[:5, :7].tolist()
expect["float64"] = np.linspace(0.0, 1.0, 11).tolist()
f.close()
with open(os.path.join(HERE, "pyramid_fixture.json"), "w") as h:
    json.dump(expect, h)
print("wrote", path, os.path.getsize(path), "bytes")
