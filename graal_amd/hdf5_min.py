"""Minimal pure-Python reader for the HDF5 files GRAAL writes (``pyramid.hdf5``, ``pyramid_sparse.py:267-324``): h5py is not
part of this image, and an existing GRAAL pyramid must still be readable.

Scope -- what ``h5py.File(path)`` with default settings produces for ``create_group(str(level))`` +
``create_dataset('data', (3, n), 'i')`` + ``create_dataset('nfrags', (1, 1), 'i')`` (+ root attributes ``str(level) = "done"``):

* superblock version 0 / 1 (and 2 / 3: root object header address only);
* version-1 object headers with continuation blocks; version-2 ("OHDR") headers with compact link messages;
* old-style groups: symbol-table message -> version-1 B-tree ("TREE") of symbol nodes ("SNOD") + local heap ("HEAP");
* datasets: fixed-point and IEEE float types, little / big endian; contiguous, compact and chunked layout (version-1 chunk
  B-tree; filters deflate and shuffle);
* attributes with fixed-length string or numeric values (variable-length strings are returned as ``None``).

Anything else raises ``Hdf5Error`` naming what was met -- the caller then asks for h5py.  When h5py IS importable,
``read_pyramid_levels`` uses it instead.  Tested against a file written by h5py 3.3 / libhdf5 1.10.6 with the reference's own
calls (tests/golden/pyramid_fixture.hdf5, generator tests/golden/make_hdf5_fixture.py).
"""
import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(RuntimeError):
    pass


class _Reader:
    def __init__(self, buf):
        self.b = buf
        self.so = self.sl = 8

    def u(self, off, n):
        return int.from_bytes(self.b[off:off + n], "little")

    def offset(self, off):
        return self.u(off, self.so)

    def length(self, off):
        return self.u(off, self.sl)


class Dataset:
    def __init__(self, f, shape, dtype, layout, filters):
        self._f, self.shape, self.dtype, self._layout, self._filters = f, tuple(shape), dtype, layout, filters

    def __getitem__(self, key):
        return self.read()[key]

    def read(self):
        r = self._f._r
        n = int(np.prod(self.shape)) if self.shape else 1
        kind = self._layout[0]
        if kind == "compact":
            raw = self._layout[1]
        elif kind == "contiguous":
            addr, size = self._layout[1], self._layout[2]
            if addr == UNDEF:      # never written: fill value 0
                return np.zeros(self.shape, dtype=self.dtype.newbyteorder("="))
            raw = r.b[addr:addr + n * self.dtype.itemsize]
        else:
            return self._read_chunked()
        return np.frombuffer(raw, dtype=self.dtype, count=n).reshape(self.shape).astype(self.dtype.newbyteorder("="))

    def _read_chunked(self):
        _, btree, chunk = self._layout
        out = np.zeros(self.shape, dtype=self.dtype.newbyteorder("="))
        if btree == UNDEF:
            return out
        rank = len(self.shape)
        for offs, size, mask, addr in self._f._chunks(btree, rank):
            raw = bytes(self._f._r.b[addr:addr + size])
            for j, (fid, cd) in reversed(list(enumerate(self._filters))):
                if mask & (1 << j):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    es = self.dtype.itemsize
                    a = np.frombuffer(raw, dtype=np.uint8)
                    k = len(a) // es
                    raw = a[:k * es].reshape(es, k).T.tobytes() + a[k * es:].tobytes()
                else:
                    raise Hdf5Error("unsupported filter id %d" % fid)
            block = np.frombuffer(raw, dtype=self.dtype, count=int(np.prod(chunk))).reshape(chunk)
            sl_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, self.shape))
            sl_in = tuple(slice(0, s.stop - s.start) for s in sl_out)
            out[sl_out] = block[sl_in]
        return out


class Group:
    def __init__(self, f, links, attrs):
        self._f, self._links, self.attrs = f, links, attrs

    def keys(self):
        return sorted(self._links)

    def __contains__(self, k):
        return k in self._links

    def __getitem__(self, name):
        node = self
        for part in [p for p in name.split("/") if p]:
            if part not in node._links:
                raise KeyError(name)
            node = node._f._object(node._links[part])
        return node


class File(Group):
    def __init__(self, path):
        with open(path, "rb") as fh:
            buf = fh.read()
        r = self._r = _Reader(memoryview(buf))
        base = None
        for off in [0] + [512 << i for i in range(12)]:     # the superblock may sit behind a user block
            if bytes(r.b[off:off + 8]) == SIGNATURE:
                base = off
                break
        if base is None:
            raise Hdf5Error("not an HDF5 file (no superblock signature)")
        ver = r.u(base + 8, 1)
        if ver in (0, 1):
            r.so, r.sl = r.u(base + 13, 1), r.u(base + 14, 1)
            p = base + 24 + (4 if ver == 1 else 0)
            p += 4 * r.so                               # base address, free-space info, end of file, driver info
            root = r.offset(p + r.so)                   # root symbol-table entry: link name offset, object header address
        elif ver in (2, 3):
            r.so, r.sl = r.u(base + 9, 1), r.u(base + 10, 1)
            root = r.offset(base + 12 + 3 * r.so)
        else:
            raise Hdf5Error("superblock version %d" % ver)
        self._cache = {}
        g = self._object(root)
        if not isinstance(g, Group):
            raise Hdf5Error("the root object is not a group")
        Group.__init__(self, self, g._links, g.attrs)

    def close(self):
        pass

    # ---- object headers -------------------------------------------------------------------------------------------
    def _messages(self, addr):
        r = self._r
        if bytes(r.b[addr:addr + 4]) == b"OHDR":
            yield from self._messages_v2(addr)
            return
        if r.u(addr, 1) != 1:
            raise Hdf5Error("object header version %d at %d" % (r.u(addr, 1), addr))
        n_msg, size = r.u(addr + 2, 2), r.u(addr + 8, 4)
        blocks = [(addr + 16, size)]
        seen = 0
        while blocks and seen < n_msg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and seen < n_msg:
                mtype, msize = r.u(p, 2), r.u(p + 2, 2)
                body = p + 8
                seen += 1
                if mtype == 0x10:
                    blocks.append((r.offset(body), r.length(body + r.so)))
                else:
                    yield mtype, body, msize
                p = body + msize

    def _messages_v2(self, addr):
        r = self._r
        flags = r.u(addr + 5, 1)
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        cs = 1 << (flags & 3)
        size = r.u(p, cs)
        p += cs
        blocks = [(p, size)]
        while blocks:
            p, left = blocks.pop(0)
            end = p + left
            while p + 4 <= end:
                mtype, msize = r.u(p, 1), r.u(p + 1, 2)
                body = p + 4 + (2 if flags & 0x04 else 0)
                if mtype == 0x10:
                    a, ln = r.offset(body), r.length(body + r.so)
                    blocks.append((a + 4, ln - 8))          # "OCHK" signature in front, checksum behind
                elif mtype != 0:
                    yield mtype, body, msize
                p = body + msize

    def _object(self, addr):
        if addr in self._cache:
            return self._cache[addr]
        r = self._r
        links, attrs = {}, {}
        shape = dtype = layout = None
        filters, is_group = [], False
        for mtype, body, msize in self._messages(addr):
            if mtype == 0x11:       # symbol table: B-tree + local heap
                is_group = True
                self._walk_group(r.offset(body), r.offset(body + r.so), links)
            elif mtype == 0x06:     # link message (new-style compact groups)
                is_group = True
                name, target = self._link(body)
                if target is not None:
                    links[name] = target
            elif mtype == 0x02:     # link info: dense storage is not supported, compact links follow as messages
                is_group = True
                if r.offset(body + 2 + (8 if r.u(body + 1, 1) & 1 else 0)) != UNDEF:
                    raise Hdf5Error("group with dense link storage (fractal heap)")
            elif mtype == 0x01:
                shape = self._dataspace(body)
            elif mtype == 0x03:
                dtype = self._datatype(body)
            elif mtype == 0x08:
                layout = self._layout(body)
            elif mtype == 0x0B:
                filters = self._filters(body)
            elif mtype == 0x0C:
                k, v = self._attribute(body)
                attrs[k] = v
        if is_group or layout is None:
            obj = Group(self, links, attrs)
        else:
            if dtype is None or shape is None:
                raise Hdf5Error("dataset without datatype / dataspace at %d" % addr)
            obj = Dataset(self, shape, dtype, layout, filters)
            obj.attrs = attrs
        self._cache[addr] = obj
        return obj

    # ---- groups ---------------------------------------------------------------------------------------------------
    def _heap_string(self, heap, off):
        r = self._r
        if bytes(r.b[heap:heap + 4]) != b"HEAP":
            raise Hdf5Error("bad local heap signature")
        data = r.offset(heap + 8 + 2 * r.sl)
        p = data + off
        q = p
        while r.b[q] != 0:
            q += 1
        return bytes(r.b[p:q]).decode("utf-8")

    def _walk_group(self, node, heap, links):
        r = self._r
        sig = bytes(r.b[node:node + 4])
        if sig == b"TREE":
            level, n = r.u(node + 5, 1), r.u(node + 6, 2)
            p = node + 8 + 2 * r.so
            for i in range(n):
                child = r.offset(p + r.sl)
                self._walk_group(child, heap, links)
                p += r.sl + r.so
        elif sig == b"SNOD":
            n = r.u(node + 6, 2)
            p = node + 8
            for i in range(n):
                links[self._heap_string(heap, r.offset(p))] = r.offset(p + r.so)
                p += 2 * r.so + 24
        else:
            raise Hdf5Error("unexpected group node signature %r" % sig)

    def _link(self, body):
        r = self._r
        flags = r.u(body + 1, 1)
        p = body + 2
        ltype = 0
        if flags & 0x08:
            ltype = r.u(p, 1)
            p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        ls = 1 << (flags & 3)
        nlen = r.u(p, ls)
        p += ls
        name = bytes(r.b[p:p + nlen]).decode("utf-8")
        p += nlen
        return name, (r.offset(p) if ltype == 0 else None)

    # ---- dataset pieces -------------------------------------------------------------------------------------------
    def _dataspace(self, body):
        r = self._r
        ver, rank = r.u(body, 1), r.u(body + 1, 1)
        p = body + (8 if ver == 1 else 4)
        return [r.length(p + i * r.sl) for i in range(rank)]

    def _datatype(self, body):
        r = self._r
        cls, bits0, size = r.u(body, 1) & 0x0F, r.u(body + 1, 1), r.u(body + 4, 4)
        order = ">" if bits0 & 1 else "<"
        if cls == 0:
            return np.dtype("%s%s%d" % (order, "i" if bits0 & 0x08 else "u", size))
        if cls == 1:
            return np.dtype("%sf%d" % (order, size))
        if cls == 3:
            return np.dtype("S%d" % size)
        if cls == 9:
            return np.dtype("O")        # variable length: not read
        raise Hdf5Error("datatype class %d" % cls)

    def _layout(self, body):
        r = self._r
        ver = r.u(body, 1)
        if ver == 3:
            cls = r.u(body + 1, 1)
            if cls == 0:
                size = r.u(body + 2, 2)
                return ("compact", bytes(r.b[body + 4:body + 4 + size]))
            if cls == 1:
                return ("contiguous", r.offset(body + 2), r.length(body + 2 + r.so))
            if cls == 2:
                nd = r.u(body + 2, 1)
                btree = r.offset(body + 3)
                dims = [r.u(body + 3 + r.so + 4 * i, 4) for i in range(nd)]
                return ("chunked", btree, tuple(dims[:-1]))
            raise Hdf5Error("layout class %d" % cls)
        if ver in (1, 2):
            nd, cls = r.u(body + 1, 1), r.u(body + 2, 1)
            p = body + 8
            if cls == 1:
                addr = r.offset(p)
                dims = [r.u(p + r.so + 4 * i, 4) for i in range(nd)]
                return ("contiguous", addr, int(np.prod(dims)))
            if cls == 2:
                addr = r.offset(p)
                dims = [r.u(p + r.so + 4 * i, 4) for i in range(nd)]
                return ("chunked", addr, tuple(dims[:-1]))
            dims = [r.u(p + 4 * i, 4) for i in range(nd)]
            size = r.u(p + 4 * nd, 4)
            return ("compact", bytes(r.b[p + 4 * nd + 4:p + 4 * nd + 4 + size]))
        raise Hdf5Error("data layout message version %d (written with libver='latest'?)" % ver)

    def _filters(self, body):
        r = self._r
        ver, n = r.u(body, 1), r.u(body + 1, 1)
        p = body + (8 if ver == 1 else 2)
        out = []
        for _ in range(n):
            fid = r.u(p, 2)
            if ver == 1 or fid >= 256:
                nlen = r.u(p + 2, 2)
                ncd = r.u(p + 6, 2)
                p += 8 + nlen + ((-nlen) % 8 if ver == 1 else 0)
            else:
                ncd = r.u(p + 4, 2)
                p += 6
            cd = [r.u(p + 4 * i, 4) for i in range(ncd)]
            p += 4 * ncd + (4 if ver == 1 and ncd % 2 else 0)
            out.append((fid, cd))
        return out

    def _chunks(self, node, rank):
        r = self._r
        if bytes(r.b[node:node + 4]) != b"TREE" or r.u(node + 4, 1) != 1:
            raise Hdf5Error("unexpected chunk index (not a version-1 B-tree)")
        level, n = r.u(node + 5, 1), r.u(node + 6, 2)
        key = 8 + 8 * (rank + 1)
        p = node + 8 + 2 * r.so
        for i in range(n):
            size, mask = r.u(p, 4), r.u(p + 4, 4)
            offs = [r.u(p + 8 + 8 * j, 8) for j in range(rank)]
            child = r.offset(p + key)
            if level == 0:
                yield offs, size, mask, child
            else:
                yield from self._chunks(child, rank)
            p += key + r.so

    def _attribute(self, body):
        r = self._r
        ver = r.u(body, 1)
        nsz, tsz, ssz = r.u(body + 2, 2), r.u(body + 4, 2), r.u(body + 6, 2)
        p = body + 8 + (1 if ver == 3 else 0)
        pad = (lambda x: (x + 7) & ~7) if ver == 1 else (lambda x: x)
        name = bytes(r.b[p:p + nsz]).split(b"\0")[0].decode("utf-8")
        p += pad(nsz)
        dtype = self._datatype(p)
        p += pad(tsz)
        shape = self._dataspace(p) if ssz >= 4 else []
        p += pad(ssz)
        if dtype == np.dtype("O"):
            return name, None
        n = int(np.prod(shape)) if shape else 1
        val = np.frombuffer(bytes(r.b[p:p + n * dtype.itemsize]), dtype=dtype, count=n)
        if dtype.kind == "S":
            val = [v.split(b"\0")[0].decode("utf-8") for v in val]
            return name, (val[0] if not shape else val)
        val = val.astype(dtype.newbyteorder("="))
        return name, (val[0] if not shape else val.reshape(shape))


def read_pyramid_levels(path):
    """{level: (data[3, nnz] int32 = id_a / id_b / count (0-based ids), nfrags)} of a GRAAL ``pyramid.hdf5``
    (``pyramid_sparse.py:313-322``, read back at ``:1216-1219``).  h5py if it can be imported, this module otherwise."""
    try:
        import h5py
        f = h5py.File(path, "r")
        keys = list(f.keys())
        get = lambda g, k: np.asarray(f[g][k])
    except ImportError:
        f = File(path)
        keys = f.keys()
        get = lambda g, k: f[g][k].read()
    out = {}
    for g in keys:
        if not g.isdigit():
            continue
        out[int(g)] = (np.asarray(get(g, "data"), dtype=np.int32), int(np.asarray(get(g, "nfrags")).ravel()[0]))
    f.close()
    return out
