"""h5lite -- a small read-only HDF5 parser, enough to load a Keras-1 model file.

The reference loads its trained classifier with `keras.models.load_model(KERAS_MODEL_FILE)`
(stone/nn_manager.py:22, 65-73); neither Keras nor h5py exist on an MI355X box, and all the loader
really needs from the file are a few attributes (layer / weight names) and a dozen float arrays.
This module reads them straight from the bytes, following the HDF5 File Format Specification for
the structures libhdf5 writes with its default ("earliest") format, which is what h5py / Keras-1
produce:

  superblock v0/v1 (v2/v3 accepted)    symbol-table groups: B-tree v1 + local heap + SNOD
  object header v1 (v2 accepted)       with continuation blocks
  dataspace v1/v2                      datatype: fixed point, IEEE float, fixed / variable strings
  data layout v1-v3                    contiguous, compact, chunked (B-tree v1; deflate + shuffle)
  attribute messages v1-v3             global heap (variable-length strings)

Not supported (raises H5Error): new-style groups (link messages / fractal heaps), dense attribute
storage, compound / array / reference types, filters other than deflate and shuffle.
"""
import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


class _Dtype:
    """decoded datatype message"""

    def __init__(self, np_dtype=None, vlen_str=False, size=0):
        self.np_dtype = np_dtype
        self.vlen_str = vlen_str
        self.size = size


class Dataset:
    def __init__(self, f, name, shape, dt, layout, filters):
        self._f, self.name, self.shape, self._dt, self._layout, self._filters = f, name, shape, dt, layout, filters
        self.attrs = {}

    @property
    def dtype(self):
        return self._dt.np_dtype

    def read(self):
        """-> numpy array (a copy) with this dataset's shape"""
        f, dt = self._f, self._dt
        if dt.vlen_str or dt.np_dtype is None:
            raise H5Error("dataset %s: only numeric and fixed-string datasets are supported" % self.name)
        n = int(np.prod(self.shape)) if self.shape else 1
        kind = self._layout[0]
        if kind == "compact":
            raw = self._layout[1]
        elif kind == "contiguous":
            addr, size = self._layout[1], self._layout[2]
            if addr == UNDEF:                     # never written: fill value (zeros)
                return np.zeros(self.shape, dt.np_dtype)
            raw = f._buf[f._base + addr: f._base + addr + n * dt.size]
        else:
            return self._read_chunked()
        return np.frombuffer(raw, dt.np_dtype, n).reshape(self.shape).copy()

    def _read_chunked(self):
        f, dt = self._f, self._dt
        _, btree, cdims = self._layout                     # cdims: chunk shape (without the element size)
        out = np.zeros(self.shape, dt.np_dtype)
        rank = len(self.shape)
        csize = int(np.prod(cdims)) * dt.size
        for offs, addr, nbytes, mask in f._chunk_leaves(btree, rank):
            raw = bytes(f._buf[f._base + addr: f._base + addr + nbytes])
            for k, (fid, cd) in reversed(list(enumerate(self._filters))):      # undo the pipeline back to front
                if mask & (1 << k):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:                                                 # shuffle: bytes of each element were transposed
                    es = cd[0] if cd else dt.size
                    raw = np.frombuffer(raw, np.uint8).reshape(es, -1).T.tobytes()
                else:
                    raise H5Error("dataset %s: unsupported filter id %d" % (self.name, fid))
            chunk = np.frombuffer(raw[:csize], dt.np_dtype).reshape(cdims)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, self.shape))
            out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out

    def __getitem__(self, key):
        return self.read()[key]


class Group:
    def __init__(self, f, name):
        self._f, self.name = f, name
        self.attrs = {}
        self._links = {}            # name -> object header address

    def keys(self):
        return list(self._links)

    def __contains__(self, k):
        return k in self._links

    def __iter__(self):
        return iter(self._links)

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(path)
            node = node._f._object(node._links[part], (node.name.rstrip("/") + "/" + part))
        return node


class File(Group):
    """File(path_or_bytes): the root group.  f["model_weights/dense_1/dense_1_W"].read(), f.attrs[...]"""

    def __init__(self, src):
        if isinstance(src, (bytes, bytearray, memoryview)):
            self._buf = memoryview(bytes(src))
        else:
            with open(src, "rb") as fh:
                self._buf = memoryview(fh.read())
        self._cache = {}
        root_addr = self._superblock()
        Group.__init__(self, self, "/")
        root = self._object(root_addr, "/", root=True)
        self.attrs, self._links = root.attrs, root._links

    # ---- primitives ---------------------------------------------------------------------------
    def _u(self, off, size):
        return int.from_bytes(self._buf[off:off + size], "little")

    def _superblock(self):
        buf = self._buf
        base = 0
        while bytes(buf[base:base + 8]) != SIGNATURE:      # the signature may sit at 0, 512, 1024, ... (user block)
            base = 512 if base == 0 else base * 2
            if base + 8 > len(buf):
                raise H5Error("not an HDF5 file")
        ver = buf[base + 8]
        if ver in (0, 1):
            self._so, self._sl = buf[base + 13], buf[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            if (self._so, self._sl) != (8, 8):
                raise H5Error("only 8-byte offsets and lengths are supported")
            self._base = self._u(p, 8)
            p += 32                                          # base, free-space, end-of-file, driver-info addresses
            # root group symbol table entry: name offset, object header address, cache type, reserved, scratch
            self._root_cache = (self._u(p + 16, 4), self._u(p + 24, 8), self._u(p + 32, 8))
            root = self._u(p + 8, 8)
        elif ver in (2, 3):
            self._so, self._sl = buf[base + 9], buf[base + 10]
            if (self._so, self._sl) != (8, 8):
                raise H5Error("only 8-byte offsets and lengths are supported")
            self._base = self._u(base + 12, 8)
            root = self._u(base + 12 + 24, 8)
            self._root_cache = (0, 0, 0)
        else:
            raise H5Error("unknown superblock version %d" % ver)
        self._sb = base
        if self._base == 0 and base:
            self._base = base                                # addresses are relative to the superblock when a user block exists
        return root

    # ---- object headers -------------------------------------------------------------------------
    def _messages(self, addr):
        """yield (type, flags, payload memoryview) of every header message of the object at addr"""
        buf, a = self._buf, self._base + addr
        if bytes(buf[a:a + 4]) == b"OHDR":
            yield from self._messages_v2(a)
            return
        if buf[a] != 1:
            raise H5Error("object header version %d at %d" % (buf[a], addr))
        nmsg = self._u(a + 2, 2)
        size = self._u(a + 8, 4)
        blocks = [(a + 16, size)]
        seen = 0
        while blocks and seen < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and seen < nmsg:
                mtype, msize, flags = self._u(p, 2), self._u(p + 2, 2), buf[p + 4]
                body = buf[p + 8:p + 8 + msize]
                p += 8 + msize
                seen += 1
                if mtype == 0x10:
                    blocks.append((self._base + self._u_of(body, 0, 8), self._u_of(body, 8, 8)))
                else:
                    yield mtype, flags, body

    def _messages_v2(self, a):
        buf = self._buf
        flags = buf[a + 5]
        p = a + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        szbytes = 1 << (flags & 3)
        chunk0 = self._u(p, szbytes)
        p += szbytes
        blocks = [(p, chunk0)]
        track = bool(flags & 0x04)
        while blocks:
            p, left = blocks.pop(0)
            end = p + left
            while p + 4 <= end:
                mtype, msize, mflags = buf[p], self._u(p + 1, 2), buf[p + 3]
                p += 4 + (2 if track else 0)
                body = buf[p:p + msize]
                p += msize
                if mtype == 0x10:
                    ca, cl = self._base + self._u_of(body, 0, 8), self._u_of(body, 8, 8)
                    blocks.append((ca + 4, cl - 8))           # skip "OCHK", drop the checksum
                elif mtype != 0:
                    yield mtype, mflags, body

    @staticmethod
    def _u_of(mv, off, size):
        return int.from_bytes(mv[off:off + size], "little")

    def _object(self, addr, name, root=False):
        if addr in self._cache:
            return self._cache[addr]
        msgs = list(self._messages(addr))
        types = {t for t, _, _ in msgs}
        if 0x08 in types:                                     # data layout -> dataset
            obj = self._dataset(name, msgs)
        else:
            obj = Group(self, name)
            for t, _, body in msgs:
                if t == 0x11:
                    self._read_symbol_table(obj, self._u_of(body, 0, 8), self._u_of(body, 8, 8))
                elif t in (0x02, 0x06):
                    raise H5Error("group %s uses new-style links (file written with libver='latest'?)" % name)
            if root and not obj._links and self._root_cache[0] == 1:
                self._read_symbol_table(obj, self._root_cache[1], self._root_cache[2])
        for t, _, body in msgs:
            if t == 0x0C:
                k, v = self._attribute(body)
                obj.attrs[k] = v
            elif t == 0x15 and self._u_of(body, 2, 8) != UNDEF:
                raise H5Error("%s stores its attributes densely (fractal heap): not supported" % name)
        self._cache[addr] = obj
        return obj

    # ---- groups ---------------------------------------------------------------------------------
    def _heap_str(self, heap_data, off):
        buf = self._buf
        end = heap_data + off
        while buf[end] != 0:
            end += 1
        return bytes(buf[heap_data + off:end]).decode("utf-8")

    def _read_symbol_table(self, group, btree, heap):
        buf = self._buf
        h = self._base + heap
        if bytes(buf[h:h + 4]) != b"HEAP":
            raise H5Error("bad local heap at %d" % heap)
        heap_data = self._base + self._u(h + 24, 8)
        stack = [btree]
        while stack:
            n = self._base + stack.pop()
            sig = bytes(buf[n:n + 4])
            if sig == b"TREE":
                if buf[n + 4] != 0:
                    raise H5Error("group B-tree node of type %d" % buf[n + 4])
                used = self._u(n + 6, 2)
                p = n + 24 + 8                                # first child follows key 0
                kids = [self._u(p + 16 * i, 8) for i in range(used)]
                stack.extend(reversed(kids))
            elif sig == b"SNOD":
                cnt = self._u(n + 6, 2)
                for i in range(cnt):
                    e = n + 8 + 40 * i
                    group._links[self._heap_str(heap_data, self._u(e, 8))] = self._u(e + 8, 8)
            else:
                raise H5Error("unexpected block %r in a group B-tree" % sig)

    # ---- datasets ---------------------------------------------------------------------------------
    def _dataspace(self, body):
        ver, rank, flags = body[0], body[1], body[2]
        if ver == 1:
            p = 8
        elif ver == 2:
            if body[3] == 2:
                return None                                   # null dataspace
            p = 4
        else:
            raise H5Error("dataspace version %d" % ver)
        return tuple(self._u_of(body, p + 8 * i, 8) for i in range(rank))

    def _datatype(self, body):
        cls, ver = body[0] & 0x0F, body[0] >> 4
        bits = self._u_of(body, 1, 3)
        size = self._u_of(body, 4, 4)
        order = ">" if bits & 1 else "<"
        if cls == 0:
            signed = bool(bits & 0x08)
            return _Dtype(np.dtype("%s%s%d" % (order, "i" if signed else "u", size)), size=size)
        if cls == 1:
            if size not in (2, 4, 8):
                raise H5Error("%d-byte floats are not supported" % size)
            return _Dtype(np.dtype("%sf%d" % (order, size)), size=size)
        if cls == 3:
            return _Dtype(np.dtype("S%d" % size), size=size)
        if cls == 9:
            if (bits & 0x0F) != 1:
                raise H5Error("variable-length sequences are not supported (only strings)")
            return _Dtype(None, vlen_str=True, size=size)
        raise H5Error("datatype class %d (version %d) is not supported" % (cls, ver))

    def _dataset(self, name, msgs):
        shape, dt, layout, filters = (), None, None, []
        for t, _, body in msgs:
            if t == 0x01:
                shape = self._dataspace(body)
            elif t == 0x03:
                dt = self._datatype(body)
            elif t == 0x0B:
                filters = self._filters(body)
            elif t == 0x08:
                ver = body[0]
                if ver == 3:
                    cls = body[1]
                    if cls == 0:
                        n = self._u_of(body, 2, 2)
                        layout = ("compact", bytes(body[4:4 + n]))
                    elif cls == 1:
                        layout = ("contiguous", self._u_of(body, 2, 8), self._u_of(body, 10, 8))
                    elif cls == 2:
                        nd = body[2]
                        dims = [self._u_of(body, 11 + 4 * i, 4) for i in range(nd)]
                        layout = ("chunked", self._u_of(body, 3, 8), tuple(dims[:-1]))
                    else:
                        raise H5Error("layout class %d" % cls)
                elif ver in (1, 2):
                    nd, cls = body[1], body[2]
                    p = 8
                    addr = None
                    if cls != 0:
                        addr = self._u_of(body, p, 8)
                        p += 8
                    dims = [self._u_of(body, p + 4 * i, 4) for i in range(nd)]
                    p += 4 * nd
                    if cls == 0:
                        n = self._u_of(body, p, 4)
                        layout = ("compact", bytes(body[p + 4:p + 4 + n]))
                    elif cls == 1:
                        layout = ("contiguous", addr, int(np.prod(dims)) if dims else 0)
                    else:
                        layout = ("chunked", addr, tuple(dims[:-1]))
                else:
                    raise H5Error("data layout version %d (virtual / v4 layouts are not supported)" % ver)
        if dt is None or layout is None or shape is None:
            raise H5Error("dataset %s: incomplete header" % name)
        return Dataset(self, name, shape, dt, layout, filters)

    def _filters(self, body):
        ver, n = body[0], body[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = self._u_of(body, p, 2)
            if ver == 1 or fid >= 256:
                nlen = self._u_of(body, p + 2, 2)
                ncd = self._u_of(body, p + 6, 2)
                p += 8 + (((nlen + 7) // 8) * 8 if ver == 1 else nlen)
            else:
                ncd = self._u_of(body, p + 4, 2)
                p += 6
            cd = [self._u_of(body, p + 4 * i, 4) for i in range(ncd)]
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            out.append((fid, cd))
        return out

    def _chunk_leaves(self, btree, rank):
        """yield (offsets, address, stored bytes, filter mask) of every chunk"""
        buf = self._buf
        stack = [btree]
        ksz = 8 + 8 * (rank + 1)
        while stack:
            a = stack.pop()
            if a == UNDEF:
                continue
            n = self._base + a
            if bytes(buf[n:n + 4]) != b"TREE" or buf[n + 4] != 1:
                raise H5Error("bad chunk B-tree node at %d" % a)
            level, used = buf[n + 5], self._u(n + 6, 2)
            p = n + 24
            for _ in range(used):
                nbytes, mask = self._u(p, 4), self._u(p + 4, 4)
                offs = tuple(self._u(p + 8 + 8 * i, 8) for i in range(rank))
                child = self._u(p + ksz, 8)
                p += ksz + 8
                if level == 0:
                    yield offs, child, nbytes, mask
                else:
                    stack.append(child)

    # ---- attributes ---------------------------------------------------------------------------------
    def _attribute(self, body):
        ver = body[0]
        nsz, dsz, ssz = self._u_of(body, 2, 2), self._u_of(body, 4, 2), self._u_of(body, 6, 2)
        if ver == 1:
            p = 8
            pad = lambda x: (x + 7) // 8 * 8                 # noqa: E731
        elif ver in (2, 3):
            p = 8 + (1 if ver == 3 else 0)
            pad = lambda x: x                                # noqa: E731
            if body[1] & 0x03:
                raise H5Error("shared attribute datatypes / dataspaces are not supported")
        else:
            raise H5Error("attribute message version %d" % ver)
        name = bytes(body[p:p + nsz]).split(b"\0")[0].decode("utf-8")
        p += pad(nsz)
        dt = self._datatype(body[p:p + dsz])
        p += pad(dsz)
        shape = self._dataspace(body[p:p + ssz])
        p += pad(ssz)
        if shape is None:
            return name, None
        n = int(np.prod(shape)) if shape else 1
        if dt.vlen_str:
            vals = []
            for i in range(n):
                q = p + 16 * i
                length, gaddr, gidx = self._u_of(body, q, 4), self._u_of(body, q + 4, 8), self._u_of(body, q + 12, 4)
                vals.append(self._global_heap(gaddr, gidx)[:length].decode("utf-8"))
            arr = np.array(vals, dtype=object).reshape(shape)
        else:
            arr = np.frombuffer(bytes(body[p:p + n * dt.size]), dt.np_dtype, n).reshape(shape).copy()
        return name, (arr[()] if shape == () else arr)

    def _global_heap(self, addr, index):
        buf, a = self._buf, self._base + addr
        if bytes(buf[a:a + 4]) != b"GCOL":
            raise H5Error("bad global heap collection at %d" % addr)
        end = a + self._u(a + 8, 8)
        p = a + 16
        while p + 16 <= end:
            idx, size = self._u(p, 2), self._u(p + 8, 8)
            if idx == 0:
                break
            if idx == index:
                return bytes(buf[p + 16:p + 16 + size])
            p += 16 + (size + 7) // 8 * 8
        raise H5Error("global heap object %d not found" % index)


def visit(group, prefix=""):
    """yield (path, object) for every dataset below `group`"""
    for k in group.keys():
        obj = group[k]
        path = prefix + "/" + k
        if isinstance(obj, Group):
            yield from visit(obj, path)
        else:
            yield path, obj
