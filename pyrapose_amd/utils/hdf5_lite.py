"""A dependency-free reader (and a minimal writer) for the subset of HDF5 that Keras 2.3.1 / h5py 2.10 weight files use.

Why: the reference loads and saves its weights as Keras `.h5` files (bin/train.py:65-68 `load_weights`, :127-143
`ModelCheckpoint`, models/__init__.py:68-71 `load_model`, models/resnet.py:89-98 the ImageNet file); this image has no
h5py / libhdf5.  What those files contain (h5py with its default `libver='earliest'`): superblock version 0, version-1
object headers (with continuation blocks), old-style groups (symbol-table message -> v1 B-tree -> symbol-table nodes ->
local heap), contiguous little-endian datasets, and attributes holding fixed-length strings / numbers.  This module follows
the HDF5 File Format Specification (version 1.x structures named above) for exactly that subset, plus "link" messages of
compact new-style groups and compact / chunk-free layouts where they are trivial; anything else raises `H5Unsupported`
(chunked or compressed datasets, variable-length data, dense attribute storage, shared messages).

STATUS: written to the published specification, round-trip tested against the writer below, and checked against the one
libhdf5-written file this image holds: scipy's `testhdf5_7.4_GLNX86.mat` (MATLAB 7.4 = HDF5 1.6 behind a 512-byte user
block: superblock version 0 found at offset 512 with base address 512, version-1 object header, old-style root group,
version-2 data layout message, version-1 attribute with a fixed-length string) -- tests/test_hdf5_lite.py.  No file written
by h5py / Keras itself exists here; `tools/h5_to_npz.py` (h5py, run where the file was made) remains the fallback route.

The writer emits the same subset (one symbol-table node per group, sized by the superblock's leaf K): enough for
`save_weights(..., format='h5')` checkpoints with Keras' layout and for the tests.
"""
import struct
from collections import OrderedDict

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Unsupported(NotImplementedError):
    pass


def _guarded(fn):
    """Structures that end early or point outside the file (a truncated download, a layout this module does not know) surface as
    H5Unsupported -- never as struct.error / IndexError / a numpy buffer error from the middle of the parser."""
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **kw):
        try:
            return fn(*a, **kw)
        except H5Unsupported:
            raise
        except KeyError:
            raise
        except (struct.error, IndexError, OverflowError, ValueError, TypeError) as e:
            raise H5Unsupported("malformed or unsupported HDF5 structure in %s: %s: %s" % (fn.__qualname__, type(e).__name__, e))
    return wrapper


# ------------------------------------------------------------------------------------------------ reader
def find_superblock(data):
    """Offset of the superblock: the signature sits at byte 0 or, behind a user block, at 512, 1024, 2048, ... (format
    specification II.A: "the superblock may begin at certain predefined offsets ... 0, 512, 1024, 2048, and so on")."""
    off = 0
    while off + 8 <= len(data):
        if bytes(data[off:off + 8]) == SIGNATURE:
            return off
        off = 512 if off == 0 else off * 2
    return -1


def is_hdf5(path):
    """True when `path` holds an HDF5 superblock at one of the offsets find_superblock() accepts (only those bytes are read)."""
    with open(path, "rb") as f:
        size = f.seek(0, 2)
        off = 0
        while off + 8 <= size:
            f.seek(off)
            if f.read(8) == SIGNATURE:
                return True
            off = 512 if off == 0 else off * 2
    return False


class _Reader(object):
    def __init__(self, data):
        self.b = data
        sb = find_superblock(data)
        if sb < 0:
            raise ValueError("not an HDF5 file (no superblock signature at 0, 512, 1024, ...)")
        self.superblock = sb
        ver = data[sb + 8]
        if ver in (0, 1):
            self.so, self.sl = data[sb + 13], data[sb + 14]
            if (self.so, self.sl) != (8, 8):
                raise H5Unsupported("offsets / lengths of %d / %d bytes" % (self.so, self.sl))
            p = sb + (24 if ver == 0 else 28)
            self.base = self.u64(p)  # every address in the file is relative to this (= the user block's size when there is one)
            root_entry = p + 32
            self.root_header = self.u64(root_entry + 8)
        elif ver in (2, 3):
            self.so, self.sl = data[sb + 9], data[sb + 10]
            if (self.so, self.sl) != (8, 8):
                raise H5Unsupported("offsets / lengths of %d / %d bytes" % (self.so, self.sl))
            self.base = self.u64(sb + 12)
            self.root_header = self.u64(sb + 12 + 24)
        else:
            raise H5Unsupported("superblock version %d" % ver)

    def u8(self, o):
        return self.b[o]

    def u16(self, o):
        return struct.unpack_from("<H", self.b, o)[0]

    def u32(self, o):
        return struct.unpack_from("<I", self.b, o)[0]

    def u64(self, o):
        return struct.unpack_from("<Q", self.b, o)[0]

    # ---- object headers -> list of (type, flags, data offset, size)
    def messages(self, addr):
        addr += self.base
        out = []
        if self.b[addr:addr + 4] == b"OHDR":  # version 2 header (libver='latest'): prefix + chunk 0
            flags = self.u8(addr + 5)
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            n = 1 << (flags & 3)
            size = int.from_bytes(self.b[p:p + n], "little")
            p += n
            blocks = [(p, size)]
            track = bool(flags & 4)
            while blocks:
                p, size = blocks.pop(0)
                end = p + size
                q = p
                while q + 4 <= end:
                    mtype, msize, mflags = self.u8(q), self.u16(q + 1), self.u8(q + 3)
                    q += 4 + (2 if track else 0)
                    if mtype == 0x10:
                        coff, clen = self.u64(q), self.u64(q + 8)
                        blocks.append((self.base + coff + 4, clen - 8))  # skip 'OCHK', drop the checksum
                    else:
                        out.append((mtype, mflags, q, msize))
                    q += msize
            return out
        if self.u8(addr) != 1:
            raise H5Unsupported("object header version %d" % self.u8(addr))
        n_msgs, size = self.u16(addr + 2), self.u32(addr + 8)
        blocks = [(addr + 16, size)]
        while blocks and len(out) < n_msgs + 64:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end:
                mtype, msize, mflags = self.u16(p), self.u16(p + 2), self.u8(p + 4)
                d = p + 8
                if mtype == 0x0010:
                    blocks.append((self.base + self.u64(d), self.u64(d + 8)))
                elif mtype != 0:
                    out.append((mtype, mflags, d, msize))
                p = d + msize
        return out

    # ---- datatype message -> (numpy dtype or None, size)
    def datatype(self, o):
        cv = self.u8(o)
        cls, bits0 = cv & 15, self.u8(o + 1)
        size = self.u32(o + 4)
        if cls == 1:
            order = ">" if bits0 & 1 else "<"
            if size not in (2, 4, 8):
                raise H5Unsupported("float of %d bytes" % size)
            return np.dtype(order + "f%d" % size), size
        if cls == 0:
            order = ">" if bits0 & 1 else "<"
            signed = bool(bits0 & 8)
            if size not in (1, 2, 4, 8):
                raise H5Unsupported("integer of %d bytes" % size)
            return np.dtype(order + ("i" if signed else "u") + str(size)), size
        if cls == 3:
            return np.dtype("S%d" % size), size
        raise H5Unsupported("datatype class %d (variable-length / compound / ... data is outside the Keras weight-file subset)" % cls)

    def dataspace(self, o):
        ver, rank = self.u8(o), self.u8(o + 1)
        if ver == 1:
            p = o + 8
        elif ver == 2:
            if self.u8(o + 3) == 2:  # null dataspace
                return None
            p = o + 4
        else:
            raise H5Unsupported("dataspace version %d" % ver)
        return tuple(self.u64(p + 8 * i) for i in range(rank))

    def attribute(self, o):
        ver = self.u8(o)
        nsz, tsz, ssz = self.u16(o + 2), self.u16(o + 4), self.u16(o + 6)
        if ver == 1:
            pad = lambda v: (v + 7) & ~7
            p = o + 8
            name = bytes(self.b[p:p + nsz]).split(b"\0")[0].decode("utf-8", "replace")
            p += pad(nsz)
            t = p
            p += pad(tsz)
            s = p
            p += pad(ssz)
        elif ver in (2, 3):
            if self.u8(o + 1) & 3:
                raise H5Unsupported("attribute %r with shared datatype / dataspace")
            p = o + 8 + (1 if ver == 3 else 0)
            name = bytes(self.b[p:p + nsz]).split(b"\0")[0].decode("utf-8", "replace")
            p += nsz
            t = p
            p += tsz
            s = p
            p += ssz
        else:
            raise H5Unsupported("attribute message version %d" % ver)
        try:
            dt, _ = self.datatype(t)
        except H5Unsupported:
            return name, None  # e.g. variable-length strings (model_config of full-model files): not needed for weights
        shape = self.dataspace(s)
        if shape is None:
            return name, None
        n = int(np.prod(shape)) if shape else 1
        arr = np.frombuffer(self.b, dtype=dt, count=n, offset=p).reshape(shape)
        return name, (arr.copy() if shape else arr.reshape(())[()])

    # ---- old-style group: B-tree of symbol-table nodes
    def group_entries(self, btree, heap):
        heap += self.base
        if self.b[heap:heap + 4] != b"HEAP":
            raise H5Unsupported("bad local heap signature")
        seg = self.base + self.u64(heap + 24)
        out = OrderedDict()

        def name_at(off):
            e = self.b.index(b"\0", seg + off)
            return bytes(self.b[seg + off:e]).decode("utf-8", "replace")

        def walk(node):
            node += self.base
            sig = self.b[node:node + 4]
            if sig == b"TREE":  # children of level-0 nodes are symbol-table nodes, of higher levels further B-tree nodes
                used = self.u16(node + 6)
                p = node + 24
                for i in range(used):
                    walk(self.u64(p + 8 + 16 * i))
            elif sig == b"SNOD":
                n = self.u16(node + 6)
                for i in range(n):
                    e = node + 8 + 40 * i
                    out[name_at(self.u64(e))] = self.u64(e + 8)
            else:
                raise H5Unsupported("bad group node signature %r" % bytes(sig))
        walk(btree)
        return out


class _Object(object):
    @_guarded
    def __init__(self, rd, addr, name):
        self._rd, self._addr, self.name = rd, addr, name
        self._msgs = rd.messages(addr)
        self.attrs = OrderedDict()
        for (t, fl, o, sz) in self._msgs:
            if t == 0x000C:
                if fl & 2:
                    raise H5Unsupported("shared attribute message")
                k, v = rd.attribute(o)
                self.attrs[k] = v
            elif t == 0x0015:  # attribute info message: fractal-heap address != UNDEF <=> attributes stored densely
                if rd.u64(o + 2 + (2 if rd.u8(o + 1) & 1 else 0)) != UNDEF:
                    raise H5Unsupported("dense attribute storage (file written with libver='latest')")

    def _is_group(self):
        return any(t in (0x0011, 0x0002, 0x0006) for (t, _, _, _) in self._msgs)


class Dataset(_Object):
    @_guarded
    def read(self):
        rd = self._rd
        dt = shape = layout = None
        for (t, fl, o, sz) in self._msgs:
            if t == 0x0003:
                if fl & 2:
                    raise H5Unsupported("committed (shared) datatype")
                dt, _ = rd.datatype(o)
            elif t == 0x0001:
                shape = rd.dataspace(o)
            elif t == 0x0008:
                layout = o
            elif t == 0x000B:
                raise H5Unsupported("filtered (compressed) dataset %s" % self.name)
        if dt is None or shape is None or layout is None:
            raise H5Unsupported("%s is not a simple dataset" % self.name)
        n = int(np.prod(shape)) if shape else 1
        ver = rd.u8(layout)
        if ver in (1, 2):
            # layout message of HDF5 <= 1.6 writers: version, dimensionality, class, 5 reserved bytes, [address], the
            # dimensions as 4-byte numbers (contiguous / chunked: rank + 1 of them, the last one the element size), then for
            # compact storage a 4-byte size and the data
            ndim, cls = rd.u8(layout + 1), rd.u8(layout + 2)
            if cls == 1:
                addr = rd.u64(layout + 8)
                dims = [rd.u32(layout + 16 + 4 * i) for i in range(ndim)]
                size = int(np.prod(dims)) if dims else 0
                if addr == UNDEF:
                    return np.zeros(shape, dt.newbyteorder("="))
                if size < n * dt.itemsize:
                    raise H5Unsupported("dataset %s: %d bytes stored, %d needed" % (self.name, size, n * dt.itemsize))
                arr = np.frombuffer(rd.b, dtype=dt, count=n, offset=rd.base + addr)
            elif cls == 0:
                o = layout + 8 + 4 * ndim
                if rd.u32(o) < n * dt.itemsize:
                    raise H5Unsupported("compact dataset %s: %d bytes stored, %d needed" % (self.name, rd.u32(o), n * dt.itemsize))
                arr = np.frombuffer(rd.b, dtype=dt, count=n, offset=o + 4)
            else:
                raise H5Unsupported("chunked dataset %s" % self.name)
            return arr.reshape(shape).astype(dt.newbyteorder("="), copy=True)
        if ver != 3:
            raise H5Unsupported("data layout message version %d" % ver)
        cls = rd.u8(layout + 1)
        if cls == 1:
            addr, size = rd.u64(layout + 2), rd.u64(layout + 10)
            if addr == UNDEF:
                return np.zeros(shape, dt.newbyteorder("="))
            if size < n * dt.itemsize:
                raise H5Unsupported("dataset %s: %d bytes stored, %d needed" % (self.name, size, n * dt.itemsize))
            arr = np.frombuffer(rd.b, dtype=dt, count=n, offset=rd.base + addr)
        elif cls == 0:
            arr = np.frombuffer(rd.b, dtype=dt, count=n, offset=layout + 4)
        else:
            raise H5Unsupported("chunked dataset %s" % self.name)
        return arr.reshape(shape).astype(dt.newbyteorder("="), copy=True)

    @property
    def shape(self):
        for (t, fl, o, sz) in self._msgs:
            if t == 0x0001:
                return self._rd.dataspace(o)


class Group(_Object):
    @_guarded
    def _children(self):
        rd = self._rd
        out = OrderedDict()
        for (t, fl, o, sz) in self._msgs:
            if t == 0x0011:
                out.update(rd.group_entries(rd.u64(o), rd.u64(o + 8)))
            elif t == 0x0006:  # link message (compact new-style group)
                flags = rd.u8(o + 1)
                p = o + 2
                ltype = 0
                if flags & 8:
                    ltype = rd.u8(p)
                    p += 1
                if flags & 4:
                    p += 8
                if flags & 16:
                    p += 1
                ln = 1 << (flags & 3)
                nlen = int.from_bytes(rd.b[p:p + ln], "little")
                p += ln
                name = bytes(rd.b[p:p + nlen]).decode("utf-8", "replace")
                p += nlen
                if ltype == 0:
                    out[name] = rd.u64(p)
            elif t == 0x0002:
                if rd.u64(o + 2 + (8 if rd.u8(o + 1) & 1 else 0)) != UNDEF:
                    raise H5Unsupported("dense link storage (file written with libver='latest')")
        return out

    def keys(self):
        return list(self._children().keys())

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group):
                raise KeyError(path)
            ch = node._children()
            if part not in ch:
                raise KeyError(path)
            obj = _Object(node._rd, ch[part], (node.name.rstrip("/") + "/" + part))
            node = (Group if obj._is_group() else Dataset)(node._rd, ch[part], obj.name)
        return node


class File(Group):
    def __init__(self, path):
        with open(path, "rb") as f:
            data = f.read()
        if find_superblock(data) < 0:
            raise ValueError("%s is not an HDF5 file (no superblock signature at 0, 512, 1024, ...)" % path)
        rd = _guarded(_Reader)(data)  # (bytes: slicing copies only the few bytes asked for; np.frombuffer reads in place)
        Group.__init__(self, rd, rd.root_header, "/")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def read_keras_weights(path):
    """-> {layer_group: {weight_name: array}} (the input of keras_names.keras_to_tensors), from a Keras-2.3.1 save_weights /
    model.save file: group 'model_weights' if present, attributes layer_names / weight_names (plain or split into chunks)."""
    def names(g, key):
        if key in g.attrs:
            vals = list(np.atleast_1d(g.attrs[key]))
        else:
            vals, i = [], 0
            while "%s%d" % (key, i) in g.attrs:
                vals.extend(np.atleast_1d(g.attrs["%s%d" % (key, i)]))
                i += 1
        return [v.decode("utf-8") if isinstance(v, (bytes, np.bytes_)) else str(v) for v in vals]
    f = File(path)
    g = f["model_weights"] if "model_weights" in f else f
    layers = OrderedDict()
    for lname in names(g, "layer_names"):
        grp = g[lname]
        ws = names(grp, "weight_names")
        if ws:
            layers[lname] = OrderedDict((w, grp[w].read()) for w in ws)
    return layers


# ------------------------------------------------------------------------------------------------ writer (same subset)
class _Writer(object):
    def __init__(self, leaf_k):
        self.buf = bytearray(96)
        self.leaf_k = leaf_k

    def alloc(self, n):
        off = (len(self.buf) + 7) & ~7
        self.buf.extend(b"\0" * (off + n - len(self.buf)))
        return off

    def put(self, off, data):
        self.buf[off:off + len(data)] = data


def _pad8(b):
    return b + b"\0" * ((-len(b)) % 8)


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == "f":
        exp, man = {2: (5, 10), 4: (8, 23), 8: (11, 52)}[dt.itemsize]
        bits = bytes([0x20, dt.itemsize * 8 - 1, 0])  # little-endian, IEEE normalisation "implied", sign bit location
        props = struct.pack("<HHBBBBI", 0, dt.itemsize * 8, man, exp, 0, man, (1 << (exp - 1)) - 1)
        return bytes([0x11]) + bits + struct.pack("<I", dt.itemsize) + props
    if dt.kind in "iu":
        bits = bytes([0x08 if dt.kind == "i" else 0x00, 0, 0])
        return bytes([0x10]) + bits + struct.pack("<I", dt.itemsize) + struct.pack("<HH", 0, dt.itemsize * 8)
    if dt.kind == "S":
        return bytes([0x13, 0x00, 0, 0]) + struct.pack("<I", dt.itemsize)  # null-terminated / padded, ASCII
    raise H5Unsupported("dtype %s" % dt)


def _space_msg(shape):
    return struct.pack("<BBBB4x", 1, len(shape), 0, 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)


def _attr_msg(name, value):
    arr = np.asarray(value)
    if arr.dtype.kind == "U":
        arr = np.char.encode(arr, "utf-8")
    if arr.dtype.kind == "S" and arr.dtype.itemsize == 0:
        arr = arr.astype("S1")
    nm = name.encode("utf-8") + b"\0"
    t, s = _dtype_msg(arr.dtype), _space_msg(arr.shape)
    body = struct.pack("<BBHHH", 1, 0, len(nm), len(t), len(s)) + _pad8(nm) + _pad8(t) + _pad8(s) + np.asarray(arr, order="C").tobytes()
    return 0x000C, body


def _header(w, msgs):
    """version-1 object header holding `msgs` = [(type, body)] in one chunk; returns its address"""
    blob = b""
    for t, body in msgs:
        body = _pad8(body)
        if len(body) > 65528:
            raise H5Unsupported("object header message of %d bytes (Keras splits long name lists into layer_names0, ...)" % len(body))
        blob += struct.pack("<HHB3x", t, len(body), 0) + body
    off = w.alloc(16 + len(blob))
    w.put(off, struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(blob)) + blob)
    return off


def _write_dataset(w, arr):
    arr = np.asarray(arr, order="C")  # (ascontiguousarray would turn a scalar into shape (1,))
    data = w.alloc(max(arr.nbytes, 1))
    w.put(data, arr.tobytes())
    msgs = [(0x0001, _space_msg(arr.shape)), (0x0003, _dtype_msg(arr.dtype)),
            (0x0005, struct.pack("<BBBB", 2, 2, 0, 0)),                               # fill value v2: late allocation, never written, undefined
            (0x0008, struct.pack("<BBQQ", 3, 1, data, arr.nbytes))]                   # layout v3, contiguous
    return _header(w, msgs)


def _write_group(w, children, attrs):
    """children: {name: ('group', children, attrs) | ('data', array)} -> (header address, btree, heap)"""
    names = sorted(children, key=lambda s: s.encode("utf-8"))
    if len(names) > 2 * w.leaf_k:
        raise H5Unsupported("group with %d members (leaf K %d)" % (len(names), w.leaf_k))
    addrs = {}
    for n in names:
        c = children[n]
        addrs[n] = _write_group(w, c[1], c[2]) if c[0] == "group" else (_write_dataset(w, c[1]), None, None)
    seg = bytearray(8)  # offset 0: the empty name
    offs = {}
    for n in names:
        offs[n] = len(seg)
        seg += _pad8(n.encode("utf-8") + b"\0")
    seg_addr = w.alloc(len(seg))
    w.put(seg_addr, bytes(seg))
    heap = w.alloc(32)
    w.put(heap, b"HEAP" + struct.pack("<B3xQQQ", 0, len(seg), 1, seg_addr))            # free-list head 1 = none
    snod = w.alloc(8 + 2 * w.leaf_k * 40)
    ent = b""
    for n in names:
        hdr, bt, hp = addrs[n]
        if bt is not None:
            ent += struct.pack("<QQII", offs[n], hdr, 1, 0) + struct.pack("<QQ", bt, hp)
        else:
            ent += struct.pack("<QQII16x", offs[n], hdr, 0, 0)
    w.put(snod, b"SNOD" + struct.pack("<BBH", 1, 0, len(names)) + ent)
    K = 16
    btree = w.alloc(24 + (2 * K + 1) * 8 + 2 * K * 8)
    last = offs[names[-1]] if names else 0
    w.put(btree, b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod, last))
    msgs = [(0x0011, struct.pack("<QQ", btree, heap))] + [_attr_msg(k, v) for k, v in attrs.items()]
    return _header(w, msgs), btree, heap


def _tree_insert(tree, parts, arr):
    if len(parts) == 1:
        tree[parts[0]] = ("data", np.asarray(arr))
    else:
        node = tree.setdefault(parts[0], ("group", OrderedDict(), OrderedDict()))
        _tree_insert(node[1], parts[1:], arr)


def _largest_group(children):
    n = len(children)
    for c in children.values():
        if c[0] == "group":
            n = max(n, _largest_group(c[1]))
    return n


def write_tree(path, children, attrs):
    """children: {name: ('group', children, attrs) | ('data', array)}, attrs: root attributes -> one HDF5 file (version-0
    superblock, old-style groups with ONE symbol-table node each: the group leaf K of the superblock is sized for the largest
    group of the tree)."""
    w = _Writer(leaf_k=max(4, (_largest_group(children) + 1) // 2))
    hdr, bt, hp = _write_group(w, children, attrs)
    eof = len(w.buf)
    sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, w.leaf_k, 16, 0) + struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQII", 0, hdr, 1, 0) + struct.pack("<QQ", bt, hp)
    assert len(sb) == 96
    w.put(0, sb)
    with open(path, "wb") as f:
        f.write(bytes(w.buf))


def _weights_group(layers):
    """{layer_group: {weight_name: array}} -> (children, attrs) of a Keras 'model_weights' group (= the root of a save_weights file)"""
    root = OrderedDict()
    for lname, ws in layers.items():
        sub = OrderedDict()
        for wname, arr in ws.items():
            _tree_insert(sub, wname.split("/"), arr)
        wn = np.array([n.encode("utf-8") for n in ws], dtype="S") if ws else np.zeros((0,), "S1")
        root[lname] = ("group", sub, OrderedDict(weight_names=wn))
    attrs = OrderedDict(layer_names=np.array([n.encode("utf-8") for n in layers], dtype="S"), backend=np.bytes_(b"tensorflow"),
                        keras_version=np.bytes_(b"2.3.1"))
    return root, attrs


def write_keras_weights(path, layers, root_attrs=None):
    """Write {layer_group: {weight_name: array}} in the layout of Keras-2.3.1 `save_weights` (layer_names / weight_names
    attributes, nested groups for the '/' of weight names, contiguous float32 datasets)."""
    root, attrs = _weights_group(layers)
    attrs.update(root_attrs or {})
    write_tree(path, root, attrs)


def write_keras_model(path, layers, optimizer_weights=None, root_attrs=None):
    """The layout of Keras-2.3.1 `model.save` (keras/engine/saving.py _serialize_model): root attributes (keras_version, backend,
    training_config, ... = root_attrs), group 'model_weights' (as write_keras_weights) and -- optimizer_weights: OrderedDict
    {weight name: array} in the optimizer's own order -- group 'optimizer_weights' with the attribute `weight_names` and one
    dataset per name ('/' nests groups)."""
    mw, mw_attrs = _weights_group(layers)
    root = OrderedDict(model_weights=("group", mw, mw_attrs))
    if optimizer_weights:
        ow = OrderedDict()
        for n, arr in optimizer_weights.items():
            _tree_insert(ow, n.split("/"), arr)
        root["optimizer_weights"] = ("group", ow, OrderedDict(weight_names=np.array([n.encode("utf-8") for n in optimizer_weights], dtype="S")))
    attrs = OrderedDict(backend=np.bytes_(b"tensorflow"), keras_version=np.bytes_(b"2.3.1"))
    attrs.update(root_attrs or {})
    write_tree(path, root, attrs)


def read_optimizer_weights(path):
    """-> (OrderedDict {weight name: array} in file order or None when the file has no 'optimizer_weights' group, root attributes)"""
    f = File(path)
    if "optimizer_weights" not in f:
        return None, f.attrs
    g = f["optimizer_weights"]
    if "weight_names" in g.attrs:
        vals = list(np.atleast_1d(g.attrs["weight_names"]))
    else:
        vals, i = [], 0
        while "weight_names%d" % i in g.attrs:
            vals.extend(np.atleast_1d(g.attrs["weight_names%d" % i]))
            i += 1
    names = [v.decode("utf-8") if isinstance(v, (bytes, np.bytes_)) else str(v) for v in vals]
    return OrderedDict((n, g[n].read()) for n in names), f.attrs
