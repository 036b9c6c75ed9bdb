"""h5lite -- the subset of HDF5 that popVAE's files need, in plain Python + numpy (no h5py, no libhdf5).

The reference writes two HDF5 files through h5py / Keras: ``<infile>.popvae.hdf5`` (popvae.py:241-248: datasets
``derived_counts`` float64 [N, L] and ``samples``, variable-length UTF-8 strings) and ``<out>_weights.hdf5``
(popvae.py:429-435: Keras' ``model_weights`` group), and reads both back (:205-209, :508).  h5py is not importable in the
interpreter this package runs in, so the format is restated here from the published HDF5 File Format Specification (v3):

writer  superblock v2, version-2 object headers, "new style" groups with compact storage (Link Info + one Link message
        per member in the object header), contiguous datasets, attributes (message v3), fixed-length and variable-length
        strings (global heap collections).  h5py / libhdf5 >= 1.8 read these files (checked by tools/check_h5lite.py
        under an interpreter that has h5py).
reader  both superblock generations; "old style" groups (symbol table: v1 B-tree + local heap + symbol nodes -- what
        h5py and Keras write by default) and compact new-style groups; version-1 and version-2 object headers with
        continuation blocks; contiguous and compact dataset layouts; integer / IEEE float / fixed string /
        variable-length string types; attributes.  Anything else (chunked or filtered datasets, dense groups, shared
        messages) raises ``H5Unsupported`` -- loudly, never a wrong array.
"""
from __future__ import annotations

import struct

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
_M = 0xFFFFFFFF


class H5Unsupported(RuntimeError):
    pass


# ------------------------------------------------------------------------------------------------
# Jenkins lookup3 (hashlittle), the metadata checksum of format version 2 structures
# ------------------------------------------------------------------------------------------------
def _rot(x, k):
    return ((x << k) | (x >> (32 - k))) & _M


def lookup3(data, initval=0):
    n = len(data)
    a = b = c = (0xDEADBEEF + n + initval) & _M
    i = 0
    while n > 12:
        w0, w1, w2 = struct.unpack_from("<III", data, i)
        a, b, c = (a + w0) & _M, (b + w1) & _M, (c + w2) & _M
        a = (a - c) & _M; a ^= _rot(c, 4); c = (c + b) & _M
        b = (b - a) & _M; b ^= _rot(a, 6); a = (a + c) & _M
        c = (c - b) & _M; c ^= _rot(b, 8); b = (b + a) & _M
        a = (a - c) & _M; a ^= _rot(c, 16); c = (c + b) & _M
        b = (b - a) & _M; b ^= _rot(a, 19); a = (a + c) & _M
        c = (c - b) & _M; c ^= _rot(b, 4); b = (b + a) & _M
        i += 12
        n -= 12
    if n == 0:
        return c
    w0, w1, w2 = struct.unpack("<III", bytes(data[i:]) + b"\0" * (12 - n))
    a, b, c = (a + w0) & _M, (b + w1) & _M, (c + w2) & _M
    c ^= b; c = (c - _rot(b, 14)) & _M
    a ^= c; a = (a - _rot(c, 11)) & _M
    b ^= a; b = (b - _rot(a, 25)) & _M
    c ^= b; c = (c - _rot(b, 16)) & _M
    a ^= c; a = (a - _rot(c, 4)) & _M
    b ^= a; b = (b - _rot(a, 14)) & _M
    c ^= b; c = (c - _rot(b, 24)) & _M
    return c


# ------------------------------------------------------------------------------------------------
# in-memory tree handed to the writer / returned by the reader
# ------------------------------------------------------------------------------------------------
class Group:
    def __init__(self, attrs=None):
        self.attrs = dict(attrs or {})
        self.children = {}

    def group(self, name, attrs=None):
        g = self.children.get(name)
        if g is None:
            g = self.children[name] = Group(attrs)
        return g

    def dataset(self, name, data, attrs=None):
        d = self.children[name] = Dataset(data, attrs)
        return d

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            node = node.children[part]
        return node

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False


class Dataset:
    """data: numeric ndarray, ndarray of dtype 'S<n>' (fixed-length strings), or a list / object array of str
    (variable-length UTF-8 strings, what h5py.string_dtype() gives)."""

    def __init__(self, data, attrs=None):
        self.data = data
        self.attrs = dict(attrs or {})


# ------------------------------------------------------------------------------------------------
# datatype / dataspace encodings
# ------------------------------------------------------------------------------------------------
def _is_vlen_str(data):
    if isinstance(data, str):
        return True
    if isinstance(data, (list, tuple)):
        return len(data) > 0 and all(isinstance(x, str) for x in data)
    return isinstance(data, np.ndarray) and data.dtype == object


def _dtype_msg(dt):
    """Datatype message body for a numpy dtype (or 'vlen_str')."""
    if isinstance(dt, str) and dt == "vlen_str":
        base = struct.pack("<B3BI", 0x10 | 0, 0, 0, 0, 1) + struct.pack("<HH", 0, 8)  # unsigned 8-bit integer
        # class 9, type = string (1), padding null-terminated (0), character set UTF-8 (1)
        return struct.pack("<B3BI", 0x10 | 9, 0x01, 0x01, 0, 16) + base
    dt = np.dtype(dt)
    if dt.kind == "f" and dt.itemsize in (4, 8):
        if dt.itemsize == 4:
            props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
            return struct.pack("<B3BI", 0x10 | 1, 0x20, 31, 0, 4) + props
        props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        return struct.pack("<B3BI", 0x10 | 1, 0x20, 63, 0, 8) + props
    if dt.kind in "iu":
        bits = 0x08 if dt.kind == "i" else 0
        return struct.pack("<B3BI", 0x10 | 0, bits, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
    if dt.kind == "S":
        return struct.pack("<B3BI", 0x10 | 3, 0x01, 0, 0, max(1, dt.itemsize))  # null-padded ASCII (numpy 'S')
    raise H5Unsupported(f"h5lite cannot write dtype {dt}")


def _space_msg(shape):
    if shape == ():
        return struct.pack("<BBBB", 2, 0, 0, 0)  # scalar
    return struct.pack("<BBBB", 2, len(shape), 0, 1) + b"".join(struct.pack("<Q", int(d)) for d in shape)


# ------------------------------------------------------------------------------------------------
# writer
# ------------------------------------------------------------------------------------------------
class _Out:
    def __init__(self, f):
        self.f = f
        self.pos = 48  # behind the version-2 superblock

    def put(self, raw):
        """Append `raw` (bytes or a C-contiguous ndarray) at the next 8-byte boundary; returns its address."""
        self.pos = (self.pos + 7) & ~7
        addr = self.pos
        self.f.seek(addr)
        if isinstance(raw, np.ndarray):
            mv = memoryview(np.ascontiguousarray(raw)).cast("B")
            step = 1 << 26
            for i in range(0, len(mv), step):
                self.f.write(mv[i:i + step])
            self.pos += len(mv)
        else:
            self.f.write(raw)
            self.pos += len(raw)
        return addr


def _vlen_payload(out, strings):
    """Global heap collection holding `strings`; returns the dataset's raw element array (16 bytes per string)."""
    enc = [s.encode("utf-8") for s in strings]
    body = bytearray()
    for i, b in enumerate(enc, start=1):
        body += struct.pack("<HHIQ", i, 1, 0, len(b)) + b + b"\0" * ((-len(b)) % 8)
    size = max(4096, 16 + len(body) + 16)
    size = (size + 7) & ~7
    free = size - 16 - len(body)
    col = bytearray(b"GCOL" + struct.pack("<B3xQ", 1, size)) + body
    col += struct.pack("<HHIQ", 0, 0, 0, free) + b"\0" * (free - 16)
    addr = out.put(bytes(col))
    return b"".join(struct.pack("<IQI", len(b), addr, i) for i, b in enumerate(enc, start=1))


def _encode_value(out, value):
    """-> (datatype message, dataspace message, raw data bytes or ndarray)."""
    if _is_vlen_str(value):
        if isinstance(value, str):
            return _dtype_msg("vlen_str"), _space_msg(()), _vlen_payload(out, [value])
        arr = np.asarray(value, dtype=object)
        return _dtype_msg("vlen_str"), _space_msg(arr.shape), _vlen_payload(out, [str(s) for s in arr.reshape(-1)])
    if isinstance(value, bytes):
        value = np.array(value, dtype=f"S{max(1, len(value))}")
    arr = np.asarray(value)
    if arr.dtype.kind == "U":
        arr = np.char.encode(arr, "utf-8")
    if arr.dtype.kind == "S" and arr.dtype.itemsize == 0:
        arr = arr.astype("S1")
    if arr.dtype.byteorder == ">":
        arr = arr.astype(arr.dtype.newbyteorder("<"))
    return _dtype_msg(arr.dtype), _space_msg(arr.shape), np.ascontiguousarray(arr)


def _msg(mtype, body, flags=0):
    return struct.pack("<BHB", mtype, len(body), flags) + body


def _attr_msgs(out, attrs):
    msgs = []
    for name, value in attrs.items():
        dt, sp, raw = _encode_value(out, value)
        if isinstance(raw, np.ndarray):
            raw = raw.tobytes()
        nm = name.encode("utf-8") + b"\0"
        body = struct.pack("<BBHHHB", 3, 0, len(nm), len(dt), len(sp), 0) + nm + dt + sp + raw
        if len(body) > 65000:
            raise H5Unsupported(f"attribute {name}: {len(body)} bytes do not fit a compact attribute message")
        msgs.append(_msg(0x0C, body))
    return msgs


def _object_header(out, msgs):
    body = b"".join(msgs)
    head = b"OHDR" + struct.pack("<BB", 2, 0x02) + struct.pack("<I", len(body))
    blob = head + body
    return out.put(blob + struct.pack("<I", lookup3(blob)))


def _emit(out, node):
    if isinstance(node, Dataset):
        dt, sp, raw = _encode_value(out, node.data)
        nbytes = raw.nbytes if isinstance(raw, np.ndarray) else len(raw)
        addr = out.put(raw) if nbytes else UNDEF
        msgs = [_msg(0x01, sp), _msg(0x03, dt, flags=1), _msg(0x05, struct.pack("<BB", 3, 0x0A)),
                _msg(0x08, struct.pack("<BBQQ", 3, 1, addr, nbytes))]
        return _object_header(out, msgs + _attr_msgs(out, node.attrs))
    links = []
    for name, child in node.children.items():
        nm = name.encode("utf-8")
        if len(nm) > 255:
            raise H5Unsupported("link names above 255 bytes")
        links.append(_msg(0x06, struct.pack("<BBB", 1, 0, len(nm)) + nm + struct.pack("<Q", _emit(out, child))))
    msgs = [_msg(0x02, struct.pack("<BBQQ", 0, 0, UNDEF, UNDEF)),                      # Link Info: compact storage
            _msg(0x0A, struct.pack("<BBHH", 0, 1, max(8, len(links)), 6))]             # Group Info
    return _object_header(out, msgs + links + _attr_msgs(out, node.attrs))


def write(path, root):
    """Write the tree under `root` (a Group) to `path`."""
    with open(path, "wb") as f:
        out = _Out(f)
        root_addr = _emit(out, root)
        eof = (out.pos + 7) & ~7
        f.seek(eof - 1)
        f.write(b"\0")  # the file ends where the superblock says it does
        sb = SIG + struct.pack("<BBBB", 2, 8, 8, 0) + struct.pack("<QQQQ", 0, UNDEF, eof, root_addr)
        f.seek(0)
        f.write(sb + struct.pack("<I", lookup3(sb)))


# ------------------------------------------------------------------------------------------------
# reader
# ------------------------------------------------------------------------------------------------
class _In:
    def __init__(self, path):
        self.mm = np.memmap(path, dtype=np.uint8, mode="r")
        self.base = 0

    def raw(self, addr, n):
        addr += self.base
        if addr < 0 or addr + n > len(self.mm):
            raise H5Unsupported(f"address {addr}+{n} beyond the end of the file ({len(self.mm)} bytes)")
        return self.mm[addr:addr + n]

    def b(self, addr, n):
        return bytes(self.raw(addr, n))


def _parse_dtype(buf):
    """-> (descriptor, bytes consumed).  descriptor: numpy dtype, or ('vlen_str',) / ('S', size)."""
    cv, b0, b1, b2, size = struct.unpack_from("<B3BI", buf, 0)
    cls, ver = cv & 0x0F, cv >> 4
    if cls == 0:
        order = ">" if b0 & 1 else "<"
        return np.dtype(f"{order}{'i' if b0 & 8 else 'u'}{size}"), 12
    if cls == 1:
        if size not in (2, 4, 8):
            raise H5Unsupported(f"float of {size} bytes")
        return np.dtype(f"{'>' if b0 & 1 else '<'}f{size}"), 20
    if cls == 3:
        return ("S", size), 8
    if cls == 9:
        if (b0 & 0x0F) != 1:
            raise H5Unsupported("variable-length sequences (only variable-length strings are read)")
        _, used = _parse_dtype(buf[8:])
        return ("vlen_str",), 8 + used
    raise H5Unsupported(f"datatype class {cls} (version {ver})")


def _parse_space(buf):
    ver = buf[0]
    if ver == 1:
        rank, flags = buf[1], buf[2]
        off = 8
    elif ver == 2:
        rank, flags, typ = buf[1], buf[2], buf[3]
        off = 4
        if typ == 2:
            return None  # null dataspace
    else:
        raise H5Unsupported(f"dataspace message version {ver}")
    return tuple(struct.unpack_from("<Q", buf, off + 8 * i)[0] for i in range(rank))


def _decode(inp, dt, shape, raw):
    n = int(np.prod(shape)) if shape else 1
    if isinstance(dt, np.dtype):
        a = np.frombuffer(raw, dtype=dt, count=n).reshape(shape)
        return a.astype(dt.newbyteorder("=")) if dt.byteorder == ">" else a
    if dt[0] == "S":
        return np.frombuffer(raw, dtype=f"S{dt[1]}", count=n).reshape(shape)
    out = np.empty(n, dtype=object)
    for i in range(n):
        ln, addr, idx = struct.unpack_from("<IQI", raw, 16 * i)
        out[i] = _heap_object(inp, addr, idx)[:ln].decode("utf-8") if ln else ""
    return out.reshape(shape) if shape else out[0]


def _heap_object(inp, addr, idx):
    head = inp.b(addr, 16)
    if head[:4] != b"GCOL":
        raise H5Unsupported("bad global heap collection signature")
    size = struct.unpack_from("<Q", head, 8)[0]
    col = inp.b(addr, size)
    pos = 16
    while pos + 16 <= size:
        i, _, _, osz = struct.unpack_from("<HHIQ", col, pos)
        if i == idx:
            return col[pos + 16:pos + 16 + osz]
        if i == 0:
            break
        pos += 16 + ((osz + 7) & ~7)
    raise H5Unsupported(f"global heap object {idx} not found")


def _messages(inp, addr):
    """All (type, flags, body) messages of the object header at `addr`, continuation blocks followed."""
    head = inp.b(addr, 16)
    out = []
    if head[:4] == b"OHDR":
        flags = head[5]
        pos = 6
        if flags & 0x20:
            pos += 16
        if flags & 0x10:
            pos += 4
        szw = 1 << (flags & 3)
        hd = inp.b(addr, pos + 8)
        chunk = int.from_bytes(hd[pos:pos + szw], "little")
        pos += szw
        blocks = [(addr + pos, chunk)]
        track = bool(flags & 0x04)
        while blocks:
            a, n = blocks.pop(0)
            buf = inp.b(a, n)
            p = 0
            while p + 4 <= n:
                t, sz, fl = struct.unpack_from("<BHB", buf, p)
                p += 4 + (2 if track else 0)
                body = buf[p:p + sz]
                p += sz
                if t == 0x10:
                    ca, cl = struct.unpack_from("<QQ", body, 0)
                    if inp.b(ca, 4) != b"OCHK":
                        raise H5Unsupported("bad object header continuation block")
                    blocks.append((ca + 4, cl - 8))
                elif t != 0:
                    out.append((t, fl, body))
        return out
    ver, _, nmsg, _, hsize = struct.unpack_from("<BBHII", head, 0)
    if ver != 1:
        raise H5Unsupported(f"object header version {ver}")
    blocks = [(addr + 16, hsize)]
    while blocks:
        a, n = blocks.pop(0)
        buf = inp.b(a, n)
        p = 0
        while p + 8 <= n:
            t, sz, fl = struct.unpack_from("<HHB", buf, p)
            body = buf[p + 8:p + 8 + sz]
            p += 8 + sz
            if t == 0x10:
                ca, cl = struct.unpack_from("<QQ", body, 0)
                blocks.append((ca, cl))
            elif t != 0:
                out.append((t, fl, body))
    return out


def _attr(inp, body):
    ver = body[0]
    nsz, dsz, ssz = struct.unpack_from("<HHH", body, 2)
    pos = 8 + (1 if ver == 3 else 0)
    pad = (lambda x: (x + 7) & ~7) if ver == 1 else (lambda x: x)
    if ver not in (1, 2, 3):
        raise H5Unsupported(f"attribute message version {ver}")
    if body[1] & 0x03 and ver >= 2:
        raise H5Unsupported("attributes with shared datatype / dataspace")
    name = body[pos:pos + nsz].split(b"\0")[0].decode("utf-8")
    pos += pad(nsz)
    dt, _ = _parse_dtype(body[pos:pos + dsz])
    pos += pad(dsz)
    shape = _parse_space(body[pos:pos + ssz])
    pos += pad(ssz)
    if shape is None:
        return name, None
    return name, _decode(inp, dt, shape, body[pos:])


def _old_group_members(inp, btree, heap):
    hh = inp.b(heap, 32)
    if hh[:4] != b"HEAP":
        raise H5Unsupported("bad local heap signature")
    dsize, _, daddr = struct.unpack_from("<QQQ", hh, 8)
    names = inp.b(daddr, dsize)
    out = []

    def walk(addr):
        nh = inp.b(addr, 24)
        if nh[:4] != b"TREE" or nh[4] != 0:
            raise H5Unsupported("bad group B-tree node")
        level, used = nh[5], struct.unpack_from("<H", nh, 6)[0]
        body = inp.b(addr + 24, (2 * used + 1) * 8)
        for i in range(used):
            child = struct.unpack_from("<Q", body, 8 + 16 * i)[0]
            if level > 0:
                walk(child)
                continue
            sh = inp.b(child, 8)
            if sh[:4] != b"SNOD":
                raise H5Unsupported("bad symbol table node")
            nsym = struct.unpack_from("<H", sh, 6)[0]
            ents = inp.b(child + 8, 40 * nsym)
            for k in range(nsym):
                noff, oaddr = struct.unpack_from("<QQ", ents, 40 * k)
                out.append((names[noff:names.index(b"\0", noff)].decode("utf-8"), oaddr))

    walk(btree)
    return out


def _load(inp, addr, lazy):
    msgs = _messages(inp, addr)
    types = {t for t, _, _ in msgs}
    attrs = {}
    for t, fl, body in msgs:
        if t == 0x0C:
            k, v = _attr(inp, body)
            attrs[k] = v
    if 0x08 in types:  # dataset
        dt = shape = layout = None
        for t, fl, body in msgs:
            if fl & 0x02 and t in (0x01, 0x03):
                raise H5Unsupported("shared datatype / dataspace messages")
            if t == 0x03:
                dt, _ = _parse_dtype(body)
            elif t == 0x01:
                shape = _parse_space(body)
            elif t == 0x08:
                layout = body
            elif t == 0x0B:
                raise H5Unsupported("filtered (compressed) datasets")
        if layout[0] not in (3, 4):
            raise H5Unsupported(f"data layout message version {layout[0]}")
        if shape is None:
            return Dataset(None, attrs)
        n = int(np.prod(shape)) if shape else 1
        esz = dt.itemsize if isinstance(dt, np.dtype) else (dt[1] if dt[0] == "S" else 16)
        if layout[1] == 0:
            sz = struct.unpack_from("<H", layout, 2)[0]
            raw = layout[4:4 + sz]
        elif layout[1] == 1:
            a, sz = struct.unpack_from("<QQ", layout, 2)
            raw = b"" if a == UNDEF else inp.raw(a, min(sz, n * esz))
        else:
            raise H5Unsupported("chunked / virtual dataset layout (only contiguous and compact are read)")
        if len(raw) < n * esz:
            raise H5Unsupported("dataset without allocated storage")
        return Dataset(_decode(inp, dt, shape, raw), attrs)
    g = Group(attrs)
    members = []
    for t, fl, body in msgs:
        if t == 0x11:
            members += _old_group_members(inp, *struct.unpack_from("<QQ", body, 0))
        elif t == 0x02:
            fl2 = body[1]
            off = 2 + (8 if fl2 & 1 else 0)
            if struct.unpack_from("<Q", body, off)[0] != UNDEF:
                raise H5Unsupported("groups with dense link storage (fractal heap)")
        elif t == 0x06:
            lf = body[1]
            p = 2
            ltype = 0
            if lf & 0x08:
                ltype = body[p]; p += 1
            if lf & 0x04:
                p += 8
            if lf & 0x10:
                p += 1
            w = 1 << (lf & 3)
            ln = int.from_bytes(body[p:p + w], "little"); p += w
            nm = body[p:p + ln].decode("utf-8"); p += ln
            if ltype != 0:
                raise H5Unsupported("soft / external links")
            members.append((nm, struct.unpack_from("<Q", body, p)[0]))
    for nm, a in members:
        g.children[nm] = _load(inp, a, lazy)
    return g


def read(path):
    """Read the whole file into a Group tree (datasets as numpy arrays; large contiguous ones are views of a memory map)."""
    inp = _In(path)
    head = inp.b(0, 8)
    if head != SIG:
        raise H5Unsupported(f"{path}: not an HDF5 file (or a user block precedes the superblock)")
    ver = inp.b(8, 1)[0]
    if ver in (0, 1):
        sb = inp.b(0, 128)
        if sb[13] != 8 or sb[14] != 8:
            raise H5Unsupported("offsets / lengths that are not 8 bytes wide")
        p = 24 + (4 if ver == 1 else 0)
        base = struct.unpack_from("<Q", sb, p)[0]
        root = struct.unpack_from("<Q", sb, p + 32 + 8)[0]  # root symbol table entry: name offset, header address
    elif ver in (2, 3):
        sb = inp.b(0, 48)
        if sb[9] != 8 or sb[10] != 8:
            raise H5Unsupported("offsets / lengths that are not 8 bytes wide")
        base, _, _, root = struct.unpack_from("<QQQQ", sb, 12)
        if lookup3(sb[:44]) != struct.unpack_from("<I", sb, 44)[0]:
            raise H5Unsupported("superblock checksum mismatch")
    else:
        raise H5Unsupported(f"superblock version {ver}")
    inp.base = base
    return _load(inp, root, False)
