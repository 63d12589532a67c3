"""A minimal HDF5 reader - just enough to read the ``.hdf5`` mirror of the matched instance-id maps.

The reference's mask-matching step writes every id map twice (/root/reference/Mask2Former_sample/match_seg.py:140-143):
``<img>.npy`` and ``h5py.File(<img>.hdf5, 'w').create_dataset('cp_instance_id_segmaps', data=output)``.  This image's
interpreter has no h5py, so rounds 1-3 read the ``.npy`` files only; this module reads the other half without the
library: the file-format subset h5py / libhdf5 produce for such files (and for the same dataset inside a BlenderProc
container): superblock version 0, version-1 object headers with continuation blocks, old-style groups (symbol-table
B-tree + local heap), contiguous, compact and chunked layouts (version-1 chunk B-trees of any depth), deflate and shuffle
filters, little- or big-endian fixed-point and IEEE floating-point element types.  Anything else raises
``NotImplementedError`` with the name of the feature, never returns wrong data silently.

Checked against files written by the real library (tests/golden/hdf5/*.hdf5, generated with h5py 3.3.0 / HDF5 1.10.6 by
tests/golden/make_hdf5_golden.py exactly as the reference writes them).  Format: "HDF5 File Format Specification
Version 2.0" (sections II.A superblock, III.A B-trees, III.B/III.C symbol tables and heaps, IV.A object headers).
"""
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class _File:
    def __init__(self, data):
        self.b = data
        base = next((o for o in (0, 512, 1024, 2048, 4096) if data[o:o + 8] == SIGNATURE), None)
        if base is None:
            raise ValueError("not an HDF5 file (no signature)")
        v = data[base + 8]
        if v not in (0, 1):
            raise NotImplementedError(f"HDF5 superblock version {v} (only 0 / 1: files written with libver='earliest')")
        self.O, self.L = data[base + 13], data[base + 14]
        if (self.O, self.L) != (8, 8):
            raise NotImplementedError(f"HDF5 offsets / lengths of {self.O} / {self.L} bytes")
        p = base + 24 + (4 if v == 1 else 0)
        self.base_addr = self.u(p, 8)
        p += 4 * 8                                      # base address, free-space info, end of file, driver info
        # root group symbol-table entry: link name offset, object header address, cache type, reserved, scratch pad
        self.root_header = self.u(p + 8, 8)
        self.root_cache = self.u(p + 16, 4)
        self.root_btree, self.root_heap = self.u(p + 24, 8), self.u(p + 32, 8)

    def u(self, off, n):
        return int.from_bytes(self.b[off:off + n], "little")

    # ---- object headers (version 1) -----------------------------------------------------------------------------
    def messages(self, addr):
        """-> [(type, flags, bytes)] of the version-1 object header at ``addr`` (continuation blocks followed)."""
        a = addr + self.base_addr
        if self.b[a:a + 4] == b"OHDR":
            raise NotImplementedError("version-2 object headers (files written with libver='latest')")
        if self.b[a] != 1:
            raise NotImplementedError(f"object header version {self.b[a]}")
        n_msg, size = self.u(a + 2, 2), self.u(a + 8, 4)
        blocks, out = [(a + 16, size)], []
        while blocks and len(out) < n_msg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < n_msg:
                t, sz, fl = self.u(p, 2), self.u(p + 2, 2), self.b[p + 4]
                body = self.b[p + 8:p + 8 + sz]
                if t == 0x0010:                          # continuation: offset, length
                    blocks.append((self.u(p + 8, 8) + self.base_addr, self.u(p + 16, 8)))
                out.append((t, fl, body))
                p += 8 + sz
        return out

    # ---- old-style groups ------------------------------------------------------------------------------------------
    def group_entries(self, btree, heap):
        """{name: object header address} of a group stored as a symbol-table B-tree + local heap."""
        h = heap + self.base_addr
        if self.b[h:h + 4] != b"HEAP":
            raise ValueError("corrupt local heap")
        seg = self.u(h + 24, 8) + self.base_addr
        out = {}

        def name_at(off):
            s = seg + off
            return self.b[s:self.b.index(b"\0", s)].decode()

        def walk(node):
            p = node + self.base_addr
            if self.b[p:p + 4] == b"TREE":
                if self.b[p + 4] != 0:
                    raise ValueError("group B-tree of the wrong node type")
                used = self.u(p + 6, 2)
                q = p + 8 + 16 + 8                        # signature, type, level, used, two siblings, key 0
                for _ in range(used):
                    walk(self.u(q, 8))
                    q += 16                               # child pointer + next key
            elif self.b[p:p + 4] == b"SNOD":
                n = self.u(p + 6, 2)
                q = p + 8
                for _ in range(n):
                    out[name_at(self.u(q, 8))] = self.u(q + 8, 8)
                    q += 40
            else:
                raise ValueError("corrupt group B-tree")
        walk(btree)
        return out

    def children(self, header_addr, cached=None):
        if cached is not None:
            return self.group_entries(*cached)
        for t, _, body in self.messages(header_addr):
            if t == 0x0011:                               # symbol table message: B-tree address, local heap address
                return self.group_entries(int.from_bytes(body[:8], "little"), int.from_bytes(body[8:16], "little"))
            if t in (0x0002, 0x0006):
                raise NotImplementedError("new-style groups (link messages; files written with libver='latest')")
        return None

    def find(self, path):
        cur, cached = self.root_header, ((self.root_btree, self.root_heap) if self.root_cache == 1 else None)
        for part in [p for p in path.split("/") if p]:
            kids = self.children(cur, cached)
            if kids is None or part not in kids:
                raise KeyError(f"{path!r}: no object {part!r}" + (f" (have {sorted(kids)})" if kids else ""))
            cur, cached = kids[part], None
        return cur

    # ---- datasets -------------------------------------------------------------------------------------------------
    @staticmethod
    def dtype_of(body):
        cls, ver = body[0] & 15, body[0] >> 4
        bits0, size = body[1], int.from_bytes(body[4:8], "little")
        if ver not in (1, 2, 3):
            raise NotImplementedError(f"datatype message version {ver}")
        order = ">" if bits0 & 1 else "<"
        if cls == 0:
            return np.dtype(f"{order}{'i' if bits0 & 8 else 'u'}{size}")
        if cls == 1:
            if size not in (2, 4, 8):
                raise NotImplementedError(f"{size}-byte floating point")
            return np.dtype(f"{order}f{size}")
        raise NotImplementedError(f"datatype class {cls} (only fixed-point and floating-point elements)")

    @staticmethod
    def shape_of(body):
        ver, rank, flags = body[0], body[1], body[2]
        p = 8 if ver == 1 else 4
        if ver not in (1, 2):
            raise NotImplementedError(f"dataspace message version {ver}")
        return tuple(int.from_bytes(body[p + 8 * i:p + 8 * i + 8], "little") for i in range(rank))

    @staticmethod
    def filters_of(body):
        ver, n = body[0], body[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid, name_len = int.from_bytes(body[p:p + 2], "little"), 0
            if ver == 1 or fid >= 256:
                name_len = int.from_bytes(body[p + 2:p + 4], "little")
                flags_at = p + 4
            else:
                flags_at = p + 2
            n_cd = int.from_bytes(body[flags_at + 2:flags_at + 4], "little")
            q = flags_at + 4 + (((name_len + 7) // 8 * 8) if ver == 1 else name_len)
            cd = [int.from_bytes(body[q + 4 * i:q + 4 * i + 4], "little") for i in range(n_cd)]
            q += 4 * n_cd
            if ver == 1 and n_cd % 2:
                q += 4
            out.append((fid, cd))
            p = q
        return out

    def read(self, header_addr):
        dtype = shape = layout = None
        filters = []
        for t, _, body in self.messages(header_addr):
            if t == 0x0001:
                shape = self.shape_of(body)
            elif t == 0x0003:
                dtype = self.dtype_of(body)
            elif t == 0x0008:
                layout = body
            elif t == 0x000B:
                filters = self.filters_of(body)
        if dtype is None or shape is None or layout is None:
            raise KeyError("the object is not a dataset")
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if layout[0] != 3:
            raise NotImplementedError(f"data layout message version {layout[0]}")
        cls = layout[1]
        if cls == 0:                                      # compact: the data sits in the header
            size = int.from_bytes(layout[2:4], "little")
            raw = bytes(layout[4:4 + size])
        elif cls == 1:                                    # contiguous
            addr, size = int.from_bytes(layout[2:10], "little"), int.from_bytes(layout[10:18], "little")
            if addr == UNDEF:
                return np.zeros(shape, dtype.newbyteorder("="))          # never written: fill value 0
            a = addr + self.base_addr
            raw = self.b[a:a + size]
        elif cls == 2:
            return self.read_chunked(layout, shape, dtype, filters)
        else:
            raise NotImplementedError(f"data layout class {cls}")
        return np.frombuffer(raw, dtype, count=n).reshape(shape).astype(dtype.newbyteorder("="))

    def read_chunked(self, layout, shape, dtype, filters):
        nd = layout[2] - 1                                # the last "dimension" is the element size
        btree = int.from_bytes(layout[3:11], "little")
        chunk = tuple(int.from_bytes(layout[11 + 4 * i:15 + 4 * i], "little") for i in range(nd))
        if nd != len(shape):
            raise ValueError("chunk rank differs from the dataset's")
        for fid, _ in filters:
            if fid not in (1, 2):
                raise NotImplementedError(f"HDF5 filter {fid} (only deflate and shuffle)")
        out = np.zeros(shape, dtype.newbyteorder("="))
        if btree == UNDEF:
            return out
        isz = dtype.itemsize
        n_chunk = int(np.prod(chunk, dtype=np.int64))

        def decode(raw, mask):
            for k in range(len(filters) - 1, -1, -1):    # undo the pipeline back to front
                if mask >> k & 1:
                    continue                              # the filter was skipped for this chunk
                fid = filters[k][0]
                if fid == 1:
                    raw = zlib.decompress(raw)
                else:                                     # shuffle: byte planes -> elements
                    raw = np.frombuffer(raw, np.uint8).reshape(isz, -1).T.tobytes()
            return np.frombuffer(raw, dtype, count=n_chunk).reshape(chunk)

        def walk(node):
            p = node + self.base_addr
            if self.b[p:p + 4] != b"TREE" or self.b[p + 4] != 1:
                raise ValueError("corrupt chunk B-tree")
            level, used = self.b[p + 5], self.u(p + 6, 2)
            q = p + 8 + 16
            key = 8 + 8 * (nd + 1)
            for _ in range(used):
                size, mask = self.u(q, 4), self.u(q + 4, 4)
                offs = tuple(self.u(q + 8 + 8 * i, 8) for i in range(nd))
                child = self.u(q + key, 8)
                if level:
                    walk(child)
                else:
                    a = child + self.base_addr
                    block = decode(self.b[a:a + size], mask)
                    sel = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, shape))
                    out[sel] = block[tuple(slice(0, s.stop - s.start) for s in sel)]
                q += key + 8
        walk(btree)
        return out


def read_dataset(path, name):
    """The dataset ``name`` (a path inside the file, e.g. 'cp_instance_id_segmaps') of the HDF5 file ``path`` as a numpy
    array in native byte order."""
    with open(path, "rb") as f:
        data = f.read()
    hf = _File(data)
    return hf.read(hf.find(name))


def list_objects(path, group="/"):
    """Names of the objects in a group of the file (old-style groups)."""
    with open(path, "rb") as f:
        data = f.read()
    hf = _File(data)
    g = hf.find(group)
    kids = hf.children(g, (hf.root_btree, hf.root_heap) if (g == hf.root_header and hf.root_cache == 1) else None)
    return sorted(kids or {})

