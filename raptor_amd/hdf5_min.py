"""A minimal, dependency-free reader for the HDF5 files rl-tools writes for its checkpoints
(SURVEY.md §5 "checkpoint / resume", §8(f) row 3): superblock version 0, old-style groups
(v1 B-tree + local heap + symbol-table nodes), version-1 object headers, contiguous
little-endian float datasets and variable-length string attributes (global heap).

That is exactly what HighFive/libhdf5 produce with default settings for
``checkpoint.h5`` (``/actor/layers/{0,1,2}/<param>/parameters`` datasets with string attributes
``type``, ``activation_function``, ``rows``/``cols`` or ``num_dims``/``dim_i``; ``/actor@meta``;
``/example/{input,output}``).  Anything else (chunking, filters, new-style groups, other
superblock versions) raises ``Hdf5FormatError`` — this is not a general HDF5 library.
"""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5FormatError(ValueError):
    pass


class Dataset:
    def __init__(self, shape, dtype, data, attrs):
        self.shape, self.dtype, self.attrs = shape, dtype, attrs
        self._data = data

    def numpy(self):
        return self._data.reshape(self.shape)


class Group:
    def __init__(self, children, attrs):
        self.children, self.attrs = children, attrs

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node.children:
                raise KeyError(path)
            node = node.children[part]
        return node

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False

    def walk(self, prefix=""):
        for name, child in self.children.items():
            p = f"{prefix}/{name}"
            yield p, child
            if isinstance(child, Group):
                yield from child.walk(p)


class File:
    def __init__(self, path):
        self.b = open(path, "rb").read()
        b = self.b
        if b[:8] != b"\x89HDF\r\n\x1a\n":
            raise Hdf5FormatError("not an HDF5 file")
        if b[8] != 0:
            raise Hdf5FormatError(f"superblock version {b[8]} not supported (only 0)")
        if b[13] != 8 or b[14] != 8:
            raise Hdf5FormatError("only 8-byte offsets/lengths supported")
        self.base = struct.unpack_from("<Q", b, 24)[0]
        # root group symbol table entry follows the four addresses (base, free space, eof, driver)
        root_header = struct.unpack_from("<Q", b, 24 + 32 + 8)[0]
        self.root = self._object(root_header)

    # ------------------------------------------------------------------ object headers
    def _messages(self, addr):
        b = self.b
        ver, _, nmsg, _refs, hsize = struct.unpack_from("<BBHII", b, addr)
        if ver != 1:
            raise Hdf5FormatError(f"object header version {ver} not supported (only 1)")
        blocks = [(addr + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            pos, size = blocks.pop(0)
            end = pos + size
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", b, pos)
                data = b[pos + 8: pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x0010:      # continuation
                    coff, clen = struct.unpack_from("<QQ", data, 0)
                    blocks.append((coff + self.base, clen))
                out.append((mtype, data))
        return out

    def _object(self, addr):
        msgs = self._messages(addr + self.base)
        attrs, stab, space, dtype, layout = {}, None, None, None, None
        for mtype, d in msgs:
            if mtype == 0x0011:
                stab = struct.unpack_from("<QQ", d, 0)
            elif mtype == 0x0001:
                space = self._dataspace(d)
            elif mtype == 0x0003:
                dtype = self._datatype(d)
            elif mtype == 0x0008:
                layout = self._layout(d)
            elif mtype == 0x000C:
                name, value = self._attribute(d)
                attrs[name] = value
        if stab is not None:
            return Group(self._group_children(*stab), attrs)
        if space is not None and dtype is not None and layout is not None:
            addr_, size = layout
            n = int(np.prod(space)) if space else 1
            np_dtype = self._numpy_dtype(dtype)
            if addr_ == UNDEF:
                data = np.zeros(n, np_dtype)
            else:
                data = np.frombuffer(self.b, np_dtype, n, addr_ + self.base).copy()
            return Dataset(tuple(space), np_dtype, data, attrs)
        raise Hdf5FormatError("object is neither an old-style group nor a contiguous dataset")

    # ------------------------------------------------------------------ groups
    def _heap_name(self, heap_addr, offset):
        b = self.b
        heap_addr += self.base
        if b[heap_addr:heap_addr + 4] != b"HEAP":
            raise Hdf5FormatError("local heap signature missing")
        data_addr = struct.unpack_from("<Q", b, heap_addr + 24)[0] + self.base
        end = b.index(b"\x00", data_addr + offset)
        return b[data_addr + offset:end].decode()

    def _group_children(self, btree_addr, heap_addr):
        children = {}

        def visit(addr):
            b = self.b
            addr += self.base
            if b[addr:addr + 4] == b"TREE":
                ntype, level, used = struct.unpack_from("<BBH", b, addr + 4)
                if ntype != 0:
                    raise Hdf5FormatError("unexpected B-tree node type")
                pos = addr + 8 + 16      # skip left/right sibling
                for i in range(used):
                    child = struct.unpack_from("<Q", b, pos + 8 + i * 16)[0]   # key, child, key, child ...
                    visit(child)
            elif b[addr:addr + 4] == b"SNOD":
                nsym = struct.unpack_from("<H", b, addr + 6)[0]
                for i in range(nsym):
                    e = addr + 8 + i * 40
                    name_off, header = struct.unpack_from("<QQ", b, e)
                    children[self._heap_name(heap_addr, name_off)] = self._object(header)
            else:
                raise Hdf5FormatError("unknown group node")
        visit(btree_addr)
        return children

    # ------------------------------------------------------------------ messages
    @staticmethod
    def _dataspace(d):
        ver, rank, flags = struct.unpack_from("<BBB", d, 0)
        if ver == 1:
            off = 8
        elif ver == 2:
            off = 4
        else:
            raise Hdf5FormatError(f"dataspace version {ver}")
        return list(struct.unpack_from(f"<{rank}Q", d, off)) if rank else []

    @staticmethod
    def _datatype(d):
        cls_ver, b0, b1, b2, size = struct.unpack_from("<BBBBI", d, 0)
        cls = cls_ver & 0x0F
        info = {"class": cls, "size": size, "bits": (b0, b1, b2)}
        if cls == 9:     # variable length: base type follows
            info["vlen_string"] = (b0 & 0x0F) == 1
        return info

    @staticmethod
    def _numpy_dtype(dt):
        if dt["class"] == 1 and dt["size"] in (4, 8) and (dt["bits"][0] & 1) == 0:
            return np.dtype("<f4" if dt["size"] == 4 else "<f8")
        if dt["class"] == 0 and (dt["bits"][0] & 1) == 0:
            signed = (dt["bits"][0] >> 3) & 1
            return np.dtype(("<i" if signed else "<u") + str(dt["size"]))
        raise Hdf5FormatError(f"unsupported datatype class {dt['class']} size {dt['size']}")

    @staticmethod
    def _layout(d):
        ver, cls = struct.unpack_from("<BB", d, 0)
        if ver != 3:
            raise Hdf5FormatError(f"data layout version {ver} not supported (only 3)")
        if cls == 1:
            return struct.unpack_from("<QQ", d, 2)
        raise Hdf5FormatError("only contiguous datasets are supported")

    def _attribute(self, d):
        ver = d[0]
        if ver != 1:
            raise Hdf5FormatError(f"attribute message version {ver}")
        name_size, dt_size, sp_size = struct.unpack_from("<HHH", d, 2)
        pad = lambda n: (n + 7) & ~7
        pos = 8
        name = d[pos:pos + name_size].split(b"\x00")[0].decode()
        pos += pad(name_size)
        dt = self._datatype(d[pos:pos + dt_size])
        pos += pad(dt_size)
        space = self._dataspace(d[pos:pos + sp_size])
        pos += pad(sp_size)
        raw = d[pos:]
        if dt["class"] == 9 and dt.get("vlen_string"):
            length, gaddr, gidx = struct.unpack_from("<IQI", raw, 0)
            return name, self._global_heap_object(gaddr, gidx)[:length].decode()
        if dt["class"] == 3:      # fixed-length string
            return name, raw[:dt["size"]].split(b"\x00")[0].decode()
        n = int(np.prod(space)) if space else 1
        return name, np.frombuffer(raw, self._numpy_dtype(dt), n).copy()

    def _global_heap_object(self, addr, index):
        b = self.b
        addr += self.base
        if b[addr:addr + 4] != b"GCOL":
            raise Hdf5FormatError("global heap signature missing")
        size = struct.unpack_from("<Q", b, addr + 8)[0]
        pos, end = addr + 16, addr + size
        while pos + 16 <= end:
            idx, _refs, _, osize = struct.unpack_from("<HHIQ", b, pos)
            if idx == index:
                return b[pos + 16: pos + 16 + osize]
            if idx == 0:
                break
            pos += 16 + ((osize + 7) & ~7)
        raise Hdf5FormatError("global heap object not found")


# ====================================================================== writer =========
# The same subset, written: superblock version 0, old-style groups (one symbol-table node per group, so
# at most 8 children each), version-1 object headers without continuation blocks, contiguous
# little-endian float32 datasets, variable-length UTF-8 string attributes in one global heap collection.
# Byte layouts follow the file rl-tools/HighFive produced for the reference checkpoint
# (tests/golden/checkpoint.h5, dumped with this module's reader); `h5dump` / `h5diff` 1.10 accept the output
# (tests/test_host_logic.py).

class GroupSpec:
    def __init__(self, children=None, attrs=None):
        self.children, self.attrs = dict(children or {}), dict(attrs or {})


class DatasetSpec:
    def __init__(self, array, attrs=None):
        self.array = np.ascontiguousarray(array, "<f4")
        self.attrs = dict(attrs or {})


_LEAF_K, _INTERNAL_K = 4, 16
_F32_DATATYPE = bytes.fromhex("11201f000400000000002000170800177f000000") + b"\x00" * 4
_VLEN_STR_DATATYPE = bytes.fromhex("1901010010000000" "100000000100000000000800") + b"\x00" * 4   # 20 B + pad
_SCALAR_DATASPACE = bytes.fromhex("0100000000000000")


def _pad8(b):
    return b + b"\x00" * (-len(b) % 8)


def _message(mtype, data, flags=0):
    assert len(data) % 8 == 0
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


class _Writer:
    def __init__(self, root):
        self.strings = []          # global heap objects, index = position + 1
        self._collect(root)
        need = 16 + sum(16 + len(_pad8(s)) for s in self.strings) + 16
        self.gheap_addr, self.gheap_size = 96, max(4096, (need + 4095) // 4096 * 4096)
        self.buf = bytearray(96 + self.gheap_size)
        self.string_index = {}
        self._write_gheap()
        root_header, btree, heap = self._group(root)
        eof = len(self.buf)
        sb = b"\x89HDF\r\n\x1a\n" + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", _LEAF_K, _INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, root_header, 1, 0) + struct.pack("<QQ", btree, heap)
        assert len(sb) == 96
        self.buf[:96] = sb

    # -------------------------------------------------------------- global heap (attribute strings)
    def _collect(self, node):
        for v in node.attrs.values():
            self.strings.append(str(v).encode("utf-8"))
        if isinstance(node, GroupSpec):
            for name in sorted(node.children):
                self._collect(node.children[name])

    def _write_gheap(self):
        out = bytearray(b"GCOL" + bytes([1, 0, 0, 0]) + struct.pack("<Q", self.gheap_size))
        for i, s in enumerate(self.strings, 1):
            out += struct.pack("<HHIQ", i, 0, 0, len(s)) + _pad8(s)
        free = self.gheap_size - len(out)
        if free >= 16:
            out += struct.pack("<HHIQ", 0, 0, 0, free)
        self.buf[self.gheap_addr:self.gheap_addr + len(out)] = out
        self._next_string = 0

    def _attribute(self, name, value):
        s = str(value).encode("utf-8")
        idx = self._next_string + 1              # same traversal order as _collect
        assert self.strings[self._next_string] == s
        self._next_string += 1
        nm = name.encode("utf-8") + b"\x00"
        data = struct.pack("<BBHHH", 1, 0, len(nm), 20, 8) + _pad8(nm) + _VLEN_STR_DATATYPE + _SCALAR_DATASPACE
        data += struct.pack("<IQI", len(s), self.gheap_addr, idx)
        return _message(0x000C, _pad8(data))

    # -------------------------------------------------------------- allocation
    def _append(self, blob):
        self.buf += b"\x00" * (-len(self.buf) % 8)
        addr = len(self.buf)
        self.buf += blob
        return addr

    def _header(self, messages):
        body = b"".join(messages)
        return self._append(struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body)

    # -------------------------------------------------------------- objects
    def _dataset(self, ds):
        attrs = [self._attribute(k, v) for k, v in ds.attrs.items()]
        shape = ds.array.shape
        space = struct.pack("<BBB5x", 1, len(shape), 1) + b"".join(struct.pack("<Q", d) for d in shape) * 2
        raw = ds.array.tobytes()
        data_addr = self._append(raw) if raw else UNDEF
        layout = _pad8(struct.pack("<BBQQ", 3, 1, data_addr, len(raw)))
        msgs = [_message(0x0001, _pad8(space)), _message(0x0003, _F32_DATATYPE, 1),
                _message(0x0005, bytes.fromhex("0202020100000000"), 1), _message(0x0008, layout)] + attrs
        return self._header(msgs)

    def _group(self, g):
        if not g.children:
            raise Hdf5FormatError("empty groups are not supported by this writer")
        if len(g.children) > 2 * _LEAF_K:
            raise Hdf5FormatError(f"more than {2 * _LEAF_K} children in one group are not supported by this writer")
        attrs = [self._attribute(k, v) for k, v in g.attrs.items()]      # before the children: _collect order
        names = sorted(g.children, key=lambda s: s.encode("utf-8"))
        entries = []
        for name in names:
            child = g.children[name]
            if isinstance(child, GroupSpec):
                entries.append((name,) + self._group(child))
            else:
                entries.append((name, self._dataset(child), None, None))
        # local heap: "" at offset 0, the names, one free block
        seg, offsets = bytearray(8), {}
        for name in names:
            offsets[name] = len(seg)
            seg += _pad8(name.encode("utf-8") + b"\x00")
        free_off = len(seg)
        seg += struct.pack("<QQ", 1, 32) + b"\x00" * 16
        seg_addr = self._append(bytes(seg))
        heap_addr = self._append(b"HEAP" + bytes(4) + struct.pack("<QQQ", len(seg), free_off, seg_addr))
        snod = bytearray(b"SNOD" + struct.pack("<BBH", 1, 0, len(names)))
        for name, header, bt, hp in entries:
            if bt is None:
                snod += struct.pack("<QQII16x", offsets[name], header, 0, 0)
            else:
                snod += struct.pack("<QQIIQQ", offsets[name], header, 1, 0, bt, hp)
        snod += b"\x00" * (8 + 40 * 2 * _LEAF_K - len(snod))
        snod_addr = self._append(bytes(snod))
        tree = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF))
        tree += struct.pack("<QQQ", 0, snod_addr, offsets[names[-1]])
        tree += b"\x00" * (24 + (2 * _INTERNAL_K + 1) * 8 + 2 * _INTERNAL_K * 8 - len(tree))
        btree_addr = self._append(bytes(tree))
        header = self._header([_message(0x0011, struct.pack("<QQ", btree_addr, heap_addr))] + attrs)
        return header, btree_addr, heap_addr


def write_file(path, root):
    """Write the tree ``root`` (GroupSpec / DatasetSpec, string attributes) as an HDF5 file."""
    w = _Writer(root)
    with open(path, "wb") as f:
        f.write(bytes(w.buf))
