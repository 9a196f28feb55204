"""Test infrastructure: writes, byte by byte, an HDF5 file with the structure libhdf5 gives the reference's dataset files
(`h5py.File(path, 'w')` + `f[name] = array`, default libver): version-0 superblock, old-style groups (version-1 object header
with a Symbol Table message, version-1 B-tree of symbol-table nodes holding at most 2 * leaf-K = 8 links each, local heap of
names), datasets with version-1 object headers (Dataspace v1, Datatype, Data Layout v3 CONTIGUOUS, plus the Fill Value and
modification-time messages libhdf5 adds, and - for every third dataset - an object-header CONTINUATION block, which libhdf5
emits whenever a header outgrows its first allocation).  HDF5 File Format Specification, sections II.A, III.A-D, IV.A.
h5py is not installed in this image, so this writer is what pins satrans_amd/h5lite.py; it follows the specification, not the
reader."""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype: int, data: bytes, flags: int = 0) -> bytes:
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _datatype(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt.kind in "iu":                                   # class 0, version 1; bit 3 of the class bits: signed
        return struct.pack("<B3BI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize) + \
            struct.pack("<HH", 0, 8 * dt.itemsize)
    if dt == np.float64:                                  # class 1: IEEE little-endian, sign bit 63
        return struct.pack("<B3BI", 0x11, 0x20, 63, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    if dt == np.float32:
        return struct.pack("<B3BI", 0x11, 0x20, 31, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
    raise ValueError(dt)


class Writer:
    def __init__(self):
        self.blob = bytearray(b"\0" * 96)                 # superblock goes here at the end

    def _alloc(self, data: bytes, align: int = 8) -> int:
        self.blob += b"\0" * (-len(self.blob) % align)
        at = len(self.blob)
        self.blob += data
        return at

    def _object_header(self, messages, split: bool = False) -> int:
        """Version-1 object header.  split: the last message moves to a continuation block elsewhere in the file."""
        cont = None
        if split and len(messages) > 1:
            tail = messages[-1]
            cont_at = self._alloc(tail)
            messages = messages[:-1] + [_msg(0x0010, struct.pack("<QQ", cont_at, len(tail)))]
            cont = tail
        body = b"".join(messages) + _msg(0x0000, b"\0" * 8)           # a NIL message as padding, as libhdf5 leaves
        n = len(messages) + 1 + (1 if cont is not None else 0)
        return self._alloc(struct.pack("<BxHII4x", 1, n, 1, len(body)) + body)

    def dataset(self, array: np.ndarray, split: bool = False) -> int:
        a = np.ascontiguousarray(array)
        data_at = self._alloc(a.tobytes()) if a.size else UNDEF
        space = struct.pack("<BBB5x", 1, a.ndim, 0) + b"".join(struct.pack("<Q", d) for d in a.shape)
        msgs = [_msg(0x0001, space), _msg(0x0003, _datatype(a.dtype), flags=1),
                _msg(0x0005, struct.pack("<BBBB", 2, 2, 2, 0)),                      # fill value: v2, never written, undefined
                _msg(0x0008, struct.pack("<BBQQ", 3, 1, data_at, a.nbytes)),         # layout v3, contiguous
                _msg(0x0012, struct.pack("<B3xI", 1, 1_700_000_000))]                # modification time
        return self._object_header(msgs, split)

    def group(self, members: dict) -> int:
        """members: {name: object header address}.  Names sorted as libhdf5 keeps them; 8 links per symbol-table node."""
        names = sorted(members)
        heap = bytearray(b"\0" * 8)                        # offset 0 = the empty name libhdf5 puts first
        offs = {}
        for nme in names:
            offs[nme] = len(heap)
            heap += _pad8(nme.encode() + b"\0")
        free_at = len(heap)
        heap += struct.pack("<QQ", 1, 64) + b"\0" * 48     # one free block, as a real heap carries
        heap_data = self._alloc(bytes(heap))
        heap_at = self._alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_at, heap_data))
        leaves = []
        for i in range(0, max(len(names), 1), 8):
            part = names[i:i + 8]
            body = b"SNOD" + struct.pack("<BxH", 1, len(part))
            for nme in part:
                body += struct.pack("<QQII16x", offs[nme], members[nme], 0, 0)
            body += b"\0" * (40 * (8 - len(part)))
            leaves.append((self._alloc(body), offs[part[-1]] if part else 0))
        tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(leaves), UNDEF, UNDEF) + struct.pack("<Q", 0)
        for at, last_key in leaves:
            tree += struct.pack("<QQ", at, last_key)
        tree += b"\0" * (16 * (32 - len(leaves)))          # node sized for 2 * internal-K = 32 children
        tree_at = self._alloc(tree)
        return self._object_header([_msg(0x0011, struct.pack("<QQ", tree_at, heap_at))]), tree_at, heap_at

    def finish(self, root) -> bytes:
        root_header, tree_at, heap_at = root
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBxBBBxHHI", 0, 0, 0, 0, 8, 8, 4, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.blob), UNDEF)
        sb += struct.pack("<QQII", 0, root_header, 1, 0) + struct.pack("<QQ", tree_at, heap_at)     # cached symbol-table entry
        assert len(sb) == 96, len(sb)
        self.blob[0:96] = sb
        return bytes(self.blob)


def write_h5(path: str, tree: dict) -> None:
    """tree: {name: array | {name: array}} - root-level datasets (the alimama.h5 layout) and one level of groups with datasets
    (ctr_train/<col>, ctr_test/<col>: the alicpp.h5 layout)."""
    w = Writer()
    top = {}
    k = 0
    for name, val in tree.items():
        if isinstance(val, dict):
            inner = {}
            for nme, arr in val.items():
                inner[nme] = w.dataset(np.asarray(arr), split=(k % 3 == 2))
                k += 1
            top[name] = w.group(inner)[0]
        else:
            top[name] = w.dataset(np.asarray(val), split=(k % 3 == 2))
            k += 1
    with open(path, "wb") as f:
        f.write(w.finish(w.group(top)))
