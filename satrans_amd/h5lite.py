"""Reader for the HDF5 files of the reference's datasets, without h5py.

The reference keeps its preprocessed datasets in HDF5 (`alicpp.h5` with the datasets `ctr_train/<column>` and
`ctr_test/<column>`, written by aliccp_dataset_processing.py:237-242; `alimama.h5` with one root-level dataset per column,
alimama_preprocessing.py:41-52) and reads them with h5py (utils.py:22-30, 266-278).  h5py is not part of this image, and the
files are plain enough not to need it: `h5py.File(path, 'w')` with its default `libver` and `f[name] = array` /
`create_dataset(name, data=array)` produce

  * a version-0 superblock (offsets and lengths of 8 bytes),
  * "old style" groups: a version-1 object header with a Symbol Table message -> a version-1 B-tree ("TREE") of symbol-table
    nodes ("SNOD") whose link names live in a local heap ("HEAP"),
  * datasets: a version-1 object header with Dataspace, Datatype and Data Layout messages; the layout is CONTIGUOUS
    (class 1: address + size) unless chunking / compression was asked for, which neither writer does.

This module parses exactly that subset (HDF5 File Format Specification v2/v3, sections II.A, III.A-D, IV.A) and hands the
columns out as numpy arrays - memory-mapped views of the file by default, so a 42 M-row column costs no host copy until a batch
is gathered (`pipeline.HostBatchFeeder`).  Anything outside the subset (new-style groups of `libver='latest'`, chunked or
compressed datasets, compound / variable-length types, big-endian data) raises `NotImplementedError` naming what was found.
"""
from __future__ import annotations

import struct
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


class H5File:
    """Read-only view of an HDF5 file of the subset described in the module docstring."""

    def __init__(self, path: str):
        self.path = path
        self.buf = np.memmap(path, dtype=np.uint8, mode="r")
        self.base = 0
        self._superblock()

    # ---- primitive reads -------------------------------------------------------------------------------------
    def _bytes(self, off: int, n: int) -> bytes:
        if off < 0 or off + n > self.buf.shape[0]:
            raise H5Error(f"{self.path}: read of {n} bytes at {off} is outside the file ({self.buf.shape[0]} bytes)")
        return self.buf[off:off + n].tobytes()

    def _u(self, off: int, n: int) -> int:
        return int.from_bytes(self._bytes(off, n), "little")

    # ---- superblock (II.A) ------------------------------------------------------------------------------------
    def _superblock(self) -> None:
        off = 0
        while True:                                      # the signature sits at 0, 512, 1024, 2048, ...
            if off + 8 > self.buf.shape[0]:
                raise H5Error(f"{self.path}: no HDF5 signature found")
            if self._bytes(off, 8) == _SIG:
                break
            off = 512 if off == 0 else off * 2
        version = self._u(off + 8, 1)
        if version not in (0, 1):
            raise NotImplementedError(f"{self.path}: superblock version {version} (a file written with libver='latest'); only the "
                                      f"default layout of h5py.File(path, 'w') is read")
        self.size_off, self.size_len = self._u(off + 13, 1), self._u(off + 14, 1)
        if self.size_off != 8 or self.size_len != 8:
            raise NotImplementedError(f"{self.path}: {self.size_off}-byte offsets / {self.size_len}-byte lengths")
        p = off + 24 + (4 if version == 1 else 0)        # v1 adds indexed-storage K + reserved
        self.base = self._u(p, 8)
        # free-space address, end-of-file address, driver-information address, then the root group's symbol-table entry
        entry = p + 32
        self.root_header = self.base + self._u(entry + 8, 8)

    # ---- object headers (IV.A.1.a, version 1) ------------------------------------------------------------------
    def _messages(self, addr: int) -> List[Tuple[int, int, int]]:
        """[(type, data offset, data size)] of the version-1 object header at `addr`, continuation blocks included."""
        if self._bytes(addr, 4) == b"OHDR":
            raise NotImplementedError(f"{self.path}: version-2 object header at {addr} (libver='latest')")
        version = self._u(addr, 1)
        if version != 1:
            raise H5Error(f"{self.path}: object header version {version} at {addr}")
        total = self._u(addr + 2, 2)
        size = self._u(addr + 8, 4)
        blocks = [(addr + 16, size)]                     # 12 bytes of prefix + 4 of padding to an 8-byte boundary
        out: List[Tuple[int, int, int]] = []
        while blocks and len(out) < total:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < total:
                mtype, msize = self._u(p, 2), self._u(p + 2, 2)
                data = p + 8
                if mtype == 0x0010:                      # continuation: offset, length
                    blocks.append((self.base + self._u(data, 8), self._u(data + 8, 8)))
                out.append((mtype, data, msize))
                p = data + msize
        return out

    # ---- groups: symbol table message -> B-tree -> symbol-table nodes, names in the local heap (III.A-D) -------------------
    def _heap_data(self, heap_addr: int) -> int:
        if self._bytes(heap_addr, 4) != b"HEAP":
            raise H5Error(f"{self.path}: no local heap at {heap_addr}")
        return self.base + self._u(heap_addr + 24, 8)

    def _name(self, heap_data: int, off: int) -> str:
        p = heap_data + off
        end = p
        while self.buf[end] != 0:
            end += 1
        return self._bytes(p, end - p).decode("utf-8")

    def _walk(self, node: int, heap_data: int, out: Dict[str, int]) -> None:
        sig = self._bytes(node, 4)
        if sig == b"TREE":
            if self._u(node + 4, 1) != 0:
                raise H5Error(f"{self.path}: B-tree node at {node} is not a group node")
            used = self._u(node + 6, 2)
            p = node + 24 + 8                            # skip key 0
            for _ in range(used):
                self._walk(self.base + self._u(p, 8), heap_data, out)
                p += 16                                  # child address + next key
        elif sig == b"SNOD":
            count = self._u(node + 6, 2)
            p = node + 8
            for _ in range(count):
                out[self._name(heap_data, self._u(p, 8))] = self.base + self._u(p + 8, 8)
                p += 40
        else:
            raise H5Error(f"{self.path}: neither TREE nor SNOD at {node}: {sig!r}")

    def members(self, header: Optional[int] = None) -> Dict[str, int]:
        """{link name: object header address} of the group whose object header is at `header` (default: the root group)."""
        header = self.root_header if header is None else header
        for mtype, data, _ in self._messages(header):
            if mtype == 0x0011:                          # symbol table: B-tree address, local heap address
                out: Dict[str, int] = {}
                self._walk(self.base + self._u(data, 8), self._heap_data(self.base + self._u(data + 8, 8)), out)
                return out
            if mtype in (0x0002, 0x0006):
                raise NotImplementedError(f"{self.path}: new-style group (link messages; libver='latest')")
        raise H5Error(f"{self.path}: object at {header} is not a group")

    def resolve(self, path: str) -> int:
        """Object header address of `/a/b/c`."""
        header = self.root_header
        for part in [p for p in path.split("/") if p]:
            m = self.members(header)
            if part not in m:
                raise KeyError(f"{self.path}: no object {part!r} in {sorted(m)[:20]}... (while resolving {path!r})")
            header = m[part]
        return header

    # ---- datasets ---------------------------------------------------------------------------------------------
    def _dtype(self, data: int) -> np.dtype:
        cls_ver = self._u(data, 1)
        cls, bits0 = cls_ver & 0x0F, self._u(data + 1, 1)
        size = self._u(data + 4, 4)
        if bits0 & 1:
            raise NotImplementedError(f"{self.path}: big-endian dataset")
        if cls == 0:                                     # fixed point: bit 3 of the class bits = signed
            return np.dtype(("<i" if bits0 & 0x08 else "<u") + str(size))
        if cls == 1:
            if size not in (2, 4, 8):
                raise NotImplementedError(f"{self.path}: {size}-byte floating point")
            return np.dtype("<f" + str(size))
        raise NotImplementedError(f"{self.path}: datatype class {cls} (only integers and IEEE floats are read)")

    def dataset(self, path: str, mmap: bool = True) -> np.ndarray:
        """The array stored at `path` (`group/name`).  mmap=True: a read-only view of the file, nothing is copied."""
        header = self.resolve(path)
        shape = dtype = None
        addr = nbytes = None
        for mtype, data, _ in self._messages(header):
            if mtype == 0x0001:                          # dataspace
                ver, rank = self._u(data, 1), self._u(data + 1, 1)
                dims = data + (8 if ver == 1 else 4)
                shape = tuple(self._u(dims + 8 * i, 8) for i in range(rank))
            elif mtype == 0x0003:
                dtype = self._dtype(data)
            elif mtype == 0x0008:                        # data layout
                ver = self._u(data, 1)
                if ver in (1, 2):                        # HDF5 1.6 and older: version, rank, class, 5 reserved, address, dims
                    cls = self._u(data + 2, 1)
                    if cls != 1:
                        raise NotImplementedError(f"{self.path}: {path}: layout class {cls} in a version-{ver} layout message")
                    addr, nbytes = self._u(data + 8, 8), None
                    continue
                if ver != 3:
                    raise NotImplementedError(f"{self.path}: data layout message version {ver} for {path}")
                cls = self._u(data + 1, 1)
                if cls == 1:                             # contiguous: address, size
                    addr, nbytes = self._u(data + 2, 8), self._u(data + 10, 8)
                elif cls == 0:                           # compact: size (2 bytes), then the data inside the header
                    nbytes = self._u(data + 2, 2)
                    addr = data + 4 - self.base
                else:
                    raise NotImplementedError(f"{self.path}: {path} is chunked (chunking / compression / resizable); rewrite it "
                                              f"with f[name] = array")
            elif mtype == 0x000B:
                raise NotImplementedError(f"{self.path}: {path} has a filter pipeline (compression)")
        if shape is None or dtype is None or addr is None:
            raise H5Error(f"{self.path}: {path} is not a dataset (dataspace / datatype / layout message missing)")
        count = int(np.prod(shape)) if shape else 1
        if addr == _UNDEF:                               # never written: HDF5 returns the fill value (zero by default)
            return np.zeros(shape, dtype=dtype)
        if nbytes is not None and nbytes < count * dtype.itemsize:
            raise H5Error(f"{self.path}: {path} holds {nbytes} bytes, its shape {shape} needs {count * dtype.itemsize}")
        off = self.base + addr
        if off + count * dtype.itemsize > self.buf.shape[0]:
            raise H5Error(f"{self.path}: {path} extends past the end of the file")
        if mmap:
            return np.memmap(self.path, dtype=dtype, mode="r", offset=off, shape=shape)
        return np.frombuffer(self._bytes(off, count * dtype.itemsize), dtype=dtype).reshape(shape).copy()

    def keys(self, group: str = "/") -> List[str]:
        return sorted(self.members(self.resolve(group)))


def read_h5_columns(path: str, group: Optional[str], columns: Optional[Iterable[str]] = None, mmap: bool = True) -> Dict[str, np.ndarray]:
    """{column: array} from the datasets `<group>/<column>` (group None or "/": root-level datasets, the `alimama.h5` layout;
    "ctr_train" / "ctr_test": the `alicpp.h5` layout).  columns None: every member of the group, as `utils.loadh52df` does."""
    f = H5File(path)
    g = group or "/"
    names = list(columns) if columns is not None else f.keys(g)
    return {c: f.dataset(f"{g.rstrip('/')}/{c}", mmap=mmap) for c in names}
