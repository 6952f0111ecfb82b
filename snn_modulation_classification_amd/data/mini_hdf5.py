"""Minimal pure-Python HDF5 reader for the one layout RadioML 2018.01A comes in (reference data/load_radio_ml.py:23-39).

The reference reads `GOLD_XYZ_OSC.0001_1024.hdf5` ('X' (N,1024,2) float32, 'Y' (N,24) int64, 'Z' (N,1) int64) and the
per-(class, SNR) files it writes itself with `h5py.File(..., 'w').create_dataset('X', data=...)` (:44-50): datasets in the
ROOT group, contiguous (or compact) storage, no filters, fixed-point / IEEE-float elements.  With h5py's default
`libver='earliest'` that is: superblock version 0 / 1, old-style groups (B-tree v1 + local heap + symbol-table nodes),
version-1 object headers, data-layout message version 3.  That subset of the HDF5 file format specification is what
is parsed here — h5py is not installed in this image; `load_radio_ml._block_2018` uses it when it is.  Anything else
(chunked / filtered datasets, version-2 object headers, nested groups) raises Hdf5Unsupported with the feature named.
Datasets are returned as numpy memmaps of the file: nothing is copied until the caller slices."""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Unsupported(RuntimeError):
    pass


class MiniHdf5:
    PAGE = 1 << 16                  # metadata is read in 64 KiB pages at the address asked for, a few of them cached

    def __init__(self, path):
        self.path = path
        self._pages = {}
        with open(path, 'rb') as f:
            self.f_size = f.seek(0, 2)
        head = self._bytes(0, 16)
        if head[:8] != b'\x89HDF\r\n\x1a\n':
            raise Hdf5Unsupported('%s: no HDF5 signature at offset 0' % path)
        ver = head[8]
        if ver > 1:
            raise Hdf5Unsupported('%s: superblock version %d (written with libver="latest"?); only 0 / 1' % (path, ver))
        if head[13] != 8 or head[14] != 8:
            raise Hdf5Unsupported('%s: offsets / lengths of %d / %d bytes; only 8 / 8' % (path, head[13], head[14]))
        pos = 24 + (4 if ver == 1 else 0)
        self.base = self._u64(pos)
        root_entry = pos + 32                   # base, free-space, end-of-file, driver-info addresses
        cache_type = struct.unpack('<I', self._bytes(root_entry + 16, 4))[0]
        if cache_type == 1:                     # B-tree / heap addresses cached in the scratch pad
            btree, heap = self._u64(root_entry + 24), self._u64(root_entry + 32)
        else:
            msgs = self._messages(self._u64(root_entry + 8))
            stab = [d for t, d in msgs if t == 0x11]
            if not stab:
                raise Hdf5Unsupported('%s: root group without a symbol table (new-style group)' % path)
            btree, heap = struct.unpack_from('<QQ', stab[0])
        self.links = {}
        self._walk_group(btree, self._heap_data(heap))
        self._pages.clear()

    # -- low-level ---------------------------------------------------------------------------------------------------
    def _bytes(self, pos, n):
        """n bytes of the file at pos.  Seek + bounded reads (64 KiB pages around the address, at most 16 cached): an object
        header, continuation block or symbol-table node that lies BEHIND a 20 GB data block (files whose metadata was
        written after the data) costs one page, not a read of everything in front of it."""
        if pos < 0 or pos + n > self.f_size:
            raise Hdf5Unsupported('%s: truncated file (metadata at %d + %d beyond the end)' % (self.path, pos, n))
        out = bytearray()
        first = pos // self.PAGE
        for pg in range(first, (pos + max(n, 1) - 1) // self.PAGE + 1):
            page = self._pages.get(pg)
            if page is None:
                with open(self.path, 'rb') as f:
                    f.seek(pg * self.PAGE)
                    page = f.read(self.PAGE)
                if len(self._pages) >= 16:
                    self._pages.pop(next(iter(self._pages)))
                self._pages[pg] = page
            out += page
        off = pos - first * self.PAGE
        return bytes(out[off:off + n])

    def _u64(self, pos):
        return struct.unpack('<Q', self._bytes(pos, 8))[0]

    def _heap_data(self, addr):
        addr += self.base
        head = self._bytes(addr, 32)
        if head[:4] != b'HEAP':
            raise Hdf5Unsupported('%s: local heap signature missing' % self.path)
        size, _, data = struct.unpack_from('<QQQ', head, 8)
        return self._bytes(self.base + data, size)

    def _walk_group(self, addr, heap):
        addr += self.base
        head = self._bytes(addr, 24)
        if head[:4] != b'TREE' or head[4] != 0:
            raise Hdf5Unsupported('%s: group B-tree node expected' % self.path)
        level, used = head[5], struct.unpack_from('<H', head, 6)[0]
        for k in range(used):
            child = self._u64(addr + 24 + 16 * k + 8)            # key k, child k, key k+1, ...
            if level > 0:
                self._walk_group(child, heap)
                continue
            node = child + self.base
            nh = self._bytes(node, 8)
            if nh[:4] != b'SNOD':
                raise Hdf5Unsupported('%s: symbol table node expected' % self.path)
            n = struct.unpack_from('<H', nh, 6)[0]
            entries = self._bytes(node + 8, 40 * n)
            for e in range(n):
                name_off, header = struct.unpack_from('<QQ', entries, 40 * e)
                name = heap[name_off:heap.index(b'\0', name_off)].decode()
                self.links[name] = header

    def _messages(self, addr):
        """(type, data) of every message of a version-1 object header, continuation blocks included."""
        addr += self.base
        head = self._bytes(addr, 16)
        if head[:4] == b'OHDR':
            raise Hdf5Unsupported('%s: version-2 object header (libver="latest"); only version 1' % self.path)
        if head[0] != 1:
            raise Hdf5Unsupported('%s: object header version %d' % (self.path, head[0]))
        n_msgs = struct.unpack_from('<H', head, 2)[0]
        blocks = [(addr + 16, struct.unpack_from('<I', head, 8)[0])]
        out = []
        while blocks and len(out) < n_msgs:
            pos, size = blocks.pop(0)
            blk = self._bytes(pos, size)
            p = 0
            while p + 8 <= size and len(out) < n_msgs:
                mtype, msize = struct.unpack_from('<HH', blk, p)
                data = blk[p + 8:p + 8 + msize]
                out.append((mtype, data))
                if mtype == 0x10:                                 # continuation: (offset, length)
                    o, l = struct.unpack_from('<QQ', data)
                    blocks.append((o + self.base, l))
                p += 8 + msize
        return out

    # -- datasets ----------------------------------------------------------------------------------------------------
    def keys(self):
        return sorted(self.links)

    def __contains__(self, name):
        return name in self.links

    def __getitem__(self, name):
        if name not in self.links:
            raise KeyError(name)
        shape = dtype = layout = None
        for mtype, d in self._messages(self.links[name]):
            if mtype == 0x01:                                     # dataspace
                ver, rank, flags = d[0], d[1], d[2]
                off = 8 if ver == 1 else 4
                shape = struct.unpack_from('<%dQ' % rank, d, off)
            elif mtype == 0x03:                                   # datatype
                cls, bits0, size = d[0] & 0x0F, d[1], struct.unpack_from('<I', d, 4)[0]
                if cls not in (0, 1):
                    raise Hdf5Unsupported('%s[%s]: datatype class %d; only fixed-point / float' % (self.path, name, cls))
                kind = 'f' if cls == 1 else ('i' if bits0 & 0x08 else 'u')
                dtype = np.dtype(('>' if bits0 & 1 else '<') + kind + str(size))
            elif mtype == 0x08:                                   # data layout
                if d[0] != 3:
                    raise Hdf5Unsupported('%s[%s]: data layout message version %d' % (self.path, name, d[0]))
                if d[1] == 1:
                    layout = ('contiguous',) + struct.unpack_from('<QQ', d, 2)
                elif d[1] == 0:
                    n = struct.unpack_from('<H', d, 2)[0]
                    layout = ('compact', d[4:4 + n])
                else:
                    raise Hdf5Unsupported('%s[%s]: chunked dataset (filters / compression); only contiguous storage — '
                                          'h5repack -l CONTI, or install h5py' % (self.path, name))
            elif mtype == 0x0B:
                raise Hdf5Unsupported('%s[%s]: filter pipeline' % (self.path, name))
        if shape is None or dtype is None or layout is None:
            raise Hdf5Unsupported('%s[%s]: not a simple dataset' % (self.path, name))
        count = int(np.prod(shape)) if len(shape) else 1
        if layout[0] == 'compact':
            return np.frombuffer(layout[1], dtype=dtype, count=count).reshape(shape)
        addr, size = layout[1], layout[2]
        if addr == UNDEF or count == 0:                           # never written: fill value 0
            return np.zeros(shape, dtype=dtype)
        if size < count * dtype.itemsize or self.base + addr + count * dtype.itemsize > self.f_size:
            raise Hdf5Unsupported('%s[%s]: truncated file' % (self.path, name))
        return np.memmap(self.path, mode='r', dtype=dtype, offset=self.base + addr, shape=tuple(shape))
