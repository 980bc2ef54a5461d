"""A small read-only HDF5 parser in pure Python/numpy — enough for the reference's perceptual-network file
`vgg16.caffemodel.h5` (a Caffe HDF5 snapshot read upstream with deepdish: imm/models/selfsup/vgg16.py:74-92,
build_vgg16.py:16), without h5py / libhdf5.

Supported (the structures libhdf5 writes with its default "earliest" format, i.e. Caffe's files): superblock versions
0/1, version-1 object headers with continuation blocks, old-style groups (symbol-table message -> version-1 B-tree ->
symbol-table nodes -> local heap), compact link messages of new-style groups, datasets with contiguous, compact or
chunked layout (version-1 chunk B-tree; filters deflate, shuffle, fletcher32), fixed-point and IEEE float types of either
byte order, dataspace versions 1/2.  Everything else (superblock 2/3, dense link storage, variable-length / compound /
string data, external links, other filters) raises NotImplementedError with the name of what was met.

Pinned against the real library: tests/golden/caffe_vgg_tiny.h5 is written by libhdf5 1.10.6 (generator
tests/golden/make_h5_golden.c); tests/test_hdf5_lite_cpu.py parses it with this module."""
import struct
import zlib
from collections import OrderedDict

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xffffffffffffffff


class H5File(object):
    def __init__(self, path):
        with open(path, 'rb') as f:
            self.buf = f.read()
        self.path = path
        self._superblock()

    # -- low level ------------------------------------------------------------------------------------------------
    def _u(self, pos, size):
        return int.from_bytes(self.buf[pos:pos + size], 'little')

    def _superblock(self):
        b = self.buf
        base = 0
        while b[base:base + 8] != SIGNATURE:               # the signature may sit at 0, 512, 1024, ...
            base = 512 if base == 0 else base * 2
            if base + 8 > len(b):
                raise ValueError('%s: not an HDF5 file' % self.path)
        ver = b[base + 8]
        if ver not in (0, 1):
            raise NotImplementedError('%s: HDF5 superblock version %d (only 0/1: files written with the default format)' % (self.path, ver))
        self.O, self.L = b[base + 13], b[base + 14]
        if self.O != 8 or self.L != 8:
            raise NotImplementedError('HDF5 offsets/lengths of %d/%d bytes' % (self.O, self.L))
        pos = base + 24 + (4 if ver == 1 else 0)
        self.base = self._u(pos, 8)
        pos += 4 * 8                                        # base, free-space, end-of-file, driver-info addresses
        # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
        self.root_header = self._u(pos + 8, 8)

    # -- object headers ---------------------------------------------------------------------------------------------
    def _messages(self, addr):
        """[(type, flags, payload bytes)] of the version-1 object header at `addr` (continuations followed)."""
        b = self.buf
        p = self.base + addr
        if b[p:p + 4] == b'OHDR':
            raise NotImplementedError('version-2 object headers (file written with libver=latest)')
        if b[p] != 1:
            raise ValueError('object header version %d at %d' % (b[p], addr))
        n_msgs = self._u(p + 2, 2)
        size = self._u(p + 8, 4)
        blocks = [(p + 16, size)]
        out = []
        while blocks and len(out) < n_msgs:
            pos, remaining = blocks.pop(0)
            end = pos + remaining
            while pos + 8 <= end and len(out) < n_msgs:
                mtype, msize, flags = self._u(pos, 2), self._u(pos + 2, 2), b[pos + 4]
                payload = b[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x10:                            # continuation: offset, length
                    blocks.append((self.base + int.from_bytes(payload[:8], 'little'), int.from_bytes(payload[8:16], 'little')))
                out.append((mtype, flags, payload))
        return out

    # -- groups --------------------------------------------------------------------------------------------------------
    def _heap_name(self, heap_addr, offset):
        p = self.base + heap_addr
        if self.buf[p:p + 4] != b'HEAP':
            raise ValueError('bad local heap at %d' % heap_addr)
        data = self.base + self._u(p + 8 + 2 * self.L, 8)
        end = self.buf.index(b'\x00', data + offset)
        return self.buf[data + offset:end].decode()

    def _group_btree(self, node_addr, heap_addr, out):
        p = self.base + node_addr
        b = self.buf
        if b[p:p + 4] == b'SNOD':
            n = self._u(p + 6, 2)
            q = p + 8
            for _ in range(n):
                out[self._heap_name(heap_addr, self._u(q, 8))] = self._u(q + 8, 8)
                q += 40
            return
        if b[p:p + 4] != b'TREE' or b[p + 4] != 0:
            raise ValueError('bad group B-tree node at %d' % node_addr)
        used = self._u(p + 6, 2)
        q = p + 8 + 2 * self.O                              # skip sibling addresses
        for i in range(used):
            child = self._u(q + self.L, 8)                  # key_i (L bytes), child_i (O bytes)
            self._group_btree(child, heap_addr, out)
            q += self.L + self.O

    def _links(self, msgs):
        """name -> object header address for a group's messages."""
        out = OrderedDict()
        for mtype, _flags, m in msgs:
            if mtype == 0x11:                               # symbol table: B-tree address, local heap address
                self._group_btree(int.from_bytes(m[:8], 'little'), int.from_bytes(m[8:16], 'little'), out)
            elif mtype == 0x06:                             # link message (compact storage of new-style groups)
                flags = m[1]
                pos = 2
                ltype = 0
                if flags & 8:
                    ltype = m[pos]; pos += 1
                if flags & 4:
                    pos += 8
                if flags & 16:
                    pos += 1
                nlen_size = 1 << (flags & 3)
                nlen = int.from_bytes(m[pos:pos + nlen_size], 'little'); pos += nlen_size
                name = m[pos:pos + nlen].decode(); pos += nlen
                if ltype != 0:
                    raise NotImplementedError('soft/external link %s' % name)
                out[name] = int.from_bytes(m[pos:pos + 8], 'little')
            elif mtype == 0x02:
                if int.from_bytes(m[-16:-8], 'little') != UNDEF and len(m) >= 18:
                    raise NotImplementedError('dense link storage (fractal heap) in a new-style group')
        return out

    # -- datasets --------------------------------------------------------------------------------------------------------
    @staticmethod
    def _dtype(m):
        cls, bits0, size = m[0] & 0x0f, m[1], int.from_bytes(m[4:8], 'little')
        order = '>' if bits0 & 1 else '<'
        if cls == 0:
            return np.dtype('%s%s%d' % (order, 'i' if bits0 & 8 else 'u', size))
        if cls == 1:
            if size not in (2, 4, 8):
                raise NotImplementedError('%d-byte floating point' % size)
            return np.dtype('%sf%d' % (order, size))
        raise NotImplementedError('HDF5 datatype class %d (only integers and IEEE floats)' % cls)

    def _shape(self, m):
        ver, rank = m[0], m[1]
        if ver == 1:
            pos = 8
        elif ver == 2:
            if m[3] == 2:
                raise NotImplementedError('null dataspace')
            pos = 4
        else:
            raise NotImplementedError('dataspace version %d' % ver)
        return tuple(int.from_bytes(m[pos + 8 * i:pos + 8 * i + 8], 'little') for i in range(rank))

    @staticmethod
    def _filters(m):
        ver, n = m[0], m[1]
        pos = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = int.from_bytes(m[pos:pos + 2], 'little'); pos += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = int.from_bytes(m[pos:pos + 2], 'little'); pos += 2
            pos += 2                                         # flags
            ncd = int.from_bytes(m[pos:pos + 2], 'little'); pos += 2
            pos += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            cd = [int.from_bytes(m[pos + 4 * i:pos + 4 * i + 4], 'little') for i in range(ncd)]
            pos += 4 * ncd
            if ver == 1 and ncd % 2:
                pos += 4
            out.append((fid, cd))
        return out

    def _chunks(self, node_addr, rank, out):
        p = self.base + node_addr
        b = self.buf
        if b[p:p + 4] != b'TREE' or b[p + 4] != 1:
            raise ValueError('bad chunk B-tree node at %d' % node_addr)
        level, used = b[p + 5], self._u(p + 6, 2)
        key_size = 8 + 8 * (rank + 1)
        q = p + 8 + 2 * self.O
        for _ in range(used):
            size, mask = self._u(q, 4), self._u(q + 4, 4)
            offs = tuple(self._u(q + 8 + 8 * i, 8) for i in range(rank))
            child = self._u(q + key_size, 8)
            if level == 0:
                out.append((offs, size, mask, child))
            else:
                self._chunks(child, rank, out)
            q += key_size + self.O

    def _dataset(self, msgs):
        shape = dtype = layout = None
        filters = []
        for mtype, _flags, m in msgs:
            if mtype == 0x01:
                shape = self._shape(m)
            elif mtype == 0x03:
                dtype = self._dtype(m)
            elif mtype == 0x08:
                layout = m
            elif mtype == 0x0b:
                filters = self._filters(m)
        if shape is None or dtype is None or layout is None:
            raise ValueError('object is not a dataset')
        n = int(np.prod(shape, dtype=np.int64))
        if layout[0] != 3:
            raise NotImplementedError('data layout message version %d' % layout[0])
        cls = layout[1]
        if cls == 0:                                          # compact: size (2), data
            size = int.from_bytes(layout[2:4], 'little')
            raw = layout[4:4 + size]
            return np.frombuffer(raw, dtype=dtype, count=n).reshape(shape).astype(dtype.newbyteorder('='))
        if cls == 1:                                          # contiguous: address, size
            addr = int.from_bytes(layout[2:10], 'little')
            if addr == UNDEF:
                return np.zeros(shape, dtype.newbyteorder('='))
            p = self.base + addr
            return np.frombuffer(self.buf, dtype=dtype, count=n, offset=p).reshape(shape).astype(dtype.newbyteorder('='))
        if cls == 2:                                          # chunked: dimensionality, B-tree address, chunk dims (+ element size)
            ndim = layout[2]
            btree = int.from_bytes(layout[3:11], 'little')
            cdims = tuple(int.from_bytes(layout[11 + 4 * i:15 + 4 * i], 'little') for i in range(ndim))
            rank = ndim - 1
            if rank != len(shape):
                raise ValueError('chunk rank %d != dataspace rank %d' % (rank, len(shape)))
            out = np.zeros(shape, dtype.newbyteorder('='))
            if btree == UNDEF:
                return out
            chunks = []
            self._chunks(btree, rank, chunks)
            csize = int(np.prod(cdims[:rank])) * dtype.itemsize
            for offs, size, mask, addr in chunks:
                raw = self.buf[self.base + addr:self.base + addr + size]
                for i, (fid, cd) in reversed(list(enumerate(filters))):
                    if mask & (1 << i):
                        continue
                    if fid == 1:
                        raw = zlib.decompress(raw)
                    elif fid == 2:
                        es = cd[0] if cd else dtype.itemsize
                        a = np.frombuffer(raw, np.uint8)
                        k = len(a) // es
                        raw = a[:k * es].reshape(es, k).T.tobytes() + a[k * es:].tobytes()
                    elif fid == 3:
                        raw = raw[:-4]
                    else:
                        raise NotImplementedError('HDF5 filter id %d' % fid)
                if len(raw) < csize:
                    raise ValueError('chunk at %s holds %d bytes, expected %d' % (offs, len(raw), csize))
                block = np.frombuffer(raw, dtype=dtype, count=csize // dtype.itemsize).reshape(cdims[:rank])
                sel_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
                sel_in = tuple(slice(0, s.stop - s.start) for s in sel_out)
                out[sel_out] = block[sel_in]
            return out
        raise NotImplementedError('data layout class %d' % cls)

    # -- tree ------------------------------------------------------------------------------------------------------------
    def _load(self, addr):
        if addr == UNDEF:
            raise NotImplementedError('symbolic link in an old-style group')
        msgs = self._messages(addr)
        types = set(t for t, _f, _m in msgs)
        if 0x08 in types:
            return self._dataset(msgs)
        if types & {0x11, 0x06, 0x02}:
            return OrderedDict((name, self._load(a)) for name, a in self._links(msgs).items())
        return OrderedDict()

    def load(self, path='/'):
        """The object at `path` as nested OrderedDicts of numpy arrays (what deepdish.io.load returns for these files)."""
        addr = self.root_header
        for part in [p for p in path.split('/') if p]:
            links = self._links(self._messages(addr))
            if part not in links:
                raise KeyError('%s: no %r in %s' % (self.path, part, path))
            addr = links[part]
        return self._load(addr)


def load(path, group='/'):
    return H5File(path).load(group)
