"""TensorFlow checkpoint bundles (`model.ckpt-N.index` + `model.ckpt-N.data-00000-of-00001`) without TensorFlow, and
the variable-name map between the reference's graph and this engine (reference:
imm/train/cnn_train_multi.py:404-439,511-513 `tf.train.Saver` save / restore, `--restore-optim`,
`--ignore-missing-vars`, `--reset-global-step`).

PARITY UNPINNED: no TensorFlow here and no checkpoint of the reference in /root/reference, so this module is written from
the published formats and checked by its own round trips plus hand-assembled byte-level known answers
(tests/test_tf_checkpoint_cpu.py), not against a file TensorFlow wrote.

Formats restated:
  * the index is a LevelDB-style sorted string table (TF lib/io/table, leveldb's table_format): data blocks of
    prefix-compressed entries [varint shared | varint non_shared | varint value_len | key suffix | value], a restart
    array (u32 offsets) + u32 count, a 5-byte trailer per block (compression type, masked CRC-32C of contents+type), a
    metaindex block, an index block (last key of each data block -> BlockHandle{varint offset, varint size}) and a
    48-byte footer (two BlockHandles, zero padding, magic 0xdb4775248b80fb57 little-endian);
  * key "" -> BundleHeaderProto {1: num_shards, 2: endianness, 3: VersionDef{1: producer}}; every other key is a
    variable name -> BundleEntryProto {1: dtype, 2: TensorShapeProto{2: Dim{1: size}}, 3: shard_id, 4: offset, 5: size,
    6: fixed32 masked CRC-32C of the tensor bytes};
  * the data shard is the tensors' little-endian bytes back to back.
CRC-32C comes from libimm_hip.so's host utility imm_crc32c (include/imm_hip.h)."""
import ctypes as C
import os
import struct
from collections import OrderedDict

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
MASK_DELTA = 0xa282ead8
# tensorflow/core/framework/types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_UINT8, DT_INT16, DT_INT8, DT_INT64, DT_BOOL, DT_HALF = 1, 2, 3, 4, 5, 6, 9, 10, 19
_NP_OF_DT = {DT_FLOAT: np.float32, DT_DOUBLE: np.float64, DT_INT32: np.int32, DT_UINT8: np.uint8, DT_INT16: np.int16,
             DT_INT8: np.int8, DT_INT64: np.int64, DT_BOOL: np.bool_, DT_HALF: np.float16}
_DT_OF_NP = {np.dtype(v): k for k, v in _NP_OF_DT.items()}


# ---- checksums ---------------------------------------------------------------------------------------------------------
def crc32c(data, crc=0):
    from .. import _lib
    buf = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    v = C.c_uint32(crc)
    _lib.call('imm_crc32c', C.c_char_p(bytes(buf)), len(buf), C.byref(v))
    return v.value


def mask_crc(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + MASK_DELTA) & 0xffffffff


def unmask_crc(masked):
    rot = (masked - MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---- varints / minimal protobuf ----------------------------------------------------------------------------------------
def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError('varint too long')


def _pb_fields(buf):
    """Yield (field number, wire type, value) of one protobuf message (value: int for varint/fixed, bytes for
    length-delimited)."""
    pos = 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]; pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n]); pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]; pos += 4
        else:
            raise ValueError('protobuf wire type %d not supported' % wt)
        yield field, wt, v


def _pb_varint(field, v):
    return _put_varint(field << 3) + _put_varint(v)


def _pb_bytes(field, b):
    return _put_varint((field << 3) | 2) + _put_varint(len(b)) + b


def encode_entry(dtype_enum, shape, shard_id, offset, size, masked_crc):
    dims = b''.join(_pb_bytes(2, _pb_varint(1, int(d))) for d in shape)
    msg = _pb_varint(1, dtype_enum) + _pb_bytes(2, dims)
    if shard_id:
        msg += _pb_varint(3, shard_id)
    if offset:
        msg += _pb_varint(4, offset)
    msg += _pb_varint(5, size)
    msg += _put_varint((6 << 3) | 5) + struct.pack('<I', masked_crc)
    return msg


def decode_entry(buf):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'slices': 0}
    for field, _wt, v in _pb_fields(buf):
        if field == 1:
            e['dtype'] = v
        elif field == 2:
            for f2, _w2, v2 in _pb_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _w3, v3 in _pb_fields(v2):
                        if f3 == 1:
                            size = v3 - (1 << 64) if v3 >> 63 else v3
                    e['shape'].append(size)
        elif field == 3:
            e['shard_id'] = v
        elif field == 4:
            e['offset'] = v
        elif field == 5:
            e['size'] = v
        elif field == 6:
            e['crc32c'] = v
        elif field == 7:
            e['slices'] += 1
    return e


def encode_header(num_shards=1, producer=1):
    return _pb_varint(1, num_shards) + _pb_bytes(3, _pb_varint(1, producer))      # endianness 0 = little (default)


# ---- snappy (raw block format) — only needed if an index was written with compression on ---------------------------
def snappy_decompress(buf):
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]; pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], 'little'); pos += nb
            ln += 1
            out += buf[pos:pos + ln]; pos += ln
            continue
        if kind == 1:
            ln = 4 + ((tag >> 2) & 7)
            off = ((tag >> 5) << 8) | buf[pos]; pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8); pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], 'little'); pos += 4
        if off == 0 or off > len(out):
            raise ValueError('snappy: bad copy offset')
        for _ in range(ln):                      # may overlap its own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('snappy: length mismatch')
    return bytes(out)


# ---- sorted string table ------------------------------------------------------------------------------------------------
def _read_block(f, offset, size, verify=True):
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) != size + 5:
        raise ValueError('truncated table block at %d' % offset)
    contents, ctype, stored = raw[:size], raw[size], struct.unpack('<I', raw[size + 1:])[0]
    if verify and unmask_crc(stored) != crc32c(raw[:size + 1]):
        raise ValueError('table block at %d: checksum mismatch' % offset)
    if ctype == 1:
        contents = snappy_decompress(contents)
    elif ctype != 0:
        raise ValueError('table block compression type %d not supported' % ctype)
    return contents


def _block_entries(block):
    n_restarts = struct.unpack('<I', block[-4:])[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b''
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared]); pos += non_shared
        yield key, bytes(block[pos:pos + vlen]); pos += vlen


def read_table(path, verify=True):
    """-> OrderedDict{key bytes: value bytes} of a LevelDB-format table file."""
    out = OrderedDict()
    with open(path, 'rb') as f:
        f.seek(0, os.SEEK_END)
        total = f.tell()
        if total < 48:
            raise ValueError('%s: too short for a table' % path)
        f.seek(total - 48)
        footer = f.read(48)
        if struct.unpack('<Q', footer[40:])[0] != TABLE_MAGIC:
            raise ValueError('%s: not a TensorFlow/LevelDB table (bad magic)' % path)
        _mo, pos = _get_varint(footer, 0)
        _ms, pos = _get_varint(footer, pos)
        io_, pos = _get_varint(footer, pos)
        is_, pos = _get_varint(footer, pos)
        for _key, handle in _block_entries(_read_block(f, io_, is_, verify)):
            bo, p2 = _get_varint(handle, 0)
            bs, _ = _get_varint(handle, p2)
            for k, v in _block_entries(_read_block(f, bo, bs, verify)):
                out[k] = v
    return out


class _BlockBuilder(object):
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.count, self.last, self.interval = bytearray(), [0], 0, b'', restart_interval

    def add(self, key, value):
        shared = 0
        if self.count % self.interval == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        self.last, self.count = key, self.count + 1

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def write_table(path, items, block_size=4096):
    """items: iterable of (key bytes, value bytes) in strictly increasing key order."""
    with open(path, 'wb') as f:
        def emit(block):
            off = f.tell()
            f.write(block + b'\x00' + struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
            return _put_varint(off) + _put_varint(len(block))

        index, cur, prev = [], _BlockBuilder(), None
        for key, value in items:
            if prev is not None and not key > prev:
                raise ValueError('table keys must be strictly increasing')
            cur.add(key, value)
            prev = key
            if len(cur.buf) >= block_size:
                index.append((key, emit(cur.finish())))
                cur = _BlockBuilder()
        if cur.count:
            index.append((prev, emit(cur.finish())))
        meta = emit(_BlockBuilder().finish())
        ib = _BlockBuilder(restart_interval=1)
        for key, handle in index:
            ib.add(key, handle)
        idx = emit(ib.finish())
        footer = meta + idx
        f.write(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC))


# ---- bundles -------------------------------------------------------------------------------------------------------------
def read_bundle(prefix, names=None, verify=True):
    """-> OrderedDict{variable name: numpy array} of the checkpoint `prefix` (e.g. 'logs/model.ckpt-2000')."""
    table = read_table(prefix + '.index', verify)
    if b'' not in table:
        raise ValueError('%s.index has no bundle header' % prefix)
    num_shards, endian = 1, 0
    for field, _wt, v in _pb_fields(table[b'']):
        if field == 1:
            num_shards = v
        elif field == 2:
            endian = v
    if endian != 0:
        raise ValueError('big-endian bundles are not supported')
    shards, out = {}, OrderedDict()
    try:
        for key, value in table.items():
            if key == b'':
                continue
            name = key.decode()
            if names is not None and name not in names:
                continue
            e = decode_entry(value)
            if e['slices']:
                raise ValueError('%s: partitioned (sliced) variables are not supported' % name)
            if e['dtype'] not in _NP_OF_DT:
                raise ValueError('%s: dtype enum %d not supported' % (name, e['dtype']))
            if e['shard_id'] not in shards:
                shards[e['shard_id']] = open('%s.data-%05d-of-%05d' % (prefix, e['shard_id'], num_shards), 'rb')
            f = shards[e['shard_id']]
            f.seek(e['offset'])
            raw = f.read(e['size'])
            dt = np.dtype(_NP_OF_DT[e['dtype']])
            if len(raw) != e['size'] or e['size'] != int(np.prod(e['shape'], dtype=np.int64)) * dt.itemsize:
                raise ValueError('%s: size %d does not match shape %s' % (name, e['size'], e['shape']))
            if verify and e['crc32c'] is not None and unmask_crc(e['crc32c']) != crc32c(raw):
                raise ValueError('%s: tensor checksum mismatch' % name)
            out[name] = np.frombuffer(raw, dtype=dt).reshape(e['shape']).copy()
    finally:
        for f in shards.values():
            f.close()
    return out


def list_bundle(prefix):
    """-> OrderedDict{name: (numpy dtype, shape)} (tf.train.NewCheckpointReader.get_variable_to_shape_map)."""
    out = OrderedDict()
    for key, value in read_table(prefix + '.index').items():
        if key:
            e = decode_entry(value)
            out[key.decode()] = (np.dtype(_NP_OF_DT[e['dtype']]), tuple(e['shape']))
    return out


def write_bundle(prefix, tensors):
    """tensors: {name: array}.  One data shard; names sorted bytewise like TensorFlow's BundleWriter."""
    items = [(b'', encode_header())]
    offset = 0
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        for name in sorted(tensors, key=lambda s: s.encode()):
            a = np.asarray(tensors[name], order='C')       # (ascontiguousarray would turn scalars into 1-element vectors)
            if a.dtype not in _DT_OF_NP:
                raise ValueError('%s: dtype %s not supported' % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder('<'), copy=False).tobytes()
            f.write(raw)
            items.append((name.encode(), encode_entry(_DT_OF_NP[a.dtype], a.shape, 0, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    write_table(prefix + '.index', items)


# ---- variable names of the reference's graph <-> this engine -----------------------------------------------------------
def tf_variable_name(engine_name):
    """Engine name -> name in the reference's checkpoints (scopes at imm_model.py:123,155,183,225,241,283,
    base_model.py:107, nn_utils.py:193,201): a conv block `<scope>/conv_N` holds its kernel and bias under a second
    `conv_N` scope and its batch norm under `batch_normalization`; the loss normalisers live in
    `SelfSupReconstructionLoss`."""
    if engine_name.startswith('loss/') and engine_name.endswith('_agg'):
        return 'SelfSupReconstructionLoss/' + engine_name[5:]
    if engine_name == COST_EMA_VAR_ENGINE:
        return COST_EMA_VAR
    scope, leaf = engine_name.rsplit('/', 1)
    if leaf in ('w', 'b'):
        return '%s/%s/%s' % (scope, scope.rsplit('/', 1)[-1], leaf)
    if leaf in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
        return '%s/batch_normalization/%s' % (scope, leaf)
    return engine_name


def optimizer_slots(engine):
    """[(engine buffer name, TF slot-variable suffix)] of the engine's optimizer (scripts/train.py:97-104 names the TF
    optimizers 'Adam', 'Adadelta', 'AdaGrad'; a second slot of one optimizer gets the suffix _1 in creation order:
    Adam m, v; Adadelta accum, accum_update; Adagrad accumulator)."""
    kind = getattr(engine, 'optim', 'adam')
    if kind == 'adadelta':
        return [('adam_v', '/Adadelta'), ('adam_m', '/Adadelta_1')]
    if kind == 'adagrad':
        return [('adam_v', '/AdaGrad')]
    return [('adam_m', '/Adam'), ('adam_v', '/Adam_1')]


LOSS_SCALE_VAR = 'imm_amd/loss_scale_state'
# The shadow variables of BaseModel._add_cost_summary's moving averages (base_model.py:52-60).  tf.train.Saver stores them under
# names derived from the graph's auto-generated op names (`<cost op name>/<name>_movavg`), which cannot be restated without the
# graph: they travel as ONE 4-vector {reconstruction_loss, weights_loss, loss_total shadows, update count} under a name of this
# build, optional on restore (a bundle written by TensorFlow does not have it: the averages then restart at 0, like a fresh run).
COST_EMA_VAR_ENGINE, COST_EMA_VAR = 'summaries/cost_movavg', 'imm_amd/cost_movavg'


def engine_to_tf(engine, with_optimizer=True):
    """{TF variable name: numpy array} of an engine: model variables, BN moving statistics, loss normalisers, global_step
    and (optionally) Adam's slots `<var>/Adam`, `<var>/Adam_1`, `beta1_power`, `beta2_power` (tf.train.AdamOptimizer)."""
    from .. import ops                 # device -> host through pinned memory, stream-synchronised (ops.download)
    out = OrderedDict()
    for k, v in engine.named_parameters().items():
        out[tf_variable_name(k)] = ops.download(v).numpy()
    for k, v in engine.named_state().items():
        out[tf_variable_name(k)] = np.asarray(ops.download(v).numpy(), dtype=np.float32)
    step = int(engine.step_count)
    out['global_step'] = np.asarray(step, dtype=np.float32)      # a float model_variable upstream (scripts/train.py:86-88)
    if with_optimizer:
        for buf, suffix in optimizer_slots(engine):
            flat = ops.download(getattr(engine, buf)).numpy()
            for i, (name, shape, _wd) in enumerate(engine.spec):
                o0, o1 = engine.tab.offsets[i], engine.tab.offsets[i + 1]
                out[tf_variable_name(name) + suffix] = flat[o0:o1].reshape(shape)
        if getattr(engine, 'loss_scale_state', None) is not None:
            # not a TensorFlow variable (the reference computes in fp32): the dynamic loss scale of f16 storage and its
            # clean-step counter, so that a resume continues at the scale the run had reached
            out[LOSS_SCALE_VAR] = ops.download(engine.loss_scale_state).numpy().astype(np.float32)
        if getattr(engine, 'optim', 'adam') != 'adam':
            return out
        # tf.train.AdamOptimizer: beta_power starts at beta and is multiplied by beta after every apply => beta^(t+1) after
        # t updates of these slots (NOT global_step: the two differ after a restore without the optimizer)
        t = int(engine.adam_t)
        out['beta1_power'] = np.asarray(float(engine.hp.beta1) ** (t + 1), dtype=np.float32)
        out['beta2_power'] = np.asarray(float(engine.hp.beta2) ** (t + 1), dtype=np.float32)
    return out


def save_tf_checkpoint(engine, prefix, with_optimizer=True):
    write_bundle(prefix, engine_to_tf(engine, with_optimizer))


def load_tf_checkpoint(engine, prefix, restore_optim=False, ignore_missing_vars=False, reset_global_step=-1):
    """Restore an engine from a TensorFlow bundle with the reference's semantics (cnn_train_multi.py:404-433): model
    variables (+ global_step) always, Adam slots with `restore_optim`, missing variables skipped only with
    `ignore_missing_vars`, `reset_global_step >= 0` overrides the step.  Returns the list of variables not found."""
    import torch
    from .. import ops
    have = list_bundle(prefix)
    want_params = OrderedDict((k, tf_variable_name(k)) for k in engine.pview)
    want_state = OrderedDict((k, tf_variable_name(k)) for k in engine.named_state())
    needed = list(want_params.values()) + [n for n in want_state.values() if n != COST_EMA_VAR]      # (optional: see COST_EMA_VAR)
    if restore_optim:
        needed += [n + s for n in want_params.values() for _b, s in optimizer_slots(engine)]
    missing = [n for n in needed if n not in have]
    if missing and not ignore_missing_vars:
        raise KeyError('%s lacks %d variables (e.g. %s); pass ignore_missing_vars to skip them' % (prefix, len(missing), missing[0]))
    data = read_bundle(prefix, names=set(needed) | {'global_step', 'beta1_power', 'beta2_power', LOSS_SCALE_VAR, COST_EMA_VAR})
    params = engine.named_parameters()
    for k, n in want_params.items():
        if n in data:
            if tuple(data[n].shape) != tuple(params[k].shape):
                raise ValueError('%s: checkpoint shape %s != %s' % (n, data[n].shape, tuple(params[k].shape)))
            params[k] = torch.from_numpy(data[n].astype(np.float32))
    state = OrderedDict((k, torch.from_numpy(np.asarray(data[n], dtype=np.float32))) for k, n in want_state.items() if n in data)
    engine.load_parameters(params, state)
    if restore_optim:
        for buf, suffix in optimizer_slots(engine):
            flat = ops.download(getattr(engine, buf)).numpy()
            for i, (name, _shape, _wd) in enumerate(engine.spec):
                o0, o1 = engine.tab.offsets[i], engine.tab.offsets[i + 1]
                n = want_params[name]
                if n + suffix in data:
                    flat[o0:o1] = data[n + suffix].reshape(-1)
            ops.upload(getattr(engine, buf), torch.from_numpy(flat), buf)      # pinned staging + read-back (ops.upload)
        # Adam's t from the saved accumulators beta^(t+1) (beta2_power first: it resolves t up to ~1e5 before float32
        # underflow, beta1_power only up to ~1e3); both underflown = the correction factors are 1 anyway: any large t
        t = 0
        for key, beta in (('beta2_power', engine.hp.beta2), ('beta1_power', engine.hp.beta1)):
            if key in data:
                bp = float(data[key])
                if 0.0 < bp < 1.0:
                    t = max(0, int(round(np.log(bp) / np.log(float(beta)))) - 1)
                    break
                if bp == 0.0:
                    t = max(t, 1 << 20)
        engine.adam_t.fill_(t)
        if getattr(engine, 'loss_scale_state', None) is not None and LOSS_SCALE_VAR in data:
            engine.loss_scale_state.copy_(torch.from_numpy(np.asarray(data[LOSS_SCALE_VAR], dtype=np.float32).reshape(-1)))
    else:
        engine.reset_optimizer_slots()
    if reset_global_step >= 0:
        engine.step_count.fill_(int(reset_global_step))
    elif 'global_step' in data:
        engine.step_count.fill_(int(round(float(data['global_step']))))
    return missing
