"""TensorFlow checkpoint (V2 "tensor bundle") interchange without TensorFlow -- SURVEY section 8 row f3.

The reference saves and restores `model.ckpt` with `tf.train.Saver` (train_multi_gpu_pc_compare_dist.py:305,354-357;
consumers: :443-453, pcrnet-registration/iterative_PCRNet_ours.py:288-290).  A V2 checkpoint is two files:

    <prefix>.index                  an SSTable (LevelDB table format, tensorflow/core/lib/io/table*) mapping
                                    ""            -> BundleHeaderProto  {num_shards, endianness, version}
                                    <tensor name> -> BundleEntryProto   {dtype, shape, shard_id, offset, size, crc32c}
    <prefix>.data-00000-of-00001    the tensors' raw little-endian bytes at those offsets

This module restates that container format (third-party: TensorFlow >= 1.14 `tensor_bundle`, not vendored in the
reference and not installable here): table blocks with prefix-compressed keys and restart arrays, 5-byte block trailers
(compression type + masked CRC32C), the 48-byte footer with magic 0xdb4775248b80fb57, and the two protobuf messages
(parsed/emitted by hand).  `read_checkpoint` returns {name: ndarray}; `write_checkpoint` emits the same format so that
weights trained here can be loaded by a TF `Saver`.  PARITY UNPINNED: no checkpoint written by real TensorFlow exists in
the reference tree or this container, so the tests pin the reader only against this writer, against hand-assembled
blocks, and against the published CRC32C / masking test vectors.  The variable names are the reference's
(`pc_compare/dpdist_local/mapper_conv{1..4}/{weights,biases}`), so `DPDistParams.load_tf_state_dict(read_checkpoint(p))`
works on a real `model.ckpt` if the format restatement is right.
"""
import os
import struct

import numpy as np

MAGIC = 0xDB4775248B80FB57
# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           14: None, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DT_OF = {np.dtype(v): k for k, v in _DTYPES.items() if v is not None}


# ---- CRC32C (Castagnoli) + LevelDB masking -------------------------------------------------------------
def _make_table():
    tab = []
    for n in range(256):
        c = n
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC_TAB = _make_table()


def _native_crc():
    """dpd_crc32c of the in-tree library when it is built (hundreds of MB/s); None otherwise.  The library is opened
    directly with ctypes: this host utility needs neither torch nor a GPU."""
    global _NATIVE
    if _NATIVE is None:
        _NATIVE = False
        so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdpdist_hip.so")
        if os.path.exists(so):
            try:
                import ctypes
                fn = ctypes.CDLL(so).dpd_crc32c
                fn.restype, fn.argtypes = ctypes.c_uint32, [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32]
                _NATIVE = fn
            except (OSError, AttributeError):
                pass
    return _NATIVE or None


_NATIVE = None


def crc32c(data, crc=0, pure_python=False):
    data = bytes(data)
    fn = None if pure_python or len(data) < 4096 else _native_crc()
    if fn is not None:
        return int(fn(data, len(data), crc))
    crc ^= 0xFFFFFFFF
    for b in data:
        crc = _CRC_TAB[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def unmask_crc(m):
    rot = (m - 0xA282EAD8) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ---- varints / protobuf ---------------------------------------------------------------------------------
def _get_varint(buf, pos):
    r, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        r |= (b & 0x7F) << shift
        if not b & 0x80:
            return r, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _pb_fields(buf):
    """Iterate (field number, wire type, value) over a protobuf message; length-delimited values are bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _get_varint(buf, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_shape(buf):
    dims = []
    for f, _, v in _pb_fields(buf):
        if f == 2:                                   # repeated Dim dim = 2
            size = 0
            for f2, _, v2 in _pb_fields(v):
                if f2 == 1:
                    size = _signed64(v2)
            dims.append(size)
    return tuple(dims)


def _parse_entry(buf):
    e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "slices": False}
    for f, _, v in _pb_fields(buf):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            e["shape"] = _parse_shape(v)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = v
        elif f == 7:
            e["slices"] = True
    return e


def _pb(field, wt, payload):
    return _put_varint((field << 3) | wt) + payload


def _emit_entry(dtype, shape, offset, size, crc):
    dims = b"".join(_pb(2, 2, _put_varint(len(d)) + d) for d in (_pb(1, 0, _put_varint(s)) for s in shape))
    out = _pb(1, 0, _put_varint(dtype)) + _pb(2, 2, _put_varint(len(dims)) + dims)
    if offset:
        out += _pb(4, 0, _put_varint(offset))
    out += _pb(5, 0, _put_varint(size)) + _pb(6, 5, struct.pack("<I", crc))
    return out


# ---- table (SSTable) ------------------------------------------------------------------------------------
def _snappy_uncompress(src):
    n, pos = _get_varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        t = tag & 3
        if t == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += src[pos:pos + ln]
            pos += ln
            continue
        if t == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif t == 2:
            ln = (tag >> 2) + 1
            off = src[pos] | (src[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 4], "little")
            pos += 4
        for _ in range(ln):                          # overlapping copies are legal
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy: length mismatch")
    return bytes(out)


def _read_block(buf, offset, size, verify=True):
    raw = buf[offset:offset + size]
    ctype = buf[offset + size]
    stored = struct.unpack_from("<I", buf, offset + size + 1)[0]
    if verify and unmask_crc(stored) != crc32c(bytes(raw) + bytes([ctype])):
        raise ValueError("table block checksum mismatch at offset %d" % offset)
    if ctype == 1:
        raw = _snappy_uncompress(raw)
    elif ctype != 0:
        raise ValueError("unknown block compression %d" % ctype)
    return bytes(raw)


def _block_entries(block):
    nrestarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        unshared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + unshared]
        pos += unshared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path, verify=True):
    """-> list of (key bytes, value bytes) of an SSTable, in key order."""
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError("%s is not a TensorFlow/LevelDB table (bad magic)" % path)
    foot = buf[len(buf) - 48:]
    _, p = _get_varint(foot, 0)                      # metaindex handle (unused)
    _, p = _get_varint(foot, p)
    ioff, p = _get_varint(foot, p)
    isize, p = _get_varint(foot, p)
    out = []
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, q = _get_varint(handle, 0)
        bsize, q = _get_varint(handle, q)
        out.extend(_block_entries(_read_block(buf, boff, bsize, verify)))
    return out


class _BlockBuilder:
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.count, self.last, self.ri = bytearray(), [0], 0, b"", restart_interval

    def add(self, key, value):
        shared = 0
        if self.count < self.ri:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def write_table(path, items, block_size=4096):
    """items: iterable of (key, value) bytes in strictly increasing key order."""
    out = bytearray()
    index = _BlockBuilder(restart_interval=1)

    def flush(bb, last_key):
        data = bb.finish()
        off = len(out)
        out.extend(data + b"\x00" + struct.pack("<I", mask_crc(crc32c(data + b"\x00"))))
        if last_key is not None:
            index.add(last_key, _put_varint(off) + _put_varint(len(data)))
        return off, len(data)

    bb, last = _BlockBuilder(), None
    for k, v in items:
        if last is not None and k <= last:
            raise ValueError("keys must be strictly increasing")
        bb.add(k, v)
        last = k
        if len(bb.buf) >= block_size:
            flush(bb, last)
            bb = _BlockBuilder()
    if bb.buf or last is None:
        flush(bb, last if last is not None else b"")
    moff, msize = flush(_BlockBuilder(), None)       # empty metaindex block
    ioff, isize = flush(index, None)
    foot = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
    out.extend(foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", MAGIC))
    with open(path, "wb") as f:
        f.write(out)


# ---- the bundle -----------------------------------------------------------------------------------------
def _data_path(prefix, shard, num_shards):
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def list_variables(prefix, verify=False):
    """-> {name: (numpy dtype, shape)} like tf.train.list_variables."""
    out = {}
    for k, v in read_table(prefix + ".index", verify):
        if k == b"":
            continue
        e = _parse_entry(v)
        out[k.decode()] = (_DTYPES.get(e["dtype"]), e["shape"])
    return out


def read_checkpoint(prefix, names=None, verify=True, verify_data=False):
    """{name: ndarray} for every (or the named) variable of the V2 checkpoint `<prefix>.index/.data-*`.
    verify: check the index blocks' CRC32C; verify_data: also the tensors' (fast when libdpdist_hip.so is built,
    ~1 s per MB in pure Python otherwise)."""
    entries, num_shards = {}, 1
    for k, v in read_table(prefix + ".index", verify):
        if k == b"":
            for f, _, val in _pb_fields(v):
                if f == 1:
                    num_shards = val
                elif f == 2 and val != 0:
                    raise ValueError("big-endian checkpoints are not supported")
            continue
        entries[k.decode()] = _parse_entry(v)
    if names is not None:
        missing = [n for n in names if n not in entries]
        if missing:
            raise KeyError("not in checkpoint: %s" % missing)
        entries = {n: entries[n] for n in names}
    shards, out = {}, {}
    for name, e in entries.items():
        dt = _DTYPES.get(e["dtype"])
        if dt is None or e["slices"]:
            raise NotImplementedError("variable %s: dtype %d / sliced tensors are not supported" % (name, e["dtype"]))
        if e["shard_id"] not in shards:
            shards[e["shard_id"]] = np.memmap(_data_path(prefix, e["shard_id"], num_shards), dtype=np.uint8, mode="r")
        raw = shards[e["shard_id"]][e["offset"]:e["offset"] + e["size"]]
        want = int(np.prod(e["shape"], dtype=np.int64)) * np.dtype(dt).itemsize
        if e["size"] != want:
            raise ValueError("variable %s: %d bytes stored, shape %s needs %d" % (name, e["size"], e["shape"], want))
        if verify_data and e["crc32c"] is not None and unmask_crc(e["crc32c"]) != crc32c(raw.tobytes()):
            raise ValueError("variable %s: data checksum mismatch" % name)
        out[name] = np.frombuffer(raw.tobytes(), dtype=np.dtype(dt).newbyteorder("<")).reshape(e["shape"]).astype(dt)
    return out


def write_checkpoint(prefix, variables, checksums=True):
    """variables: {name: ndarray}.  Writes `<prefix>.index` and `<prefix>.data-00000-of-00001` (one shard, little endian).
    checksums=False stores 0 as the tensors' CRC32C (TF's reader would reject the data; this module's accepts it unless
    verify_data is set) -- the pure-Python CRC costs ~1 s per MB."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    version = _pb(1, 0, _put_varint(1))                                        # VersionDef.producer = 1
    header = _pb(1, 0, _put_varint(1)) + _pb(2, 0, _put_varint(0)) + _pb(3, 2, _put_varint(len(version)) + version)
    items, offset = [(b"", header)], 0
    with open(_data_path(prefix, 0, 1), "wb") as f:
        for name in sorted(variables, key=lambda s: s.encode()):
            arr = np.asarray(variables[name])
            if arr.ndim and not arr.flags.c_contiguous:          # (ascontiguousarray would turn a scalar into shape (1,))
                arr = np.ascontiguousarray(arr)
            if arr.dtype not in _DT_OF:
                raise NotImplementedError("dtype %s" % arr.dtype)
            raw = arr.astype(arr.dtype.newbyteorder("<")).tobytes()
            f.write(raw)
            crc = mask_crc(crc32c(raw)) if checksums else 0
            items.append((name.encode(), _emit_entry(_DT_OF[arr.dtype], arr.shape, offset, len(raw), crc)))
            offset += len(raw)
    write_table(prefix + ".index", items)
