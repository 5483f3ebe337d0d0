"""Row f3: the TF V2 checkpoint container restated in dpdist_amd/tf_checkpoint.py.  No TensorFlow-written file is
available (parity unpinned, see the module header): the reader is pinned against published CRC32C vectors, hand-assembled
table bytes, a hand-assembled snappy stream, and round trips through the writer."""
import struct

import numpy as np
import pytest

from dpdist_amd import tf_checkpoint as T


def test_crc32c_published_vectors():
    assert T.crc32c(b"123456789") == 0xE3069283                      # RFC 3720 B.4 check value
    assert T.crc32c(bytes(32)) == 0x8A9136AA                         # RFC 3720: 32 bytes of zeros
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43                # 32 bytes of ones
    assert T.crc32c(bytes(range(32))) == 0x46DD794E                  # incrementing
    for x in (0, 1, 0xE3069283, 0xFFFFFFFF):
        assert T.unmask_crc(T.mask_crc(x)) == x and T.mask_crc(x) != x
    assert T.crc32c(b"6789", T.crc32c(b"12345")) == 0xE3069283       # incremental form


def test_varint_and_protobuf_entry_roundtrip():
    for v in (0, 1, 127, 128, 300, 2 ** 32, 2 ** 63 - 1):
        b = T._put_varint(v)
        assert T._get_varint(b, 0) == (v, len(b))
    e = T._parse_entry(T._emit_entry(1, (1, 2503, 1, 1024), 4096, 2503 * 1024 * 4, 0xDEADBEEF))
    assert e["dtype"] == 1 and e["shape"] == (1, 2503, 1, 1024) and e["offset"] == 4096
    assert e["size"] == 2503 * 1024 * 4 and e["crc32c"] == 0xDEADBEEF and e["shard_id"] == 0
    assert T._parse_entry(T._emit_entry(9, (), 0, 8, 1))["shape"] == ()          # scalar (global_step)


def test_hand_assembled_table(tmp_path):
    """Bytes built here by hand (not by write_table): one data block with a prefix-compressed second key."""
    def block(payload):
        return payload + b"\x00" + struct.pack("<I", T.mask_crc(T.crc32c(payload + b"\x00")))
    data = (bytes([0, 3, 2]) + b"abc" + b"v1"          # shared 0, unshared 3, value 2
            + bytes([2, 2, 1]) + b"de" + b"w"           # key "ab" + "de" = "abde"
            + struct.pack("<II", 0, 1))                 # restart array [0], count 1
    meta = struct.pack("<II", 0, 1)
    d_off, m_off = 0, len(data) + 5
    i_off = m_off + len(meta) + 5
    index = bytes([0, 4, 2]) + b"abde" + bytes([d_off, len(data)]) + struct.pack("<II", 0, 1)
    foot = bytes([m_off, len(meta), i_off, len(index)])
    blob = block(data) + block(meta) + block(index) + foot + bytes(40 - len(foot)) + struct.pack("<Q", T.MAGIC)
    p = tmp_path / "t.index"
    p.write_bytes(blob)
    assert T.read_table(str(p)) == [(b"abc", b"v1"), (b"abde", b"w")]
    bad = bytearray(blob)
    bad[4] ^= 1                                          # corrupt the data block
    p.write_bytes(bytes(bad))
    with pytest.raises(ValueError, match="checksum"):
        T.read_table(str(p))
    p.write_bytes(blob[:-1] + b"\x00")
    with pytest.raises(ValueError, match="magic"):
        T.read_table(str(p))


def test_snappy_stream():
    # "abcdabcdabcdabcd!" = literal "abcd" + copy(offset 4, len 12) + literal "!"
    src = bytes([17]) + bytes([(4 - 1) << 2]) + b"abcd" + bytes([((12 - 1) << 2) | 2, 4, 0]) + bytes([0 << 2]) + b"!"
    assert T._snappy_uncompress(src) == b"abcdabcdabcdabcd!"


def test_table_roundtrip_many_keys_and_blocks(tmp_path):
    items = [(("scope/layer%03d/weights" % i).encode(), bytes([i % 251]) * (i % 37 + 1)) for i in range(400)]
    items.sort()
    p = str(tmp_path / "many.index")
    T.write_table(p, items, block_size=256)              # forces dozens of data blocks and restart points
    assert T.read_table(p) == items


def test_checkpoint_roundtrip_with_reference_variable_names(tmp_path):
    from dpdist_amd import synth
    W = synth.make_weights("wide", mlp=(64, 64, 64))
    extra = {"global_step": np.array(153600, dtype=np.int64), "beta1_power": np.array(0.9, dtype=np.float32),
             "pc_compare/dpdist_local/mapper_conv1/weights/Adam": np.zeros((1, 2503, 1, 64), np.float32)}
    prefix = str(tmp_path / "log" / "model.ckpt")
    T.write_checkpoint(prefix, {**W, **extra})
    lv = T.list_variables(prefix)
    assert lv["global_step"] == (np.int64, ()) and lv["pc_compare/dpdist_local/mapper_conv1/weights"][1] == (1, 2503, 1, 64)
    back = T.read_checkpoint(prefix, verify_data=True)
    assert set(back) == set(W) | set(extra)
    for n, a in {**W, **extra}.items():
        assert back[n].dtype == a.dtype and back[n].shape == a.shape and np.array_equal(back[n], a), n
    only = T.read_checkpoint(prefix, names=["global_step"])
    assert list(only) == ["global_step"] and int(only["global_step"]) == 153600
    with pytest.raises(KeyError):
        T.read_checkpoint(prefix, names=["nope"])
    # flip one data byte: caught by the tensor checksum
    data = T._data_path(prefix, 0, 1)
    raw = bytearray(open(data, "rb").read())
    raw[100] ^= 0x40
    open(data, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="data checksum"):
        T.read_checkpoint(prefix, verify_data=True)


def test_checkpoint_feeds_the_parameter_loader(tmp_path):
    """read_checkpoint -> DPDistParams.load_tf_state_dict -> tf_state_dict -> write_checkpoint is the identity."""
    from dpdist_amd import synth
    from dpdist_amd.model import DPDistParams
    W = synth.make_weights("xavier_tf", mlp=(64, 64, 64))
    p1, p2 = str(tmp_path / "a.ckpt"), str(tmp_path / "b.ckpt")
    T.write_checkpoint(p1, W)
    P = DPDistParams(k=5, mlp=(64, 64, 64), device="cpu", init=None)
    P.load_tf_state_dict(T.read_checkpoint(p1))
    T.write_checkpoint(p2, P.tf_state_dict())
    a, b = T.read_checkpoint(p1), T.read_checkpoint(p2)
    assert set(a) == set(b) and all(np.array_equal(a[n], b[n]) for n in a)


def test_native_crc_matches_pure_python():
    rng = np.random.default_rng(0)
    for n in (4096, 4097, 65536 + 3):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert T.crc32c(b) == T.crc32c(b, pure_python=True)
        assert T.crc32c(b[100:], T.crc32c(b[:100])) == T.crc32c(b, pure_python=True)


def test_hand_assembled_sharded_bundle(tmp_path):
    """A bundle as tensor_bundle.proto / tensor_bundle.cc describe it when `tf.train.Saver(sharded=True)` (or a merge of per-device
    bundles) wrote it, assembled here field by field -- NOT by write_checkpoint, which only ever emits one shard:
      * key "" first: BundleHeaderProto {num_shards = 2, endianness = LITTLE (0), version {producer = 1}};
      * the remaining keys in BYTEWISE order ("A/..." < "a/..." < "a/b" < "a0": uppercase and '/' sort before digits and lowercase),
        each a BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c (masked, fixed32)}; shard_id 0 and offset 0 are
        omitted on the wire (proto3 defaults);
      * tensor bytes in <prefix>.data-0000S-of-00002, S = shard_id, at `offset`.
    Still not a TensorFlow-written file (row f3 stays "parity unpinned"); it pins the reader's handling of the fields the
    one-shard writer never produces."""
    import struct
    prefix = str(tmp_path / "model.ckpt")
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    b = np.array([7, 8, 9], dtype=np.int32)
    c = np.array(153600, dtype=np.int64)
    shard0 = a.tobytes()                                   # "a/b" at offset 0 of shard 0
    shard1 = b"\xAA" * 8 + b.tobytes() + c.tobytes()       # 8 bytes of something else first: offsets 8 and 20 in shard 1
    open(T._data_path(prefix, 0, 2), "wb").write(shard0)
    open(T._data_path(prefix, 1, 2), "wb").write(shard1)

    def entry(dtype, shape, shard, offset, raw):
        dims = b"".join(T._pb(2, 2, T._put_varint(len(d)) + d) for d in (T._pb(1, 0, T._put_varint(x)) for x in shape))
        out = T._pb(1, 0, T._put_varint(dtype)) + T._pb(2, 2, T._put_varint(len(dims)) + dims)
        if shard:
            out += T._pb(3, 0, T._put_varint(shard))
        if offset:
            out += T._pb(4, 0, T._put_varint(offset))
        return out + T._pb(5, 0, T._put_varint(len(raw))) + T._pb(6, 5, struct.pack("<I", T.mask_crc(T.crc32c(raw))))

    version = T._pb(1, 0, T._put_varint(1))
    header = T._pb(1, 0, T._put_varint(2)) + T._pb(3, 2, T._put_varint(len(version)) + version)     # endianness 0 omitted
    items = [(b"", header),
             (b"A/global_step", entry(9, (), 1, 20, c.tobytes())),        # DT_INT64 = 9, scalar
             (b"a/b", entry(1, (2, 3), 0, 0, a.tobytes())),               # DT_FLOAT = 1
             (b"a0", entry(3, (3,), 1, 8, b.tobytes()))]                  # DT_INT32 = 3
    assert [k for k, _ in items] == sorted(k for k, _ in items)          # the table requires (and TF writes) bytewise key order
    T.write_table(prefix + ".index", items)
    lv = T.list_variables(prefix)
    assert lv == {"A/global_step": (np.int64, ()), "a/b": (np.float32, (2, 3)), "a0": (np.int32, (3,))}
    got = T.read_checkpoint(prefix, verify_data=True)
    assert np.array_equal(got["a/b"], a) and np.array_equal(got["a0"], b) and int(got["A/global_step"]) == 153600
    # a big-endian header is refused, a slice-spec entry (partitioned variable) is refused
    T.write_table(prefix + ".index", [(b"", header + T._pb(2, 0, T._put_varint(1)))] + items[1:])
    with pytest.raises(ValueError, match="big-endian"):
        T.read_checkpoint(prefix)
    T.write_table(prefix + ".index", [items[0], (b"a/b", items[2][1] + T._pb(7, 2, T._put_varint(0)))])
    with pytest.raises(NotImplementedError):
        T.read_checkpoint(prefix)
