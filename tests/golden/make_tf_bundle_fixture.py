"""Assemble a TensorFlow V2 checkpoint bundle BY HAND from the published formats -- deliberately sharing NO code with
flickering_adversarial_video_amd/tf_checkpoint.py or tfrecord_io.py (nothing of the package is imported), so that the reader is
checked against bytes its own writer did not produce.  Output (committed): tests/golden/tf_bundle_handmade.{index,data-0000N-of-00002}.

Formats restated (public documentation of LevelDB's table format and TensorFlow's tensor_bundle.proto / types.proto / tensor_shape.proto):
  * table file   = data blocks, metaindex block, index block, 48-byte footer
                   footer = metaindex BlockHandle, index BlockHandle (each: varint64 offset, varint64 size), zero padding to 40 bytes,
                            magic 0xdb4775248b80fb57 little-endian
  * block        = entries, uint32 restart offsets[], uint32 num_restarts;   followed on disk by a 1-byte compression type (0 = none)
                   and a 4-byte MASKED crc32c of (block + type byte)
  * entry        = varint32 shared key bytes, varint32 unshared key bytes, varint32 value length, key delta, value
                   (shared = 0 at every restart point)
  * index block  = one entry per data block: key >= last key of the block (here: the last key itself), value = its BlockHandle
  * masked crc   = ((crc >> 15) | (crc << 17)) + 0xa282ead8  (mod 2^32), crc = CRC-32C (Castagnoli, reflected polynomial 0x82F63B78)
  * key ""       -> BundleHeaderProto { 1: num_shards (varint), 2: endianness (varint, 0 = little), 3: VersionDef { 1: producer } }
  * key <name>   -> BundleEntryProto  { 1: dtype, 2: TensorShapeProto { 2: Dim { 1: size } ... }, 3: shard_id, 4: offset, 5: size,
                                        6: crc32c (fixed32, masked, of the tensor bytes) }
  * dtypes       : DT_FLOAT 1, DT_INT32 3, DT_INT64 9, DT_BFLOAT16 14
Differences from what the repository's writer emits (so that they are exercised): TWO shards, restart interval 2 (shared-prefix entries
between restarts, several restart points per block), three data blocks, a NON-EMPTY metaindex block, index keys that are shortened
separators rather than full keys for the first block, proto fields in a different order with an unknown field in the header and an
explicit shard_id / zero offset written out."""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PREFIX = os.path.join(HERE, "tf_bundle_handmade")


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def crc32c_bitwise(data):
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 & -(crc & 1))
    return crc ^ 0xFFFFFFFF


def mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def tag(field, wire):
    return varint(field << 3 | wire)


def ld(field, payload):
    return tag(field, 2) + varint(len(payload)) + payload


def block(entries, restart_interval):
    out, restarts, prev = bytearray(), [], b""
    for n, (k, v) in enumerate(entries):
        if n % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def main():
    assert crc32c_bitwise(b"123456789") == 0xE3069283                      # the CRC-32C check value
    assert crc32c_bitwise(bytes(32)) == 0x8A9136AA                         # RFC 3720 B.4 test vectors
    assert crc32c_bitwise(bytes([0xFF] * 32)) == 0x62A8AB43
    assert crc32c_bitwise(bytes(range(32))) == 0x46DD794E

    # the tensors (values are written out again, literally, in tests/test_tf_checkpoint_cpu.py)
    w = (np.arange(2 * 3 * 4, dtype=np.float32) * 0.25 - 1.0).reshape(1, 1, 2, 3, 4)
    beta = np.array([0.5, -0.25, 3.0, 1e-3], np.float32).reshape(1, 1, 1, 1, 4)
    mean = np.array([-1.5, 2.0, 0.0, 7.75], np.float32).reshape(1, 1, 1, 1, 4)
    eps = (np.arange(6 * 3, dtype=np.float32) - 9.0).reshape(6, 1, 1, 3) / 64.0
    step = np.array(31337, np.int64)
    flow = np.array([[1, -2, 3], [40000, -50000, 60000]], np.int32)
    bf = np.array([1.0, -2.5, 0.15625, 384.0], np.float32)
    bf_raw = (bf.view(np.uint32) >> 16).astype("<u2").tobytes()
    tensors = [   # (name, dtype code, shape, raw little-endian bytes, shard)
        (b"Flow/other", 3, flow.shape, flow.astype("<i4").tobytes(), 1),
        (b"RGB/eps", 1, eps.shape, eps.astype("<f4").tobytes(), 1),
        (b"RGB/half_precision", 14, (4,), bf_raw, 0),
        (b"RGB/inception_i3d/Conv3d_1a_7x7/batch_norm/beta", 1, beta.shape, beta.astype("<f4").tobytes(), 0),
        (b"RGB/inception_i3d/Conv3d_1a_7x7/batch_norm/moving_mean", 1, mean.shape, mean.astype("<f4").tobytes(), 0),
        (b"RGB/inception_i3d/Conv3d_1a_7x7/conv_3d/w", 1, w.shape, w.astype("<f4").tobytes(), 0),
        (b"global_step", 9, (), step.astype("<i8").tobytes(), 1),
    ]
    assert [t[0] for t in tensors] == sorted(t[0] for t in tensors)
    shards, entries = {0: bytearray(), 1: bytearray(b"\xAB" * 5)}, []        # shard 1 starts with 5 bytes nobody references
    for name, dt, shape, raw, sid in tensors:
        off = len(shards[sid])
        shards[sid] += raw
        dims = b"".join(ld(2, tag(1, 0) + varint(d)) for d in shape)
        # field order 3, 1, 4, 5, 2, 6 -- protobuf parsers must not depend on it; offset 0 written explicitly
        e = tag(3, 0) + varint(sid) + tag(1, 0) + varint(dt) + tag(4, 0) + varint(off) + tag(5, 0) + varint(len(raw)) + ld(2, dims)
        e += tag(6, 5) + struct.pack("<I", mask(crc32c_bitwise(raw)))
        entries.append((name, e))
    header = ld(3, tag(1, 0) + varint(1) + tag(2, 0) + varint(0)) + tag(2, 0) + varint(0) + tag(1, 0) + varint(2)   # version, endianness, num_shards = 2
    header += ld(15, b"ignored")                                                                                     # unknown field
    items = [(b"", header)] + entries

    groups = [items[:3], items[3:6], items[6:]]                            # three data blocks
    f, index = bytearray(), []

    def emit(blk):
        off = len(f)
        f.extend(blk + b"\x00" + struct.pack("<I", mask(crc32c_bitwise(blk + b"\x00"))))
        return varint(off) + varint(len(blk))

    for gi, g in enumerate(groups):
        h = emit(block(g, restart_interval=2))
        # index key: any key >= the block's last key and < the next block's first key; block 0 uses a shortened separator
        # ("RGB/f" sorts after "RGB/eps" and before "RGB/half_precision"), the others the last key itself
        index.append((b"RGB/f" if gi == 0 else g[-1][0], h))
    meta = emit(block([(b"filter.none", b"")], restart_interval=16))       # a non-empty metaindex block (its handle is ignored by readers)
    idx = emit(block(index, restart_interval=1))
    footer = meta + idx
    f.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57))
    open(PREFIX + ".index", "wb").write(bytes(f))
    for sid, raw in shards.items():
        open(f"{PREFIX}.data-{sid:05d}-of-00002", "wb").write(bytes(raw))
    print("wrote", PREFIX + ".index", len(f), "bytes;", {k: len(v) for k, v in shards.items()})


if __name__ == "__main__":
    main()
