"""TensorFlow checkpoint ("tensor bundle", V2) reader / writer without TensorFlow.

The reference restores I3D with ``tf.train.Saver(var_list=rgb_variable_map).restore(sess, ckpt_path)``
(utils/kinetics_i3d_utils.py:41-62; ``ckpt_path`` = ``.../rgb_imagenet/model.ckpt``, run_config.yml:6-7) and saves the
universal perturbation with ``saver.save(sess, 'model_step_%05d')`` (i3d_adversarial_main_universal.py:176-201).  A bundle is

  <prefix>.index                  an SSTable (LevelDB table format, uncompressed blocks): key "" -> BundleHeaderProto,
                                  key <variable name> -> BundleEntryProto{dtype, shape, shard_id, offset, size, crc32c}
  <prefix>.data-0000N-of-0000M    the raw little-endian tensor bytes

``read_bundle(prefix)`` returns ``{variable name: ndarray}`` -- exactly the dict ``FlickerI3D(weights=...)`` takes (names as
in the checkpoint: ``RGB/inception_i3d/Conv3d_1a_7x7/conv_3d/w`` ...).  ``write_bundle`` produces files TensorFlow can
restore.  Format restated from the published LevelDB table / tensor_bundle descriptions; no checkpoint ships with the
reference, so the reader is pinned by round trips through the writer and by hand-built tables (tests/test_tf_checkpoint_cpu.py).
"""
import os
import struct

import numpy as np

from .tfrecord_io import _enc_varint, _fields, _ld, _varint, masked_crc

TABLE_MAGIC = 0xDB4775248B80FB57
# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DT_BFLOAT16 = 14
_DTYPE_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


# ---- SSTable -----------------------------------------------------------------------------------------------------------
def _block_entries(block):
    """(key, value) pairs of one table block: prefix-compressed entries, then the restart array and its length."""
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    i, key = 0, b""
    while i < end:
        shared, i = _varint(block, i)
        unshared, i = _varint(block, i)
        vlen, i = _varint(block, i)
        key = key[:shared] + bytes(block[i:i + unshared])
        i += unshared
        yield key, bytes(block[i:i + vlen])
        i += vlen


def _read_block(buf, offset, size, verify):
    block, ctype = buf[offset:offset + size], buf[offset + size]
    if verify:
        want = struct.unpack_from("<I", buf, offset + size + 1)[0]
        if masked_crc(bytes(buf[offset:offset + size + 1])) != want:
            raise ValueError(f"table block at {offset}: crc mismatch")
    if ctype != 0:
        raise NotImplementedError("snappy-compressed table blocks are not supported (TensorFlow writes bundles uncompressed)")
    return block


def read_table(path, verify=False):
    """-> list of (key bytes, value bytes) of an SSTable file in key order"""
    buf = memoryview(open(path, "rb").read())
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise ValueError(f"{path}: not an SSTable (bad magic)")
    footer = buf[len(buf) - 48:]
    _, i = _varint(footer, 0)            # metaindex handle (unused)
    _, i = _varint(footer, i)
    ioff, i = _varint(footer, i)
    isize, i = _varint(footer, i)
    out = []
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        off, j = _varint(handle, 0)
        size, _ = _varint(handle, j)
        out.extend(_block_entries(_read_block(buf, off, size, verify)))
    return out


def _build_block(items, restart_interval=16):
    out, restarts, prev = bytearray(), [], b""
    for n, (k, v) in enumerate(items):
        shared = 0
        if n % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _enc_varint(shared) + _enc_varint(len(k) - shared) + _enc_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_table(path, items, block_size=4096):
    """items: iterable of (key bytes, value bytes), sorted by key"""
    items = list(items)
    assert all(items[i][0] < items[i + 1][0] for i in range(len(items) - 1)), "keys must be strictly increasing"
    f, index, pos = bytearray(), [], 0

    def emit(block):
        nonlocal pos
        f.extend(block + b"\x00" + struct.pack("<I", masked_crc(block + b"\x00")))
        handle = _enc_varint(pos) + _enc_varint(len(block))
        pos += len(block) + 5
        return handle

    cur, cur_bytes = [], 0
    for k, v in items:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 3
        if cur_bytes >= block_size:
            index.append((cur[-1][0], emit(_build_block(cur))))
            cur, cur_bytes = [], 0
    if cur or not items:
        index.append((cur[-1][0] if cur else b"", emit(_build_block(cur))))
    meta = emit(_build_block([]))
    idx = emit(_build_block(index, restart_interval=1))
    footer = meta + idx
    f.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    with open(path, "wb") as fh:
        fh.write(bytes(f))


# ---- bundle protos -------------------------------------------------------------------------------------------------------
def _parse_entry(buf):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for fn, wt, v in _fields(memoryview(buf)):
        if fn == 1:
            e["dtype"] = v
        elif fn == 2:
            for f2, _, v2 in _fields(v):
                if f2 == 2:                       # TensorShapeProto.Dim
                    size = 0
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            size = v3 - (1 << 64) if v3 >= (1 << 63) else v3
                    e["shape"].append(size)
        elif fn == 3:
            e["shard_id"] = v
        elif fn == 4:
            e["offset"] = v
        elif fn == 5:
            e["size"] = v
        elif fn == 6:
            e["crc32c"] = struct.unpack("<I", bytes(v))[0]
        elif fn == 7:
            e["sliced"] = True
    return e


def _enc_entry(dtype_code, shape, offset, size, crc):
    dims = b"".join(_ld(2, bytes([0x08]) + _enc_varint(int(d))) for d in shape)
    out = bytes([0x08]) + _enc_varint(dtype_code) + _ld(2, dims)
    if offset:
        out += bytes([0x20]) + _enc_varint(offset)
    out += bytes([0x28]) + _enc_varint(size)
    if crc is not None:
        out += bytes([0x35]) + struct.pack("<I", crc)
    return out


def _parse_header(buf):
    h = dict(num_shards=1, endianness=0)
    for fn, _, v in _fields(memoryview(buf)):
        if fn == 1:
            h["num_shards"] = v
        elif fn == 2:
            h["endianness"] = v
    return h


def list_variables(prefix):
    """-> {name: (dtype code, shape)} like tf.train.list_variables"""
    return {k.decode(): (e["dtype"], tuple(e["shape"])) for k, e in ((k, _parse_entry(v)) for k, v in read_table(prefix + ".index") if k)}


def read_bundle(prefix, names=None, verify_crc=False):
    """-> {variable name: ndarray}.  ``names``: optional filter (iterable of names or a predicate on the name)."""
    entries = read_table(prefix + ".index", verify=verify_crc)
    header = _parse_header(dict(entries).get(b"", b""))
    if header["endianness"] != 0:
        raise NotImplementedError("big-endian bundles are not supported")
    want = names if callable(names) or names is None else set(names).__contains__
    shards, out = {}, {}
    for k, v in entries:
        name = k.decode()
        if not k or (want is not None and not want(name)):
            continue
        e = _parse_entry(v)
        if e["sliced"]:
            raise NotImplementedError(f"{name}: partitioned (sliced) variables are not supported")
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap(f"{prefix}.data-{sid:05d}-of-{header['num_shards']:05d}", dtype=np.uint8, mode="r")
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        if verify_crc and e["crc32c"] is not None and masked_crc(raw.tobytes()) != e["crc32c"]:
            raise ValueError(f"{name}: tensor crc mismatch")
        if e["dtype"] == _DT_BFLOAT16:
            arr = (np.frombuffer(raw.tobytes(), dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)
        elif e["dtype"] in _DTYPES:
            arr = np.frombuffer(raw.tobytes(), dtype=_DTYPES[e["dtype"]])
        else:
            raise NotImplementedError(f"{name}: dtype code {e['dtype']} (strings / resources) is not supported")
        out[name] = arr.reshape(e["shape"]).copy()
    return out


def write_bundle(prefix, tensors, with_crc=True):
    """Write ``{name: ndarray}`` as a one-shard V2 checkpoint (``prefix.index`` + ``prefix.data-00000-of-00001``)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items, offset = [], 0
    header = bytes([0x08, 0x01]) + _ld(3, bytes([0x08, 0x01]))          # num_shards = 1, version.producer = 1
    items.append((b"", header))
    # a reader discovers a bundle by its .index file (latest_checkpoint globs for it): the data shard is complete before the index
    # appears, and both appear atomically (temporary name + rename), so a concurrent reader never sees a half-written bundle
    data_path, index_path = prefix + ".data-00000-of-00001", prefix + ".index"
    with open(data_path + ".tmp", "wb") as f:
        for name in sorted(tensors, key=lambda s: s.encode()):
            a = np.asarray(tensors[name])                            # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in _DTYPE_CODES:
                raise TypeError(f"{name}: unsupported dtype {a.dtype}")
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            f.write(raw)
            items.append((name.encode(), _enc_entry(_DTYPE_CODES[a.dtype], a.shape, offset, len(raw), masked_crc(raw) if with_crc else None)))
            offset += len(raw)
    os.replace(data_path + ".tmp", data_path)
    write_table(index_path + ".tmp", items)
    os.replace(index_path + ".tmp", index_path)


def load_i3d_checkpoint(ckpt_path, scope="RGB"):
    """The variables ``init_model`` restores (kinetics_i3d_utils.py:41-62): everything under ``<scope>/``, including the
    batch-norm moving averages; returns the dict FlickerI3D takes."""
    return read_bundle(ckpt_path, names=lambda n: n.split("/")[0] == scope)
