"""uint8 Kinetics TFRecords without TensorFlow.

The reference writes one ``tf.train.Example`` per video with two features (kinetics_to_tf_record_uint8.py:82-95):
``train/label`` (int64) and ``train/video`` (bytes: the LAST 90 frames, uint8 [T,224,224,3]); readers decode with
``tf.decode_raw`` and scale ``x = u8/128 - 1`` (pre_process_rgb_flow.py:211-236).  Here the records stay uint8 all the way
into HBM -- the scaling happens inside the apply kernel (flk_perturb_apply_s2d, x_is_u8).

TFRecord framing: u64 length | u32 masked-crc32c(length) | payload | u32 masked-crc32c(payload).
Example proto:   Example{1: Features{1: map<string, Feature{1: BytesList | 2: FloatList | 3: Int64List}>}}.
"""
import glob
import os
import struct

import numpy as np

# ---- crc32c (Castagnoli), needed only to WRITE valid files and to verify on request ------------------------------
_TABLE = None


def _table():
    global _TABLE
    if _TABLE is None:
        t = np.zeros(256, dtype=np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t[i] = c
        _TABLE = t
    return _TABLE


def crc32c(data: bytes) -> int:
    t, c = _table(), 0xFFFFFFFF
    for b in data:
        c = int(t[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protobuf wire format (just what Example needs) -----------------------------------------------------------------
def _varint(buf, i):
    shift = v = 0
    while True:
        b = buf[i]
        i += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, i
        shift += 7


def _fields(buf):
    """yield (field number, wire type, value) of one message; value = int (varint) or memoryview (length-delimited)"""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 2:
            ln, i = _varint(buf, i)
            v = buf[i:i + ln]
            i += ln
        elif wt == 1:
            v = buf[i:i + 8]; i += 8
        elif wt == 5:
            v = buf[i:i + 4]; i += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fn, wt, v


def parse_example(payload, copy=True):
    """-> {feature name: bytes | np.int64 array | np.float32 array}.  copy=False leaves byte features as memoryviews into ``payload``
    (a clip is 9.6-13.5 MB: the batch loader copies it exactly once, into the batch array)"""
    buf = memoryview(payload)
    out = {}
    for fn, _, features in _fields(buf):
        if fn != 1:
            continue
        for fn2, _, entry in _fields(features):              # map entries
            if fn2 != 1:
                continue
            key, feat = None, None
            for fn3, _, v in _fields(entry):
                if fn3 == 1:
                    key = bytes(v).decode()
                elif fn3 == 2:
                    feat = v
            for kind, _, lst in _fields(feat):
                if kind == 1:                                  # BytesList
                    vals = [bytes(v) if copy else v for f, _, v in _fields(lst) if f == 1]
                    out[key] = vals[0] if len(vals) == 1 else vals
                elif kind == 3:                                # Int64List (packed or not)
                    vals = []
                    for f, wt, v in _fields(lst):
                        if f != 1:
                            continue
                        if wt == 0:
                            vals.append(v)
                        else:
                            j = 0
                            while j < len(v):
                                x, j = _varint(v, j)
                                vals.append(x)
                    out[key] = np.array([x - (1 << 64) if x >> 63 else x for x in vals], dtype=np.int64)
                elif kind == 2:                                # FloatList
                    vals = []
                    for f, wt, v in _fields(lst):
                        if f == 1:
                            vals.append(np.frombuffer(bytes(v), dtype="<f4"))
                    out[key] = np.concatenate(vals) if vals else np.zeros(0, np.float32)
    return out


def read_records(path, verify_crc=False):
    """yield raw Example payloads of one .tfrecords file"""
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise ValueError(f"{path}: truncated record header")
            (length,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if masked_crc(head[:8]) != lcrc:
                raise ValueError(f"{path}: corrupt record length")
            data = f.read(length)
            tail = f.read(4)
            if len(data) < length or len(tail) < 4:
                raise ValueError(f"{path}: truncated record")
            if verify_crc and masked_crc(data) != struct.unpack("<I", tail)[0]:
                raise ValueError(f"{path}: payload crc mismatch")
            yield data


def parse_example_uint8(payload, frames=90, size=224, copy=True):
    """pre_process_rgb_flow.py:211-236 without the float conversion: (uint8 [T,size,size,3], int64 label); copy=False: the clip is a
    view into ``payload``"""
    ex = parse_example(payload, copy)
    video = np.frombuffer(ex["train/video"], dtype=np.uint8)
    per = size * size * 3
    if video.size % per:
        raise ValueError(f"train/video has {video.size} bytes: not a whole number of {size}x{size}x3 frames")
    video = video.reshape(-1, size, size, 3)
    if video.shape[0] < frames:
        raise ValueError(f"record holds {video.shape[0]} frames, {frames} requested")
    return video[-frames:], int(ex["train/label"][0])


def list_tfrecords(paths, limit=None):
    """glob + sort + truncate, as the scripts do (i3d_adversarial_main_single_class_gen.py:111-120)"""
    files = []
    for p in ([paths] if isinstance(paths, str) else paths):
        files += sorted(glob.glob(os.path.join(p, "*.tfrecords")))
    return files[:limit] if limit else files


def record_spans(path):
    """yield (payload offset, payload length) of every record of one .tfrecords file WITHOUT reading the payloads
    (a 90-frame clip is 13.5 MB: a rank only reads the records it owns)"""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        pos = 0
        while pos < size:
            f.seek(pos)
            head = f.read(12)
            if len(head) < 12:
                raise ValueError(f"{path}: truncated record header")
            (length,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if masked_crc(head[:8]) != lcrc:
                raise ValueError(f"{path}: corrupt record length")
            if pos + 12 + length + 4 > size:
                raise ValueError(f"{path}: truncated record")
            yield pos + 12, length
            pos += 12 + length + 4


def _read_span(path, off, length):
    with open(path, "rb") as f:
        f.seek(off)
        return f.read(length)


def batches(files, batch_size, frames=90, rank=0, world=1, drop_remainder=True, buffers=None):
    """uint8 batches [B,T,224,224,3] + labels.  ``buffers``: optional iterator of ``(clips, labels)`` numpy arrays to fill instead of
    allocating a new pair per batch (prefetch.DeviceBatches hands in pinned buffers); a batch is then yielded as those arrays.

    Data-parallel sharding: the record stream is cut into GLOBAL batches of ``world * batch_size`` consecutive records and
    rank r takes records [r*B, (r+1)*B) of each; the global remainder is dropped.  Every rank therefore yields the SAME number
    of batches by construction (each step is followed by a collective: unequal counts would dead-lock the all-reduce), and
    reads only the payloads it owns."""
    if world > 1 and not drop_remainder:
        raise ValueError("drop_remainder=False is only meaningful for a single rank (ranks must see equal batch counts)")
    per_global = world * batch_size
    spans = []

    def load(span_list):
        # one read + one copy per clip: the payload is parsed in place and the frames go straight into the batch array
        # (np.stack of 8 clips took 2 s here -- 30 MB/s -- against 0.03 s for slice assignment)
        if buffers is not None and len(span_list) == batch_size:
            clips, labels = next(buffers)
        else:
            clips = np.empty((len(span_list), frames, 224, 224, 3), dtype=np.uint8)
            labels = np.empty(len(span_list), dtype=np.int64)
        for i, (path, off, ln) in enumerate(span_list):
            v, labels[i] = parse_example_uint8(_read_span(path, off, ln), frames, copy=False)
            clips[i] = v
        return clips, labels

    for path in files:
        for off, ln in record_spans(path):
            spans.append((path, off, ln))
            if len(spans) == per_global:
                yield load(spans[rank * batch_size:(rank + 1) * batch_size])
                spans = []
    if spans and not drop_remainder:
        yield load(spans)


def count_batches(files, batch_size, world=1):
    """number of batches every rank gets from ``batches`` (headers only)"""
    return sum(1 for path in files for _ in record_spans(path)) // (world * batch_size)


# ---- writer (tests / dataset preparation) -----------------------------------------------------------------------------
def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(fn, payload):
    return _enc_varint((fn << 3) | 2) + _enc_varint(len(payload)) + payload


def make_example(video_u8, label):
    """kinetics_to_tf_record_uint8.py:82-95"""
    def entry(key, feature):
        return _ld(1, _ld(1, key.encode()) + _ld(2, feature))
    f_label = _ld(3, _ld(1, _enc_varint(label & ((1 << 64) - 1))))          # Int64List, packed
    f_video = _ld(1, _ld(1, np.ascontiguousarray(video_u8, dtype=np.uint8).tobytes()))
    return _ld(1, entry("train/label", f_label) + entry("train/video", f_video))


def write_records(path, payloads, with_payload_crc=True):
    with open(path, "wb") as f:
        for data in payloads:
            head = struct.pack("<Q", len(data))
            f.write(head + struct.pack("<I", masked_crc(head)) + data + struct.pack("<I", masked_crc(data) if with_payload_crc else 0))
