"""TFRecord / tf.train.Example reader (no TensorFlow): round trip, framing errors, rank sharding."""
import numpy as np
import pytest

from flickering_adversarial_video_amd import tfrecord_io as tio


def test_crc32c_known_answers():
    assert tio.crc32c(b"123456789") == 0xE3069283            # RFC 3720 check value
    assert tio.crc32c(b"") == 0
    # TFRecord masking: ((crc >> 15) | (crc << 17)) + 0xa282ead8
    assert tio.masked_crc(b"123456789") == ((((0xE3069283 >> 15) | (0xE3069283 << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _records(n, T=3, size=4, seed=0):
    rng = np.random.default_rng(seed)
    return [(rng.integers(0, 256, (T, size, size, 3), dtype=np.uint8), int(rng.integers(0, 400))) for _ in range(n)]


def test_round_trip_and_last_frames(tmp_path):
    recs = _records(5)
    p = tmp_path / "a.tfrecords"
    tio.write_records(str(p), [tio.make_example(v, l) for v, l in recs])
    got = [tio.parse_example_uint8(d, frames=2, size=4) for d in tio.read_records(str(p), verify_crc=True)]
    assert len(got) == 5
    for (v, l), (gv, gl) in zip(recs, got):
        assert gl == l
        np.testing.assert_array_equal(gv, v[-2:])             # the LAST frames, like the reference writer/reader pair
    with pytest.raises(ValueError):
        tio.parse_example_uint8(next(tio.read_records(str(p))), frames=9, size=4)


def test_negative_label_and_unpacked_int64():
    ex = tio.parse_example(tio.make_example(np.zeros((1, 2, 2, 3), np.uint8), -3))
    assert ex["train/label"][0] == -3


def test_corruption_is_detected(tmp_path):
    p = tmp_path / "a.tfrecords"
    tio.write_records(str(p), [tio.make_example(*_records(1)[0])])
    raw = bytearray(p.read_bytes())
    raw[20] ^= 0xFF
    p.write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        list(tio.read_records(str(p), verify_crc=True))
    p.write_bytes(bytes(raw[:30]))
    with pytest.raises(ValueError):
        list(tio.read_records(str(p)))


def test_batches_shard_over_ranks(tmp_path):
    recs = _records(11, T=2, size=224, seed=3)
    for i in range(2):
        tio.write_records(str(tmp_path / f"f{i}.tfrecords"), [tio.make_example(v, l) for v, l in recs[i * 6:(i + 1) * 6]], with_payload_crc=False)
    files = tio.list_tfrecords(str(tmp_path))
    assert len(files) == 2 and tio.list_tfrecords([str(tmp_path)], limit=1) == files[:1]
    single = list(tio.batches(files, 2, frames=2))
    assert len(single) == 5 and single[0][0].shape == (2, 2, 224, 224, 3) and single[0][0].dtype == np.uint8
    # data-parallel: global batches of world*B consecutive records, rank r takes slice r; the global remainder is dropped,
    # so every rank sees the SAME number of batches (each step ends in a collective)
    r0, r1 = list(tio.batches(files, 2, frames=2, rank=0, world=2)), list(tio.batches(files, 2, frames=2, rank=1, world=2))
    assert len(r0) == len(r1) == 2 == tio.count_batches(files, 2, world=2)
    assert [int(l) for b in r0 for l in b[1]] == [recs[i][1] for i in (0, 1, 4, 5)]
    assert [int(l) for b in r1 for l in b[1]] == [recs[i][1] for i in (2, 3, 6, 7)]
    np.testing.assert_array_equal(r1[1][0][0], recs[6][0])
    with pytest.raises(ValueError):
        list(tio.batches(files, 2, frames=2, rank=0, world=2, drop_remainder=False))
    assert len(list(tio.batches(files, 2, frames=2, drop_remainder=False))) == 6


@pytest.mark.parametrize("n,world,B", [(28, 8, 2), (12, 8, 2), (11, 3, 2), (16, 8, 2), (5, 2, 4)])
def test_equal_batch_counts_per_rank(tmp_path, n, world, B):
    """the advisor's cases: N not divisible by world*B must not give ranks different batch counts"""
    recs = _records(n, T=1, size=224, seed=n)
    tio.write_records(str(tmp_path / "a.tfrecords"), [tio.make_example(v, l) for v, l in recs], with_payload_crc=False)
    files = tio.list_tfrecords(str(tmp_path))
    per_rank = [list(tio.batches(files, B, frames=1, rank=r, world=world)) for r in range(world)]
    counts = [len(x) for x in per_rank]
    assert counts == [n // (world * B)] * world == [tio.count_batches(files, B, world)] * world
    seen = [int(l) for k in range(counts[0]) for r in range(world) for l in per_rank[r][k][1]]
    assert seen == [recs[i][1] for i in range(counts[0] * world * B)]        # rank-major within a global batch, nothing read twice


def test_prefetcher_yields_the_same_batches_in_order(tmp_path):
    """prefetch.DeviceBatches (reader thread -> ring of reusable buffers) against the plain loader: same batches, same order, for
    both ranks of a 2-rank shard; buffers are reused, so a batch is compared before the next one is requested"""
    import numpy as np
    from flickering_adversarial_video_amd import prefetch, tfrecord_io as tio
    rng = np.random.default_rng(3)
    T = 4
    clips = [rng.integers(0, 256, (T, 224, 224, 3), dtype=np.uint8) for _ in range(9)]
    path = str(tmp_path / "a.tfrecords")
    tio.write_records(path, [tio.make_example(c, i) for i, c in enumerate(clips)])
    for rank in (0, 1):
        ref = list(tio.batches([path], 2, frames=T, rank=rank, world=2))
        assert len(ref) == 2                                   # 9 records -> 2 global batches of 4, remainder dropped
        n = 0
        for (x, y), (xr, yr) in zip(prefetch.DeviceBatches([path], 2, T, rank, 2, device="cpu", depth=2), ref):
            assert (x.numpy() == xr).all() and y.tolist() == yr.tolist()
            n += 1
        assert n == 2
    # two passes over the same object (epochs) restart the reader
    db = prefetch.DeviceBatches([path], 4, T, device="cpu")
    assert [y.tolist() for _, y in db] == [y.tolist() for _, y in db] == [[0, 1, 2, 3], [4, 5, 6, 7]]


def test_prefetcher_consumer_may_stop_early(tmp_path):
    """the attack loop leaves the epoch when MAX_NUM_STEP is reached: the reader thread must end instead of blocking, and the next
    pass over the same object starts from the first record again"""
    import threading
    import time
    import numpy as np
    from flickering_adversarial_video_amd import prefetch, tfrecord_io as tio
    T = 2
    rng = np.random.default_rng(4)
    path = str(tmp_path / "b.tfrecords")
    tio.write_records(path, [tio.make_example(rng.integers(0, 256, (T, 224, 224, 3), dtype=np.uint8), i) for i in range(12)])
    db = prefetch.DeviceBatches([path], 2, T, device="cpu", depth=2)
    before = threading.active_count()
    for n, (x, y) in enumerate(db):
        if n == 1:
            break
    assert y.tolist() == [2, 3]
    # the generator is closed when the for loop is left (refcount): its finally block JOINS the reader, so that a second pass can
    # never share the ring of pinned buffers with a reader that is still filling one -- no grace period needed
    assert threading.active_count() == before, "reader thread still alive after the consumer stopped"
    # many early exits in a row, then a full pass: contents (not only labels) must be those of the plain loader
    for _ in range(5):
        for x, y in db:
            break
        assert threading.active_count() == before
    ref = list(tio.batches([path], 2, T))
    got = [(x.numpy().copy(), y.copy()) for x, y in db]
    assert len(got) == len(ref) == 6
    for (x, y), (xr, yr) in zip(got, ref):
        assert y.tolist() == yr.tolist() and np.array_equal(x, xr)
    (time)
