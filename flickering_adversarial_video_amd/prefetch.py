"""Input prefetch for the dataset-driven attacks: the step before the hot path.

The reference feeds each step through ``tf.data`` (``TFRecordDataset.map(parse).batch(8).prefetch``,
i3d_adversarial_main_single_class_gen.py:130-144) and a ``feed_dict`` host-to-device copy per ``sess.run``.  Here a background
thread reads and parses the next uint8 batches straight into a ring of PINNED host buffers (no allocation per batch: first-touch
page faults of fresh 77 MB arrays cost more than the read), and the consumer turns a filled buffer into a device tensor with one
asynchronous copy on its own stream -- the uint8 bytes (77 MB for 8 x 64 x 224 x 224 x 3) are what crosses PCIe, the
``u8/128 - 1`` conversion happens in the apply kernel.  At 7 ms per step the attack consumes 8 clips x 144 steps/s = 11 GB/s per
GPU, more than one PCIe Gen5 x16 link or any disk delivers: with real data the loop is input-bound, and the loader's job is to
stay out of the way (measured on the MI355X box, tools/input_pipeline_rate.py: 19 GB/s from the page cache, and 6.91 ms per step with
a new batch every step against 6.90 ms with the batch resident)."""
import queue
import threading

import numpy as np
import torch

from . import tfrecord_io as tio


class _ConsumerGone(Exception):
    pass


class DeviceBatches:
    """iterate ``(clips uint8 [B,T,224,224,3] on the device, labels int64 [B] on the host as numpy)`` over TFRecord files.

    ``depth`` buffers circulate: reader thread fills -> consumer copies to the device -> buffer returns to the reader once the
    copy has completed (an event per buffer).  The yielded device tensor stays valid until ``depth - 1`` further batches have been
    requested (the attack consumes it within the same step)."""

    def __init__(self, files, batch_size, frames, rank=0, world=1, device=None, depth=3):
        self.args = (files, batch_size, frames, rank, world)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None and torch.cuda.is_available() else device
        cuda = self.device is not None and torch.device(self.device).type == "cuda"
        shape = (batch_size, frames, 224, 224, 3)
        self.host = [torch.empty(shape, dtype=torch.uint8, pin_memory=cuda) for _ in range(depth)]
        self.host_np = [h.numpy() for h in self.host]
        self.labels = [np.empty(batch_size, dtype=np.int64) for _ in range(depth)]
        self.dev = [torch.empty(shape, dtype=torch.uint8, device=self.device) for _ in range(depth)] if cuda else None
        self.done = [torch.cuda.Event() for _ in range(depth)] if cuda else None
        self.used = [None] * depth          # consumer-side events: dev[k] may be overwritten only after them
        self.depth = depth

    def __iter__(self):
        free, full = queue.Queue(), queue.Queue(maxsize=self.depth)
        for k in range(self.depth):
            free.put(k)
        stop = threading.Event()

        def reader():
            """fills free buffers in order; ends quietly when the consumer stops early (a None arrives on the free queue)"""
            def bufs():
                while True:
                    k = free.get()
                    if k is None:
                        raise _ConsumerGone
                    yield self.host_np[k], self.labels[k]

            try:
                for k_clips, _ in tio.batches(*self.args, buffers=bufs()):
                    if stop.is_set():
                        return
                    full.put(next(i for i, h in enumerate(self.host_np) if h is k_clips))
                full.put(None)
            except RuntimeError as e:                       # PEP 479 wraps the generator's _ConsumerGone
                if not isinstance(e.__cause__, _ConsumerGone) and not stop.is_set():
                    full.put(e)
            except _ConsumerGone:
                pass
            except BaseException as e:                      # noqa: BLE001 -- hand the error to the consumer instead of dying silently
                if not stop.is_set():
                    full.put(e)

        th = threading.Thread(target=reader, daemon=True)
        th.start()
        cuda = self.dev is not None
        copy_stream = torch.cuda.Stream(device=self.device) if cuda else None
        in_copy = []                                   # host buffers whose H2D copy may still be running (oldest first)

        def fetch():
            """next filled buffer -> (k, labels); on a GPU the H2D copy of its clips is issued on the copy stream right away, so
            it runs while the consumer still computes on the previous batch"""
            k = full.get()
            if k is None:
                return None
            if isinstance(k, BaseException):
                raise k
            labels = self.labels[k].copy()
            if cuda:
                if self.used[k] is not None:
                    copy_stream.wait_event(self.used[k])          # the consumer's kernels that read dev[k] `depth` batches ago
                with torch.cuda.stream(copy_stream):
                    self.dev[k].copy_(self.host[k], non_blocking=True)
                    self.done[k].record(copy_stream)
                in_copy.append(k)
                while len(in_copy) > 1:                            # a host buffer returns to the reader once its copy has finished
                    j = in_copy.pop(0)
                    self.done[j].synchronize()
                    free.put(j)
            return k, labels

        try:
            nxt = fetch()
            while nxt is not None:
                k, labels = nxt
                nxt = fetch()                                      # issue the NEXT copy before handing out this batch
                if cuda:
                    torch.cuda.current_stream().wait_event(self.done[k])
                    x = self.dev[k]
                else:
                    x = self.host[k].clone()
                    free.put(k)
                yield x, labels
                if cuda:                                           # the consumer is back: everything it enqueued on dev[k] is ordered before this
                    self.used[k] = torch.cuda.Event()
                    self.used[k].record()
        finally:
            stop.set()
            free.put(None)
            # The reader may still be filling a pinned buffer (it looks at `stop` only between batches) or be blocked on a full
            # queue: keep making room until it has ENDED.  A DeviceBatches object is iterated again every epoch and the ring of
            # pinned buffers is shared between iterations -- a second reader must never start while the first can still write.
            while th.is_alive():
                try:
                    while True:
                        full.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)
            if cuda:
                copy_stream.synchronize()                      # copies still reading host buffers that the next iteration refills
