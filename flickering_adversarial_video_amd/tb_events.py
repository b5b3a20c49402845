"""TensorBoard scalar summaries without TensorFlow.

The universal attack logs its scalars every 50 steps into ``<model_dir>/train`` (``tf.summary.scalar`` +
``SummarySaverHook(save_steps=50)``, i3d_adversarial_main_universal.py:176-201).  An event file is a TFRecord stream of
``Event`` protos: ``Event{1: wall_time double, 2: step int64, 3: file_version string | 5: Summary{1: Value{1: tag, 2: simple_value float}}}``.
``SCALAR_TAGS`` maps the reference's tag names onto the keys of ``StepResult``.
"""
import os
import socket
import struct
import time

from .tfrecord_io import _enc_varint, _fields, _ld, masked_crc, read_records

# tag (i3d_adversarial_main_universal.py:176-194) -> StepResult key
SCALAR_TAGS = {
    "Loss/total": "total_loss",
    "Loss/adversarial_loss": "adv_loss",
    "Loss/regularizer_loss": "reg_loss",
    "Loss/thickness": "norm_reg",
    "Loss/first_order_temporal_diff": "diff_norm_reg",
    "Loss/second_order_temporal_diff": "laplacian_norm_reg",
    "Perturbation/thickness_%": "thickness_relative",
    "Perturbation/roughness_%": "roughness_relative",
    "Perturbation/max": "pert_max",
    "Perturbation/min": "pert_min",
    "Probability/prob_to_min": "prob_to_min",
    "Probability/prob_to_max": "prob_to_max",
}


def _event(wall_time, step, payload):
    return bytes([0x09]) + struct.pack("<d", wall_time) + bytes([0x10]) + _enc_varint(step & ((1 << 64) - 1)) + payload


class SummaryWriter:
    """``tf.summary.FileWriter`` for scalars: ``add_scalars(step, {tag: value})``; files are named like TensorFlow's
    (``events.out.tfevents.<time>.<host>``) so TensorBoard picks them up."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}")
        self._f = open(self.path, "wb")
        self._write(_event(time.time(), 0, _ld(3, b"brain.Event:2")))

    def _write(self, data):
        head = struct.pack("<Q", len(data))
        self._f.write(head + struct.pack("<I", masked_crc(head)) + data + struct.pack("<I", masked_crc(data)))

    def add_scalars(self, step, scalars, wall_time=None):
        values = b"".join(_ld(1, _ld(1, tag.encode()) + bytes([0x15]) + struct.pack("<f", float(v))) for tag, v in scalars.items())
        self._write(_event(time.time() if wall_time is None else wall_time, int(step), _ld(5, values)))

    def add_step_result(self, step, host_result, beta0=None):
        """host_result: ``StepResult.host()``; writes every reference tag whose key is present (+ the weighted regulariser)"""
        sc = {tag: float(host_result[key]) for tag, key in SCALAR_TAGS.items() if key in host_result}
        if beta0 is not None and "reg_loss" in host_result:
            sc["Loss/regularizer_loss_weighted"] = float(beta0) * float(host_result["reg_loss"])
        self.add_scalars(step, sc)

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_scalars(path):
    """-> list of (step, {tag: value}) of an event file (tests / post-processing)"""
    out = []
    for payload in read_records(path, verify_crc=True):
        step, vals = 0, {}
        for fn, _, v in _fields(memoryview(payload)):
            if fn == 2:
                step = v
            elif fn == 5:
                for f2, _, v2 in _fields(v):
                    if f2 == 1:
                        tag, val = None, None
                        for f3, _, v3 in _fields(v2):
                            if f3 == 1:
                                tag = bytes(v3).decode()
                            elif f3 == 2:
                                val = struct.unpack("<f", bytes(v3))[0]
                        vals[tag] = val
        if vals:
            out.append((step, vals))
    return out
