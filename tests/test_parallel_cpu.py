"""world_size-2 gloo test of the data-parallel protocol (flickering_adversarial_video_amd/parallel.py) on CPU.

Each rank computes the adversarial delta-gradient of ITS shard of videos (the oracle's attack maths on a tiny
conv3d victim stands in for the HIP network), the payload is sum-all-reduced, every rank adds the regulariser gradient
once and runs TF-Adam.  The result must equal the single-process update on the concatenated batch: for the summed
margin loss and for the mean CE loss.  Fooling-rate counters are reduced the same way."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from flickering_adversarial_video_amd import parallel
from oracle import attack_math as am

T, B_PER = 8, 3


def victim(x_cl, W):
    x = x_cl.permute(0, 4, 1, 2, 3)
    h = F.relu(F.conv3d(x, W["w1"], padding=1))
    return F.linear(h.mean(dim=(2, 3, 4)), W["fw"], W["fb"])


def make_data(world):
    rng = np.random.default_rng(11)
    W = {"w1": torch.from_numpy(rng.standard_normal((6, 3, 3, 3, 3)).astype(np.float32) * 0.3),
         "fw": torch.from_numpy(rng.standard_normal((400, 6)).astype(np.float32)), "fb": torch.zeros(400)}
    x = torch.from_numpy(rng.uniform(-1, 1, (world * B_PER, T, 6, 6, 3)).astype(np.float32))
    delta = torch.from_numpy(rng.uniform(-0.05, 0.05, (T, 1, 1, 3)).astype(np.float32))
    labels = victim(x, W).argmax(-1)
    return W, x, delta, labels


def local_payload(W, x, labels, delta, improve, gbatch):
    d = delta.clone().requires_grad_(True)
    lg = victim(am.tf_apply(x, d), W)
    if improve:
        adv, to_min, to_max = am.tf_improve_adversarial_loss(lg, labels, 0.05, False, False)
        per = torch.stack([torch.zeros(len(x)), torch.softmax(lg, -1).gather(1, labels.view(-1, 1))[:, 0],
                           (torch.softmax(lg, -1) - F.one_hot(labels, 400)).max(-1)[0]], 1).detach()
        per[:, 0] = adv.item() / len(x)
    else:
        adv = (-torch.log(1 - torch.softmax(lg, -1).gather(1, labels.view(-1, 1)) + 1e-6)).sum() / gbatch   # mean over the GLOBAL batch
        per = torch.zeros(len(x), 3)
        per[:, 0] = adv.item() / len(x)
    (g,) = torch.autograd.grad(adv, d)
    payload = torch.zeros(parallel.payload_size(T))
    payload[:3 * T] = g.reshape(-1)
    parallel.pack_scalars(payload, T, per)
    return payload, lg.detach()


def update(payload, delta, gbatch):
    g_adv, adv_sum, p_min, p_max = parallel.unpack(payload, T, gbatch)
    d = delta.clone().requires_grad_(True)
    _, reg = am.tf_total_loss(torch.zeros(()), d, 1.0, 0.5, 0.5, 0.5)
    (g_reg,) = torch.autograd.grad(1.0 * reg, d)                       # regulariser gradient added ONCE, after the reduce
    z = torch.zeros_like(delta)
    new, _, _ = am.tf_adam_step(delta, g_adv.reshape(delta.shape) + g_reg, z, z, 1)
    return new, float(adv_sum)


def worker(rk, world, port, improve, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rk, world_size=world)
    try:
        W, x, delta, labels = make_data(world)
        lo, hi = parallel.shard_range(len(x), rk, world)
        assert (lo, hi) == (rk * B_PER, (rk + 1) * B_PER)
        payload, lg = local_payload(W, x[lo:hi], labels[lo:hi], delta, improve, len(x))
        parallel.allreduce_sum_(payload)
        new, adv = update(payload, delta, len(x))
        cnt = parallel.FoolingCounter()
        clean = victim(x[lo:hi], W).argmax(-1)
        cnt.update(lg.argmax(-1), clean, labels[lo:hi])
        out.put((rk, new.numpy(), adv, cnt.result()))
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("improve", [True, False], ids=["margin_sum", "ce_mean"])
def test_two_rank_update_equals_single_process(improve):
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, improve, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    # single process on the concatenated batch
    W, x, delta, labels = make_data(world)
    payload, lg = local_payload(W, x, labels, delta, improve, len(x))
    ref_new, ref_adv = update(payload, delta, len(x))
    ref_cnt = parallel.FoolingCounter()
    ref_cnt.update(lg.argmax(-1), victim(x, W).argmax(-1), labels)
    for rk, new, adv, fool in res:
        np.testing.assert_allclose(new, ref_new.numpy(), rtol=1e-5, atol=1e-8)      # identical replicas on every rank
        assert adv == pytest.approx(ref_adv, rel=1e-5, abs=1e-8)
        assert fool[1] == ref_cnt.result()[1] and fool[0] == pytest.approx(ref_cnt.result()[0], nan_ok=True)
    np.testing.assert_array_equal(res[0][1], res[1][1])


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 50):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_single_process_is_a_noop():
    p = torch.arange(float(parallel.payload_size(T)))
    assert parallel.world_size() == 1 and parallel.rank() == 0
    assert torch.equal(parallel.allreduce_sum_(p.clone()), p)
