"""Data-parallel protocol of the universal / class-generalisation attacks.

One process per GPU.  Videos are sharded; weights are frozen and resident per rank; delta and the Adam state are
replicated and updated identically everywhere (deterministic replicated optimiser, no parameter broadcast).  The ONLY
data-path exchange per iteration is one sum all-reduce of a small fp32 payload:

    [ d(sum_b adv_b)/d(delta)  (T*3) | sum_b adv_b | sum_b to_min_prob | sum_b to_max_prob ]

after which every rank adds the regulariser gradient ONCE and runs Adam.  This equals the single-process result with the
concatenated batch because the margin loss is a SUM over the batch (kinetics_i3d_utils.py:285); mean losses (CE,
kinetics_i3d_utils.py:305) are scaled by 1/global_batch inside the loss kernel.  On MI355X the backend is "nccl"
(= RCCL over xGMI); the same code runs on "gloo" for CPU tests.  (The reference's nn.DataParallel re-broadcasts all
31 M frozen parameters every forward, model.py:576-578; its TF MirroredStrategy is dead code,
i3d_adversarial_main_universal.py:309-312.)
"""
import torch
import torch.distributed as dist


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def payload_size(T):
    return 3 * T + 3


def pack_scalars(payload, T, per_clip):
    """per_clip: [B,>=3] = {loss_b, label_prob, max_non_label_prob}; writes the three batch sums behind the gradient."""
    payload[3 * T:3 * T + 3] = per_clip[:, :3].sum(0)
    return payload


def force_collective():
    """FLK_FORCE_COLLECTIVE=1: issue the collectives even in a 1-rank process group (they are the identity there) -- lets a
    one-GPU box execute the real RCCL all-reduce calls of the data-parallel path"""
    import os
    return os.environ.get("FLK_FORCE_COLLECTIVE", "0") == "1" and dist.is_available() and dist.is_initialized()


def allreduce_sum_(payload, group=None):
    """in-place sum over ranks (no-op for a single process unless FLK_FORCE_COLLECTIVE=1)"""
    if world_size(group) > 1 or force_collective():
        dist.all_reduce(payload, op=dist.ReduceOp.SUM, group=group)
    return payload


def unpack(payload, T, global_batch):
    g = payload[:3 * T].view(T, 3)
    return g, payload[3 * T], payload[3 * T + 1] / global_batch, payload[3 * T + 2] / global_batch


def shard_range(n_items, rk, world):
    """contiguous shard [lo, hi) of n_items records for rank rk; remainders go to the first ranks"""
    base, rem = divmod(n_items, world)
    lo = rk * base + min(rk, rem)
    return lo, lo + base + (1 if rk < rem else 0)


class FoolingCounter:
    """(miss, valid) counters of kinetics_i3d.evaluate (kinetics_i3d_utils.py:217-250), summed over ranks."""

    def __init__(self, device="cpu"):
        self.cnt = torch.zeros(2, dtype=torch.float64, device=device)

    def update(self, adv_argmax, clean_argmax, labels, targeted=False, target=None, exclude_misclassify=True):
        miss = (adv_argmax == target) if targeted else (adv_argmax != labels)
        if exclude_misclassify:
            valid = clean_argmax == labels
            self.cnt[0] += (miss & valid).sum()
            self.cnt[1] += valid.sum()
        else:
            self.cnt[0] += miss.sum()
            self.cnt[1] += miss.numel()

    def result(self, group=None):
        c = self.cnt.clone()
        allreduce_sum_(c, group)
        miss, total = c.tolist()
        return (miss / total if total else float("nan")), int(total)
