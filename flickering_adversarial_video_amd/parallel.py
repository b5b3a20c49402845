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


def launch_ranks(n, script, argv):
    """Start `script argv` as n ranks of ONE node (one process per GPU, `python -m torch.distributed.run`, rendezvous on 127.0.0.1)
    from a parent that has NOT touched the GPU, wait, and return the launcher's exit code (non-zero if any rank failed).  This is what
    `--gpus N` means in bench.py and the universal / class-generalisation scripts when no launcher has set WORLD_SIZE -- the
    reference's multi-GPU entry is one command too (nn.DataParallel over DEVICES_IDS, r2plus1d_main_universal_attack.py:30-33,
    model.py:576-578).  The ranks are CHILD processes: a process that has initialised HIP must never be replaced by exec."""
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(script)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def ranks_from_env(gpus):
    """(world, rank, local_rank) from the launcher's environment; refuses (SystemExit 2) a world size that differs from an explicit
    --gpus request instead of silently running a different job."""
    import os
    import sys
    world, rk, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if gpus is not None and gpus != world:
        print(f"--gpus {gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to run", file=sys.stderr)
        raise SystemExit(2)
    return world, rk, local


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


class DirectRccl:
    """The all-reduce through the C ABI (flk_allreduce_sum_f32: RCCL on the caller's HIP stream, no event hop between the attack
    kernels and the collective) instead of torch.distributed's own NCCL binding.  Opt-in (FLK_RCCL_DIRECT=1): torch.distributed is
    still what launches the ranks and carries the 128-byte RCCL id from rank 0 to the others."""

    def __init__(self, device, group=None):
        import ctypes as C
        from . import _lib
        self._lib, self._C = _lib, C
        lib = _lib.load()
        rk, world = rank(group), world_size(group)
        uid = torch.zeros(128, dtype=torch.uint8)
        if rk == 0:
            buf = (C.c_char * 128)()
            _lib.check(lib.flk_comm_unique_id(buf))
            uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if world > 1:
            dev_uid = uid.cuda(device) if dist.get_backend(group) == "nccl" else uid
            dist.broadcast(dev_uid, src=0, group=group)
            uid = dev_uid.cpu()
        self.handle = C.c_void_p()
        raw = (C.c_char * 128).from_buffer_copy(bytes(uid.numpy().tobytes()))
        _lib.check(lib.flk_comm_create(raw, rk, world, int(device), C.byref(self.handle)))
        self.world = world

    def allreduce_sum_(self, t):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self._lib.check(self._lib.load().flk_allreduce_sum_f32(self.handle, self._C.c_void_p(t.data_ptr()), t.numel(),
                                                                self._C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return t

    def close(self):
        if self.handle:
            self._lib.load().flk_comm_destroy(self.handle)
            self.handle = None


_direct = {}


def direct_rccl(device, group=None):
    """the process-wide DirectRccl communicator of ``group`` (created on first use) when FLK_RCCL_DIRECT=1, else None"""
    import os
    if os.environ.get("FLK_RCCL_DIRECT", "0") != "1":
        return None
    key = id(group)
    if key not in _direct:
        _direct[key] = DirectRccl(device, group)
    return _direct[key]


def allreduce_sum_(payload, group=None):
    """in-place sum over ranks (no-op for a single process unless FLK_FORCE_COLLECTIVE=1)"""
    if world_size(group) > 1 or force_collective():
        # (fp32 payloads: the gradient / loss sums; the float64 fooling counters of evaluate() stay on torch.distributed)
        direct_ok = payload.is_cuda and payload.dtype == torch.float32 and payload.is_contiguous()
        comm = direct_rccl(payload.device.index, group) if direct_ok else None
        if comm is not None:
            return comm.allreduce_sum_(payload)
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
