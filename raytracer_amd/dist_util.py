"""Small torch.distributed helpers for the one-process-per-GPU launch.

The data path has no collective (independent op streams per GPU).  Communication
is of two kinds, and they use different process groups on purpose:

  control   barrier around a timed region, max / sum of per-rank scalars, gathering
            small python objects.  Always the DEFAULT group, backend gloo, host
            tensors: it cannot fail for a GPU reason, so the sharded op streams
            (which need nothing else) produce their numbers whether or not RCCL
            comes up on the node.
  data      the one real exchange, the all-gather of k-NN partial lists.  A second
            group, backend "nccl" (RCCL over xGMI on ROCm), created on first use by
            data_group() and checked with one tiny all-reduce; every rank then
            agrees over the control group whether it works.  If it does not, the
            exchange goes through host memory over gloo (raytracer_amd.knn handles
            a gloo group with device tensors) and data_backend() says so.
"""
import datetime
import os

import torch
import torch.distributed as dist

_DATA = {"tried": False, "group": None, "backend": "none", "error": None}


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend="gloo", timeout_s=900):
    """Initialise the default (control) process group from the torchrun environment. Returns (world, rank, local).

    `backend` is the control backend; anything but gloo is accepted for callers that insist, but the
    helpers below always move host tensors."""
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend or "gloo", timeout=datetime.timedelta(seconds=timeout_s))
    return world, rank, local


def _active():
    return dist.is_available() and dist.is_initialized()


def barrier():
    if _active():
        dist.barrier()


def _reduce(value, op):
    if not _active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)  # host tensor: the control group is gloo
    if dist.get_backend() != "gloo":
        t = t.cuda()
    dist.all_reduce(t, op=op)
    return float(t.item())


def max_over_ranks(value, device=None):  # `device` is ignored (kept for older callers): control data stays on the host
    return _reduce(value, dist.ReduceOp.MAX)


def min_over_ranks(value):
    return _reduce(value, dist.ReduceOp.MIN)


def sum_over_ranks(value, device=None):
    return _reduce(value, dist.ReduceOp.SUM)


def gather_objects(obj):
    """List of every rank's object on every rank (control-plane sized data only)."""
    if not _active():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def data_group(timeout_s=120):
    """The group for device-to-device exchanges: RCCL when it works on every rank, otherwise None (= the default
    gloo group; the exchange is then staged through host memory).  Collective: every rank must call it."""
    if not _active() or dist.get_world_size() == 1:
        return None
    if _DATA["tried"]:
        return _DATA["group"]
    _DATA["tried"] = True
    ok, group = 1.0, None
    if os.environ.get("RT_DP_REHEARSE_ONE_GPU") == "1" or os.environ.get("RT_DP_NO_RCCL") == "1" or not torch.cuda.is_available():
        ok, _DATA["error"] = 0.0, "RCCL not attempted (several ranks on one GPU, RT_DP_NO_RCCL=1, or no GPU)"
    else:
        try:
            group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s))
            probe = torch.ones(1, device="cuda")
            dist.all_reduce(probe, group=group)
            torch.cuda.synchronize()
            if int(probe.item()) != dist.get_world_size():
                raise RuntimeError("RCCL all-reduce returned %r" % probe.item())
        except Exception as e:  # noqa: BLE001 - whatever RCCL raises, the control plane decides what happens next
            ok, _DATA["error"] = 0.0, "%s: %s" % (type(e).__name__, e)
    if min_over_ranks(ok) < 1.0:   # some rank could not: nobody uses RCCL
        group = None
        _DATA["backend"] = "gloo (host staging)"
    else:
        _DATA["backend"] = "nccl (RCCL)"
    _DATA["group"] = group
    return group


def data_backend():
    """What data_group() resolved to, for reports: 'nccl (RCCL)', 'gloo (host staging)' or 'none' (single process /
    not asked yet); data_error() holds this rank's reason when RCCL was not used."""
    return _DATA["backend"]


def data_error():
    return _DATA["error"]


def shutdown():
    if _active():
        try:
            dist.barrier()
        finally:
            dist.destroy_process_group()
