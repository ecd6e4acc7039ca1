"""One process per GPU over RCCL/xGMI (torch.distributed backend "nccl" IS RCCL on ROCm).

The reference is single-GPU (train.py:75 SimpleTrainer; no collective anywhere).  Every hot-path op
is independent per cloud, so the concatenated Siamese batch [anchors, positives, negatives,
other-negatives] (core/model.py:139-146) shards over ranks by cloud with NO data-path collective;
the only exchange is the all-gather of the [clouds_per_rank, 256] global descriptors that the
triplet / quadruplet loss needs in role order (core/losses.py:175-178).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun). Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:  # (DH3D_DIST_BACKEND=gloo: dev runs of several ranks on ONE GPU -- RCCL refuses two ranks per device)
            backend = os.environ.get("DH3D_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def shard_plan(num_clouds, world):
    """Contiguous block partition of the role-ordered batch, padded to equal counts.

    Returns (per_rank, padded_total); cloud i lives on rank i // per_rank at slot i % per_rank.
    RCCL all-gather needs equal counts, so the tail rank carries masked padding slots."""
    per_rank = (num_clouds + world - 1) // world
    return per_rank, per_rank * world


def local_slice(num_clouds, rank, world):
    """[start, stop) of the real clouds owned by `rank` (may be empty on tail ranks)."""
    per_rank, _ = shard_plan(num_clouds, world)
    start = min(rank * per_rank, num_clouds)
    return start, min(start + per_rank, num_clouds)


def shard_batch(points, rank, world):
    """points [Bt, N, 3] (role-ordered) -> this rank's [per_rank, N, 3] block, zero-padded + mask."""
    Bt = points.shape[0]
    per_rank, _ = shard_plan(Bt, world)
    start, stop = local_slice(Bt, rank, world)
    out = points.new_zeros((per_rank,) + tuple(points.shape[1:]))
    if stop > start:
        out[: stop - start] = points[start:stop]
        if stop - start < per_rank:  # padding clouds repeat a real one so kernels see valid geometry
            out[stop - start:] = points[start:start + 1]
    mask = torch.zeros(per_rank, dtype=torch.bool, device=points.device)
    mask[: stop - start] = True
    return out, mask


# The sharded code path with a single rank (bench.py "sharded_path_ms", tests): every collective of the step is issued
# on the (1-rank) process group instead of being skipped.
FORCE_COLLECTIVES = False
# calls of all_reduce_sum_ / all_gather_rows since the last reset (bench.py reports collectives per step)
COLLECTIVE_CALLS = [0]


def collectives_active():
    """Whether the step issues its collectives: a process group with more than one rank, or the forced 1-rank path."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def _staged(t):
    """gloo moves CUDA tensors only for broadcast / all_reduce; everything else is staged through the host there
    (the CPU-test / single-GPU multi-process path -- RCCL takes device tensors directly)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_gather_rows(local):
    """local [n, ...] (same n on every rank) -> [world*n, ...] in rank order."""
    world = dist.get_world_size()
    COLLECTIVE_CALLS[0] += 1
    if _staged(local):
        host = local.detach().cpu().contiguous()
        out = torch.empty((world * host.shape[0],) + tuple(host.shape[1:]), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host)
        return out.to(local.device)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def all_reduce_sum_(t):
    """In-place SUM all-reduce (device tensors over RCCL; host-staged under gloo)."""
    COLLECTIVE_CALLS[0] += 1
    if _staged(t):
        host = t.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_gather_descriptors(local_desc, num_clouds):
    """local_desc [per_rank, D] -> [num_clouds, D] in the original role order on every rank.

    One all-gather of per_rank*D floats per rank (24.6 KB total for the Oxford-shaped batch): latency
    bound, no reduction.  With world == 1 this is the identity."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_desc[:num_clouds]
    return all_gather_rows(local_desc)[:num_clouds]


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    """Scalar max over ranks (bench timing contract)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if dist.get_backend() != "gloo" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
