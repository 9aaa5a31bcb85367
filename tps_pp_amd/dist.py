"""Batch sharding across the GPUs of one node (one process per GPU) and the one collective the
inference path has: the all-gather of per-image recognition outputs.

Reference: data-parallel only: `MMDistributedDataParallel` for training
(`mmocr/apis/train.py:59-67`) and `multi_gpu_test`'s result gather for evaluation
(`tools/test.py:202-207`).  Every image is independent (no BatchNorm in TPS_PP; backbone BN in eval
mode), so rank r simply owns the contiguous slice [lo, hi) of the batch; the rectification path
itself needs NO collective.  `all_gather_rows` is what carries the decoder logits
(N_local, 40, 93) fp32 to every rank: RCCL ("nccl" backend on ROCm) over xGMI for GPU tensors, gloo
for the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world_size: int):
    """Contiguous, balanced slice [lo, hi) of `n_items` owned by `rank` (first `n % world` ranks get
    one extra item)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(t: torch.Tensor, rank: int, world_size: int):
    lo, hi = shard_bounds(t.shape[0], rank, world_size)
    return t[lo:hi]


def _fused_gather(local: torch.Tensor, group) -> bool:
    """True when this tensor's collective runs on RCCL: only then is the fused `all_gather_into_tensor` taken -- some gloo
    builds lack it, also for CUDA tensors on a gloo-only group.  The answer depends only on the tensor's device type and the
    group's (possibly per-device) backend string, both the same on every rank, so every rank takes the same branch; a failing
    backend query PROPAGATES (a rank that silently fell back to the list-based all_gather while its peers issue
    all_gather_into_tensor would desynchronise the communicator)."""
    if not local.is_cuda:
        return False
    backend = str(dist.get_backend(group)).lower()
    # "nccl" (RCCL registers under that name on ROCm; "rccl" accepted as well), or a per-device map such as
    # "cpu:gloo,cuda:nccl"
    for part in backend.split(","):
        dev, _, name = part.rpartition(":")
        if name in ("nccl", "rccl") and dev in ("", "cuda"):
            return True
    return False


def all_gather_rows(local: torch.Tensor, n_total: int, group=None):
    """Gather the per-rank row blocks produced by `shard_batch` back into the (n_total, ...) tensor,
    on every rank.  Ragged shards are padded to the largest shard for the collective (one
    `all_gather_into_tensor`: a single large message per link instead of world_size small ones) and
    trimmed afterwards."""
    if not dist.is_available() or not dist.is_initialized():
        if local.shape[0] != n_total:
            raise RuntimeError("all_gather_rows: no process group and local != total")
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(n_total, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}: local has {local.shape[0]} rows, shard is {hi - lo}")
    cap = (n_total + world - 1) // world
    tail = tuple(local.shape[1:])
    send = local.contiguous()
    if send.shape[0] != cap:
        pad = torch.zeros((cap - send.shape[0],) + tail, dtype=local.dtype, device=local.device)
        send = torch.cat([send, pad], dim=0)
    # the code path is chosen up front from where the tensor lives (a group may carry per-device backends,
    # "cpu:gloo,cuda:nccl": the backend string alone does not say which one this tensor gets): a collective that fails
    # on one rank (comm abort, timeout, size mismatch) must propagate, not be followed by a different collective on a
    # desynchronised communicator
    if not _fused_gather(local, group):                  # gloo (CPU tests, or a CUDA tensor on a gloo-only group)
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(parts, send, group=group)
        recv = torch.cat(parts, dim=0)
    else:                                                # RCCL: one large message per link
        recv = torch.empty((world * cap,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(recv, send, group=group)
    if n_total == world * cap:
        return recv
    pieces = []
    for r in range(world):
        rlo, rhi = shard_bounds(n_total, r, world)
        pieces.append(recv[r * cap: r * cap + (rhi - rlo)])
    return torch.cat(pieces, dim=0)


def recognize_sharded(decode_local, n_total: int, convertor, group=None):
    """Multi-GPU evaluation step (the role of `multi_gpu_test`'s result gather, tools/test.py:202-207):
    rank r decodes its contiguous shard, the per-step scores (n_local, max_seq_len, num_classes-1) are
    all-gathered (the only collective of the inference path), and every rank converts
    the full tensor to strings.  `decode_local(lo, hi)` returns the local scores tensor."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    lo, hi = shard_bounds(n_total, rank, world)
    local = decode_local(lo, hi)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}: decode_local returned {local.shape[0]} rows for a shard of {hi - lo}")
    scores = all_gather_rows(local, n_total, group)
    # this rank's decoder status word (NRTRDecoder attaches it to its output) rides along to the one device->host copy
    # `tensor2idx` makes: a barrier timeout of the persistent decode raises here, on the rank it happened on, instead of
    # NaN scores being decoded into strings (a raising rank fails the job)
    status = getattr(local, "_tpspp_status", None)
    if status is not None and scores is not local:
        scores._tpspp_status = status
    if hasattr(convertor, "tensor2str"):
        texts, char_scores = convertor.tensor2str(scores)
    else:
        indexes, char_scores = convertor.tensor2idx(scores)
        texts = convertor.idx2str(indexes)
    return [dict(text=t, score=s) for t, s in zip(texts, char_scores)]
