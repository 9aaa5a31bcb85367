"""N > 1 path on CPU: world_size-2 gloo process group (the GPU path uses the same code over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tps_pp_amd import dist as tdist


def test_shard_bounds_cover_the_batch_exactly():
    for n in (0, 1, 2, 7, 512, 513, 2048):
        for world in (1, 2, 3, 4, 8):
            spans = [tdist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        tdist.shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(n_total * 40 * 93, dtype=torch.float32).reshape(n_total, 40, 93)
        local = tdist.shard_batch(full, rank, world).clone()       # "logits" of this rank's images
        got = tdist.all_gather_rows(local, n_total)
        ok = torch.equal(got, full)
        # per-rank work is independent: a rank-local transform commutes with the gather
        got2 = tdist.all_gather_rows(local * 2 + rank * 0, n_total)
        ok = ok and torch.equal(got2, full * 2)
        q.put((rank, bool(ok), tuple(got.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7, 1])
def test_all_gather_rows_world2_gloo(n_total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    assert all(r[2] == (n_total, 40, 93) for r in res)


def test_all_gather_rows_without_process_group_is_identity():
    t = torch.randn(5, 3)
    assert tdist.all_gather_rows(t, 5) is t
    with pytest.raises(RuntimeError):
        tdist.all_gather_rows(t, 6)


def _recognize_worker(rank, world, port, q):
    """Sharded evaluation step on the reference's golden decoder scores: every rank ends up with the
    strings of the WHOLE batch, equal to the unsharded conversion."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import cases
    from tps_pp_amd import AttnConvertor
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        G = cases.load("nrtr_head_full")
        scores = torch.from_numpy(G["out_dec"]).repeat(3, 1, 1)[:5]          # 5 images: ragged shards
        conv = AttnConvertor()
        res = tdist.recognize_sharded(lambda lo, hi: scores[lo:hi].clone(), scores.shape[0], conv)
        want = [str(G["text"][i % 2]) for i in range(5)]
        q.put((rank, [r["text"] for r in res] == want, len(res)))
    finally:
        dist.destroy_process_group()


def test_recognize_sharded_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_recognize_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res) and all(r[2] == 5 for r in res), res
