"""CPU, world_size 2, gloo: the N>1 path of the engine -- prompt-group sharding from the shared seeded order, the packed
metrics all-reduce (mean with NaN handling / min / max / sum) and the bucketed gradient mean all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from video_r2_amd import dist as vdist
    from video_r2_amd.host.sampler import RepeatSampler, unique_prompts
    vdist.init(backend="gloo")
    assert vdist.rank() == rank and vdist.world_size() == world
    # every rank draws the SAME permutation (seeded), then takes its contiguous share of each step's prompt groups
    G, groups_per_step = 4, 4
    sampler = RepeatSampler(range(23), mini_repeat_count=G, batch_size=groups_per_step, repeat_count=1, seed=42)
    order = list(unique_prompts(iter(sampler), G))
    mine = [vdist.shard_prompts(order[s:s + groups_per_step]) for s in range(0, len(order) - groups_per_step + 1, groups_per_step)]
    # metrics
    red = vdist.MetricsReducer("cpu").reduce(
        {"loss": 1.0 + rank, "kl": float("nan") if rank == 0 else 3.0, "len/min": 5 + rank, "len/max": 5 + rank, "tokens": 10 * (rank + 1)},
        {"len/min": "min", "len/max": "max", "tokens": "sum"})
    # gradient buckets: flat buffer of 1000 elements, tiny buckets to force several collectives
    flat = torch.full((1000,), float(rank + 1))
    vdist.allreduce_mean_(flat, bucket_bytes=4 * 96)
    q.put((rank, order, mine, red, flat[:3].tolist(), float(flat.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_sharding_metrics_and_grad_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, order0, mine0, red0, f0, s0), (r1, order1, mine1, red1, f1, s1) = res
    assert order0 == order1 and len(order0) == 20          # 23 prompts, chunks of 4: tail of 3 dropped
    for a, b, s in zip(mine0, mine1, range(0, 20, 4)):
        assert a == order0[s:s + 2] and b == order0[s + 2:s + 4]          # disjoint, contiguous, complete
    assert red0 == red1
    assert red0["loss"] == pytest.approx(1.5) and red0["kl"] == pytest.approx(3.0)    # NaN ignored like nanmean
    assert red0["len/min"] == 5 and red0["len/max"] == 6 and red0["tokens"] == 30
    assert f0 == f1 == [1.5, 1.5, 1.5] and s0 == s1 == pytest.approx(1500.0)


def test_single_process_helpers_are_noops():
    from video_r2_amd import dist as vdist
    assert vdist.world_size() == 1 and vdist.rank() == 0
    assert vdist.shard_prompts([3, 1, 2]) == [3, 1, 2]
    t = torch.ones(5)
    assert vdist.allreduce_mean_(t) is t
    assert vdist.MetricsReducer().reduce({"a": 2.0, "b": float("nan")})["a"] == 2.0


def _bf16_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from video_r2_amd import dist as vdist
    vdist.init(backend="gloo")
    # what train.Trainer._exchange does with grad_exchange="bf16": round the fp32 bucket to bf16, SUM all-reduce the bf16 copy,
    # widen it back -- every rank must end up with the SAME values (replicas stay bit-identical) and they are the bf16-rounded sum
    g = torch.Generator().manual_seed(100 + rank)
    arena = torch.randn(4096, generator=g)
    wire = arena.to(torch.bfloat16)
    work = dist.all_reduce(wire, op=dist.ReduceOp.SUM, async_op=True)
    watch = vdist.ExchangeWatch(timeout=60)
    watch.add("layer 3", work, wire.numel() * 2)
    work.wait()
    arena.copy_(wire)
    q.put((rank, arena.tolist()))           # (plain floats: a tensor in the queue is a shared-memory handle of a process that exits)
    dist.barrier()
    watch.close()
    dist.destroy_process_group()


def test_world2_bf16_gradient_exchange_gives_identical_replicas():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_bf16_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = {r: torch.tensor(v) for r, v in (q.get(timeout=120) for _ in range(world))}
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(res[0], res[1])
    want = sum(torch.randn(4096, generator=torch.Generator().manual_seed(100 + r)).to(torch.bfloat16).float() for r in range(2))
    assert (res[0] - want).abs().max().item() <= 2 ** -7 * want.abs().max().item()       # one bf16 rounding of the sum


def _stuck_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), VR2_COLLECTIVE_TIMEOUT="3")
    import time
    from video_r2_amd import dist as vdist
    vdist.init(backend="gloo")
    dist.barrier()
    if rank == 1:
        time.sleep(30)                     # the peer never issues its half of the exchange (stuck in a reward call, say)
        os._exit(0)
    buf = torch.ones(1 << 16)
    work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
    vdist.ExchangeWatch().add("layer 12", work, buf.numel() * 4)
    time.sleep(25)                         # the step goes on enqueueing work; the watch ends the process long before this returns
    os._exit(0)


def test_a_stuck_gradient_bucket_ends_the_rank_with_its_name(capfd):
    """a peer that never joins a bucket's all-reduce: the waiting rank exits non-zero (ExchangeWatch.EXIT_CODE) within the
    configured timeout, and says which bucket -- it is not left hanging in its next host sync"""
    import time
    from video_r2_amd import dist as vdist
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=_stuck_worker, args=(r, world, port)) for r in range(world)]
    t0 = time.monotonic()
    for p in ps:
        p.start()
    ps[0].join(timeout=20)
    took = time.monotonic() - t0
    assert ps[0].exitcode == vdist.ExchangeWatch.EXIT_CODE, ps[0].exitcode
    assert took < 20
    ps[1].kill()
    ps[1].join(timeout=10)
    err = capfd.readouterr().err
    assert "gradient bucket 'layer 12'" in err and "rank 0 of 2" in err
