"""N>1 path on CPU: world_size-2 gloo run of the benchmark's sharding/timing helpers
(one process per GPU, independent streams, barrier + MAX-over-ranks timing, no data-path collective)."""
import os
import socket
import sys

import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import time
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gst-plugins-rs_amd"))
    from mi355fx import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.streams_for_rank(7, rank, world)
    # rank 1 is deliberately slower: the reported time must be the max over ranks
    dt = sharding.timed_region(lambda: time.sleep(0.05 + 0.15 * rank), dist=dist)
    q.put((rank, mine, dt))
    dist.destroy_process_group()


def test_two_rank_gloo_timing_and_stream_partition():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, t0), (r1, s1, t1) = res
    assert sorted(s0 + s1) == list(range(7)) and not set(s0) & set(s1)   # disjoint cover of the streams
    assert s0 == [0, 2, 4, 6] and s1 == [1, 3, 5]
    assert abs(t0 - t1) < 1e-9 and t0 >= 0.2                               # both ranks see the MAX


def test_aggregate_throughput_is_whole_job():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gst-plugins-rs_amd"))
    from mi355fx import sharding
    assert sharding.aggregate_throughput(400, 8, 2.0) == 1600.0
    with pytest.raises(ValueError):
        sharding.streams_for_rank(4, 3, 2)
