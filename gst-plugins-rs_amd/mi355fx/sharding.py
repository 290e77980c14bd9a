"""Multi-GPU sharding of the hot path: independent streams, one process per GPU, no data-path collective.

Every element instance owns its state (settings, LUT, echo ring — e.g. video/hsv/src/hsvfilter/imp.rs:54-57,
audio/audiofx/src/audioecho/imp.rs:57-66) and streams never exchange data, so the
partition is stream -> GPU and the only cross-rank traffic is the timing reduction of the benchmark
(barrier + MAX over ranks). `torch.distributed` is used for exactly that and nothing else, over a CPU ("gloo") group on
the GPU box as well as in the CPU tests: neither the data path nor the harness opens an RCCL communicator (north_star:
"no RCCL: there is no cross-stream collective").
"""
import time


def streams_for_rank(n_streams, rank, world):
    """Round-robin stream -> rank map (`gpu = stream_index mod n_gpus`)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_streams, world))


def timed_region(run, dist=None, device_sync=None, reduce_device=None, keep_busy=None):
    """barrier + sync, run(), sync + barrier; returns the MAX elapsed seconds over all ranks.
    keep_busy: optional callable that ENQUEUES (does not wait for) a few milliseconds of untimed device work; it is called
    right before the opening barrier so that the device does not sit idle - and drop its clocks - while the ranks meet (a
    host-side barrier over 8 processes takes about a millisecond); the sync that follows the barrier drains it."""
    def sync():
        if device_sync is not None:
            device_sync()

    sync()
    if keep_busy is not None:
        keep_busy()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    run()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device=reduce_device if reduce_device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def aggregate_throughput(units_per_rank, world, seconds):
    """Whole-job rate: units processed by all ranks / max-over-ranks time (weak scaling)."""
    return units_per_rank * world / seconds
