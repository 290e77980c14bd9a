#!/usr/bin/env python3
"""bench.py — headline benchmark: 3840x2160 RGBA frames/s through hsvfilter -> colorlut (33^3, trilinear)
on MI355X, with the HBM roofline fraction of the dominant kernel and the CPU oracle timed beside it.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched by
torch.distributed.run with one rank per GPU. One JSON line on rank 0.

A "step" = one pass of the hot path over one batch of `--batch` device-resident 4K RGBA frames:
one hsvfilter launch (in place, hue-shift=90) + one colorlut launch (33^3 LUT) — the two-kernel mode
of SURVEY.md §8d (algorithmic 16 B/pixel/frame = 132,710,400 B per frame). hsvfilter is AlwaysInPlace in the
reference (video/hsv/src/hsvfilter/imp.rs:315-320): upstream hands it a fresh frame every time. The bench does the
same: every warm-up and timed step filters its OWN never-touched pristine batch (steps + warmup distinct source
batches, far larger than the 256 MiB Infinity Cache, so every step streams from HBM); only the untimed clock ramp
re-filters scratch batches.
Streams are independent: ranks share nothing (no collective on the data path), scaling = "weak".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))

W, H = 3840, 2160
FRAME_BYTES = W * H * 4
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_FRAME_PER_KERNEL = 2 * FRAME_BYTES  # 4 B read + 4 B written per pixel


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=75)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--pairs-per-step", type=int, default=4,
                    help="(hsvfilter, colorlut) launch pairs per step, each on its own pristine batch of --batch frames: a step is\n"
                         "pairs x batch = 32 frames. One pair is 0.19 ms; the driver times 20 steps, and a 3.8 ms bracket spends\n"
                         "5-6 %% of itself starting and draining the queue (measured: 39.5 k frames/s at 20 steps of one pair against\n"
                         "42.1 k at 300) - 20 steps of four pairs are 15 ms")
    ap.add_argument("--rewarm-steps", type=int, default=150,
                    help="untimed steps on scratch batches right before every timed bracket (the pristine batches' fills and statistics leave the GPU at low clocks)")
    ap.add_argument("--ramp-seconds", type=float, default=0.25,
                    help="untimed preamble before the W warm-up steps of each measured leg: the same launches for this long,\n"
                         "so that the GPU has left its idle clocks (a cold 50-step run measures 35 k frames/s, a warm one 43 k)")
    ap.add_argument("--batch", type=int, default=8,
                    help="4K frames per launch (one step = one batch). 8 frames = 265 MB: the batch hsvfilter has just written\n"
                         "in place is still in the 256 MiB Infinity Cache when the colorlut launch reads it (32.0 k frames/s; 32 frames\n"
                         "per launch: 28-29 k; 1 frame per launch: 24.6 k)")
    ap.add_argument("--ring", type=int, default=4, help="destination batches cycled through, and scratch source batches of the untimed ramp\n"
                                                         "(the timed and warm-up steps have one pristine source batch EACH)")
    ap.add_argument("--max-source-gib", type=float, default=96.0,
                    help="cap on the pristine source batches held at once; a run needing more is timed in chunks with the sources\n"
                         "regenerated between chunks outside the timed region (reported as config.source_chunks)")
    ap.add_argument("--content", default="smooth", choices=["smooth", "noise"], help="headline frame content")
    ap.add_argument("--lut-variant", type=int, default=0,
                    help="MI355_FLAG_LUT_VARIANT: 0 auto (default), 6 interpolating kernel only, 5 table kernel only")
    ap.add_argument("--hsv-blocks-per-cu", type=int, default=0, help="MI355_FLAG_HSV_BLOCKS_PER_CU (tuning; 0 = library default)")
    ap.add_argument("--ctx-flag", action="append", default=[], metavar="FLAG=VALUE",
                    help="mi355_ctx_set_flag on the main context (numeric MI355_FLAG_* id = value); A/B knob, repeatable")
    ap.add_argument("--streams", type=int, default=32, help="concurrent streams of the secondary many-streams measurement (0 / 1 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary (other-content) measurement")
    ap.add_argument("--config", type=int, default=0,
                    help="0 = the headline chain (default). 5 = BASELINE config 5: `--streams` concurrent 4K streams per GPU through the\n"
                         "videocompare SSIM (dssim) engine, one comparison per stream and step; prints its own JSON line (comparisons/s)")
    ap.add_argument("--workers", type=int, default=8, help="config 5: host threads / contexts per GPU that serve the streams in turn")
    ap.add_argument("--shared-reference", action="store_true",
                    help="config 5: the streams are the non-reference pads of --workers videocompare elements (reference frame hashed once per aggregate)")
    ap.add_argument("--group", action="store_true",
                    help="config 5: every stream's pair through the process's dispatcher (mi355_group_submit_compare / _wait_compare, one launch\n"
                         "sequence per interval; --workers host threads stand in for the element threads) instead of --workers contexts taking the\n"
                         "streams in turn")
    ap.add_argument("--no-group", action="store_true", help="config 5: accepted, the default (see --group)")
    ap.add_argument("--hash-algo", default="dssim", choices=["dssim", "blockhash"],
                    help="config 5: videocompare's hash-algorithm: dssim (BASELINE's 'SSIM', default) or blockhash (the element's own default)")
    ap.add_argument("--rendezvous", type=int, default=0, help="config 5: pairs that make the dispatcher launch (0 = all streams of the rank: one launch sequence per interval)")
    ap.add_argument("--compare-lanes", type=int, default=0, help="config 5: HIP streams of the dispatcher's compare queue (0 = library default, 8)")
    ap.add_argument("--dssim-clock-ghz", type=float, default=2.4,
                    help="config 5: shader clock of the VALU roofline's peak (default: the part's 2.4 GHz peak engine clock; under this load the\n"
                         "SQ counters show less - SQ_BUSY_CYCLES / 32 SEs / kernel duration, profiles/r06_dssim_sq_counters.txt - so the fraction is conservative)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 --pmc child runs at the end (then it is quoted from\n"
                         "profiles/pmc_latest.json when that profile carries this code's fingerprint, else null)")
    ap.add_argument("--dssim-two-step", action="store_true",
                    help="config 5: hash every frame with mi355_dssim_create_image and compare the two images (the round-2 form) instead of\n"
                         "hashing + comparing the non-reference frames in one pass (mi355_dssim_compare_frames)")
    ap.add_argument("--stub", action="store_true",
                    help="TEST SCAFFOLDING, no GPU: tiny CPU frames and a fake context that sleeps instead of launching kernels, so that\n"
                         "the N>1 control flow (gloo group, barriers, MAX over ranks, aggregation, rank-0 JSON) can be exercised by the CPU\n"
                         "test suite (tests/test_bench_multirank.py); the line it prints says \"data\": \"stub\" and means nothing")
    return ap.parse_args()


class SourcePool:
    """Pristine source batches. Batch number `index` is a pure function of (content, index): a seeded base frame
    (synth.smooth_frame / noise_frame, one of N_BASE seeds) and whole-pixel rotations along the row, different for every
    frame of every batch, so that all frames are distinct while keeping the content statistics of the base frame."""
    N_BASE = 4

    def __init__(self, torch, synth, dev, batch, content):
        self.torch, self.batch = torch, batch

        def gen(w, h, seed):
            if content == "noise":
                return synth.noise_frame(w, h, seed=seed)
            f = synth.smooth_frame(w, h, seed=seed)
            if content.startswith("smooth+"):   # "smooth+4": the natural-like frame plus uniform noise of +-4 per colour channel
                amp = int(content.split("+")[1])
                rng = np.random.default_rng(seed + 1000)
                px = f.reshape(h, w, 4).astype(np.int16)
                px[..., :3] += rng.integers(-amp, amp + 1, size=(h, w, 3), dtype=np.int16)
                f = np.clip(px, 0, 255).astype(np.uint8).reshape(h, w * 4)
            return f
        self.bases = [torch.from_numpy(gen(W, H, seed=synth.SEED + 17 * r)).to(dev) for r in range(self.N_BASE)]  # (H, W*4) u8

    def fill(self, buf, index):
        base, k = self.bases[index % self.N_BASE], index // self.N_BASE
        for i in range(self.batch):
            buf[i].copy_(self.torch.roll(base, shifts=4 * ((97 * i + 389 * k) % W), dims=1))
        return buf

    def new(self, index):
        return self.fill(self.torch.empty((self.batch, H, W * 4), dtype=self.torch.uint8, device=self.bases[0].device), index)


def batch_stats(torch, b):
    """Byte sum of a batch and the number of distinct colours of its first frame (content fingerprint)."""
    rgb = b[0].reshape(-1).view(torch.int32) & 0x00FFFFFF
    return {"byte_sum": int(b.sum(dtype=torch.int64).item()), "distinct_colours_frame0": int(torch.unique(rgb).numel())}


EVENT_EVERY = 4  # steps between per-kernel event samples inside the timed region
HOST_TRACE = [] if os.environ.get("BENCH_HOST_TRACE") else None


class EventPool:
    """Timestamped events, created AND recorded once before any timed region. A process's first ~170 recorded timing events
    end in a host-side stall of 35-55 ms (the runtime grows a pool behind them; torch creates its events lazily at the
    first record): with the events of a 75-step region created inside it, that stall sat at launch pair 224 of 300, where the host
    is only 39 ms of device work ahead - 3 % of the region on a quiet host and 20-50 % on a loaded one (value 20-37 k instead
    of 43-46 k on the same box in the same minute; tools/trace_big_gaps.py, BENCH_HOST_TRACE=1)."""

    def __init__(self, torch, n):
        self.torch = torch
        self.events = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        for e in self.events:      # all outstanding at once: whatever pool the runtime keeps behind them grows to this size now
            e.record()
        torch.cuda.synchronize()
        self.next = 0

    def take(self):
        if not self.events:
            return self.torch.cuda.Event(enable_timing=True)
        e = self.events[self.next]
        self.next = (self.next + 1) % len(self.events)
        return e


EVENTS = None   # set once the stream exists (main)


def run_region(torch, ctx, srcs, dsts, settings, steps, batch, record, every=EVENT_EVERY, served=None):
    """`steps` steps on ctx's stream (== torch current stream); step k filters srcs[k % len(srcs)] in place and maps it into
    dsts[k % len(dsts)]. With `record`, every `every`-th step has its two launches bracketed by in-stream events (the
    timestamps cost a few microseconds each and keep consecutive kernels from overlapping their tails, so bracketing every
    launch would lower the throughput being measured); returns the event triples of the sampled steps."""
    evs = []
    pitch = FRAME_BYTES
    for k in range(steps):
        if HOST_TRACE is not None:   # BENCH_HOST_TRACE=1: when the host was where (diagnostic; a list append per step)
            HOST_TRACE.append(time.perf_counter())
        s = srcs[k % len(srcs)]
        d = dsts[k % len(dsts)]
        sample = record and (k % every == 0)
        if sample:
            e0, e1, e2 = (EVENTS.take() for _ in range(3))
            e0.record()
        ctx.hsvfilter_frames_device(s.data_ptr(), batch, pitch, W, H, W * 4, "RGBA", settings)
        if sample:
            e1.record()
        ctx.colorlut_frames_device(s.data_ptr(), pitch, W * 4, d.data_ptr(), pitch, W * 4, batch, W, H, "RGBA")
        if served is not None:   # which colorlut kernel the library picked for this launch (a host-side query, no device work)
            name = ctx.colorlut_kernel_name()
            served[name] = served.get(name, 0) + 1
        if sample:
            e2.record()
            evs.append((e0, e1, e2))
    return evs


def cpu_baseline(synth, settings, cube_text, seconds_target=12.0):
    """The CPU oracle ("port" of the reference's scalar loops) on the host cores of this box:
    hsvfilter then colorlut on 4K smooth frames, 1 thread (what one reference pipeline does: both
    elements run on the single upstream streaming thread)."""
    from oracle import oracle as O
    cube = O.Cube.parse(cube_text)
    frame = synth.smooth_frame(W, H)
    out = np.zeros_like(frame)
    n, t0 = 0, time.perf_counter()
    while True:
        buf = frame.copy().reshape(-1)
        O.hsvfilter(buf, W, W * 4, 4, 0, False, settings)
        O.colorlut_rgba8(cube, buf, W * 4, out, W * 4, W, H)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_target or n >= 64:
            break
    one = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d 4K smooth frames, hsvfilter(hue-shift=90)+colorlut(33^3), oracle C -O3 -ffp-contract=off, 1 thread" % n}
    # all host cores: one independent stream (frame) per core, each single-threaded — the reference's
    # only scaling axis is more pipelines. Capped so the sample stays within a few seconds and GiB.
    nc = min(os.cpu_count() or 1, 64)
    frames = np.stack([frame] * nc).copy()
    outs = np.zeros_like(frames)
    n2, t0 = 0, time.perf_counter()
    while True:
        O.chain_streams(cube, frames, outs, nc, W, H, settings, nc)
        n2 += nc
        dt2 = time.perf_counter() - t0
        if dt2 >= seconds_target / 2 or n2 >= 1024:
            break
    return one, {"value": n2 / dt2, "unit": "frames/s", "cores": nc, "kind": "port",
                 "sample": "%d 4K smooth frames as %d concurrent single-threaded streams" % (n2, nc)}


def _install_stub(torch, mi355fx, rank):
    """--stub: replace the device with the CPU and the context with a sleeper (test scaffolding for the N>1 control flow)."""
    global W, H, FRAME_BYTES, BYTES_PER_FRAME_PER_KERNEL
    W, H = 64, 16
    FRAME_BYTES = W * H * 4
    BYTES_PER_FRAME_PER_KERNEL = 2 * FRAME_BYTES
    import contextlib

    class Ev:
        def __init__(self, enable_timing=True):
            self.t = 0.0

        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3 + 0.02

    class St:
        def __init__(self, device=None):
            self.cuda_stream = 0

    torch.cuda.Event = Ev
    torch.cuda.Stream = St
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.mem_get_info = lambda *a, **k: (1 << 30, 1 << 30)
    torch.cuda.empty_cache = lambda: None

    class Ctx:
        def __init__(self, device):
            self.lat = 0.001 * (1 + rank)  # rank 1 is slower: the reported time must be the MAX over ranks

        def set_stream(self, s): pass
        def colorlut_load(self, *a): pass
        def set_flag(self, *a): pass
        def hsvfilter_frames_device(self, *a): time.sleep(self.lat)
        def colorlut_frames_device(self, *a): time.sleep(self.lat)
        def hsv_colorlut_frames_device(self, *a): time.sleep(self.lat)
        def colorlut_kernel_choice(self, fused=False): return (False, 0.0, 0.0)
        def colorlut_kernel_name(self): return "stub"
        def synchronize(self): pass
        def close(self): pass

    mi355fx.Context = Ctx


def live_pmc_traffic(args, kernel_prefix):
    """HBM bytes per launch of the dominant kernel, measured on THIS box: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE - they
    do not fit one pass) over a short run of this file's main leg, as child processes after the timed work is done. gfx950
    corrections as in tools/summarize_profiles.py (MI355X_MICROARCH.md, HBM section): the counters are in KiB and FETCH_SIZE
    reports half the bytes of wide coalesced streaming reads. A third child run (--kernel-trace --stats) gives the kernel's average
    duration as the profiler sees it. Returns (bytes, rocprof_avg_ms, provenance) or (None, rocprof_avg_ms or None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not rocprof:
        return None, None, "rocprofv3 not found"
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ):
        return None, None, "this run is itself being profiled"
    out = tempfile.mkdtemp(prefix="mi355fx_pmc_", dir="/tmp")
    # THIS interpreter, resolved to the binary: a `python3` from PATH may be a wrapper script (pyenv, conda) = one more exec
    # under the profiler's preloaded, GPU-initialised library, which this pool forbids
    child = [os.path.realpath(sys.executable), os.path.abspath(__file__), "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-extra", "--no-live-pmc",
             "--batch", str(args.batch), "--content", args.content, "--lut-variant", str(args.lut_variant), "--pairs-per-step", str(args.pairs_per_step)]
    for f in args.ctx_flag:
        child += ["--ctx-flag", f]
    if args.hsv_blocks_per_cu:
        child += ["--hsv-blocks-per-cu", str(args.hsv_blocks_per_cu)]
    avg = {}
    rocprof_ms = None

    def run_child(cmd):
        """Own session, so that a timeout ends rocprofv3 AND the GPU process under it (subprocess.run would only kill rocprofv3)."""
        import signal
        p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            p.wait(timeout=240)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)   # exactly the group this call started
            except OSError:
                pass
            p.wait()
            raise
        return p

    try:
        # the kernel's average duration as rocprofv3 --kernel-trace --stats sees it on this box (must agree with avg_launch_ms)
        d = os.path.join(out, "trace")
        r = run_child([rocprof, "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "t", "--"] + child)
        if r.returncode == 0:
            for path in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
                for row in csv.DictReader(open(path)):
                    if "mi355::" in row["Name"] and row["Name"].split("mi355::")[1].startswith(kernel_prefix) and int(row["Calls"]) >= 20:
                        rocprof_ms = float(row["AverageNs"]) * 1e-6
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, counter)
            r = run_child([rocprof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--"] + child)
            if r.returncode != 0:
                return None, rocprof_ms, "rocprofv3 --pmc %s pass failed (rc %d)" % (counter, r.returncode)
            vals = []
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(path)):
                    if row.get("Counter_Name") == counter and "mi355::" in row["Kernel_Name"] and \
                            row["Kernel_Name"].split("mi355::")[1].startswith(kernel_prefix):
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, rocprof_ms, "no %s rows for %s" % (counter, kernel_prefix)
            avg[counter] = (sum(vals) / len(vals), len(vals))
    except (OSError, subprocess.SubprocessError, ValueError, KeyError) as e:
        return None, rocprof_ms, "live PMC pass failed: %r" % (e,)
    finally:
        shutil.rmtree(out, ignore_errors=True)
    traffic = avg["FETCH_SIZE"][0] * 1024 * 2 + avg["WRITE_SIZE"][0] * 1024
    return traffic, rocprof_ms, ("live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (child processes of this run, after the timed legs) over "
                     "`bench.py --steps 10 --warmup 2 --no-extra`: %d / %d launches averaged; FETCH_SIZE(KiB) * 1024 * 2 + WRITE_SIZE(KiB) * 1024"
                     % (avg["FETCH_SIZE"][1], avg["WRITE_SIZE"][1]))


def source_fingerprint():
    """sha256 over the kernel sources, the C ABI header and this file: ties a committed PMC profile to the code it was taken from
    (the GPU boxes get a snapshot without .git, so a commit id cannot be read there)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "gst-plugins-rs_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "gst-plugins-rs_amd", "csrc", "*.hpp")))
    for f in files + [os.path.join(ROOT, "include", "mi355fx.h"), os.path.abspath(__file__)]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


# Measured VALU cost of the Dssim kernels (profiles/r06_dssim_sq_counters.txt, 4K scale 0 launches, SQ_INSTS_VALU x 64 lanes / pixels):
# dssim_scale_fused_kernel (the reference frame's hash) 522 and dssim_hash_compare_kernel (the other frame hashed + compared) 772
# VALU instructions per pixel of the scale; the five scales together are 1 + 1/4 + ... = 1.332 scale-0 images. SQ_ACTIVE_INST_VALU x 4
# / SQ_INSTS_VALU = 4.07 cycles per wave instruction (exact f32: unfused mul / add, IEEE division sequences, v_pk_* pairs).
DSSIM_VALU_PER_PIXEL_PAIR = 522.0 + 772.0
DSSIM_SCALE_SUM = 1.0 + 0.25 + 0.0625 + 0.015625 + 0.00390625
DSSIM_CYCLES_PER_VALU = 4.07
N_SIMD = 1024          # 256 CUs x 4 SIMDs


def _valu_roofline(comps_per_gpu, clock_ghz):
    """The Dssim pair against the VALU issue roofline: wave instructions per second the comparisons need / what 1024 SIMDs issue."""
    wave_instr = DSSIM_VALU_PER_PIXEL_PAIR * DSSIM_SCALE_SUM * W * H / 64.0
    achieved = comps_per_gpu * wave_instr / 1e9                 # G wave-instructions / s
    peak = N_SIMD * clock_ghz / DSSIM_CYCLES_PER_VALU           # G wave-instructions / s
    return {"bound": "valu", "kernel": "dssim_scale_fused_kernel + dssim_hash_compare_kernel", "instr_per_pixel": DSSIM_VALU_PER_PIXEL_PAIR,
            "wave_instr_per_comparison": wave_instr, "cycles_per_instr": DSSIM_CYCLES_PER_VALU, "clock_ghz": clock_ghz,
            "achieved": achieved, "peak": peak, "unit": "G wave-instructions/s", "frac": achieved / peak,
            "source": "SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU per launch, profiles/r06_dssim_sq_counters.txt; clock = SQ_BUSY_CYCLES / 32 SEs / kernel duration of the same pass"}


def _install_stub_config5(mi355fx, rank):
    """--stub for config 5: contexts and a dispatcher that sleep instead of launching (CPU test of the N > 1 control flow)."""
    global W, H, FRAME_BYTES
    W, H = 64, 16
    FRAME_BYTES = W * H * 4
    import itertools
    import threading

    class Ctx:
        def __init__(self, device):
            self.h = object()

        def alloc(self, n): return 1
        def h2d(self, *a): pass
        def free(self, *a): pass
        def close(self): pass
        def videocompare_hash_frames_device(self, *a): time.sleep(0.0005 * (1 + rank)); return [0]
        def videocompare_distance(self, *a): return 0.0
        def dssim_create_image_device(self, *a): return 1
        def dssim_free_image(self, *a): pass
        def dssim_compare(self, *a): time.sleep(0.001 * (1 + rank)); return 0.25
        def dssim_compare_frames_device(self, x, frames, *a): time.sleep(0.001 * (1 + rank) * len(frames)); return [0.25] * len(frames)

    class Grp:
        def __init__(self, device=0, max_batch=0):
            self.tk, self.lock, self.n = itertools.count(1), threading.Lock(), 0

        def set_rendezvous(self, *a): pass
        def set_compare_lanes(self, *a): pass
        def submit_compare(self, *a): return next(self.tk)

        def wait_compare(self, t):
            time.sleep(0.001 * (1 + rank))   # rank 1 is slower: the reported time must be the MAX over ranks
            with self.lock:
                self.n += 1
            return 0.25, 0, 0

        def compare_stats(self): return (self.n, self.n, 1)
        def close(self): pass

    mi355fx.Context, mi355fx.Group = Ctx, Grp


def run_config5(args, rank, local_rank, world):
    """BASELINE config 5: 256 concurrent 4K streams through videocompare's SSIM engine = 32 streams per GPU. Every stream is a
    pair of resident 4K RGBA frames (the natural-like frame and the same frame + noise of sigma 2, SURVEY.md 8d synthetic (8));
    a step = one comparison per stream the way the element does it (hash_image of the reference frame; the other frame hashed and compared,
    video/videofx/src/videocompare/hashed_image.rs:24-79). The streams are independent and sharded over the ranks.
    Default (round 6): every stream is its own two-pad element on its own host thread with its own context, submitting its pair to
    the process's dispatcher and waiting for its score (mi355_group_submit_compare / _wait_compare: the interval's pairs are ONE
    launch sequence). --no-group: the round-5 form, `--workers` host threads per rank, each with its own context and HIP stream,
    taking the rank's streams in turn. Timing: gloo barrier + MAX over ranks."""
    import threading
    import mi355fx
    from mi355fx import sharding, synth
    stub = bool(getattr(args, "stub", False))
    if stub:
        _install_stub_config5(mi355fx, rank)
        device_sync = lambda: None
    else:
        import torch
        torch.cuda.set_device(local_rank)
        device_sync = torch.cuda.synchronize
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("gloo", rank=rank, world_size=world)
        dist = dist_mod
    shared_ref = bool(getattr(args, "shared_reference", False))
    two_step = bool(getattr(args, "dssim_two_step", False))   # round-2 form: create_image for every frame, then compare
    use_group = bool(getattr(args, "group", False)) and not shared_ref and not two_step
    n_streams = args.streams
    n_workers = max(1, min(args.workers, args.streams))
    rng = np.random.default_rng(1234 + rank)
    base = synth.smooth_frame(W, H)
    noisy = base.reshape(H, W, 4).astype(np.int16)
    noisy[..., :3] += np.rint(rng.normal(0.0, 2.0, size=(H, W, 3))).astype(np.int16)
    noisy = np.clip(noisy, 0, 255).astype(np.uint8).reshape(H, W * 4)
    base[:, 3::4] = 255
    noisy[:, 3::4] = 255
    ctxs = [mi355fx.Context(local_rank) for _ in range(n_workers)]
    frames = []   # per stream: (device reference frame, device secondary frame), all distinct by a row rotation
    for s in range(n_streams):
        c = ctxs[s % n_workers]
        a, b = np.roll(base, 4 * 61 * s, axis=1).copy(), np.roll(noisy, 4 * 61 * s, axis=1).copy()
        da, db = c.alloc(a.nbytes), c.alloc(b.nbytes)
        c.h2d(da, a.reshape(-1)); c.h2d(db, b.reshape(-1))
        frames.append((da, db))
    results = [0.0] * n_streams
    algo_code = 4 if getattr(args, "hash_algo", "dssim") == "blockhash" else 5
    group = None
    if use_group:
        group = mi355fx.Group(local_rank)
        if getattr(args, "compare_lanes", 0):
            group.set_compare_lanes(args.compare_lanes)
        group.set_rendezvous(getattr(args, "rendezvous", 0) or n_streams, 5000)   # every element submits + waits at once; a straggler is waited for 5 ms at most

    def serve(w, steps):
        c = ctxs[w]
        for _ in range(steps):
            if use_group:
                # Two-pad videocompare elements: aggregate() hands its pair to the dispatcher and waits for the score. This host
                # thread stands in for the element threads of ITS streams (an interpreter's threads wake one after the other: 32 of
                # them cost more than the comparisons; native element threads: tools/agroup_bench.cpp): every element's submit,
                # then every element's wait - on the device exactly what those elements' own threads produce
                mine = list(range(w, n_streams, n_workers))
                tk = [group.submit_compare(c, frames[s][0], frames[s][1], W * 4, W, H, "RGBA", algo_code) for s in mine]
                for s, t in zip(mine, tk):
                    results[s] = group.wait_compare(t)[0]
                continue
            if algo_code == 4:
                # hash-algorithm=blockhash without the dispatcher: two hash calls (one synchronisation each) + the distance per stream
                for s in range(w, n_streams, n_workers):
                    da, db = frames[s]
                    h0 = c.videocompare_hash_frames_device(da, FRAME_BYTES, W * 4, 1, W, H)[0]
                    h1 = c.videocompare_hash_frames_device(db, FRAME_BYTES, W * 4, 1, W, H)[0]
                    results[s] = c.videocompare_distance(h0, h1)
                continue
            if shared_ref:
                # one videocompare element per worker: its reference pad's frame is hashed ONCE per aggregate, every other pad's
                # frame is hashed and compared with it (videocompare/imp.rs:316-345) - the worker's streams are those pads
                x = c.dssim_create_image_device(frames[w][0], W * 4, W, H)
                pads = list(range(w, n_streams, n_workers))
                if two_step:
                    for s in pads:
                        y = c.dssim_create_image_device(frames[s][1], W * 4, W, H)
                        results[s] = c.dssim_compare(x, y)
                        c.dssim_free_image(y)
                else:   # the pads' frames are hashed and compared in one pass each, one synchronisation per aggregate
                    for s, v in zip(pads, c.dssim_compare_frames_device(x, [frames[s][1] for s in pads], W * 4, W, H)):
                        results[s] = v
                c.dssim_free_image(x)
                continue
            for s in range(w, n_streams, n_workers):
                da, db = frames[s]
                x = c.dssim_create_image_device(da, W * 4, W, H)
                if two_step:
                    y = c.dssim_create_image_device(db, W * 4, W, H)
                    results[s] = c.dssim_compare(x, y)
                    c.dssim_free_image(y)
                else:
                    results[s] = c.dssim_compare_frames_device(x, [db], W * 4, W, H)[0]
                c.dssim_free_image(x)

    def run(steps):
        ts = [threading.Thread(target=serve, args=(w, steps)) for w in range(n_workers)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    t_end = time.perf_counter() + args.ramp_seconds
    while time.perf_counter() < t_end:
        run(2 if use_group else 1)
    run(max(1, args.warmup))
    dt = sharding.timed_region(lambda: run(args.steps), dist=dist, device_sync=device_sync)
    comps = sharding.aggregate_throughput(args.steps * n_streams, world, dt)
    gstats = group.compare_stats() if group else None
    form = ("one two-pad element per stream, the pairs of an interval through the process's dispatcher as ONE launch sequence "
            "(mi355_group_submit_compare / _wait_compare, rendezvous of %d; %d host threads stand in for the element threads)" % (n_streams, n_workers)) if use_group else (
            "streams = non-reference pads of %d aggregators: reference hashed once per step, every pad hashed + compared" % n_workers if shared_ref else
            "one two-pad element per stream, %d worker contexts taking the streams in turn: hash the reference frame, hash + compare the other" % n_workers)

    def cleanup():
        if group:
            group.close()
        for s_, (da, db) in enumerate(frames):
            ctxs[s_ % n_workers].free(da)
            ctxs[s_ % n_workers].free(db)
        for c in ctxs:
            c.close()

    if getattr(args, "as_leg", False):
        cleanup()
        leg = {"comparisons_per_s": comps, "streams_per_gpu": n_streams, "worker_contexts_per_gpu": n_workers, "steps": args.steps,
               "seconds": dt, "real_time_need": 30 * n_streams, "frac_of_hbm_peak": comps * 2 * FRAME_BYTES / 1e9 / HBM_PEAK_GBS,
               "algorithmic_bytes_per_comparison": 2 * FRAME_BYTES, "dssim_of_stream_0": results[0],
               "what": "BASELINE config 5 on this GPU: videocompare hash-algorithm=dssim, 3840x2160 RGBA; " + form}
        if gstats:
            leg["dispatcher"] = {"pairs": gstats[0], "launch_sequences": gstats[1], "pairs_in_the_largest": gstats[2]}
        if not shared_ref and not two_step:
            leg["roofline"] = _valu_roofline(comps, getattr(args, "dssim_clock_ghz", 2.4))
        return leg
    if rank == 0:
        algo = 2 * FRAME_BYTES  # two 4K RGBA frames read per comparison
        out = {"metric": "videocompare SSIM comparisons/sec, 32 concurrent 4K streams per GPU (BASELINE config 5)", "value": comps, "unit": "comparisons/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "stub" if stub else "synthetic",
               "config": {"workload": "videocompare hash-algorithm=%s, 3840x2160 RGBA, " % getattr(args, "hash_algo", "dssim") + form,
                          "streams_per_gpu": n_streams, "worker_contexts_per_gpu": n_workers, "frames": "natural-like frame vs the same + N(0, 2) noise, resident",
                          "timing_group": "gloo (CPU) barrier + MAX; no RCCL" if world > 1 else "single process",
                          "real_time_need": "%d streams x 30 frames/s = %d comparisons/s per GPU" % (n_streams, 30 * n_streams)},
               "roofline": {"bound": "hbm", "kernel": "blockhash_rgba_kernel", "achieved": comps / world * algo / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": comps / world * algo / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_comparison": algo} if algo_code == 4 else
                           dict(_valu_roofline(comps / world, getattr(args, "dssim_clock_ghz", 2.4)), traffic=None,
                                hbm={"achieved_GBps": comps / world * algo / 1e9, "peak_GBps": HBM_PEAK_GBS, "frac": comps / world * algo / 1e9 / HBM_PEAK_GBS,
                                     "algorithmic_bytes_per_comparison": algo,
                                     "note": "the HBM fraction is reported for completeness: the engine is VALU-bound (exact f32, ~1,300 instructions per pixel pair)"}),
               "dssim_of_stream_0": results[0]}
        if gstats:
            out["dispatcher"] = {"pairs": gstats[0], "launch_sequences": gstats[1], "pairs_in_the_largest": gstats[2]}
        if not args.no_cpu_baseline and not stub:
            from oracle import dssim_restate as R
            hw, hh = W // 2, H // 2
            fa = np.ascontiguousarray(base.reshape(H, W, 4)[:hh, :hw]).reshape(hh, hw * 4)
            fb = np.ascontiguousarray(noisy.reshape(H, W, 4)[:hh, :hw]).reshape(hh, hw * 4)
            t0 = time.perf_counter()
            v = R.compare(R.DssimImage(fa, hw, hh, hw * 4, 4), R.DssimImage(fb, hw, hh, hw * 4, 4))
            d1 = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": 1.0 / (4.0 * d1), "unit": "comparisons/s", "cores": 1, "kind": "port",
                                   "sample": "one 1920x1080 comparison by the numpy restatement (oracle/dssim_restate.py), scaled by 4 to 4K; dssim %.6f" % v}
        print(json.dumps(out), flush=True)
    cleanup()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    if args.config == 5:
        if world != args.gpus and world > 1:
            args.gpus = world
        return run_config5(args, rank, local_rank, world)

    import torch
    import mi355fx
    from mi355fx import sharding, synth

    if args.stub:
        _install_stub(torch, mi355fx, rank)
        args.rewarm_steps = min(args.rewarm_steps, 2)
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (torch.cuda unavailable); there is no CPU fallback")
    if os.environ.get("MI355_BENCH_TEST_SHARE_GPU"):
        # test scaffolding only: exercise the N>1 control flow (barriers, max-over-ranks, aggregation) on a 1-GPU box by
        # putting every rank on device 0; the numbers of such a run mean nothing
        local_rank = 0
    if not args.stub:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cpu") if args.stub else torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        # Streams are independent (north_star: "no RCCL: there is no cross-stream collective"): the only cross-rank
        # traffic of the whole run is the barrier and the MAX of one double around the timed region, over a CPU (gloo)
        # group. Neither the data path nor the harness touches RCCL / xGMI.
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("gloo", rank=rank, world_size=world)
        dist = dist_mod

    settings = synth.HSV_SETTINGS["hue90"]
    cube_text = synth.cube_text_3d(33)
    # product-side LUT parse (the host mirror of CubeLut::parse; no oracle on this path)
    from mi355fx.cube import parse_cube
    lut = parse_cube(cube_text)

    per_batch = args.batch * FRAME_BYTES
    ctx = mi355fx.Context(local_rank)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx.set_stream(stream.cuda_stream)
        global EVENTS
        EVENTS = EventPool(torch, max(1024, 3 * (args.steps * args.pairs_per_step // EVENT_EVERY + 1) + 512))
        ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
        if args.lut_variant:
            ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, args.lut_variant)
        if args.hsv_blocks_per_cu:
            ctx.set_flag(mi355fx.FLAG_HSV_BLOCKS_PER_CU, args.hsv_blocks_per_cu)
        for kv in args.ctx_flag:
            k_, v_ = kv.split("=")
            ctx.set_flag(int(k_), int(v_))

        def ramp(run_n):
            """Untimed: keep the device busy with the leg's own launches for --ramp-seconds (clock ramp-up)."""
            t_end = time.perf_counter() + args.ramp_seconds
            while time.perf_counter() < t_end:
                run_n(20)
                torch.cuda.synchronize()

        marker = {}

        def marker_ms():
            """Median interval between two back-to-back timestamped events on the launch stream while it is busy."""
            if "ms" not in marker:
                pairs = []
                for _ in range(32):
                    # keep the queue busy; byte order xRGB = another instantiation (hsvfilter_flat_kernel<5, 1, false>), so that these
                    # 1-frame launches do not dilute the 8-frame launches' average in a rocprofv3 kernel trace of this command
                    ctx.hsvfilter_frames_device(cal.data_ptr(), 1, FRAME_BYTES, W, H, W * 4, "xRGB", settings)
                    a, b = EVENTS.take(), EVENTS.take()
                    a.record(); b.record()
                    pairs.append((a, b))
                torch.cuda.synchronize()
                vals = sorted(a.elapsed_time(b) for a, b in pairs)
                marker["ms"] = vals[len(vals) // 2]
            return marker["ms"]

        def hold_count(need, nb):
            free, _total = torch.cuda.mem_get_info(dev)
            cap = int(min(args.max_source_gib * 2 ** 30, 0.6 * free) // (nb * FRAME_BYTES))
            return max(1, min(need, cap))

        def consume(pool, srcs, n_total, first_index, body):
            """Feeds body(src_slice, n) with pristine batches first_index .. first_index+n_total-1, refilling `srcs`
            (outside anything body times) whenever it runs out. Returns the number of refills."""
            done, fills = 0, 0
            while done < n_total:
                n = min(len(srcs), n_total - done)
                for j in range(n):
                    pool.fill(srcs[j], first_index + done + j)
                torch.cuda.synchronize()
                fills += 1
                body(srcs[:n], n, first_index + done)
                done += n
            return fills

        def measure(content, steps, warmup, record, fused=False, streams=None, native_round=None, solo=False):
            """One leg: ramp on scratch batches, W warm-up steps and K timed steps, each on its own pristine batch.
            solo: this rank alone (no barrier, no MAX over ranks) - a secondary leg of rank 0 in an N > 1 run.
            streams: list of per-stream contexts - frame i of every batch then belongs to stream i and is processed by
            that stream's own context with one-frame launches (what N independent pipelines issue)."""
            nb = len(streams) if streams else args.batch
            # a step = `pairs` launch pairs (micro-steps), each on its own pristine batch; the stream leg's step is one frame per stream
            served, in_timed = {}, [False]   # colorlut kernels picked during the timed region (and the bracketed pass)
            pairs = 1 if streams else args.pairs_per_step
            steps, warmup = steps * pairs, warmup * pairs
            pool = SourcePool(torch, synth, dev, nb, content)
            dsts = [torch.empty((nb, H, W * 4), dtype=torch.uint8, device=dev) for _ in range(args.ring)]
            scratch = [pool.new(10_000 + r) for r in range(args.ring)]
            pitch = FRAME_BYTES

            def region(srcs_, n, rec, every=EVENT_EVERY):
                if streams:
                    for k in range(n):
                        s_, d_ = srcs_[k % len(srcs_)], dsts[k % len(dsts)]
                        if native_round is not None:
                            # the 2 x n launches of a round from one native loop (what n streaming threads written in C cost),
                            # not from n x 2 interpreter calls
                            native_round.issue([s_[i].data_ptr() for i in range(nb)], [d_[i].data_ptr() for i in range(nb)])
                            continue
                        for i, c in enumerate(streams):
                            p_ = s_[i].data_ptr()
                            c.hsvfilter_frames_device(p_, 1, pitch, W, H, W * 4, "RGBA", settings)
                            c.colorlut_frames_device(p_, pitch, W * 4, d_[i].data_ptr(), pitch, W * 4, 1, W, H, "RGBA")
                    return []
                if not fused:
                    return run_region(torch, ctx, srcs_, dsts, settings, n, args.batch, rec, every, served=served if in_timed[0] else None)
                evs_ = []
                for k in range(n):
                    s_, d_ = srcs_[k % len(srcs_)], dsts[k % len(dsts)]
                    sample = rec and (k % every == 0)
                    if sample:
                        e0, e1 = EVENTS.take(), EVENTS.take()
                        e0.record()
                    ctx.hsv_colorlut_frames_device(s_.data_ptr(), pitch, W * 4, d_.data_ptr(), pitch, W * 4, args.batch, W, H, settings)
                    if in_timed[0]:
                        name = ctx.colorlut_kernel_name()
                        served[name] = served.get(name, 0) + 1
                    if sample:
                        e1.record()
                        evs_.append((e0, e1))
                return evs_

            def ramp_body(n):
                # the ramp's in-place hsvfilter passes would grind the scratch batches down to a fixed point (a different,
                # colour-poor picture): refresh them from the pool every few steps so the ramp sees the leg's content too
                for k0 in range(0, n, 2 * len(scratch)):
                    for r, sb in enumerate(scratch):
                        pool.fill(sb, 10_000 + r)
                    region(scratch, min(2 * len(scratch), n - k0), False)

            ramp(ramp_body)
            if fused:
                # the fused entry point builds its table only after 8 calls with unchanged hsv settings, then measures two
                # launches of each kind: keep that learning phase out of the timed region (out-of-place: scratch stays pristine)
                region(scratch, 14, False)
            srcs = [torch.empty((nb, H, W * 4), dtype=torch.uint8, device=dev) for _ in range(hold_count(steps + warmup, nb))]
            stats = {}
            evs, dts = [], []

            def warm(sl, n, first):
                region(sl, n, False)

            def timed(sl, n, first):
                if "first" not in stats:
                    stats["first"] = batch_stats(torch, sl[0])
                if first + n == warmup + steps:
                    stats["last"] = batch_stats(torch, sl[n - 1])
                # Filling the pristine batches and taking their statistics (copies, a sort) leaves the GPU at low clocks; it then
                # needs ~10 ms of this leg's work to be back at speed - 15 % of a 20-step timed region, nothing of a 300-step
                # one. So the device is put back to work on the scratch batches right before the bracket (untimed, declared
                # as config.rewarm_steps); the W warm-up steps on pristine batches have run before.
                ramp_body(args.rewarm_steps)
                # (short regions carry no event markers: ~5 us each on the stream, 2 % of 20 steps; the dedicated pass below
                # then supplies every per-kernel sample)
                def body():
                    in_timed[0] = True
                    evs.extend(region(sl, n, record and steps >= 64))
                    in_timed[0] = False
                if HOST_TRACE is not None:
                    del HOST_TRACE[:]
                dts.append(sharding.timed_region(body, dist=None if solo else dist, device_sync=torch.cuda.synchronize, keep_busy=lambda: ramp_body(min(16, args.rewarm_steps))))
                if HOST_TRACE is not None and len(HOST_TRACE) > 1:
                    gaps = [(i, (b - a) * 1e3) for i, (a, b) in enumerate(zip(HOST_TRACE, HOST_TRACE[1:])) if b - a > 1e-3]
                    print("host trace: %d launch pairs enqueued in %.1f ms of a %.1f ms region; iterations over 1 ms: %s" %
                          (len(HOST_TRACE), (HOST_TRACE[-1] - HOST_TRACE[0]) * 1e3, dts[-1] * 1e3, ["pair %d: %.1f ms" % g for g in gaps][:10]), file=sys.stderr)

            consume(pool, srcs, warmup, 0, warm)
            chunks = consume(pool, srcs, steps, warmup, timed)
            dt = sum(dts)
            res = {"dt": dt, "chunks": chunks, "held": len(srcs), "source_stats": stats, "samples": len(evs), "frames": steps * nb, "launch_pairs": steps, "colorlut_kernels_served": served}
            if record:
                if len(evs) < 32:
                    # few in-region samples (small K): a dedicated bracketed pass on fresh pristine batches, every launch timed
                    extra = []

                    def bracketed(sl, n, first):
                        ramp_body(args.rewarm_steps)
                        extra.extend(region(sl, n, True, 1))
                    consume(pool, srcs, 64, warmup + steps, bracketed)
                    torch.cuda.synchronize()
                    evs = evs + extra
                    res["samples"] = len(evs)
                    res["dedicated_pass_samples"] = len(extra)
                torch.cuda.synchronize()
                # an event bracket is longer than the kernel inside it by the cost of one timestamped marker; that cost is
                # calibrated live (empty brackets on the same stream, GPU busy), reported, and subtracted
                if fused:
                    res["raw_ms"] = (sum(a.elapsed_time(b) for a, b in evs) / len(evs),)
                else:
                    res["raw_ms"] = (sum(a.elapsed_time(b) for a, b, _ in evs) / len(evs), sum(b.elapsed_time(c) for _, b, c in evs) / len(evs))
                res["ms"] = tuple(v - marker_ms() for v in res["raw_ms"])
            del srcs, dsts, scratch, pool
            torch.cuda.empty_cache()
            return res

        cal = torch.zeros(FRAME_BYTES, dtype=torch.uint8, device=dev)
        main_leg = measure(args.content, args.steps, args.warmup, True)
        main_kernel_name = ctx.colorlut_kernel_name()   # the kernel that served the main leg's last colorlut launch
        dt = main_leg["dt"]
        hsv_ms, lut_ms = main_leg["ms"]
        lut_tab, lut_tc, lut_tt = ctx.colorlut_kernel_choice()
        lut_win, lut_gather_t, lut_window_t = ctx.colorlut_kernel_choice(fused=10)
        hsv_tab, hsv_tc, hsv_tt = ctx.colorlut_kernel_choice(fused=2)
        lb = BYTES_PER_FRAME_PER_KERNEL * args.batch
        interp = None
        if not args.no_extra and args.lut_variant == 0 and world == 1:
            # the same two-launch chain with the interpolating (arithmetic) colorlut kernel pinned: no memoised table
            ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 6)
            n_i = max(10, args.steps // 2)
            leg = measure(args.content, n_i, 4, True)
            ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 0)
            h_i, l_i = leg["ms"]
            interp = {"frames_per_s": sharding.aggregate_throughput(leg["frames"], world, leg["dt"]), "colorlut_kernel": ctx.colorlut_kernel_name(),
                      "colorlut_kernels_served": leg["colorlut_kernels_served"],
                      "colorlut_ms_per_launch": l_i, "colorlut_GBps": lb / (l_i * 1e-3) / 1e9,
                      "colorlut_frac_of_hbm_peak": lb / (l_i * 1e-3) / 1e9 / HBM_PEAK_GBS, "hsvfilter_ms_per_launch": h_i}
        fused = None
        if not args.no_extra and world == 1:
            leg = measure(args.content, args.steps, args.warmup, True, fused=True)
            fused_fps = sharding.aggregate_throughput(leg["frames"], world, leg["dt"])
            fused_ms = leg["ms"][0]
            f_tab, f_tc, f_tt = ctx.colorlut_kernel_choice(fused=True)
            f_win, f_gather_t, f_window_t = ctx.colorlut_kernel_choice(fused=11)   # inside the table path: gather kernel / LDS-cached kernel
            fused = {"frames_per_s": fused_fps, "ms_per_launch": fused_ms,
                     "kernel": (max(leg["colorlut_kernels_served"], key=leg["colorlut_kernels_served"].get) + " (composed hsv+lut table)") if f_tab and leg.get("colorlut_kernels_served") else ("memoised table kernel (composed hsv+lut table)" if f_tab else "fused compute kernel"),
                     "kernels_served": leg.get("colorlut_kernels_served"),
                     "auto_ms_per_mpx": {"compute": f_tc, "table": f_tt},
                     "table_kernel_ms_per_mpx": {"gather": f_gather_t, "lds_cached": f_window_t, "lds_cached_in_use": bool(f_win)},
                     "algorithmic_bytes_per_launch": lb, "GBps": lb / (fused_ms * 1e-3) / 1e9,
                     # two different quantities: the kernel inside its event bracket, and the leg's wall-clock throughput
                     # (launch gaps included) against the 8 B/pixel roofline of 120,563 frames/s
                     "kernel_frac_of_hbm_peak": lb / (fused_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "throughput_frac_of_hbm_peak": fused_fps * BYTES_PER_FRAME_PER_KERNEL / 1e9 / HBM_PEAK_GBS,
                     "note": "one launch per batch, 8 B/px algorithmic (SURVEY 8d fused accounting); bit-identical to the two-kernel chain"}
        nt_ab = None
        if not args.no_extra and world == 1 and not args.stub:
            # A/B: the same two launches with hsvfilter's output written past the Infinity Cache (non-temporal stores). What the
            # colorlut launch loses is what it gets from reading the batch hsvfilter has just written on-die instead of from HBM:
            # the 16 B/pixel of the headline are algorithmic bytes, and 4 of them need not touch HBM (roofline.on_die_bytes_per_launch).
            ctx.set_flag(mi355fx.FLAG_HSV_NT, 1)
            leg = measure(args.content, max(10, args.steps // 2), 4, True)
            ctx.set_flag(mi355fx.FLAG_HSV_NT, 0)
            nt_ab = {"frames_per_s": leg["frames"] / leg["dt"], "hsvfilter_ms_per_launch": leg["ms"][0], "colorlut_ms_per_launch": leg["ms"][1],
                     "colorlut_kernels_served": leg["colorlut_kernels_served"],
                     "what": "the headline's two launches with MI355_FLAG_HSV_NT = 1: hsvfilter stores non-temporally, colorlut reads its input from HBM"}
        extra = None
        if not args.no_extra and world == 1:
            other = "noise" if args.content == "smooth" else "smooth"
            n_o = max(10, args.steps // 2)
            # warm-up long enough for the colorlut kernel choice to follow the change of content (sampled every 8th launch)
            leg = measure(other, n_o, 18, True)
            extra = {"content": other, "frames_per_s": leg["frames"] / leg["dt"],
                     "hsvfilter_ms_per_launch": leg["ms"][0], "colorlut_ms_per_launch": leg["ms"][1],
                     "colorlut_kernel": ctx.colorlut_kernel_name(), "source_stats": leg["source_stats"]}

        streams_leg = None
        if not args.no_extra and world == 1 and args.streams > 1:
            # N independent streams per GPU (config 5 runs 32): one context + HIP stream per stream, one 4K frame per launch;
            # the streams load the same LUT, so they share one memoised table (mi355_shared_table_count)
            sctx, sstreams = [], []
            for _ in range(args.streams):
                c = mi355fx.Context(local_rank)
                if not args.stub:
                    st_ = torch.cuda.Stream(device=dev)
                    c.set_stream(st_.cuda_stream)
                    sstreams.append(st_)
                c.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
                if args.lut_variant:
                    c.set_flag(mi355fx.FLAG_LUT_VARIANT, args.lut_variant)
                sctx.append(c)
            torch.cuda.synchronize()
            n_s = max(8, args.steps // 4)
            leg_py = measure(args.content, n_s, 24, False, streams=sctx)
            if args.stub:
                leg = leg_py
            else:
                leg = measure(args.content, n_s, 24, False, streams=sctx, native_round=mi355fx.StreamsRound(sctx, W, H, W * 4, "RGBA", settings))
            streams_leg = {"streams_per_gpu": args.streams, "frames_per_s": leg["frames"] / leg["dt"], "launches": "one 4K frame per launch and stream",
                           "issued_by": "one native loop per round of 2 x %d launches (mi355_issue_streams_round)" % args.streams,
                           "frames_per_s_issued_from_python": leg_py["frames"] / leg_py["dt"],
                           "colorlut_kernel": sctx[0].colorlut_kernel_name(),
                           "memoised_tables_alive": mi355fx.load_library().mi355_shared_table_count() if not args.stub else 0}
            if not args.stub:
                # the same streams, each still handing over ONE 4K frame per call, through the multi-stream dispatcher
                # (mi355_group_*, csrc/group.hip): frames that agree in geometry, settings and LUT share a launch pair of up to
                # eight frames; bit-identical per stream (tests/test_gpu_group.py). This is the figure for many streams; the
                # one above is what they get when every element launches for itself.
                group = mi355fx.Group(local_rank)

                class GroupRound:
                    def issue(self, src_ptrs, dst_ptrs):
                        group.submit_round(sctx, src_ptrs, dst_ptrs, W, H, W * 4, "RGBA", settings)

                leg_g = measure(args.content, n_s, 24, False, streams=sctx, native_round=GroupRound())
                group.wait_all()
                g_frames, g_pairs, g_single = group.stats()
                streams_leg["frames_per_s_separate_launches"] = streams_leg["frames_per_s"]
                streams_leg["frames_per_s"] = leg_g["frames"] / leg_g["dt"]
                streams_leg["launches"] = "one 4K frame per CALL and stream; launches: up to 8 frames of different streams each (mi355_group_*)"
                streams_leg["issued_by"] = "one native loop per round: %d x mi355_group_submit_chain + flush (mi355_group_submit_round)" % args.streams
                streams_leg["group"] = {"frames": g_frames, "batched_launch_pairs": g_pairs, "frames_through_own_context": g_single,
                                        "frames_per_launch_pair": g_frames / max(g_pairs, 1)}
                group.close()
                # ... and as the FUSED pair (mi355_group_submit_fused: one launch per batch through the composed table, sources
                # left untouched - what the shim's hsvfilter -> colorlut hand-over submits): 8 B/pixel for the chain
                group_f = mi355fx.Group(local_rank)

                class GroupRoundFused:
                    def issue(self, src_ptrs, dst_ptrs):
                        group_f.submit_round(sctx, src_ptrs, dst_ptrs, W, H, W * 4, "RGBA", settings, fused=True)

                leg_f = measure(args.content, n_s, 24, False, streams=sctx, native_round=GroupRoundFused())
                group_f.wait_all()
                f_frames, f_launches, f_single = group_f.stats()
                streams_leg["fused"] = {"frames_per_s": leg_f["frames"] / leg_f["dt"], "launches": "one per batch of up to 8 streams' frames (mi355_group_submit_fused)",
                                        "frames": f_frames, "batched_launches": f_launches, "frames_through_own_context": f_single}
                group_f.close()
            for c in sctx:
                c.close()

        sweep = None
        if not args.no_extra and world == 1:
            # what the chain does on other content statistics, auto kernel choice (default flags) and the arithmetic brick /
            # three-pass path pinned (variant 6): natural-like frame + uniform noise of +-amp per channel, and uniform noise
            sweep = {}
            n_c = max(8, args.steps // 5)
            for label, content in (("amp0", "smooth"), ("amp4", "smooth+4"), ("amp8", "smooth+8"), ("uniform", "noise")):
                row = {}
                for tag, variant in (("auto", 0), ("interpolating", 6)):
                    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, variant)
                    leg = measure(content, n_c, 18, True)
                    row[tag] = {"frames_per_s": leg["frames"] / leg["dt"], "hsvfilter_ms_per_launch": leg["ms"][0], "colorlut_ms_per_launch": leg["ms"][1],
                                "colorlut_frac_of_hbm_peak": lb / (leg["ms"][1] * 1e-3) / 1e9 / HBM_PEAK_GBS, "colorlut_kernels_served": leg["colorlut_kernels_served"]}
                row["distinct_colours_frame0"] = leg["source_stats"]["first"]["distinct_colours_frame0"]
                sweep[label] = row
            ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, args.lut_variant)
        elif not args.no_extra and world > 1 and rank == 0:
            # N > 1: the secondary legs are dropped, but the scaling record should show more than the amp-0 number - rank 0 alone
            # (the other ranks wait at the final barrier, their GPUs idle) measures the chain on the natural-like frame + uniform
            # noise of +-4 per channel with the default flags. One GPU's figure, not the job's.
            leg = measure("smooth+4", max(8, args.steps // 5), 18, True, solo=True)
            sweep = {"amp4": {"auto": {"frames_per_s": leg["frames"] / leg["dt"], "hsvfilter_ms_per_launch": leg["ms"][0], "colorlut_ms_per_launch": leg["ms"][1],
                                       "colorlut_frac_of_hbm_peak": lb / (leg["ms"][1] * 1e-3) / 1e9 / HBM_PEAK_GBS, "colorlut_kernels_served": leg["colorlut_kernels_served"]},
                              "distinct_colours_frame0": leg["source_stats"]["first"]["distinct_colours_frame0"],
                              "measured_on": "rank 0 alone after the timed region (one GPU's frames/s; the other ranks idle at the barrier)"}}

    fps = sharding.aggregate_throughput(main_leg["frames"], world, dt)
    ms_per_step = dt / args.steps * 1e3
    config5 = None
    if not args.no_extra and world == 1 and not args.stub and args.streams > 1:
        # BASELINE config 5 (32 concurrent 4K streams per GPU through the SSIM engine) as a short leg of the default line
        leg_args = argparse.Namespace(**vars(args))
        leg_args.as_leg, leg_args.streams, leg_args.workers, leg_args.steps, leg_args.warmup = True, 32, 8, 48, 4
        leg_args.group = False
        config5 = run_config5(leg_args, rank, local_rank, 1)     # 8 worker contexts take the 32 streams in turn (the form of rounds 3-5)
        # the same elements through the process's dispatcher (8 host threads stand in for the 32 element threads) ...
        leg_args = argparse.Namespace(**vars(leg_args))
        leg_args.group, leg_args.steps = True, 24
        config5["through_the_dispatcher"] = run_config5(leg_args, rank, local_rank, 1)
        leg_args = argparse.Namespace(**vars(leg_args))
        leg_args.group, leg_args.steps = False, 48
        # ... and as what BASELINE configures literally: 32 two-pad elements on 32 NATIVE threads, each with its own context, against
        # the same threads handing their pairs to the dispatcher (tools/agroup_bench.cpp; an interpreter cannot host 32 streaming threads)
        native = os.path.join(ROOT, "tools", "agroup_bench")
        if os.path.exists(native):
            import subprocess
            try:
                r_ = subprocess.run([native, "32", "dssim"], capture_output=True, text=True, timeout=300)
                for l_ in r_.stdout.splitlines():
                    if l_.startswith("{") and '"dssim"' in l_:
                        config5["native_element_threads"] = json.loads(l_)
            except Exception as e_:      # noqa: BLE001
                config5["native_element_threads"] = {"error": str(e_)}
        # the same 32 streams as the non-reference pads of 4 videocompare elements (8 pads + one reference pad each): the element
        # hashes its reference frame once per aggregate, so a comparison costs one hash + one compare (+ 1/8 of a hash)
        leg_args = argparse.Namespace(**vars(leg_args))
        leg_args.shared_reference, leg_args.workers = True, 4
        config5["as_pads_of_four_aggregators"] = run_config5(leg_args, rank, local_rank, 1)
        # ... and of 2 elements with 16 pads each (fewer reference hashes, one synchronisation per 16 comparisons)
        leg_args = argparse.Namespace(**vars(leg_args))
        leg_args.workers, leg_args.steps = 2, 32
        config5["as_pads_of_two_aggregators"] = run_config5(leg_args, rank, local_rank, 1)

    if rank == 0:
        # dominant kernel = the longer of the two launches
        per_launch_bytes = lb
        # the kernel that served most of the main leg's colorlut launches (auto may sample the other kinds in between)
        served = main_leg.get("colorlut_kernels_served") or {}
        lut_name = max(served, key=served.get) if served else main_kernel_name
        if lut_ms >= hsv_ms:
            dom, dom_ms = lut_name, lut_ms
        else:
            dom, dom_ms = ("colorlut_table_tiled_kernel (hsvfilter table)" if hsv_tab else "hsvfilter_flat_kernel"), hsv_ms
        achieved = per_launch_bytes / (dom_ms * 1e-3) / 1e9
        # HBM bytes per launch cannot be read from inside the process: `traffic` comes from rocprofv3 --pmc passes run as child
        # processes at the end of this run (live_pmc_traffic); if that is not possible (no rocprofv3, N > 1, --no-extra), it is the
        # per-launch figure of the committed passes of THIS command (tools/collect_profiles.sh -> profiles/pmc_latest.json) when that
        # profile was taken with the same batch size, content, kernel and code; otherwise null. The provenance is spelled out next to it.
        traffic, traffic_src, live_note, rocprof_ms = None, None, None, None
        if world == 1 and not args.stub and not args.no_extra and not args.no_live_pmc:
            # measured on this box, by this run (a committed file is not a measurement of the driver's run)
            traffic, rocprof_ms, traffic_src = live_pmc_traffic(args, dom.split(" ")[0].split("<")[0])
            if traffic is None:
                live_note, traffic_src = traffic_src, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json"))) if traffic is None else {}
            if (pmc.get("frames_per_launch", pmc.get("frames_per_step")) == args.batch and pmc.get("content", "smooth") == args.content
                    and pmc.get("pristine_sources") and pmc.get("lut_variant", 0) == args.lut_variant
                    and pmc.get("source_fingerprint") == source_fingerprint()):   # taken from THIS code, or null
                for kname, rec in pmc.get("kernels", {}).items():
                    if kname.startswith(dom.split(" ")[0].split("<")[0]):
                        traffic = rec["hbm_bytes"]
                        traffic_src = "profiles/pmc_latest.json: separate rocprofv3 --pmc passes of this command (%s)" % pmc.get("collected", "?")
        except (OSError, ValueError, KeyError):
            pass
        chain_gbs = 2 * per_launch_bytes / ((hsv_ms + lut_ms) * 1e-3) / 1e9
        out = {
            "metric": "4K RGBA frames/sec through hsvfilter+colorlut at 1 GPU; % HBM roofline",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "stub" if args.stub else "synthetic",
            "config": {"workload": "hsvfilter(hue-shift=90) -> colorlut(33^3 trilinear), 3840x2160 RGBA, two kernels",
                       "frames_per_step": args.batch * args.pairs_per_step, "frames_per_launch": args.batch,
                       "launches_per_step": 2 * args.pairs_per_step, "content": args.content,
                       "algorithmic_bytes_per_frame": 2 * BYTES_PER_FRAME_PER_KERNEL, "streams_per_gpu": 1,
                       "lut_variant": args.lut_variant, "ctx_flags": args.ctx_flag, "ramp_seconds": args.ramp_seconds, "rewarm_steps": args.rewarm_steps, "event_marker_ms": marker_ms(),
                       "sources": "pristine: every warm-up/timed step filters its own never-touched batch (in place)",
                       "source_batches_held": main_leg["held"], "source_chunks": main_leg["chunks"],
                       "source_stats": main_leg["source_stats"], "dst_ring_batches": args.ring,
                       "timing_group": "gloo (CPU) barrier + MAX; no RCCL" if world > 1 else "single process"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_live_note": live_note,
                         "source_fingerprint": source_fingerprint(),
                         "algorithmic_bytes_per_launch": per_launch_bytes, "avg_launch_ms": dom_ms,
                         # of the algorithmic bytes of a colorlut launch, its input (half) was written by the hsvfilter launch just
                         # before it and can be served by the 256 MiB Infinity Cache instead of HBM: an upper bound, not a counter
                         # (FETCH_SIZE counts Infinity-Cache hits too); the measured effect is the hsvfilter_nontemporal_ab leg
                         "on_die_bytes_per_launch": (min(per_launch_bytes // 2, 256 << 20) if lut_ms >= hsv_ms else 0),
                         "on_die_note": "upper bound: the part of this launch's input that the preceding hsvfilter launch left in the Infinity Cache (256 MiB); "
                                        "with hsvfilter's stores made non-temporal the same launch reads it from HBM - see hsvfilter_nontemporal_ab",
                         "rocprof_avg_launch_ms": rocprof_ms,   # the same kernel in a rocprofv3 --kernel-trace --stats child run on this box
                         "avg_launch_ms_raw_bracket": main_leg["raw_ms"][1] if lut_ms >= hsv_ms else main_leg["raw_ms"][0],
                         "launch_samples": main_leg["samples"]},
            "kernels": {"hsvfilter_ms_per_launch": hsv_ms, "colorlut_ms_per_launch": lut_ms,
                        "hsvfilter_ms_raw_bracket": main_leg["raw_ms"][0], "colorlut_ms_raw_bracket": main_leg["raw_ms"][1],
                        "hsvfilter_GBps": per_launch_bytes / (hsv_ms * 1e-3) / 1e9,
                        "colorlut_GBps": per_launch_bytes / (lut_ms * 1e-3) / 1e9,
                        "chain_GBps": chain_gbs, "chain_frac_of_hbm_peak": chain_gbs / HBM_PEAK_GBS,
                        "colorlut_kernel": lut_name, "colorlut_kernels_served": main_leg["colorlut_kernels_served"],
                        "colorlut_auto_ms_per_mpx": {"compute": lut_tc, "table": lut_tt},
                        "colorlut_table_kernel_ms_per_mpx": {"gather": lut_gather_t, "lds_cached": lut_window_t, "lds_cached_in_use": bool(lut_win)},
                        "hsvfilter_kernel": "colorlut_table_tiled_kernel (hsvfilter table)" if hsv_tab else "hsvfilter_flat_kernel",
                        "hsvfilter_auto_ms_per_mpx": {"compute": hsv_tc, "table": hsv_tt},
                        "event_marker_ms_subtracted": marker_ms(), "samples": main_leg["samples"],
                        "dedicated_pass_samples": main_leg.get("dedicated_pass_samples", 0)},
        }
        if interp:
            out["interpolating_kernel_only"] = interp
        if fused:
            out["fused_chain"] = fused
        if nt_ab:
            out["hsvfilter_nontemporal_ab"] = nt_ab
        if extra:
            out["other_content"] = extra
        if streams_leg:
            out["concurrent_streams"] = streams_leg
        if sweep:
            out["content_sweep"] = sweep
        if config5:
            out["config5"] = config5
        if not args.no_cpu_baseline:
            one, mt = cpu_baseline(synth, settings, cube_text, seconds_target=0.5 if args.stub else 12.0)
            out["cpu_baseline"] = one
            out["cpu_baseline_all_cores"] = mt
        print(json.dumps(out), flush=True)

    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
