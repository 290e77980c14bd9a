"""bench.py's N>1 control flow end to end on the CPU: two ranks under torch.distributed.run with `--stub` (tiny CPU frames, a
context that sleeps instead of launching kernels). Checks what the driver relies on: one JSON line from rank 0 only, n_gpus = 2,
`value` = the units ALL ranks processed / the MAX-over-ranks time (rank 1 is made twice as slow), the timing group is gloo (no
RCCL anywhere), and cpu_baseline is present at N > 1."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(nproc, steps=6, warmup=2, extra=()):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", str(steps), "--warmup", str(warmup),
           "--stub", "--ramp-seconds", "0.02"] + list(extra)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                     # rank 0 prints, nobody else
    return json.loads(lines[0])


def test_two_ranks_aggregate_over_the_slowest_rank():
    d = _run(2)
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["data"] == "stub" and d["scaling"] == "weak"
    assert "gloo" in d["config"]["timing_group"] and "RCCL" in d["config"]["timing_group"]
    # rank 1 sleeps 2 ms per launch, rank 0 1 ms: a step is four (hsvfilter, colorlut) pairs = 8 launches -> >= 16 ms on the slowest rank
    assert d["config"]["launches_per_step"] == 8 and d["config"]["frames_per_step"] == 4 * d["config"]["frames_per_launch"]
    assert d["ms_per_step"] >= 16.0
    frames = d["steps"] * d["config"]["frames_per_step"] * 2     # both ranks' frames
    assert abs(d["value"] - frames / (d["ms_per_step"] * 1e-3 * d["steps"])) / d["value"] < 1e-6
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] == "port"
    assert "roofline" in d and d["roofline"]["bound"] == "hbm"
    # N > 1 keeps one secondary leg: rank 0 alone on the +-4-noise content (the scaling record then shows more than amp 0)
    amp4 = d["content_sweep"]["amp4"]
    assert amp4["auto"]["frames_per_s"] > 0 and "rank 0 alone" in amp4["measured_on"]
    assert sorted(d["content_sweep"]) == ["amp4"]                # ... and only that one: the other legs are N = 1 business
    for key in ("interpolating_kernel_only", "fused_chain", "other_content", "concurrent_streams", "config5"):
        assert key not in d, key


def test_single_process_stub_has_all_legs():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--stub", "--ramp-seconds", "0.02"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    for key in ("interpolating_kernel_only", "fused_chain", "other_content", "cpu_baseline", "cpu_baseline_all_cores"):
        assert key in d, key
    assert d["config"]["sources"].startswith("pristine") and d["config"]["source_chunks"] >= 1
    # the driver's contract (one JSON line with these keys) and the two objects this tier adds
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "u8"
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in d["cpu_baseline"], key
    assert "workload" in d["config"] and "model" not in d["config"]
    for key in ("ramp_seconds", "rewarm_steps", "event_marker_ms"):   # everything untimed that precedes the bracket is declared
        assert key in d["config"], key
    assert sum(d["kernels"]["colorlut_kernels_served"].values()) == d["steps"] * d["config"]["launches_per_step"] // 2


def test_config5_two_ranks_aggregate_over_the_slowest_rank():
    """BASELINE config 5 is the config that names 8 GPUs: its own barrier / aggregation path (bench.py run_config5) as two real ranks
    on the CPU. One JSON line from rank 0, n_gpus 2, value = both ranks' comparisons / the MAX-over-ranks time (rank 1's stub
    dispatcher takes 2 ms per comparison, rank 0's 1 ms), gloo only."""
    d = _run(2, steps=5, warmup=1, extra=("--config", "5", "--streams", "4", "--group"))
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 1 and d["data"] == "stub" and d["scaling"] == "weak"
    assert d["unit"] == "comparisons/s" and "config 5" in d["metric"]
    assert "gloo" in d["config"]["timing_group"] and "RCCL" in d["config"]["timing_group"]
    assert d["config"]["streams_per_gpu"] == 4 and "dispatcher" in d["config"]["workload"]
    assert d["ms_per_step"] >= 2.0                                   # the slower rank's 2 ms per (concurrent) comparison
    comps = d["steps"] * 4 * 2                                       # both ranks' comparisons
    assert abs(d["value"] - comps / (d["ms_per_step"] * 1e-3 * d["steps"])) / d["value"] < 1e-6
    assert d["roofline"]["bound"] == "valu" and d["roofline"]["instr_per_pixel"] > 1000 and "hbm" in d["roofline"]
    assert d["dispatcher"]["pairs"] >= 5 * 4


def test_config5_single_process_stub_without_the_dispatcher():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "5", "--steps", "3", "--warmup", "1", "--stub", "--ramp-seconds", "0.02",
                          "--streams", "4", "--workers", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["worker_contexts_per_gpu"] == 2 and "dispatcher" not in d
