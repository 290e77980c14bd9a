"""rsaudioecho, BASELINE config 1 shape (48 kHz stereo f32, 250 ms delay, feedback 0.4) as a deployment: S independent streams
fed 10 ms buffers. One launch set for the whole batch (mi355_echo_process_batch_device) against one context per stream
(mi355_echo_process_device in a loop), device-resident buffers; the CPU oracle (scalar C restatement) timed on the same
buffers for scale. Prints one JSON line. Run on the GPU box: python tools/bench_echo.py [streams]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd")); sys.path.insert(0, ROOT)
import mi355fx


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    rate, ch, n = 48000, 2, 960  # 10 ms of interleaved stereo
    ring = rate * ch  # max-delay 1 s
    delay, inten, fb = 24000, 0.6, 0.4
    x = np.random.default_rng(0).standard_normal((S, n)).astype(np.float32)
    iters = 300
    # batch
    ctx = mi355fx.Context(0)
    ctx.echo_setup_batch(S, ring)
    dev = ctx.alloc(x.nbytes); ctx.h2d(dev, x.reshape(-1).view(np.uint8))
    run = lambda: ctx.echo_process_batch_device(dev, n, n, False, [delay] * S, [inten] * S, [fb] * S)
    for _ in range(30): run()
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): run()
    ctx.synchronize(); t_batch = (time.perf_counter() - t0) / iters
    ctx.free(dev); ctx.echo_reset()
    # one context per stream, same thread
    ctxs = [mi355fx.Context(0) for _ in range(S)]
    devs = []
    for c, row in zip(ctxs, x):
        c.echo_setup(ring); d = c.alloc(row.nbytes); c.h2d(d, row.view(np.uint8)); devs.append(d)
    def run1():
        for c, d in zip(ctxs, devs):
            c._ck(c.L.mi355_echo_process_device(c.h, d, n, 0, delay, inten, fb))
    for _ in range(10): run1()
    for c in ctxs: c.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters // 3): run1()
    for c in ctxs: c.synchronize()
    t_single = (time.perf_counter() - t0) / (iters // 3)
    for c, d in zip(ctxs, devs): c.free(d); c.close()
    # CPU oracle, one core
    from oracle import oracle
    es = [oracle.Echo(10 ** 9, rate, ch) for _ in range(S)]
    rows = [np.ascontiguousarray(r) for r in x]
    t0 = time.perf_counter()
    for _ in range(iters):
        for e, r in zip(es, rows): e.process(r, 250 * 10 ** 6, inten, fb)
    t_cpu = (time.perf_counter() - t0) / iters
    buf_s = n / (rate * ch)
    print(json.dumps({"config": "rsaudioecho, %d streams x 10 ms buffers (48 kHz stereo f32, delay 250 ms, feedback 0.4), device-resident" % S,
                      "batch_ms_per_step": t_batch * 1e3, "batch_realtime_factor_per_stream": buf_s / t_batch,
                      "one_context_per_stream_ms_per_step": t_single * 1e3, "batch_speedup": t_single / t_batch,
                      "cpu_oracle_1core_ms_per_step": t_cpu * 1e3}))


if __name__ == "__main__":
    main()
