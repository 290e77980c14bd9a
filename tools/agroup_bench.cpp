// tools/agroup_bench.cpp — N independent element instances (audio, then videocompare) as N NATIVE threads (what N GStreamer streaming threads are), each
// handing one buffer per interval to (a) its own single-instance context (the shim's path until round 5) and (b) the process's
// mi355_agroup (round 6). Prints one JSON line per element kind. Build: make -C tools agroup_bench (g++, links libmi355fx.so).
// Run on the GPU box: tools/agroup_bench [instances]
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/mi355fx.h"

using clk = std::chrono::steady_clock;
static double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

// a reusable barrier (C++17: no std::barrier)
struct Barrier {
  std::atomic<int> count{0}, gen{0};
  int n;
  explicit Barrier(int n_) : n(n_) {}
  void wait() {
    const int g = gen.load();
    if (count.fetch_add(1) + 1 == n) { count.store(0); gen.fetch_add(1); }
    else while (gen.load() == g) std::this_thread::yield();
  }
};

template <typename F>
static double run_threads(int n, int iters, F body) {
  Barrier bar(n + 1);
  std::vector<std::thread> th;
  for (int m = 0; m < n; m++)
    th.emplace_back([&, m] {
      bar.wait();
      for (int i = 0; i < iters; i++) body(m, i);
      bar.wait();
    });
  bar.wait();
  const auto t0 = clk::now();
  bar.wait();
  const double dt = secs(t0, clk::now());
  for (auto &t : th) t.join();
  return dt;
}

#define CK(x) do { int rc_ = (x); if (rc_) { std::fprintf(stderr, "%s -> %d\n", #x, rc_); std::exit(1); } } while (0)

int main(int argc, char **argv) {
  const int S = argc > 1 ? std::atoi(argv[1]) : 32;
  const std::string section = argc > 2 ? argv[2] : "all";   // all | audio | videocompare | dssim
  const bool do_audio = section == "all" || section == "audio", do_video = section != "audio", dssim_only = section == "dssim";
  int st = 0;
  // ------------------------------------------------------------ rsaudioecho: 48 kHz stereo f32, 10 ms buffers, delay 250 ms, feedback 0.4
  if (do_audio) {
    const size_t n = 960, ring = 96000;
    const int iters = 400;
    std::vector<mi355_ctx *> ctxs(S);
    std::vector<std::vector<float>> buf(S, std::vector<float>(n));
    std::vector<void *> dev(S);
    for (int m = 0; m < S; m++) {
      ctxs[m] = mi355_ctx_create(0, &st);
      if (!ctxs[m]) { std::fprintf(stderr, "no context: %d\n", st); return 1; }
      CK(mi355_echo_setup(ctxs[m], ring));
      for (size_t i = 0; i < n; i++) buf[m][i] = 0.1f * std::sin(0.01f * (float)(i + 17 * m));
      dev[m] = mi355_device_alloc(ctxs[m], n * 4);
      CK(mi355_memcpy_h2d(ctxs[m], dev[m], buf[m].data(), n * 4));
    }
    // warm-up + (a) one context per instance, host buffers (the shim's call) and device buffers
    run_threads(S, 50, [&](int m, int) { CK(mi355_echo_process_f32(ctxs[m], buf[m].data(), n, 24000, 0.6, 0.4)); });
    const double own_host = run_threads(S, iters, [&](int m, int) { CK(mi355_echo_process_f32(ctxs[m], buf[m].data(), n, 24000, 0.6, 0.4)); }) / iters;
    const double own_dev = run_threads(S, iters, [&](int m, int) {
      CK(mi355_echo_process_device(ctxs[m], dev[m], n, 0, 24000, 0.6, 0.4));
      CK(mi355_ctx_synchronize(ctxs[m]));   // transform_ip returns a finished buffer
    }) / iters;
    // (b) the group
    mi355_agroup *g = mi355_agroup_create_echo(0, S, ring, &st);
    if (!g) { std::fprintf(stderr, "no agroup: %d\n", st); return 1; }
    CK(mi355_agroup_set_linger(g, 2000, 0));
    auto via_group = [&](int device) {
      return [&, device](int m, int) {
        uint64_t t = 0;
        CK(mi355_agroup_submit_echo(g, m, device ? dev[m] : (void *)buf[m].data(), n, 0, 24000, 0.6, 0.4, device, &t));
        CK(mi355_agroup_wait(g, t, nullptr));
      };
    };
    run_threads(S, 50, via_group(0));
    const double grp_host = run_threads(S, iters, via_group(0)) / iters;
    const double grp_dev = run_threads(S, iters, via_group(1)) / iters;
    uint64_t stats[3];
    CK(mi355_agroup_stats(g, stats));
    // the same from ONE thread (no thread wake-ups in the number): all members submitted, then waited for
    const auto t0 = clk::now();
    for (int i = 0; i < iters; i++) {
      std::vector<uint64_t> tk(S);
      for (int m = 0; m < S; m++) CK(mi355_agroup_submit_echo(g, m, dev[m], n, 0, 24000, 0.6, 0.4, 1, &tk[m]));
      for (int m = 0; m < S; m++) CK(mi355_agroup_wait(g, tk[m], nullptr));
    }
    const double grp_one_thread = secs(t0, clk::now()) / iters;
    std::printf("{\"element\": \"rsaudioecho\", \"instances\": %d, \"buffer\": \"10 ms of 48 kHz stereo f32 (960 samples), delay 250 ms, feedback 0.4\", "
                "\"own_context_host_buffers_ms_per_interval\": %.4f, \"own_context_device_buffers_ms_per_interval\": %.4f, "
                "\"agroup_host_buffers_ms_per_interval\": %.4f, \"agroup_device_buffers_ms_per_interval\": %.4f, \"agroup_device_buffers_one_thread_ms_per_interval\": %.4f, "
                "\"agroup_launch_sets\": %llu, \"agroup_buffers\": %llu, \"agroup_largest_set\": %llu, \"threads\": \"one native thread per instance\"}\n",
                S, own_host * 1e3, own_dev * 1e3, grp_host * 1e3, grp_dev * 1e3, grp_one_thread * 1e3, (unsigned long long)stats[1], (unsigned long long)stats[0],
                (unsigned long long)stats[2]);
    std::fflush(stdout);
    mi355_agroup_destroy(g);
    for (int m = 0; m < S; m++) { mi355_device_free(ctxs[m], dev[m]); mi355_ctx_destroy(ctxs[m]); }
  }
  // ------------------------------------------------------------ ebur128level: 48 kHz stereo f32, 100 ms buffers, mode M | S | I | LRA | sample + true peak
  if (do_audio) {
    const unsigned rate = 48000, ch = 2;
    const size_t frames = 4800;
    const int iters = 100;
    std::vector<mi355_ctx *> ctxs(S);
    std::vector<std::vector<float>> buf(S, std::vector<float>(frames * ch));
    for (int m = 0; m < S; m++) {
      ctxs[m] = mi355_ctx_create(0, &st);
      CK(mi355_ebur128_setup(ctxs[m], ch, rate, 63, nullptr));
      for (size_t i = 0; i < frames * ch; i++) buf[m][i] = 0.1f * std::sin(0.02f * (float)(i + 31 * m));
    }
    auto own = [&](int m, int) {
      double v;
      CK(mi355_ebur128_add_frames(ctxs[m], buf[m].data(), frames, 2));
      CK(mi355_ebur128_loudness_momentary(ctxs[m], &v));
    };
    run_threads(S, 10, own);
    const double t_own = run_threads(S, iters, own) / iters;
    mi355_agroup *g = mi355_agroup_create_ebur128(0, S, ch, rate, 63, nullptr, &st);
    if (!g) { std::fprintf(stderr, "no agroup: %d\n", st); return 1; }
    CK(mi355_agroup_set_linger(g, 2000, 0));   // independent meters: what gst/gstebur128level.c sets
    auto grp = [&](int m, int) {
      uint64_t t = 0;
      double v;
      CK(mi355_agroup_submit_ebur128(g, m, buf[m].data(), frames, 2, 0, &t));
      CK(mi355_agroup_wait(g, t, nullptr));
      CK(mi355_agroup_ebur128_loudness(g, m, 0, &v));
    };
    run_threads(S, 10, grp);
    const double t_grp = run_threads(S, iters, grp) / iters;
    std::printf("{\"element\": \"ebur128level\", \"instances\": %d, \"buffer\": \"100 ms of 48 kHz stereo f32, all modes, momentary loudness read per buffer\", "
                "\"own_context_ms_per_interval\": %.4f, \"agroup_ms_per_interval\": %.4f, \"own_context_realtime_aggregate\": %.1f, \"agroup_realtime_aggregate\": %.1f}\n",
                S, t_own * 1e3, t_grp * 1e3, S * 0.1 / t_own, S * 0.1 / t_grp);
    std::fflush(stdout);
    mi355_agroup_destroy(g);
    for (int m = 0; m < S; m++) mi355_ctx_destroy(ctxs[m]);
  }
  // ------------------------------------------------------------ audioloudnorm: 192 kHz stereo f64, 8 s per instance, whole frames
  if (do_audio) {
    const unsigned ch = 2;
    const size_t rate = 192000, total = 8 * rate;
    std::vector<std::vector<double>> x(S, std::vector<double>(total * ch));
    for (int m = 0; m < S; m++)
      for (size_t i = 0; i < total; i++) {
        const double t = (double)i / (double)rate;
        double a = 0.05 * std::sin(2 * M_PI * (440.0 + m) * t) * (1.0 + 0.5 * std::sin(2 * M_PI * 0.2 * t));
        if ((i / 9600 + (size_t)m) % 37 == 0 && i % 9600 < 1500) a *= 14.0;   // bursts for the limiter
        x[m][2 * i] = a; x[m][2 * i + 1] = 0.8 * a;
      }
    std::vector<std::vector<double>> out(S, std::vector<double>((size_t)31 * 19200 * ch));
    // (a) one context per instance
    std::vector<mi355_ctx *> ctxs(S);
    for (int m = 0; m < S; m++) { ctxs[m] = mi355_ctx_create(0, &st); CK(mi355_loudnorm_setup(ctxs[m], ch, -24.0, 7.0, -2.0, 0.0)); }
    const double t_own = run_threads(S, 1, [&](int m, int) {
      size_t pos = 0, n_out = 0;
      std::vector<double> big((size_t)(3 * rate / 19200 + 40) * 19200 * ch);
      // the element pushes what arrives: here 100 ms buffers (the adapter behind push collects the first 3 s)
      while (pos < total) { CK(mi355_loudnorm_push(ctxs[m], x[m].data() + pos * ch, 19200, big.data(), big.size() / ch, &n_out)); pos += 19200; }
      int eos = 0;
      CK(mi355_loudnorm_drain(ctxs[m], big.data(), big.size() / ch, &n_out, &eos));
    });
    for (int m = 0; m < S; m++) mi355_ctx_destroy(ctxs[m]);
    // (b) the group
    mi355_agroup *g = mi355_agroup_create_loudnorm(0, S, ch, -24.0, 7.0, -2.0, 0.0, &st);
    if (!g) { std::fprintf(stderr, "no agroup: %d\n", st); return 1; }
    // independent members. A run that goes as fast as it can (a file transcode, this bench) lingers long enough for the slowest thread of
    // an interval - 100 ms: a straggler then joins the set instead of opening one of its own; a live pipeline (100 ms frames every
    // 100 ms) takes the shim's 2 ms (MI355_GROUP_LINGER_US)
    CK(mi355_agroup_set_linger(g, 100000, 0));
    const double t_grp = run_threads(S, 1, [&](int m, int) {
      size_t pos = 0, n_out = 0;
      uint64_t t = 0;
      for (;;) {
        const size_t fs = mi355_agroup_loudnorm_frame_size(g, m);
        if (total - pos < fs) break;
        CK(mi355_agroup_submit_loudnorm(g, m, x[m].data() + pos * ch, fs, out[m].data(), out[m].size() / ch, 0, 0, &t));
        CK(mi355_agroup_wait(g, t, &n_out));
        pos += fs;
      }
      CK(mi355_agroup_submit_loudnorm(g, m, x[m].data() + pos * ch, total - pos, out[m].data(), out[m].size() / ch, 1, 0, &t));
      CK(mi355_agroup_wait(g, t, &n_out));
    });
    std::printf("{\"element\": \"audioloudnorm\", \"instances\": %d, \"stream\": \"8 s of 192 kHz stereo f64 per instance, host buffers, one native thread per instance\", "
                "\"own_context_wall_s\": %.3f, \"agroup_wall_s\": %.3f, \"own_context_realtime_aggregate\": %.1f, \"agroup_realtime_aggregate\": %.1f}\n",
                S, t_own, t_grp, S * 8.0 / t_own, S * 8.0 / t_grp);
    std::fflush(stdout);
    mi355_agroup_destroy(g);
  }
  // ------------------------------------------------------------ videocompare: N two-pad elements on N native threads, 4K RGBA, device-resident frames
  if (do_video) {
    const int W = 3840, H = 2160;
    const size_t fb = (size_t)W * H * 4;
    std::vector<mi355_ctx *> ctxs(S);
    std::vector<uint8_t *> fa(S), fbuf(S);
    std::vector<uint8_t> host(fb), host2(fb);
    for (size_t i = 0; i < fb; i++) { host[i] = (uint8_t)((i * 2654435761u) >> 24); if ((i & 3) == 3) host[i] = 255; }
    for (size_t i = 0; i < fb; i++) { host2[i] = (uint8_t)(host[i] ^ ((i >> 7) & 3)); if ((i & 3) == 3) host2[i] = 255; }
    for (int m = 0; m < S; m++) {
      ctxs[m] = mi355_ctx_create(0, &st);
      fa[m] = (uint8_t *)mi355_device_alloc(ctxs[m], fb);
      fbuf[m] = (uint8_t *)mi355_device_alloc(ctxs[m], fb);
      CK(mi355_memcpy_h2d(ctxs[m], fa[m], host.data(), fb));
      CK(mi355_memcpy_h2d(ctxs[m], fbuf[m], host2.data(), fb));
    }
    std::vector<double> own_v(S), grp_v(S);
    for (int algo : {MI355_HASH_BLOCKHASH, MI355_HASH_DSSIM}) {
      if (dssim_only && algo != MI355_HASH_DSSIM) continue;
      const int iters = algo == MI355_HASH_DSSIM ? 12 : 200;
      auto own = [&](int m, int) {
        if (algo == MI355_HASH_DSSIM) {
          mi355_dssim_image *img = nullptr;
          const uint8_t *one[1] = {fbuf[m]};
          CK(mi355_dssim_create_image_device(ctxs[m], fa[m], W * 4, W, H, MI355_FMT_RGBA, &img));
          CK(mi355_dssim_compare_frames_device(ctxs[m], img, one, 1, W * 4, W, H, MI355_FMT_RGBA, &own_v[m]));
          mi355_dssim_free_image(ctxs[m], img);
        } else {
          uint64_t h0 = 0, h1 = 0;
          CK(mi355_videocompare_hash_frames_device(ctxs[m], fa[m], fb, W * 4, 1, W, H, MI355_FMT_RGBA, algo, &h0));
          CK(mi355_videocompare_hash_frames_device(ctxs[m], fbuf[m], fb, W * 4, 1, W, H, MI355_FMT_RGBA, algo, &h1));
          own_v[m] = mi355_videocompare_distance(algo, h0, h1);
        }
      };
      run_threads(S, 3, own);
      const double t_own = run_threads(S, iters, own) / iters;
      mi355_group *g = mi355_group_create(0, 0, &st);
      CK(mi355_group_set_rendezvous(g, S, 5000));
      auto grp = [&](int m, int) {
        uint64_t t = 0;
        CK(mi355_group_submit_compare(g, ctxs[m], fa[m], fbuf[m], W * 4, W, H, MI355_FMT_RGBA, algo, &t));
        CK(mi355_group_wait_compare(g, t, &grp_v[m], nullptr));
      };
      run_threads(S, 3, grp);
      const double t_grp = run_threads(S, iters, grp) / iters;
      uint64_t stats[3];
      CK(mi355_group_compare_stats(g, stats));
      bool same = true;
      for (int m = 0; m < S; m++) same &= own_v[m] == grp_v[m];
      std::printf("{\"element\": \"videocompare\", \"hash_algorithm\": \"%s\", \"instances\": %d, \"frames\": \"3840x2160 RGBA, device-resident, one two-pad element per native thread\", "
                  "\"own_context_comparisons_per_s\": %.1f, \"dispatcher_comparisons_per_s\": %.1f, \"dispatcher_launch_sequences\": %llu, \"dispatcher_pairs\": %llu, "
                  "\"results_identical\": %s}\n",
                  algo == MI355_HASH_DSSIM ? "dssim" : "blockhash", S, S / t_own, S / t_grp, (unsigned long long)stats[1], (unsigned long long)stats[0], same ? "true" : "false");
      std::fflush(stdout);
      mi355_group_destroy(g);
    }
    for (int m = 0; m < S; m++) { mi355_device_free(ctxs[m], fa[m]); mi355_device_free(ctxs[m], fbuf[m]); mi355_ctx_destroy(ctxs[m]); }
  }
  return 0;
}
