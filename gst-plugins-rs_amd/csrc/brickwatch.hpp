// brickwatch.hpp — content watch of the interpolating colorlut path: which of the three exact kernels serves a stream.
// HIP-free (policy only), unit-tested on the CPU through mi355_selftest_brickwatch (tests/test_brickwatch.py).
//
//   level 0  brick-cache kernel, 32 sets x 2 ways per wave (16 waves per CU): fastest on locally coherent content
//   level 1  round 4: the block-shared brick cache (colorlut3d_shared_kernel: 512 sets x 2 ways for the 16 waves of a CU
//            together) where the launch is large enough for it (RGBA8, plain colorlut, "shared1" below); otherwise the
//            brick-cache kernel with 64 sets x 2 ways per wave (8 waves per CU). For content whose 256-pixel steps spread
//            over more LUT cells than a wave's 32 sets hold (edges between colour regions, noise up to +-24 levels)
//   level 2  three-pass whole-plane kernel: indifferent to content, slowest on coherent frames
//
// The brick kernels count the 256-pixel steps that needed cache fills ("miss") and those that could not be served from
// the cache even after filling ("slow"). Every kSnapEvery-th launch at one level a snapshot of the counters becomes
// available (asynchronously; the caller polls). A snapshot that is bad for its level moves the stream up a level at once;
// a better level is probed - with a single launch - again after `period` launches (64, doubling to 256 while the answer stays the same: a probe costs
// four launches on the wrong kernel, i.e. at most a few percent, and a stream whose content calms down is back on the
// faster kernel within a few hundred buffers).
#pragma once

namespace mi355 {

struct BrickWatch {
  int home = 0;            // level currently believed best
  int probing = -1;        // level whose one-launch probe is in flight (-1: none)
  unsigned retry_in = 0;   // launches left before the next lower level is probed
  unsigned period = 0;     // current probe period
  double last_miss = 0.0, last_slow = 0.0;
  int last_level = 0;
};

constexpr unsigned kWatchSnapEvery = 4;
constexpr unsigned kWatchPeriodMin = 64, kWatchPeriodMax = 256;

inline bool watch_bad(int level, double miss, double slow, bool shared1 = false) {
  // Measured on 8x4K (profiles/r02_brick_sweep.txt; miss / slow = share of 256-pixel steps). Level 0 (32 hashed sets, 16
  // waves per CU) against level 1 (64 sets, 10 waves): 6 % miss steps 0.118-0.127 ms against 0.133-0.139; 18 % 0.136-0.143
  // against 0.148-0.153; 28 % 0.165 against 0.161; 46 % with 1 % slow steps 0.209 against 0.169; 73 % 0.27 against 0.18.
  // Level 1 against the three-pass kernel's 0.23-0.27 ms: 29 % miss steps 0.226 ms (better), 35 % 0.25-0.28 (even), 38 %
  // with 1 % slow steps 0.30 (worse), 53 % with 10 % slow steps 0.44.
  if (!shared1) return level == 0 ? (slow > 0.01 || miss > 0.35) : (slow > 0.02 || miss > 0.32);
  // Level 1 = the block-shared cache (profiles/r04_noise_probe.txt; 8 x 4K per launch, bench.py's batches). Level 0 against
  // it, by level 0's miss steps: 8 % 0.132 ms against 0.149; 26 % 0.151 against 0.153; 88 % 0.342 against 0.185 (the 64-set
  // per-wave cache: 0.276). On 1 / 2 x 4K per launch it is ahead from 20 % / 40 % (0.0241 against 0.0230, 0.0452 against 0.0422).
  // It against the three-pass kernel, by its own counters: 41 % miss / 21 % slow steps 0.284 ms against 0.33 (better);
  // 40 % / 11 % 0.340 against 0.370 (better); 55 % / 41 % 0.423 against 0.359 (worse); 78 % / 55 % 0.461 against 0.378 (worse).
  return level == 0 ? (slow > 0.01 || miss > 0.30) : (slow > 0.30 || miss > 0.50);
}

// Level for the next launch (call once per launch). A probe of the next lower level is ONE launch: its counters are
// snapshotted right behind it in stream order and the stream goes straight back to `home`; the verdict is applied whenever
// the snapshot arrives. (A host that enqueues hundreds of launches ahead of the device would otherwise run the kernel
// believed slower for as long as its queue is deep.) can_probe: no other snapshot is in flight.
// min_level: the lowest level worth running for this launch (1 for launches of one or two frames where level 1 is the
// block-shared cache: 16 waves warm ONE cache there instead of one each - 0.0248 against 0.0278 ms on a clean 4K frame,
// 0.033 against 0.068 at +-8) - never probed below.
inline int watch_level(BrickWatch &W, bool can_probe, int min_level = 0) {
  if (W.home < min_level) { W.home = min_level; W.probing = -1; W.retry_in = 0; W.period = 0; }
  if (W.probing >= 0 || W.home <= min_level) return W.home;
  if (W.retry_in > 0) { W.retry_in--; return W.home; }
  if (!can_probe) return W.home;
  W.probing = W.home - 1;
  return W.probing;
}

// a snapshot taken over launches that all ran at `level` (0 or 1) has arrived; shared1: level 1 is the block-shared cache for this stream
inline void watch_snapshot(BrickWatch &W, int level, double miss, double slow, bool shared1 = false) {
  W.last_miss = miss; W.last_slow = slow; W.last_level = level;
  const bool bad = watch_bad(level, miss, slow, shared1);
  if (W.probing == level) {
    W.probing = -1;
    if (!bad) {               // the lower level is fine again: move down, forget the back-off
      W.home = level;
      W.period = 0;
      W.retry_in = level > 0 ? kWatchPeriodMin : 0;
      if (level > 0) W.period = kWatchPeriodMin;
    } else {                  // still bad: stay, back off
      W.period = W.period ? (W.period * 2 > kWatchPeriodMax ? kWatchPeriodMax : W.period * 2) : kWatchPeriodMin;
      W.retry_in = W.period;
    }
    return;
  }
  if (level != W.home) return;  // stale snapshot of a level the stream has already left
  if (bad) {
    W.home = level + 1;
    W.period = kWatchPeriodMin;
    W.retry_in = W.period;
  }
}

}  // namespace mi355
