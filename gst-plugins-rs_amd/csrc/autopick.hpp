// autopick.hpp - the policy of the run-time choice between an interpolating ("compute") kernel and the memoised-table
// kernel (colorlut_kernels.hip: auto_launch), free of HIP so that it can be exercised on a CPU-only box through
// mi355_selftest_autopick (tests/test_autopick.py). The mechanism around it (events on the stream, table builds) lives in
// colorlut_kernels.hip.
//
//   * learning: compute, compute, table, table - four measured launches, the first of each kind discarded (its interval
//     holds one-off costs: code upload, cold caches). A measurement is only ever POLLED (hipEventQuery), never waited for:
//     while one is in flight the launches go on, unmeasured, with the kind being learned, so on a host that enqueues far
//     ahead of the device the learning simply spans more launches;
//   * afterwards the kind with the smaller time per pixel group serves the launches (3 % hysteresis); every n-th launch
//     of it is measured (n = 8..32, about one sample per 128 Mpixel) unless the previous sample is still in flight;
//   * the kind NOT in use is tried again after probe_period launches (64, doubling up to 1024 while the answer stays the
//     same, back to 64 when it changes). A probe is TWO consecutive launches of that kind, the second one measured: what the
//     kind not in use keeps in the caches (the table's lines, the brick table) has been evicted by the launches in between,
//     and a single cold launch of the table kernel measures 0.156 ms against 0.09-0.11 ms warm - enough to lose against the
//     interpolating kernel for ever once a disturbance has flipped the choice (measured: 34 k frames/s instead of 41 k);
//   * a decisive answer (the other kind at least twice as slow) sends its next probe straight to the longest period; a sample of
//     the kind in use that differs from the previous one by more than 25 % (the content changed) brings the probe forward to now;
//   * a change of launch size by more than 2x restarts the learning;
//   * a LUT reload keeps what was learnt as the prior (lut_upload): the choice depends on the content, hardly on the LUT.
#pragma once
#include <cstddef>

namespace mi355 {

constexpr unsigned kProbeMin = 64, kProbeMax = 1024;
constexpr double kDecisive = 2.0;       // the kind not in use measured this much slower: its next probe waits kProbeMax launches
constexpr double kRegimeChange = 1.25;  // the kind in use changed by this factor between two samples: probe the other kind at once

struct AutoPolicy {
  unsigned calls = 0, since_probe = 0, probe_period = kProbeMin;
  bool table = false;                     // kind in use once both times are known
  double t_compute = 0.0, t_table = 0.0;  // ms per 16-byte pixel group, last measurement (0 = none yet)
  size_t vec = 0;                         // launch size the measurements belong to
  int pending_kind = -1;                  // measurement in flight: -1 none, 0 compute, 1 table
  size_t pending_vec = 0;
  unsigned pending_call = 0;              // `calls` when the sample in flight was recorded
  bool pending_probe = false, pending_discard = false;
  unsigned learn = 0;                     // learning phase step: 0,1 compute; 2,3 table; 4 = done
  bool probe_second = false;              // the previous launch was the warming half of a probe: this one is the measured half
  bool table_unavailable = false;         // the table could not be allocated / built: compute kernel only
  // how much faster the kind not in use must measure to take over; samples of the kind in use averaged with the previous
  // one (smooth). The
  // nested choice between the two table kernels sets 10 % + smoothing: behind hsvfilter they are within 5-10 % of each other,
  // single in-stream samples scatter by as much, and every flip costs 64 launches of probing at the short period
  double hysteresis = 0.03;
  // ... and how much faster the kind in the "compute" role must measure to take the launches BACK from the table role (< 0: the
  // same as `hysteresis`). The nested choice sets 0: the LDS-cached kernel keeps the launches only while it measures faster at all -
  // a wrong turn towards it (behind hsvfilter the two are 5-10 % apart and single samples scatter by as much) is undone by the next
  // sample instead of surviving until the gather kernel leads by the full margin.
  double hysteresis_back = -1.0;
  bool smooth = false;
  bool last_probe = false;                // the launch decided last was a learning / probe launch (auto_decide)
};

struct AutoDecision {
  int kind;        // 0 compute, 1 table
  bool measure;    // bracket this launch with events
  bool probe, discard;
};

// a sampled launch costs ~10 us more (two timestamped events): every 8th launch at most
inline unsigned auto_sample_every(size_t n_vec) {
  unsigned n = (unsigned)(((size_t)1 << 25) / (n_vec ? n_vec : 1));
  return n < 8 ? 8 : (n > 32 ? 32 : n);
}

// the measurement in flight has completed: ms <= 0 means it could not be read
inline void auto_complete(AutoPolicy &A, double ms) {
  if (A.pending_kind < 0) return;
  if (!A.pending_discard && ms > 0.0) {
    const double per_vec = ms / (double)A.pending_vec;
    // (smoothing is for the kind in use, which is sampled every few launches; the other kind's rare probe samples count as they
    //  are - averaged with a first, cold one they would take three probe periods to say what the second already said)
    double &t = A.pending_kind == 0 ? A.t_compute : A.t_table;
    const bool in_use = A.learn >= 4 && (A.pending_kind == 1) == A.table;
    const double before = t;
    const bool both_before = A.t_compute > 0.0 && A.t_table > 0.0;
    t = A.smooth && in_use && t > 0.0 ? 0.5 * (t + per_vec) : per_vec;
    bool regime = false;
    if (in_use && before > 0.0 && (per_vec > kRegimeChange * before || per_vec * kRegimeChange < before)) {
      // The kind in use has just become much slower or much faster than it was: the content (or where the pixels come from) has
      // changed, and what is known about the OTHER kind is as old as its last probe - ask it again now, not in up to 1024 launches.
      A.probe_period = kProbeMin;
      A.since_probe = kProbeMin;
      regime = true;
    }
    if (A.t_compute > 0.0 && A.t_table > 0.0) {
      // hysteresis (3 % by default): measurements of near-equal kernels must not flip the choice back and forth
      const double back = A.hysteresis_back < 0.0 ? A.hysteresis : A.hysteresis_back;
      const bool table = A.t_table < A.t_compute * (A.table ? 1.0 + back : 1.0 - A.hysteresis);
      if (table != A.table) { A.probe_period = kProbeMin; A.since_probe = 0; }
      else if ((A.pending_probe || !both_before) && !regime) {
        // the answer stays: ask again later. A probe is two launches of the SLOWER kind (on uniform noise the table kernel takes three
        // times the interpolating one: the five probes of the 64 ... 1024 ladder cost BENCH_r05's 32-launch uniform leg 10 %), so a
        // decisive answer - the other kind at least twice as slow - goes straight to the longest period; a close one climbs the ladder.
        const double mine = table ? A.t_table : A.t_compute, other = table ? A.t_compute : A.t_table;
        if (other >= kDecisive * mine) A.probe_period = kProbeMax;
        else if (A.pending_probe) A.probe_period = A.probe_period * 2 > kProbeMax ? kProbeMax : A.probe_period * 2;
      }
      A.table = table;
    }
  }
  A.pending_kind = -1;
}

// what this call runs (after the harvest step); updates the counters
inline AutoDecision auto_decide(AutoPolicy &A, size_t n_vec) {
  if (A.vec && (n_vec > 2 * A.vec || 2 * n_vec < A.vec) && A.pending_kind < 0) {
    A.t_compute = A.t_table = 0.0;
    A.learn = 0;
    A.probe_period = kProbeMin;
    A.since_probe = 0;
  }
  if (A.pending_kind < 0) A.vec = n_vec;
  AutoDecision D{0, false, false, false};
  A.last_probe = A.learn < 4 || A.probe_second || (A.since_probe + 1 >= A.probe_period && A.pending_kind < 0);
  if (A.learn < 4) {
    D.kind = A.learn < 2 ? 0 : 1;
    D.discard = (A.learn & 1) == 0;
    D.probe = true;
    if (A.pending_kind < 0) A.learn++;
  } else {
    D.kind = A.table ? 1 : 0;
    bool warming = false;
    if (A.probe_second) {  // second half of a probe: measured if nothing else is in flight (otherwise the probe is lost)
      A.probe_second = false;
      D.kind ^= 1;
      D.probe = true;
    } else if (++A.since_probe >= A.probe_period && A.pending_kind < 0) {  // first half: warms the other kind's working set
      A.since_probe = 0;
      A.probe_second = true;
      D.kind ^= 1;
      warming = true;
    }
    if (warming) { A.calls++; return D; }  // never measured: it is neither a sample of the kind in use nor a fair one of the other
  }
  D.measure = A.pending_kind < 0 && (D.probe || (A.calls % auto_sample_every(n_vec)) == 0);
  A.calls++;
  return D;
}

// the launch decided by D has been enqueued between two events
inline void auto_sampled(AutoPolicy &A, const AutoDecision &D, size_t n_vec) {
  A.pending_kind = D.kind;
  A.pending_call = A.calls;
  A.pending_vec = n_vec;
  A.pending_probe = D.probe && A.t_compute > 0.0 && A.t_table > 0.0;
  A.pending_discard = D.discard;
}

// the table cannot be had: compute kernel for good
inline void auto_give_up_table(AutoPolicy &A) {
  A.table_unavailable = true;
  A.learn = 4;
  A.table = false;
}

}  // namespace mi355
