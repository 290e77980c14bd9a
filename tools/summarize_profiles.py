#!/usr/bin/env python3
"""tools/summarize_profiles.py <tag> — condense gpurun_out/profiles_<tag>/ into tracked files under profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim) of `bench.py --no-cpu-baseline --no-extra`
  profiles/<tag>_kernel_stats_full.csv   the same for the plain `bench.py` command (all legs)
  profiles/<tag>_pmc.json           per-kernel FETCH_SIZE / WRITE_SIZE averages and corrected HBM bytes/launch
  profiles/<tag>_bench.json         the bench.py line of the same run
  profiles/pmc_latest.json          copy of <tag>_pmc.json that bench.py reads for roofline.traffic
gfx950 corrections (MI355X_MICROARCH.md §HBM): counters are in KiB; FETCH_SIZE reports exactly half the
bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is used as is (it matches the
algorithmic store bytes of the streaming kernels to 0.0%).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def avg_counter(path, counter):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row.get("Counter_Name") == counter:
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
    full = glob.glob(os.path.join(src, "trace_full", "**", "*kernel_stats.csv"), recursive=True)
    if full:
        shutil.copy(full[0], os.path.join(dst, tag + "_kernel_stats_full.csv"))
    fetch = glob.glob(os.path.join(src, "pmc_fetch", "**", "*counter_collection.csv"), recursive=True)
    write = glob.glob(os.path.join(src, "pmc_write", "**", "*counter_collection.csv"), recursive=True)
    out = {"tag": tag, "units": "bytes per launch", "fetch_correction": "FETCH_SIZE(KiB) * 1024 * 2", "kernels": {}}
    if fetch and write:
        f, w = avg_counter(fetch[0], "FETCH_SIZE"), avg_counter(write[0], "WRITE_SIZE")
        for k in f:
            if "mi355::" not in k:
                continue
            short = k.split("mi355::")[1].split("(")[0]
            fb = f[k][0] * 1024 * 2
            wb = w.get(k, (0.0, 0))[0] * 1024
            out["kernels"][short] = {"fetch_size_kib_raw": f[k][0], "write_size_kib_raw": w.get(k, (0.0, 0))[0],
                                     "hbm_read_bytes": fb, "hbm_write_bytes": wb, "hbm_bytes": fb + wb, "launches_sampled": f[k][1]}
    # L2 (TCC) hit rate and L1 (TCP) miss traffic per kernel, where the extra passes exist
    for sub, names in (("pmc_tcc", ("TCC_HIT_sum", "TCC_MISS_sum")), ("pmc_tcp", ("TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"))):
        files = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        for name in names:
            for k, (v, _) in avg_counter(files[0], name).items():
                if "mi355::" in k:
                    short = k.split("mi355::")[1].split("(")[0]
                    out["kernels"].setdefault(short, {})[name] = v
    for rec in out["kernels"].values():
        if "TCC_HIT_sum" in rec and rec["TCC_HIT_sum"] + rec.get("TCC_MISS_sum", 0) > 0:
            rec["l2_hit_rate"] = rec["TCC_HIT_sum"] / (rec["TCC_HIT_sum"] + rec["TCC_MISS_sum"])
    b0 = os.path.join(src, "bench.json")
    try:
        cfg = json.load(open(b0))["config"]
        out["frames_per_step"] = cfg["frames_per_step"]   # the PMC passes ran the same default workload
        out["frames_per_launch"] = cfg.get("frames_per_launch", cfg["frames_per_step"])
        out["lut_variant"] = cfg.get("lut_variant", 0)
        out["content"] = cfg.get("content", "smooth")
        out["pristine_sources"] = str(cfg.get("sources", "")).startswith("pristine")
        # which code the counters were taken from: bench.py only reports them as roofline.traffic for the same fingerprint
        out["source_fingerprint"] = json.load(open(b0))["roofline"].get("source_fingerprint")
        out["collected"] = tag + ": tools/collect_profiles.sh"
    except (OSError, ValueError, KeyError):
        out["frames_per_step"] = None
    json.dump(out, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
    # only the default bench.py command's counters feed roofline.traffic (tags like r02_interp are other commands)
    if "_" not in tag:
        shutil.copy(os.path.join(dst, tag + "_pmc.json"), os.path.join(dst, "pmc_latest.json"))
    b = os.path.join(src, "bench.json")
    if os.path.exists(b) and os.path.getsize(b):
        shutil.copy(b, os.path.join(dst, tag + "_bench.json"))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
