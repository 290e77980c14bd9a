"""Prints the figures of a bench.py JSON line that the round's targets are stated in. usage: python tools/bench_summary.py bench.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
g = lambda o, *ks: (g(o.get(ks[0], {}), *ks[1:]) if len(ks) > 1 else o.get(ks[0])) if isinstance(o, dict) else None
print("value %.0f frames/s, %.4f ms/step" % (d["value"], d["ms_per_step"]))
r = d["roofline"]
print("roofline: %s frac %.4f avg %.4f ms (rocprof %s) traffic %s on-die %s" % (r["kernel"], r["frac"], r["avg_launch_ms"], r.get("rocprof_avg_launch_ms"), r.get("traffic"), r.get("on_die_bytes_per_launch")))
k = d["kernels"]
print("kernels: hsv %.4f lut %.4f %s" % (k["hsvfilter_ms_per_launch"], k["colorlut_ms_per_launch"], k["colorlut_kernels_served"]))
for name in ("interpolating_kernel_only", "fused_chain", "hsvfilter_nontemporal_ab", "other_content", "concurrent_streams"):
    v = d.get(name)
    if v:
        print("%s: %.0f fps %s" % (name, v["frames_per_s"], {kk: v[kk] for kk in ("ms_per_launch", "colorlut_ms_per_launch", "kernel", "kernels_served", "colorlut_kernel", "colorlut_kernels_served") if kk in v}))
cs = d.get("concurrent_streams") or {}
if cs.get("fused"):
    print("concurrent_streams fused through the group: %.0f fps (%s launches for %s frames); separate launches: %s" %
          (cs["fused"]["frames_per_s"], cs["fused"].get("batched_launches"), cs["fused"].get("frames"), cs.get("frames_per_s_separate_launches")))
for kk, v in (d.get("content_sweep") or {}).items():
    a, i = v.get("auto"), v.get("interpolating")
    print("sweep %s: auto %.0f fps lut %.4f %s | interpolating %s" % (kk, a["frames_per_s"], a["colorlut_ms_per_launch"], a["colorlut_kernels_served"], ("%.0f fps lut %.4f" % (i["frames_per_s"], i["colorlut_ms_per_launch"])) if i else None))
c5 = d.get("config5")
if c5:
    print("config5: %.0f comparisons/s (4 aggregators %.0f, 2 aggregators %.0f)" % (c5["comparisons_per_s"], g(c5, "as_pads_of_four_aggregators", "comparisons_per_s") or 0, g(c5, "as_pads_of_two_aggregators", "comparisons_per_s") or 0))
cb = d.get("cpu_baseline")
if cb:
    print("cpu_baseline: %.3f %s on %d core(s)" % (cb["value"], cb["unit"], cb["cores"]))
