"""Shared by tests/test_oracle_ebur128.py (CPU oracle) and tests/test_gpu_ebur128.py (device): runs one case of
tests/golden/ebu_tech_334x.json (EBU Tech 3341 / 3342 minimum-requirement signals, published recipes and tolerances) through a
meter object that offers add(samples), momentary(), shortterm(), integrated(), lra(), true_peak(channel)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ebu_tech_334x.json")


def load_cases():
    doc = json.load(open(GOLDEN))
    return doc["rate"], doc["cases"]


def check_case(case, rate, meter, synth, dtype=np.float32):
    x, ch = synth.ebu_signal(case, rate=rate, dtype=dtype)
    ex = case["expect"]
    hold = [k for k in ("shortterm_constant_after_s", "momentary_constant_after_s") if k in ex]
    if hold:
        # "constant after t seconds": fed in 100 ms buffers, every reading from t on lies within the tolerance
        key = "shortterm" if "shortterm" in ex else "momentary"
        step, pos, vals = rate // 10, 0, []
        while pos < x.size // ch:
            meter.add(x[pos * ch:(pos + step) * ch])
            pos += step
            if pos / rate >= ex[hold[0]]:
                vals.append(meter.shortterm() if key == "shortterm" else meter.momentary())
        assert len(vals) >= 5
        assert max(abs(v - ex[key]) for v in vals) <= case["tolerance_lu"], (case["id"], min(vals), max(vals))
        return {key: (min(vals), max(vals))}
    meter.add(x)
    got = {}
    for key, fn in (("momentary", meter.momentary), ("shortterm", meter.shortterm), ("integrated", meter.integrated), ("lra", meter.lra)):
        if key in ex:
            got[key] = fn()
            assert abs(got[key] - ex[key]) <= case["tolerance_lu"], (case["id"], key, got[key], ex[key])
    if "true_peak_dbtp" in ex:
        got["true_peak_dbtp"] = 20.0 * np.log10(max(meter.true_peak(c) for c in range(ch)))
        lo, hi = case["tolerance_db"]
        assert lo <= got["true_peak_dbtp"] - ex["true_peak_dbtp"] <= hi, (case["id"], got["true_peak_dbtp"])
    return got
