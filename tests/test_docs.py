"""Documentation stays in step with the ABI: every entry point declared in include/mi355fx.h is named in INTEGRATION.md
(the reference-side binding document), and every profile artefact listed in profiles/README.md exists."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_entry_point_is_documented():
    header = open(os.path.join(ROOT, "include", "mi355fx.h")).read()
    names = sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", header)))
    assert len(names) > 60
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in names if n not in doc]
    assert not missing, missing


def test_profile_artefacts_listed_exist():
    readme = open(os.path.join(ROOT, "profiles", "README.md")).read()
    files = set(re.findall(r"`(r0[0-9][a-z0-9_]*\.(?:json|csv|jsonl|txt))`", readme)) | {"pmc_latest.json"}
    assert len(files) >= 6
    for f in files:
        assert os.path.exists(os.path.join(ROOT, "profiles", f)), f
