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


def _top_level_args(arglist):
    depth, n, seen = 0, 0, False
    for ch in arglist:
        if ch in "([<":
            depth += 1
        elif ch in ")]>":
            depth -= 1
        elif ch == "," and depth == 0:
            n += 1
        if not ch.isspace():
            seen = True
    return n + 1 if seen and arglist.strip() != "void" else 0


def test_rust_binding_block_matches_the_header_argument_counts():
    """Every `pub fn mi355_*` of INTEGRATION.md's extern block takes as many arguments as the C prototype of the same name."""
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mi355fx.h")).read(), flags=re.S)
    c_protos = {m.group(1): _top_level_args(m.group(2)) for m in re.finditer(r"\b(mi355_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S)}
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rust = {m.group(1): _top_level_args(m.group(2)) for m in re.finditer(r"pub fn (mi355_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->[^;]*)?;", doc, flags=re.S)}
    assert len(rust) >= 40
    wrong = {n: (rust[n], c_protos.get(n)) for n in rust if c_protos.get(n) != rust[n]}
    assert not wrong, wrong
