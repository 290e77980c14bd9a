"""CPU-side checks of the drop-in boundary: libmi355fx.so loads and exports every symbol that
include/mi355fx.h declares, and it fails loudly (no fallback) when there is no GPU."""
import os
import re

import pytest


def _declared_symbols():
    import mi355fx
    text = open(mi355fx.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = _declared_symbols()
    for must in ("mi355_ctx_create", "mi355_hsvfilter_frame_ip", "mi355_hsvfilter_frames_device", "mi355_colorlut_load",
                 "mi355_colorlut_frame", "mi355_colorlut_frames_device", "mi355_echo_setup", "mi355_echo_process_f32",
                 "mi355_hsvdetect_frame"):
        assert must in syms
    assert len(syms) >= 30


def test_library_exports_every_declared_symbol(mi355lib):
    for name in _declared_symbols():
        assert hasattr(mi355lib, name), "libmi355fx.so does not export %s" % name


def test_abi_version_and_status_strings(mi355lib):
    assert mi355lib.mi355_abi_version() == 1
    assert mi355lib.mi355_status_string(0) == b"ok"
    assert b"device" in mi355lib.mi355_status_string(-2)


def test_host_library_exports_cube_reader():
    from mi355fx.cube import load_host_library
    L = load_host_library()
    for name in ("mi355h_cube_parse", "mi355h_cube_parse_file", "mi355h_cube_free", "mi355h_cube_table", "mi355h_cube_domain"):
        assert hasattr(L, name)


def test_no_gpu_means_loud_failure_not_fallback(mi355lib):
    """On a box without a gfx950 device a context cannot be created; nothing computes on the CPU."""
    import mi355fx
    if mi355lib.mi355_device_count() > 0:
        pytest.skip("GPU present: covered by the gpu-marked tests")
    with pytest.raises(mi355fx.Mi355Error) as e:
        mi355fx.Context(0)
    assert e.value.status == mi355fx.ERR_NO_DEVICE


def test_product_package_never_imports_the_oracle():
    """The product path must not route through oracle/ (only tests, smoke() and bench's cpu_baseline may)."""
    import mi355fx
    root = mi355fx.PKG_ROOT
    offenders = []
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".c")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|liboracle|#include\s+\"[^\"]*oracle", txt, flags=re.M):
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders


def test_ctypes_signatures_match_the_header_argument_counts(mi355lib):
    """The bindings the tests and the bench go through declare as many arguments as the C prototypes (a drifted argtypes list
    would silently pass garbage)."""
    import mi355fx
    text = re.sub(r"/\*.*?\*/", "", open(mi355fx.HEADER_PATH).read(), flags=re.S)

    def count(arglist):
        depth, n, seen = 0, 0, False
        for ch in arglist:
            if ch in "([":
                depth += 1
            elif ch in ")]":
                depth -= 1
            elif ch == "," and depth == 0:
                n += 1
            if not ch.isspace():
                seen = True
        return n + 1 if seen and arglist.strip() != "void" else 0

    protos = {m.group(1): count(m.group(2)) for m in re.finditer(r"\b(mi355_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S)}
    checked, wrong = 0, {}
    for name, n in protos.items():
        fn = getattr(mi355lib, name)
        if fn.argtypes is None:
            continue
        checked += 1
        if len(fn.argtypes) != n:
            wrong[name] = (len(fn.argtypes), n)
    assert checked >= 80 and not wrong, (checked, wrong)
