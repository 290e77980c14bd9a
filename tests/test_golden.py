"""Golden fixtures. tests/golden/pixel_crc.json holds CRC-32s of full all-colours (2^24) frames produced by the NUMPY
restatement (oracle/np_restate.py, written from the reference source independently of the C oracle). Three things must
reproduce them: the numpy restatement itself (so it cannot drift unnoticed), the C oracle (CPU) and the HIP path (GPU) —
two independently written CPU restatements and the device agree bit for bit on every colour."""
import importlib.util
import json
import os
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_DOC = json.load(open(os.path.join(HERE, "golden", "pixel_crc.json")))
GOLDEN = _DOC["crc32"]


def _make_golden():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg


def test_golden_provenance_is_the_numpy_restatement():
    assert "np_restate" in _DOC["provenance"]["pixels"] and "2^24" in _DOC["provenance"]["pixels"]
    assert len(GOLDEN) == 8


def test_c_oracle_reproduces_committed_crcs():
    """The C oracle (the checker of every GPU parity test) against CRCs it did not produce."""
    got = {k: zlib.crc32(v) for k, v in _make_golden().crc_cases().items()}
    assert got == GOLDEN


def test_numpy_restatement_reproduces_committed_crcs():
    """The numpy restatement over all 2^24 colours, every case (about a minute), and its block-wise echo loop on config 1:
    if it drifts, this fails, not the goldens."""
    got = {k: zlib.crc32(v) for k, v in _make_golden().np_crc_cases().items()}
    assert got == GOLDEN


def test_exact_decimal_to_f32():
    """f32_from_decimal (the goldens' LUT text reader) is correctly rounded where float()-then-cast rounds twice."""
    mg = _make_golden()
    import numpy as np
    assert mg.f32_from_decimal("0.1") == np.float32(0.1) and mg.f32_from_decimal("1") == np.float32(1)
    # 1 + 2^-24 + 2^-60 lies just above the midpoint of 1 and 1 + 2^-23: f32 must round up; via double it rounds to the
    # midpoint first and then to even (down)
    tok = "1.00000005960464477626738148115280148089681454002857208251953125"
    assert mg.f32_from_decimal(tok) == np.nextafter(np.float32(1), np.float32(2))
    assert np.float32(float(tok)) == np.float32(1)


@pytest.mark.gpu
def test_gpu_reproduces_committed_crcs(ctx, synth):
    from mi355fx.cube import parse_cube
    ac = synth.allcolors()
    got = {}
    for name, st in synth.HSV_SETTINGS.items():
        buf = ac.copy().reshape(-1)
        ctx.hsvfilter_frame_ip(buf, 4096, 4096 * 4, "RGBA", st)
        got["hsvfilter_allcolors_rgba_" + name] = zlib.crc32(buf.tobytes())
    buf = ac.copy().reshape(-1)
    ctx.hsvfilter_frame_ip(buf, 4096, 4096 * 4, "xBGR", synth.HSV_SETTINGS["mixed"])
    got["hsvfilter_allcolors_xbgr_mixed"] = zlib.crc32(buf.tobytes())
    for tag, text in (("lut33", synth.cube_text_3d(33)), ("lut17_domain", synth.cube_text_3d(17, amp=0.07, domain=((-0.25, 0.0, 0.1), (1.5, 1.0, 0.9)))),
                      ("lut1d64", synth.cube_text_1d(64))):
        lut = parse_cube(text)
        ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
        dst = np.zeros_like(ac)
        ctx.colorlut_frame(ac, 4096 * 4, dst, 4096 * 4, 4096, 4096, "RGBA")
        got["colorlut_allcolors_" + tag] = zlib.crc32(dst.tobytes())
    ctx.echo_setup(96000)
    x = synth.sine_stereo_f32()
    ctx.echo_process(x, 24000, 0.6, 0.4)
    got["echo_config1_f32"] = zlib.crc32(x.tobytes())
    assert got == GOLDEN


def test_cubetool_roundtrip(tmp_path):
    """tools/cubetool.py: identity writer -> product host reader -> identity table; HRIR info on the reference fixture."""
    import subprocess, sys, os
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "cubetool.py")
    out = tmp_path / "id9.cube"
    assert subprocess.call([sys.executable, tool, "cube-identity", "9", str(out)]) == 0
    sys.path.insert(0, os.path.join(root, "gst-plugins-rs_amd"))
    from mi355fx.cube import parse_cube_file
    lut = parse_cube_file(str(out))
    assert lut.is3d and lut.size == 9
    t = lut.table.reshape(9, 9, 9, 4)
    g = (np.arange(9) / 8.0).astype(np.float32)
    assert np.allclose(t[..., 0], g[None, None, :], atol=1e-6) and np.allclose(t[..., 2], g[:, None, None], atol=1e-6)
    r = subprocess.run([sys.executable, tool, "cube-validate", str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and "3D LUT, size 9" in r.stdout
    r = subprocess.run([sys.executable, tool, "hrir-info", os.path.join(root, "tests", "golden", "test.hrir")], capture_output=True, text=True)
    assert r.returncode == 0 and "187 vertices, 370 faces" in r.stdout
    bad = tmp_path / "bad.cube"
    bad.write_text("LUT_3D_SIZE 2\n0 0 0\n")
    assert subprocess.call([sys.executable, tool, "cube-validate", str(bad)], stdout=subprocess.DEVNULL) == 1
