"""ctypes binding of the host-side .cube reader (gst-plugins-rs_amd/host/cube_lut.cpp), the mirror of
CubeLut::parse (video/colorlut/src/parser.rs:110-281). Product code: never touches oracle/."""
import ctypes as C
import os

import numpy as np

from . import PKG_ROOT

HOST_LIB_PATH = os.path.join(PKG_ROOT, "libmi355fx_host.so")
_host = None


class CubeParseError(ValueError):
    pass


def load_host_library():
    global _host
    if _host is not None:
        return _host
    if not os.path.exists(HOST_LIB_PATH):
        raise ImportError("libmi355fx_host.so not built: run `make -C %s`" % PKG_ROOT)
    L = C.CDLL(HOST_LIB_PATH)
    L.mi355h_cube_parse.restype = C.c_void_p
    L.mi355h_cube_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.mi355h_cube_parse_file.restype = C.c_void_p
    L.mi355h_cube_parse_file.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.mi355h_cube_free.argtypes = [C.c_void_p]
    L.mi355h_cube_is3d.argtypes = [C.c_void_p]
    L.mi355h_cube_size.restype = C.c_size_t
    L.mi355h_cube_size.argtypes = [C.c_void_p]
    L.mi355h_cube_table.restype = C.POINTER(C.c_float)
    L.mi355h_cube_table.argtypes = [C.c_void_p]
    L.mi355h_cube_table_len.restype = C.c_size_t
    L.mi355h_cube_table_len.argtypes = [C.c_void_p]
    L.mi355h_cube_domain.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    _host = L
    return L


class CubeLut:
    """Parsed LUT: is3d, size, table (flat float32), domain_scale, domain_offset."""

    def __init__(self, is3d, size, table, scale, offset):
        self.is3d, self.size, self.table = is3d, size, table
        self.domain_scale, self.domain_offset = scale, offset


def _wrap(L, h, err):
    if not h:
        raise CubeParseError(err.value.decode("utf-8", "replace"))
    try:
        n = L.mi355h_cube_table_len(h)
        table = np.ctypeslib.as_array(L.mi355h_cube_table(h), shape=(n,)).copy()
        sc, of = np.zeros(3, np.float32), np.zeros(3, np.float32)
        fp = C.POINTER(C.c_float)
        L.mi355h_cube_domain(h, sc.ctypes.data_as(fp), of.ctypes.data_as(fp))
        return CubeLut(bool(L.mi355h_cube_is3d(h)), L.mi355h_cube_size(h), table, sc, of)
    finally:
        L.mi355h_cube_free(h)


def parse_cube(text):
    L = load_host_library()
    if isinstance(text, str):
        text = text.encode("utf-8")
    err = C.create_string_buffer(512)
    return _wrap(L, L.mi355h_cube_parse(text, len(text), err, 512), err)


def parse_cube_file(path):
    L = load_host_library()
    err = C.create_string_buffer(512)
    return _wrap(L, L.mi355h_cube_parse_file(os.fsencode(path), err, 512), err)
