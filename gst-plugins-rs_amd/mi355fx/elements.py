"""ctypes binding of the host-side element layer (gst-plugins-rs_amd/host/elements.cpp): the C++ mirror
of the reference elements' GObject/BaseTransform surface. Tests drive the hot path through it the way
gst_check::Harness drives the real elements (audio/hrtf/tests/hrtfrender.rs:58-92)."""
import ctypes as C

import numpy as np

from . import FMT
from .cube import load_host_library

FLOW_OK, FLOW_EOS, FLOW_NOT_NEGOTIATED, FLOW_ERROR = 0, -3, -4, -5
PROP_FLOAT, PROP_DOUBLE, PROP_UINT64, PROP_STRING = 0, 1, 2, 3
FMT_NAME = {v: k for k, v in FMT.items()}

_bound = False


def _lib():
    global _bound
    L = load_host_library()
    if not _bound:
        vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
        sig = {
            "mi355el_factory_make": (vp, [C.c_char_p, i, C.c_char_p, sz]),
            "mi355el_free": (None, [vp]),
            "mi355el_last_error": (C.c_char_p, [vp]),
            "mi355el_type_name": (C.c_char_p, [vp]),
            "mi355el_factory_name": (C.c_char_p, [vp]),
            "mi355el_klass": (C.c_char_p, [vp]),
            "mi355el_long_name": (C.c_char_p, [vp]),
            "mi355el_n_properties": (i, [vp]),
            "mi355el_property_name": (C.c_char_p, [vp, i]),
            "mi355el_property_info": (i, [vp, i, C.POINTER(i), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i)]),
            "mi355el_n_formats": (i, [vp, i]),
            "mi355el_format": (i, [vp, i, i]),
            "mi355el_set_double": (i, [vp, C.c_char_p, C.c_double]),
            "mi355el_get_double": (i, [vp, C.c_char_p, C.POINTER(C.c_double)]),
            "mi355el_set_u64": (i, [vp, C.c_char_p, C.c_uint64]),
            "mi355el_get_u64": (i, [vp, C.c_char_p, C.POINTER(C.c_uint64)]),
            "mi355el_set_string": (i, [vp, C.c_char_p, C.c_char_p]),
            "mi355el_start": (i, [vp]),
            "mi355el_stop": (i, [vp]),
            "mi355el_transform_frame_ip": (i, [vp, i, i, i, i, vp, sz]),
            "mi355el_transform_frame": (i, [vp, i, i, i, i, vp, sz, i, i, vp, sz]),
            "mi355el_audio_setup": (i, [vp, i, i, i]),
            "mi355el_audio_transform_ip": (i, [vp, vp, sz]),
            "mi355el_ebur128_setup": (i, [vp, i, i, i, i, C.POINTER(i)]),
            "mi355el_ebur128_push": (i, [vp, vp, C.POINTER(vp), sz, C.c_uint64]),
            "mi355el_ebur128_reset_signal": (None, [vp]),
            "mi355el_ebur128_pop_message": (i, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                C.POINTER(C.c_double), i, C.POINTER(i)]),
            "mi355el_roundedcorners_set_caps": (i, [vp, i, i, i]),
            "mi355el_roundedcorners_prepare": (i, [vp, vp, sz, C.POINTER(sz), C.POINTER(i), C.POINTER(i)]),
            "mi355el_roundedcorners_src_formats": (i, [vp]),
            "mi355el_roundedcorners_prepare_device": (i, [vp, vp, sz, sz, i]),
            "mi355el_loudnorm_set_caps": (i, [vp, i, i]),
            "mi355el_loudnorm_chain": (i, [vp, vp, sz, i, vp, sz, C.POINTER(sz)]),
            "mi355el_loudnorm_drain": (i, [vp, i, vp, sz, C.POINTER(sz)]),
            "mi355el_videocompare_aggregate": (i, [vp, i, C.POINTER(vp), i, i, i, i, C.c_uint64, C.POINTER(i), C.POINTER(C.c_double), i]),
            "mi355el_hrtf_set_hrir_raw": (i, [vp, vp, sz]),
            "mi355el_hrtf_set_objects": (i, [vp, i, C.POINTER(C.c_float), C.POINTER(i)]),
            "mi355el_hrtf_get_objects": (i, [vp, i, C.POINTER(C.c_float), C.POINTER(i)]),
            "mi355el_hrtf_set_caps": (i, [vp, i, i, C.POINTER(i)]),
            "mi355el_hrtf_transform_size": (sz, [vp, sz]),
            "mi355el_hrtf_transform": (i, [vp, vp, sz, vp, sz, C.POINTER(sz)]),
            "mi355el_hrtf_drain": (i, [vp, vp, sz, C.POINTER(sz)]),
            "mi355el_hrtf_flush_stop": (None, [vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _bound = True
    return L


class ElementError(RuntimeError):
    pass


class Element:
    """gst::ElementFactory::make(factory) on the mirror layer."""

    def __init__(self, factory, device=0):
        L = _lib()
        err = C.create_string_buffer(512)
        self.h = L.mi355el_factory_make(factory.encode(), device, err, 512)
        if not self.h:
            raise ElementError(err.value.decode("utf-8", "replace"))
        self.L = L

    def close(self):
        if getattr(self, "h", None):
            self.L.mi355el_free(self.h)
            self.h = None

    __del__ = close

    # ---- introspection (gst-inspect surface)
    @property
    def type_name(self):
        return self.L.mi355el_type_name(self.h).decode()

    @property
    def klass(self):
        return self.L.mi355el_klass(self.h).decode()

    @property
    def last_error(self):
        return self.L.mi355el_last_error(self.h).decode("utf-8", "replace")

    def properties(self):
        out = {}
        for k in range(self.L.mi355el_n_properties(self.h)):
            t, m = C.c_int(), C.c_int()
            d, lo, hi = C.c_double(), C.c_double(), C.c_double()
            self.L.mi355el_property_info(self.h, k, C.byref(t), C.byref(d), C.byref(lo), C.byref(hi), C.byref(m))
            out[self.L.mi355el_property_name(self.h, k).decode()] = dict(type=t.value, default=d.value, min=lo.value, max=hi.value,
                                                                         mutable="playing" if m.value else "ready")
        return out

    def formats(self, src=False):
        return [FMT_NAME.get(self.L.mi355el_format(self.h, int(src), k)) for k in range(self.L.mi355el_n_formats(self.h, int(src)))]

    # ---- g_object_set / get
    def set_property(self, name, value):
        n = name.encode()
        if isinstance(value, bool):
            value = 1.0 if value else 0.0
        if isinstance(value, str):
            rc = self.L.mi355el_set_string(self.h, n, value.encode())
        elif isinstance(value, int) and not isinstance(value, bool) and self.properties().get(name, {}).get("type") == PROP_UINT64:
            rc = self.L.mi355el_set_u64(self.h, n, value)
        else:
            rc = self.L.mi355el_set_double(self.h, n, float(value))
        return rc == 0

    def get_property(self, name):
        n = name.encode()
        if self.properties()[name]["type"] == PROP_UINT64:
            v = C.c_uint64()
            assert self.L.mi355el_get_u64(self.h, n, C.byref(v)) == 0
            return v.value
        v = C.c_double()
        assert self.L.mi355el_get_double(self.h, n, C.byref(v)) == 0
        return v.value

    # ---- vfuncs
    def start(self):
        return self.L.mi355el_start(self.h) == 0

    def stop(self):
        return self.L.mi355el_stop(self.h) == 0

    def transform_frame_ip(self, fmt, width, height, stride, data):
        return self.L.mi355el_transform_frame_ip(self.h, FMT[fmt], width, height, stride, data.ctypes.data, data.nbytes)

    def transform_frame(self, in_fmt, width, height, in_stride, src, out_fmt, out_stride, dst):
        return self.L.mi355el_transform_frame(self.h, FMT[in_fmt], width, height, in_stride, src.ctypes.data, src.nbytes,
                                              FMT[out_fmt], out_stride, dst.ctypes.data, dst.nbytes)

    def audio_setup(self, rate, channels, f64=False):
        return self.L.mi355el_audio_setup(self.h, rate, channels, int(f64)) == 0

    def audio_transform_ip(self, data):
        assert isinstance(data, np.ndarray) and data.flags.c_contiguous
        return self.L.mi355el_audio_transform_ip(self.h, data.ctypes.data, data.nbytes)

    # ---- ebur128level
    _EB_FMT = {np.dtype(np.int16): 0, np.dtype(np.int32): 1, np.dtype(np.float32): 2, np.dtype(np.float64): 3}

    def ebur128_setup(self, rate, channels, dtype, planar=False, channel_class=None):
        cc = (C.c_int * channels)(*channel_class) if channel_class is not None else None
        self._eb_channels = channels
        return self.L.mi355el_ebur128_setup(self.h, rate, channels, self._EB_FMT[np.dtype(dtype)], int(planar), cc) == 0

    def ebur128_push(self, data, pts_ns, planar=False):
        a = np.ascontiguousarray(data)
        if planar:
            ptrs = (C.c_void_p * a.shape[0])(*[a[c].ctypes.data for c in range(a.shape[0])])
            return self.L.mi355el_ebur128_push(self.h, None, ptrs, a.shape[1], pts_ns)
        ch = self._eb_channels
        return self.L.mi355el_ebur128_push(self.h, a.ctypes.data, None, a.size // ch, pts_ns)

    def ebur128_reset_signal(self):
        self.L.mi355el_ebur128_reset_signal(self.h)

    def ebur128_pop_messages(self):
        out = []
        while True:
            ts, fields, nch = C.c_uint64(), C.c_uint(), C.c_int()
            sc = (C.c_double * 5)()
            sp, tp = (C.c_double * 64)(), (C.c_double * 64)()
            if not self.L.mi355el_ebur128_pop_message(self.h, C.byref(ts), C.byref(fields), sc, sp, tp, 64, C.byref(nch)):
                return out
            m = {"timestamp": ts.value}
            f = fields.value
            if f & 1:
                m["momentary-loudness"] = sc[0]
            if f & 2:
                m["shortterm-loudness"] = sc[1]
            if f & 4:
                m["global-loudness"], m["relative-threshold"] = sc[2], sc[3]
            if f & 8:
                m["loudness-range"] = sc[4]
            if f & 16:
                m["sample-peak"] = [sp[c] for c in range(nch.value)]
            if f & 32:
                m["true-peak"] = [tp[c] for c in range(nch.value)]
            out.append(m)


    # ---- hrtfrender
    COORD = {"cartesian": 0, "left-handed": 1, "right-handed": 2}

    def hrtf_set_hrir_raw(self, data):
        b = bytes(data)
        buf = (C.c_uint8 * len(b)).from_buffer_copy(b)
        return self.L.mi355el_hrtf_set_hrir_raw(self.h, C.cast(buf, C.c_void_p), len(b)) == 0

    def hrtf_set_spatial_objects(self, objs):
        """objs: list of dicts {x, y, z, distance-gain (default 1.0), coordinate-system (default left-handed)}"""
        n = len(objs)
        arr = (C.c_float * (4 * max(n, 1)))()
        cs = (C.c_int * max(n, 1))()
        for k, o in enumerate(objs):
            arr[4 * k], arr[4 * k + 1], arr[4 * k + 2] = o["x"], o["y"], o["z"]
            arr[4 * k + 3] = o.get("distance-gain", 1.0)
            cs[k] = self.COORD[o.get("coordinate-system", "left-handed")]
        return self.L.mi355el_hrtf_set_objects(self.h, n, arr, cs) == 0

    def hrtf_spatial_objects(self):
        arr = (C.c_float * 256)()
        cs = (C.c_int * 64)()
        n = self.L.mi355el_hrtf_get_objects(self.h, 64, arr, cs)
        names = {v: k for k, v in self.COORD.items()}
        return [{"x": arr[4 * k], "y": arr[4 * k + 1], "z": arr[4 * k + 2], "distance-gain": arr[4 * k + 3],
                 "coordinate-system": names[cs[k]]} for k in range(n)]

    def hrtf_set_caps(self, rate, channels, positions=None):
        p = (C.c_int * channels)(*positions) if positions is not None else None
        return self.L.mi355el_hrtf_set_caps(self.h, rate, channels, p) == 0

    def hrtf_transform_size(self, in_bytes):
        return self.L.mi355el_hrtf_transform_size(self.h, in_bytes)

    def hrtf_transform(self, data):
        """Push one input buffer (interleaved f32); returns (flow, stereo f32 output of every block completed)."""
        a = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)
        cap = self.hrtf_transform_size(a.nbytes)
        out = np.zeros(max(cap // 4, 1), np.float32)
        nb = C.c_size_t(0)
        flow = self.L.mi355el_hrtf_transform(self.h, a.ctypes.data, a.nbytes, out.ctypes.data, cap, C.byref(nb))
        return flow, out[: nb.value // 4]

    def hrtf_drain(self, max_frames=1 << 20):
        out = np.zeros(max_frames * 2, np.float32)
        nb = C.c_size_t(0)
        flow = self.L.mi355el_hrtf_drain(self.h, out.ctypes.data, out.nbytes, C.byref(nb))
        return flow, out[: nb.value // 4]

    def hrtf_flush_stop(self):
        self.L.mi355el_hrtf_flush_stop(self.h)

    # ---- videocompare
    def videocompare_aggregate(self, frames, fmt, width, height, stride, running_time=0):
        """frames[0] = reference pad; returns (flow, posted, [distance of sink_1, ...])."""
        arrs = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
        ptrs = (C.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
        posted = C.c_int(0)
        dist = (C.c_double * 64)()
        flow = self.L.mi355el_videocompare_aggregate(self.h, len(arrs), ptrs, FMT[fmt], width, height, stride, running_time,
                                                     C.byref(posted), dist, 64)
        return flow, bool(posted.value), [dist[k] for k in range(max(len(arrs) - 1, 0))] if posted.value else []

    # ---- audioloudnorm
    def loudnorm_set_caps(self, rate, channels):
        self._ln_ch = channels
        return self.L.mi355el_loudnorm_set_caps(self.h, rate, channels) == 0

    def loudnorm_chain(self, data):
        a = np.ascontiguousarray(data, dtype=np.float64).reshape(-1)
        ch = getattr(self, "_ln_ch", 1)
        frames = a.size // ch
        cap = (frames // 19200 + 32) * 19200
        out = np.zeros(cap * ch, np.float64)
        n = C.c_size_t(0)
        flow = self.L.mi355el_loudnorm_chain(self.h, a.ctypes.data, frames, ch, out.ctypes.data, cap, C.byref(n))
        return flow, out[: n.value * ch]

    def loudnorm_drain(self):
        ch = getattr(self, "_ln_ch", 1)
        cap = 31 * 19200 + 3 * 192000
        out = np.zeros(cap * ch, np.float64)
        n = C.c_size_t(0)
        flow = self.L.mi355el_loudnorm_drain(self.h, ch, out.ctypes.data, cap, C.byref(n))
        return flow, out[: n.value * ch]

    # ---- roundedcorners (host only: the mask is rendered with the system libcairo exactly as the reference does)
    def roundedcorners_set_caps(self, width, height, a420=True):
        self._rc_dims = (width, height)
        return self.L.mi355el_roundedcorners_set_caps(self.h, width, height, int(a420)) == 0

    def roundedcorners_src_formats(self):
        m = self.L.mi355el_roundedcorners_src_formats(self.h)
        return [f for f, bit in (("I420", 1), ("A420", 2)) if m & bit]

    def roundedcorners_prepare_device(self, d_frames, frame_pitch, alpha_offset, n_frames):
        """prepare_output_buffer for a device-resident A420 batch: returns the flow value."""
        return self.L.mi355el_roundedcorners_prepare_device(self.h, d_frames, frame_pitch, alpha_offset, n_frames)

    def roundedcorners_prepare(self):
        """-> (flow, passthrough, alpha plane as (rows, stride) uint8 array or None)"""
        w, h = self._rc_dims
        cap = ((w + 3) & ~3) * ((h + 1) & ~1)
        out = np.zeros(cap, np.uint8)
        n, st, pt = C.c_size_t(0), C.c_int(0), C.c_int(0)
        flow = self.L.mi355el_roundedcorners_prepare(self.h, out.ctypes.data, cap, C.byref(n), C.byref(st), C.byref(pt))
        if flow != 0 or pt.value or n.value == 0:
            return flow, bool(pt.value), None
        return flow, False, out[: n.value].reshape(-1, st.value)
