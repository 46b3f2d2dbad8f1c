"""gst-plugins-bad_amd -- MI355X-native `bayer2rgb` (hot path of gst-plugins-bad's bayer plugin).

The product is native code:

  libmibayer.so   hand-written HIP kernels for gfx950 + the C ABI of include/mibayer.h
                  (csrc/mibayer_kernels.hip, csrc/mibayer_abi.hip)
  libgstbayer.so  GStreamer plugin `bayer`, element `bayer2rgb`: registration, pad templates and
                  caps negotiation identical to reference gst/bayer/gstbayer2rgb.c, `transform`
                  calling the C ABI (gst/gstmibayerelement.c)

This Python module is only the ctypes harness that tests/ and bench.py use to drive the C ABI
(the directory name is not an importable identifier; load it through
`__graft_entry__.load_package()`, which registers it as `gst_plugins_bad_amd`).  There is no Python
or CPU compute path here: every compute call goes to libmibayer.so and raises `MibayerError` if the
library or a GPU is missing.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.environ.get("MIBAYER_LIB_PATH") or os.path.join(HERE, "libmibayer.so")   # override: A/B of two builds
LAB_LIB_PATH = os.path.join(HERE, "libmibayer_lab.so")      # `make lab`: + experiment arms and tuning environment
PLUGIN_PATH = os.path.join(HERE, "libgstbayer.so")
HEADER_PATH = os.path.join(ROOT, "include", "mibayer.h")

# Bayer orders, numbering of reference gstbayer2rgb.c:95-101
PATTERNS = {"bggr": 0, "gbrg": 1, "grbg": 2, "rggb": 3}
# (r_off, g_off, b_off) = GST_VIDEO_INFO_COMP_OFFSET of the 8 src-template formats,
# reference gstbayer2rgb.c:134-135, :268-271
FORMATS = {
    "RGBx": (0, 1, 2), "xRGB": (1, 2, 3), "BGRx": (2, 1, 0), "xBGR": (3, 2, 1),
    "RGBA": (0, 1, 2), "ARGB": (1, 2, 3), "BGRA": (2, 1, 0), "ABGR": (3, 2, 1),
}

PLAN_DEFAULT, PLAN_MEASURED, PLAN_CACHED, PLAN_SET = 0, 1, 2, 3
FRAME_QUEUES = 4                # MIBAYER_FRAME_QUEUES
OK = 0
ERR_ARG, ERR_GEOMETRY, ERR_LAYOUT, ERR_NO_DEVICE, ERR_HIP, ERR_NOMEM, ERR_BUSY, ERR_EMPTY, ERR_TIMEOUT = (
    -1, -2, -3, -4, -5, -6, -7, -8, -9)


class MibayerError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        msg = "mibayer status %d" % status
        try:
            L = lib()
            msg = L.mibayer_strerror(status).decode()
            hip = L.mibayer_last_hip_error().decode()
            if status == ERR_HIP and hip:
                msg += " [" + hip + "]"
        except Exception:  # pragma: no cover - library itself unavailable
            pass
        super().__init__(("%s: " % where if where else "") + msg)


class Cfg(ctypes.Structure):
    """struct mibayer_cfg (include/mibayer.h)."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("width", ctypes.c_int32), ("height", ctypes.c_int32),
        ("src_stride", ctypes.c_int32), ("dst_stride", ctypes.c_int32),
        ("pattern", ctypes.c_int32),
        ("r_off", ctypes.c_int32), ("g_off", ctypes.c_int32), ("b_off", ctypes.c_int32),
        ("device", ctypes.c_int32), ("inflight", ctypes.c_int32), ("variant", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
    ]


MAX_SHARDS = 16
FLAG_HIPGRAPH = 1
FLAG_RGB2BAYER = 2
FLAG_HIPGRAPH_CHAIN = 4


class HostStats(ctypes.Structure):
    """struct mibayer_host_stats (include/mibayer.h)."""
    _fields_ = [("submits", ctypes.c_uint64), ("waits", ctypes.c_uint64), ("polls", ctypes.c_uint64),
                ("naps", ctypes.c_uint64), ("submit_cpu_ms", ctypes.c_double), ("wait_cpu_ms", ctypes.c_double),
                ("wait_wall_ms", ctypes.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PoolCfg(ctypes.Structure):
    """struct mibayer_pool_cfg (include/mibayer.h)."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("stream", Cfg), ("ndevices", ctypes.c_int32),
                ("devices", ctypes.c_int32 * MAX_SHARDS)]


_u8p = ctypes.POINTER(ctypes.c_uint8)
_vp = ctypes.c_void_p
_lib = None

# every symbol include/mibayer.h declares: name -> (restype, argtypes)
ABI = {
    "mibayer_abi_version": (ctypes.c_int, []),
    "mibayer_device_pci_bus_id": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]),
    "mibayer_is_lab_build": (ctypes.c_int, []),
    "mibayer_device_count": (ctypes.c_int, []),
    "mibayer_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "mibayer_last_hip_error": (ctypes.c_char_p, []),
    "mibayer_create": (ctypes.c_int, [ctypes.POINTER(Cfg), ctypes.POINTER(_vp)]),
    "mibayer_destroy": (None, [_vp]),
    "mibayer_get_cfg": (ctypes.c_int, [_vp, ctypes.POINTER(Cfg)]),
    "mibayer_process_host": (ctypes.c_int, [_vp, _vp, _vp]),
    "mibayer_submit": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "mibayer_wait": (ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    "mibayer_pending": (ctypes.c_int, [_vp]),
    "mibayer_set_wait_timeout": (ctypes.c_int, [_vp, ctypes.c_int]),
    "mibayer_pool_set_wait_timeout": (ctypes.c_int, [_vp, ctypes.c_int]),
    "mibayer_set_wait_spin": (ctypes.c_int, [_vp, ctypes.c_int]),
    "mibayer_pool_set_wait_spin": (ctypes.c_int, [_vp, ctypes.c_int]),
    "mibayer_get_host_stats": (ctypes.c_int, [_vp, ctypes.POINTER(HostStats)]),
    "mibayer_pool_get_host_stats": (ctypes.c_int, [_vp, ctypes.POINTER(HostStats)]),
    "mibayer_pool_reclaim": (ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    "mibayer_pool_lost": (ctypes.c_int, [_vp]),
    "mibayer_deferred_frees": (ctypes.c_int, []),
    "mibayer_wedged_contexts": (ctypes.c_int, []),
    "mibayer_pool_inject_stall": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "mibayer_pool_create": (ctypes.c_int, [ctypes.POINTER(PoolCfg), ctypes.POINTER(_vp)]),
    "mibayer_pool_destroy": (None, [_vp]),
    "mibayer_pool_capacity": (ctypes.c_int, [_vp]),
    "mibayer_pool_pending": (ctypes.c_int, [_vp]),
    "mibayer_pool_submit": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "mibayer_pool_wait": (ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    "mibayer_pool_alive": (ctypes.c_int, [_vp]),
    "mibayer_pool_take_failure": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                                 ctypes.c_char_p, ctypes.c_size_t]),
    "mibayer_pool_inject_fault": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_longlong]),
    "mibayer_process_device": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t,
                                              ctypes.c_int, _vp]),
    "mibayer_process_device_list": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_int, _vp]),
    "mibayer_ctx_stream": (_vp, [_vp]),
    "mibayer_sync": (ctypes.c_int, [_vp]),
    "mibayer_time_device": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.POINTER(ctypes.c_float)]),
    "mibayer_autotune": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t, ctypes.c_int,
                                        ctypes.c_char_p, ctypes.c_size_t]),
    "mibayer_autotune_list": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_int,
                                             ctypes.c_char_p, ctypes.c_size_t]),
    "mibayer_copy_plan": (ctypes.c_int, [_vp, _vp]),
    "mibayer_get_plan": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                        ctypes.POINTER(ctypes.c_int)]),
    "mibayer_set_plan": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mibayer_get_plan_for": (ctypes.c_int, [_vp, ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 4),
    "mibayer_set_plan_for": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mibayer_ctx_frame_queue": (_vp, [_vp, ctypes.c_int]),
    "mibayer_plan_source": (ctypes.c_int, [_vp]),
    "mibayer_plan_from_cache": (ctypes.c_int, [_vp]),
    "mibayer_plan_cache_clear": (None, []),
    "mibayer_host_alloc": (_vp, [ctypes.c_size_t]),
    "mibayer_host_free": (None, [_vp]),
    "mibayer_host_alloc_near": (_vp, [ctypes.c_int, ctypes.c_size_t]),
    "mibayer_device_numa_node": (ctypes.c_int, [ctypes.c_int]),
    "mibayer_host_numa_node": (ctypes.c_int, [_vp]),
    "mibayer_host_is_pinned": (ctypes.c_int, [_vp]),
    "mibayer_device_alloc": (_vp, [_vp, ctypes.c_size_t]),
    "mibayer_device_free": (None, [_vp, _vp]),
    "mibayer_copy_to_device": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    "mibayer_copy_from_device": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    "mibayer_dev_alloc": (_vp, [ctypes.c_int, ctypes.c_size_t]),
    "mibayer_dev_free": (None, [ctypes.c_int, _vp]),
    "mibayer_dev_upload": (ctypes.c_int, [ctypes.c_int, _vp, _vp, ctypes.c_size_t]),
    "mibayer_dev_download": (ctypes.c_int, [ctypes.c_int, _vp, _vp, ctypes.c_size_t]),
    "mibayer_dev_stream_create": (_vp, [ctypes.c_int]),
    "mibayer_dev_stream_destroy": (None, [ctypes.c_int, _vp]),
    "mibayer_dev_upload_async": (ctypes.c_int, [ctypes.c_int, _vp, _vp, ctypes.c_size_t, _vp]),
    "mibayer_dev_download_async": (ctypes.c_int, [ctypes.c_int, _vp, _vp, ctypes.c_size_t, _vp]),
    "mibayer_dev_event_query": (ctypes.c_int, [ctypes.c_int, _vp]),
    "mibayer_dev_event_create": (_vp, [ctypes.c_int]),
    "mibayer_dev_event_destroy": (None, [ctypes.c_int, _vp]),
    "mibayer_dev_event_record": (ctypes.c_int, [ctypes.c_int, _vp, _vp]),
    "mibayer_dev_event_wait": (ctypes.c_int, [ctypes.c_int, _vp]),
    "mibayer_dev_stream_wait_event": (ctypes.c_int, [ctypes.c_int, _vp, _vp]),
    "mibayer_fill_synthetic": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.c_uint32,
                                              ctypes.c_int, ctypes.c_uint32, _vp]),
    "mibayer_variant_count": (ctypes.c_int, []),
    "mibayer_variant_name": (ctypes.c_char_p, [ctypes.c_int]),
    "mibayer_auto_variant": (ctypes.c_int, [ctypes.c_int]),
    "mibayer_frame_class_variant": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mibayer_known_width_plan": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "mibayer_ctx_variant_name": (ctypes.c_char_p, [_vp]),
    "mibayer_plan_selectors": (ctypes.c_int, [ctypes.POINTER(Cfg), ctypes.POINTER(ctypes.c_uint32 * 4),
                                              ctypes.POINTER(ctypes.c_int)]),
    "mibayer_launch_geometry": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                               ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                               ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int),
                                               ctypes.POINTER(ctypes.c_int64)]),
    "mibayer_block_to_tile": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int]),
}


def device_pci_bus_id(device):
    """'0000:c1:00.0' of a HIP ordinal (None if the runtime does not say)."""
    buf = ctypes.create_string_buffer(32)
    if lib().mibayer_device_pci_bus_id(device, buf, len(buf)) != OK:
        return None
    return buf.value.decode()


def build(quiet=True):
    """Compile libmibayer.so (hipcc, gfx950), the lab build beside it and, when GStreamer headers exist, libgstbayer.so."""
    subprocess.run(["make", "-C", HERE, "all", "lab"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def lib():
    """The C ABI.  Fails loudly when the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MibayerErrorNoLib(
                "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in ABI.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.mibayer_abi_version() != 5:
            raise MibayerErrorNoLib("libmibayer.so ABI version mismatch")
        _lib = L
    return _lib


class MibayerErrorNoLib(RuntimeError):
    pass


class FrameLost(MibayerError):
    """mibayer_pool_wait handed a frame back as lost (MIBAYER_ERR_TIMEOUT with its tag): the frame was in flight on a
    device that ran into the wait deadline; its buffers stay the device's until Pool.reclaim().  The stream carries on
    as long as Pool.alive() > 0."""

    def __init__(self, tag):
        super().__init__(ERR_TIMEOUT, "mibayer_pool_wait (frame lost)")
        self.tag = tag


def device_count():
    return lib().mibayer_device_count()


def variant_names():
    L = lib()
    return [L.mibayer_variant_name(i).decode() for i in range(L.mibayer_variant_count())]


def _check(rc, where):
    if rc != OK:
        raise MibayerError(rc, where)


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def make_cfg(width, height, pattern="bggr", fmt="RGBx", src_stride=0, dst_stride=0, device=-1,
             inflight=0, variant=0, flags=0):
    r, g, b = FORMATS[fmt] if isinstance(fmt, str) else fmt
    pat = PATTERNS[pattern] if isinstance(pattern, str) else int(pattern)
    return Cfg(ctypes.sizeof(Cfg), width, height, src_stride, dst_stride, pat, r, g, b,
               device, inflight, variant, flags)


class Pool:
    """Round-robin frame sharding over HIP devices (mibayer_pool): frame g -> devices[g % N]."""

    def __init__(self, devices, width, height, pattern="bggr", fmt="RGBx", inflight=2, flags=0):
        pc = PoolCfg()
        pc.struct_size = ctypes.sizeof(PoolCfg)
        pc.stream = make_cfg(width, height, pattern, fmt, inflight=inflight, flags=flags)
        pc.ndevices = len(devices)
        for i, d in enumerate(devices):
            pc.devices[i] = d
        self._h = _vp()
        _check(lib().mibayer_pool_create(ctypes.byref(pc), ctypes.byref(self._h)), "mibayer_pool_create")
        self.capacity = lib().mibayer_pool_capacity(self._h)

    def submit(self, src, dst, tag=0):
        _check(lib().mibayer_pool_submit(self._h, _ptr(src), _ptr(dst), _vp(tag)), "mibayer_pool_submit")

    def wait(self):
        tag = _vp()
        rc = lib().mibayer_pool_wait(self._h, ctypes.byref(tag))
        if rc == ERR_TIMEOUT and tag.value:
            raise FrameLost(tag.value)          # (whether the stream carries on is alive() > 0)
        _check(rc, "mibayer_pool_wait")
        return tag.value or 0

    def reclaim(self):
        """Tags of lost frames whose device has let go of their buffers since (non-blocking)."""
        out = []
        while True:
            tag = _vp()
            if lib().mibayer_pool_reclaim(self._h, ctypes.byref(tag)) != OK:
                return out
            out.append(tag.value or 0)

    def lost(self):
        return lib().mibayer_pool_lost(self._h)

    def set_wait_spin(self, us):
        _check(lib().mibayer_pool_set_wait_spin(self._h, us), "mibayer_pool_set_wait_spin")

    def host_stats(self):
        st = HostStats()
        _check(lib().mibayer_pool_get_host_stats(self._h, ctypes.byref(st)), "mibayer_pool_get_host_stats")
        return st.as_dict()

    def pending(self):
        return lib().mibayer_pool_pending(self._h)

    def alive(self):
        return lib().mibayer_pool_alive(self._h)

    def set_wait_timeout(self, ms):
        _check(lib().mibayer_pool_set_wait_timeout(self._h, ms), "mibayer_pool_set_wait_timeout")

    def inject_stall(self, shard, ms):
        _check(lib().mibayer_pool_inject_stall(self._h, shard, ms), "mibayer_pool_inject_stall")

    def inject_fault(self, shard, after_frames):
        _check(lib().mibayer_pool_inject_fault(self._h, shard, after_frames), "mibayer_pool_inject_fault")

    def take_failure(self):
        """(devices dropped since the last call, ordinal of the latest, devices left, message)"""
        dev, alive = ctypes.c_int(-1), ctypes.c_int(0)
        buf = ctypes.create_string_buffer(300)
        n = lib().mibayer_pool_take_failure(self._h, ctypes.byref(dev), ctypes.byref(alive), buf, len(buf))
        return n, dev.value, alive.value, buf.value.decode()

    def close(self):
        if getattr(self, "_h", None):
            lib().mibayer_pool_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One negotiated stream == one mibayer_ctx (what the element creates in set_caps).

    `fmt` is a GStreamer video format name of the src template (RGBx, BGRx, ...) or an explicit
    (r_off, g_off, b_off) triple.
    """

    def __init__(self, width, height, pattern="bggr", fmt="RGBx", src_stride=0, dst_stride=0,
                 device=-1, inflight=0, variant=0, flags=0):
        cfg = make_cfg(width, height, pattern, fmt, src_stride, dst_stride, device, inflight, variant, flags)
        self._h = _vp()
        _check(lib().mibayer_create(ctypes.byref(cfg), ctypes.byref(self._h)), "mibayer_create")
        out = Cfg()
        _check(lib().mibayer_get_cfg(self._h, ctypes.byref(out)), "mibayer_get_cfg")
        self.cfg = out
        self.width, self.height = out.width, out.height
        self.src_stride, self.dst_stride = out.src_stride, out.dst_stride
        self.src_bytes = out.src_stride * out.height
        self.dst_bytes = out.dst_stride * out.height
        self.variant_name = lib().mibayer_ctx_variant_name(self._h).decode()

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            lib().mibayer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- host path --------------------------------------------------------------------------
    def process_host(self, src, dst=None):
        """src: uint8 array of src_stride*height bytes -> (height, dst_stride) uint8."""
        src = np.ascontiguousarray(src, dtype=np.uint8)
        assert src.size == self.src_bytes, (src.size, self.src_bytes)
        if dst is None:
            dst = np.full((self.height, self.dst_stride), 0xA5, np.uint8)
        assert dst.size == self.dst_bytes and dst.flags.c_contiguous
        _check(lib().mibayer_process_host(self._h, _ptr(src), _ptr(dst)), "mibayer_process_host")
        return dst

    def submit(self, src, dst, tag=0):
        _check(lib().mibayer_submit(self._h, _ptr(src), _ptr(dst), _vp(tag)), "mibayer_submit")

    def wait(self):
        tag = _vp()
        _check(lib().mibayer_wait(self._h, ctypes.byref(tag)), "mibayer_wait")
        return tag.value or 0

    def pending(self):
        return lib().mibayer_pending(self._h)

    def set_wait_timeout(self, ms):
        _check(lib().mibayer_set_wait_timeout(self._h, ms), "mibayer_set_wait_timeout")

    def set_wait_spin(self, us):
        _check(lib().mibayer_set_wait_spin(self._h, us), "mibayer_set_wait_spin")

    def host_stats(self):
        st = HostStats()
        _check(lib().mibayer_get_host_stats(self._h, ctypes.byref(st)), "mibayer_get_host_stats")
        return st.as_dict()

    def stall(self, ms):
        """Drill (csrc/mibayer_hooks.h): occupy the context's compute queue for `ms` milliseconds."""
        if not hasattr(lib(), "mibayer_internal_stall"):
            raise MibayerErrorNoLib("the stall drill is exported by the lab build only (make lab; MIBAYER_LIB_PATH)")
        fn = lib().mibayer_internal_stall
        fn.restype, fn.argtypes = ctypes.c_int, [_vp, ctypes.c_int]
        _check(fn(self._h, ms), "mibayer_internal_stall")

    # -- device path ------------------------------------------------------------------------
    def device_alloc(self, nbytes):
        p = lib().mibayer_device_alloc(self._h, nbytes)
        if not p:
            raise MibayerError(ERR_NOMEM, "mibayer_device_alloc(%d)" % nbytes)
        return p

    def device_free(self, p):
        lib().mibayer_device_free(self._h, _vp(p))

    def to_device(self, d_dst, arr):
        arr = np.ascontiguousarray(arr)
        _check(lib().mibayer_copy_to_device(self._h, _vp(d_dst), _ptr(arr), arr.nbytes),
               "mibayer_copy_to_device")

    def from_device(self, d_src, nbytes):
        out = np.empty(nbytes, np.uint8)
        _check(lib().mibayer_copy_from_device(self._h, _ptr(out), _vp(d_src), nbytes),
               "mibayer_copy_from_device")
        return out

    @property
    def stream(self):
        """The context's own compute stream (hipStream_t as an int)."""
        return lib().mibayer_ctx_stream(self._h)

    def process_device(self, d_src, d_dst, nframes=1, src_frame_bytes=None, dst_frame_bytes=None,
                       stream="ctx"):
        """stream: "ctx" = the context's compute stream, else a hipStream_t value (0/None = null stream)."""
        s = self.stream if stream == "ctx" else (stream or 0)
        _check(lib().mibayer_process_device(
            self._h, _vp(d_src), src_frame_bytes or self.src_bytes, _vp(d_dst),
            dst_frame_bytes or self.dst_bytes, nframes, _vp(s)), "mibayer_process_device")

    def process_device_list(self, d_srcs, d_dsts, stream="ctx"):
        """One launch (per 16 frames) over frames that are separate device allocations."""
        s = self.stream if stream == "ctx" else (stream or 0)
        n = len(d_srcs)
        a, b = (_vp * n)(*d_srcs), (_vp * n)(*d_dsts)
        _check(lib().mibayer_process_device_list(self._h, a, b, n, _vp(s)), "mibayer_process_device_list")

    def sync(self):
        _check(lib().mibayer_sync(self._h), "mibayer_sync")

    def time_device(self, d_src, d_dst, nframes, warmup=2, reps=10, src_frame_bytes=None,
                    dst_frame_bytes=None):
        ms = ctypes.c_float()
        _check(lib().mibayer_time_device(
            self._h, _vp(d_src), src_frame_bytes or self.src_bytes, _vp(d_dst),
            dst_frame_bytes or self.dst_bytes, nframes, warmup, reps, ctypes.byref(ms)),
            "mibayer_time_device")
        return ms.value

    def autotune(self, d_src, d_dst, nframes, src_frame_bytes=None, dst_frame_bytes=None):
        """Measured plan selection (mibayer_autotune); returns the one-line report."""
        buf = ctypes.create_string_buffer(1024)
        _check(lib().mibayer_autotune(
            self._h, _vp(d_src), src_frame_bytes or self.src_bytes, _vp(d_dst),
            dst_frame_bytes or self.dst_bytes, nframes, buf, len(buf)), "mibayer_autotune")
        self.variant_name = lib().mibayer_ctx_variant_name(self._h).decode()
        return buf.value.decode()

    def autotune_list(self, d_srcs, d_dsts):
        """mibayer_autotune_list over separately allocated frames; returns the one-line report."""
        n = len(d_srcs)
        a, b = (_vp * n)(*d_srcs), (_vp * n)(*d_dsts)
        buf = ctypes.create_string_buffer(1024)
        _check(lib().mibayer_autotune_list(self._h, a, b, n, buf, len(buf)), "mibayer_autotune_list")
        self.variant_name = lib().mibayer_ctx_variant_name(self._h).decode()
        return buf.value.decode()

    def get_plan(self):
        """(variant id, band override, store alignment)"""
        v, b, a = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(lib().mibayer_get_plan(self._h, ctypes.byref(v), ctypes.byref(b), ctypes.byref(a)), "mibayer_get_plan")
        return v.value, b.value, a.value

    def variant_name_for(self, nframes):
        """The kernel variant launches of `nframes` frames run (`variant_name` describes the batch-class plan
        only; a context that launches frame by frame -- the host path, stream mode -- runs the frame-class plan)."""
        return variant_names()[self.get_plan_for(nframes)[0]]

    def get_plan_for(self, nframes):
        """(variant id, band override, store alignment, source) of the launch class an nframes-frame launch falls into"""
        v, b, a, s = (ctypes.c_int() for _ in range(4))
        _check(lib().mibayer_get_plan_for(self._h, nframes, ctypes.byref(v), ctypes.byref(b), ctypes.byref(a),
                                          ctypes.byref(s)), "mibayer_get_plan_for")
        return v.value, b.value, a.value, s.value

    def set_plan_for(self, nframes, variant, band, align=0):
        _check(lib().mibayer_set_plan_for(self._h, nframes, variant, band, align), "mibayer_set_plan_for")

    def frame_queue(self, k):
        """Frame queue k of the context's device (hipStream_t as an int): FRAME_QUEUES compute streams on hardware
        queues of their own, for independent one-frame launches; created on first use."""
        s = lib().mibayer_ctx_frame_queue(self._h, k)
        if not s:
            raise MibayerError(ERR_HIP, "mibayer_ctx_frame_queue(%d)" % k)
        return s

    @property
    def frame_queues(self):
        return [self.frame_queue(k) for k in range(FRAME_QUEUES)]

    def set_plan(self, variant, band, align=0):
        _check(lib().mibayer_set_plan(self._h, variant, band, align), "mibayer_set_plan")
        self.variant_name = lib().mibayer_ctx_variant_name(self._h).decode()

    def plan_from_cache(self):
        hit = lib().mibayer_plan_from_cache(self._h) == 1
        self.variant_name = lib().mibayer_ctx_variant_name(self._h).decode()
        return hit

    @property
    def plan_source(self):
        return lib().mibayer_plan_source(self._h)

    def copy_plan_from(self, other):
        _check(lib().mibayer_copy_plan(self._h, other._h), "mibayer_copy_plan")
        self.variant_name = lib().mibayer_ctx_variant_name(self._h).decode()

    def fill_synthetic(self, d_src, nframes, seed, first_frame=0, src_frame_bytes=None, stream="ctx"):
        s = self.stream if stream == "ctx" else (stream or 0)
        _check(lib().mibayer_fill_synthetic(
            self._h, _vp(d_src), src_frame_bytes or self.src_bytes, first_frame, nframes, seed,
            _vp(s)), "mibayer_fill_synthetic")

    def launch_geometry(self, nframes=1):
        tw, th, tx, band = (ctypes.c_int() for _ in range(4))
        rows, grid = ctypes.c_int64(), ctypes.c_int64()
        _check(lib().mibayer_launch_geometry(self._h, nframes, ctypes.byref(tw), ctypes.byref(th),
                                             ctypes.byref(tx), ctypes.byref(rows), ctypes.byref(band),
                                             ctypes.byref(grid)), "mibayer_launch_geometry")
        return {"tile_w": tw.value, "tile_h": th.value, "tiles_x": tx.value, "tile_rows": rows.value,
                "band": band.value, "grid_blocks": grid.value}

    # -- convenience for tests ---------------------------------------------------------------
    def process_batch_via_device(self, frames):
        """frames: (N, height, src_stride) uint8 on the host -> (N, height, dst_stride) uint8,
        through ONE device-resident batch launch (mibayer_process_device)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n = frames.shape[0]
        assert frames[0].size == self.src_bytes
        d_src = self.device_alloc(n * self.src_bytes)
        d_dst = self.device_alloc(n * self.dst_bytes)
        try:
            self.to_device(d_src, frames)
            self.process_device(d_src, d_dst, n)
            self.sync()
            out = self.from_device(d_dst, n * self.dst_bytes)
        finally:
            self.device_free(d_src)
            self.device_free(d_dst)
        return out.reshape(n, self.height, self.dst_stride)
