"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

ctypes front-end of the CPU oracle (oracle/bayer2rgb_oracle.c) plus the NumPy
closed form (oracle/bayer2rgb_np.py).  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes
import os
import subprocess

import numpy as np

from . import bayer2rgb_np as np_oracle  # noqa: F401
from .bayer2rgb_np import LAYOUTS, PATTERNS, synthetic_frames  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_bayer.so")
REF_ROWS_PATH = os.path.join(_HERE, "_ref", "libbayerorc_ref.so")
REF_FRAME_PATH = os.path.join(_HERE, "_ref", "libbayer_frame_ref.so")
_lib = None
_ref_frame = None

_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(quiet=True):
    """Compile the oracle (and oracle/_ref when /root/reference is mounted)."""
    subprocess.run(["make", "-C", _HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        frame_args = [_u8p, ctypes.c_int, _u8p, ctypes.c_int] + [ctypes.c_int] * 6
        L.oracle_bayer2rgb.argtypes = frame_args
        L.oracle_bayer2rgb_refrows.argtypes = frame_args
        L.oracle_rgb2bayer.argtypes = frame_args
        L.oracle_load_ref_rows.argtypes = [ctypes.c_char_p]
        L.oracle_bayer2rgb_batch.argtypes = [
            _u8p, ctypes.c_size_t, ctypes.c_int, _u8p, ctypes.c_size_t, ctypes.c_int,
        ] + [ctypes.c_int] * 9
        L.oracle_bayer2rgb_mode.argtypes = frame_args + [ctypes.c_int] * 3
        L.oracle_bayer2rgb_batch_bands.argtypes = [
            _u8p, ctypes.c_size_t, ctypes.c_int, _u8p, ctypes.c_size_t, ctypes.c_int,
        ] + [ctypes.c_int] * 10
        L.oracle_bayer2rgb_batch_bands_repeat.argtypes = [
            _u8p, ctypes.c_size_t, ctypes.c_int, _u8p, ctypes.c_size_t, ctypes.c_int,
        ] + [ctypes.c_int] * 11
        L.oracle_fill_synthetic.argtypes = [
            _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t,
            ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32]
        L.oracle_fill_synthetic.restype = None
        _lib = L
    return _lib


def have_ref_rows():
    return os.path.exists(REF_ROWS_PATH)


def load_ref_rows():
    if lib().oracle_load_ref_rows(REF_ROWS_PATH.encode()) != 0:
        raise RuntimeError("cannot load " + REF_ROWS_PATH)


def _p(a):
    return a.ctypes.data_as(_u8p)


def have_ref_frame():
    """oracle/_ref/libbayer_frame_ref.so: the reference's own gst_bayer2rgb_process and gst_rgb2bayer_transform,
    compiled from the lines where they lie by `make -C oracle ref_frame` (needs the GStreamer libraries at run time)."""
    return os.path.exists(REF_FRAME_PATH)


def ref_frame_lib():
    global _ref_frame
    if _ref_frame is None:
        L = ctypes.CDLL(REF_FRAME_PATH)
        L.ref_frame_bayer2rgb.argtypes = [_u8p, ctypes.c_int, _u8p, ctypes.c_int] + [ctypes.c_int] * 6
        L.ref_frame_rgb2bayer.argtypes = [_u8p, _u8p] + [ctypes.c_int] * 4
        L.ref_frame_describe.restype = ctypes.c_char_p
        _ref_frame = L
    return _ref_frame


def ref_frame_bayer2rgb(src, width, pattern, r_off, g_off, b_off, dst_stride=None):
    """The REFERENCE's frame driver (gstbayer2rgb.c:387-451) on one frame: src (H, src_stride) -> (H, dst_stride);
    bytes beyond 4*width of a row keep the 0xA5 guard fill."""
    if isinstance(pattern, str):
        pattern = PATTERNS[pattern]
    src = np.ascontiguousarray(src, dtype=np.uint8)
    H, sstride = src.shape
    dstride = 4 * width if dst_stride is None else dst_stride
    dst = np.full((H, dstride), 0xA5, np.uint8)
    rc = ref_frame_lib().ref_frame_bayer2rgb(_p(dst), dstride, _p(src), sstride, width, H, pattern,
                                             r_off, g_off, b_off)
    if rc != 0:
        raise ValueError("reference frame driver: geometry outside its defined domain (rc=%d)" % rc)
    return dst


def ref_frame_rgb2bayer(src, width, pattern):
    """The REFERENCE's gst_rgb2bayer_transform (gstrgb2bayer.c:229-278) on one ARGB frame: src (H, src_stride >= 4W)
    -> (H, ROUND_UP_4(W)) mosaic; destination bytes beyond `width` keep the 0xA5 guard fill."""
    if isinstance(pattern, str):
        pattern = PATTERNS[pattern]
    src = np.ascontiguousarray(src, dtype=np.uint8)
    H, sstride = src.shape
    dst = np.full((H, (width + 3) & ~3), 0xA5, np.uint8)
    rc = ref_frame_lib().ref_frame_rgb2bayer(_p(dst), _p(src), sstride, width, H, pattern)
    if rc != 0:
        raise ValueError("reference rgb2bayer transform failed (rc=%d)" % rc)
    return dst


def bayer2rgb(src, width, pattern, r_off, g_off, b_off, dst_stride=None, ref_rows=False):
    """src: (H, src_stride) uint8 -> (H, dst_stride) uint8 via the C oracle.

    Bytes of a destination row beyond 4*width keep the 0xA5 guard fill.
    """
    if isinstance(pattern, str):
        pattern = PATTERNS[pattern]
    src = np.ascontiguousarray(src, dtype=np.uint8)
    H, sstride = src.shape
    dstride = 4 * width if dst_stride is None else dst_stride
    dst = np.full((H, dstride), 0xA5, np.uint8)
    if ref_rows:
        load_ref_rows()
        fn = lib().oracle_bayer2rgb_refrows
    else:
        fn = lib().oracle_bayer2rgb
    rc = fn(_p(dst), dstride, _p(src), sstride, width, H, pattern, r_off, g_off, b_off)
    if rc != 0:
        raise ValueError("oracle rejected geometry/layout (rc=%d)" % rc)
    return dst


def rgb2bayer(src, width, pattern, r_off=1, g_off=2, b_off=3, dst_stride=None):
    """src: (H, src_stride) uint8 with 4 B/pixel rows -> (H, dst_stride) uint8 mosaic via the C oracle.
    Destination bytes beyond `width` keep the 0xA5 guard fill (the reference never writes them)."""
    if isinstance(pattern, str):
        pattern = PATTERNS[pattern]
    src = np.ascontiguousarray(src, dtype=np.uint8)
    H, sstride = src.shape
    dstride = ((width + 3) & ~3) if dst_stride is None else dst_stride
    dst = np.full((H, dstride), 0xA5, np.uint8)
    rc = lib().oracle_rgb2bayer(_p(dst), dstride, _p(src), sstride, width, H, pattern, r_off, g_off, b_off)
    if rc != 0:
        raise ValueError("oracle rejected geometry (rc=%d)" % rc)
    return dst


def bayer2rgb_batch(src, width, pattern, r_off, g_off, b_off, nthreads=1, ref_rows=False):
    """src: (N, H, src_stride) uint8 -> (N, H, 4*width) uint8."""
    if isinstance(pattern, str):
        pattern = PATTERNS[pattern]
    src = np.ascontiguousarray(src, dtype=np.uint8)
    N, H, sstride = src.shape
    dst = np.empty((N, H, 4 * width), np.uint8)
    if ref_rows:
        load_ref_rows()
    rc = lib().oracle_bayer2rgb_batch(
        _p(dst), H * 4 * width, 4 * width, _p(src), H * sstride, sstride,
        width, H, pattern, r_off, g_off, b_off, N, nthreads, int(ref_rows))
    if rc != 0:
        raise ValueError("oracle rejected geometry/layout (rc=%d)" % rc)
    return dst


ROWS = {"own": 0, "ref": 1, "sse2": 2, "avx2": 3}


def simd_isas():
    """Row-kernel restatements of the ORC programs this CPU can run ("sse2", and "avx2" where present)."""
    return ["sse2", "avx2"] if lib().oracle_simd_best_isa() >= 2 else ["sse2"]


def bayer2rgb_mode(src, width, pattern, r_off, g_off, b_off, mode="own", y0=0, y1=-1, dst=None):
    """One frame with the chosen row kernels ("own", "ref", "sse2", "avx2"); output rows y0..y1-1 only
    (rows outside keep the 0xA5 guard fill, or whatever `dst` held)."""
    if isinstance(pattern, str):
        pattern = PATTERNS[pattern]
    src = np.ascontiguousarray(src, dtype=np.uint8)
    H, sstride = src.shape
    if dst is None:
        dst = np.full((H, 4 * width), 0xA5, np.uint8)
    if mode == "ref":
        load_ref_rows()
    rc = lib().oracle_bayer2rgb_mode(_p(dst), dst.shape[1], _p(src), sstride, width, H, pattern,
                                     r_off, g_off, b_off, ROWS[mode], y0, y1)
    if rc != 0:
        raise ValueError("oracle rejected geometry/layout/mode (rc=%d)" % rc)
    return dst


def bayer2rgb_batch_bands(src, width, pattern, r_off, g_off, b_off, nbands=1, nthreads=1, mode="own", dst=None,
                          repeat=1):
    """src: (N, H, src_stride) uint8 -> (N, H, 4*width) uint8; N x nbands jobs over nthreads pthreads; `repeat` > 1
    converts the batch that many times inside the worker threads (timing: amortises thread creation)."""
    if isinstance(pattern, str):
        pattern = PATTERNS[pattern]
    src = np.ascontiguousarray(src, dtype=np.uint8)
    N, H, sstride = src.shape
    if dst is None:
        dst = np.empty((N, H, 4 * width), np.uint8)
    if mode == "ref":
        load_ref_rows()
    rc = lib().oracle_bayer2rgb_batch_bands_repeat(
        _p(dst), H * 4 * width, 4 * width, _p(src), H * sstride, sstride,
        width, H, pattern, r_off, g_off, b_off, N, nbands, nthreads, ROWS[mode], repeat)
    if rc != 0:
        raise ValueError("oracle rejected geometry/layout/mode (rc=%d)" % rc)
    return dst


def fill_synthetic(width, height, nframes, seed, first_frame=0, stride=None):
    stride = width if stride is None else stride
    buf = np.empty((nframes, height, stride), np.uint8)
    lib().oracle_fill_synthetic(_p(buf), width, height, stride, height * stride,
                                first_frame, nframes, seed)
    return buf
