"""CPU check of the kernel's algorithmic restatement: the NumPy lane model (tests/kernel_model.py), fed with the
selectors the library really computes, must equal the oracle for every order x format x edge geometry."""
import numpy as np

from kernel_model import alignbit, bayer2rgb_model, bayer2rgb_model_aligned, lerp_u8, perm

PATTERNS = ("bggr", "gbrg", "grbg", "rggb")
ALL_FORMATS = ("RGBx", "xRGB", "BGRx", "xBGR", "RGBA", "ARGB", "BGRA", "ABGR")


def test_primitive_models():
    a = np.array([0x00FF7F01, 0xFFFFFFFF], np.uint32)
    b = np.array([0x01FF8002, 0x00000000], np.uint32)
    assert lerp_u8(a, b).tolist() == [0x01FF8002, 0x80808080]
    assert alignbit(np.array([0xAABBCCDD], np.uint32), np.array([0x11223344], np.uint32), 3).tolist() == [0xBBCCDD11]
    assert alignbit(np.array([0xAABBCCDD], np.uint32), np.array([0x11223344], np.uint32), 1).tolist() == [0xDD112233]
    s0, s1 = np.array([0x44332211], np.uint32), np.array([0x88776655], np.uint32)
    assert perm(s0, s1, 0x0D040007).tolist() == [0xFF115544]
    assert perm(s0, s1, 0x03020100).tolist() == [0x88776655]
    assert perm(s0, s1, 0x07060504).tolist() == [0x44332211]
    assert perm(s0, s1, 0x0C0D0C0D).tolist() == [0x00FF00FF]


def test_lane_model_equals_oracle(pkg, oracle):
    rng = np.random.default_rng(12)
    for (w, h) in [(4, 3), (4, 4), (6, 4), (8, 5), (10, 7), (64, 48), (66, 50), (130, 9), (258, 6), (1024, 4)]:
        stride = (w + 3) & ~3
        src = rng.integers(0, 256, (h, stride), dtype=np.uint8)
        for pat in PATTERNS:
            for fmt in ALL_FORMATS:
                r, g, b = pkg.FORMATS[fmt]
                want = oracle.bayer2rgb(src, w, pat, r, g, b)
                got = bayer2rgb_model(pkg, src, w, pat, fmt)
                assert np.array_equal(got, want), (w, h, pat, fmt)


def test_aligned_lane_map_equals_oracle(pkg, oracle):
    """The per-row shifted lane map of bayer2rgb_lds_aligned_kernel (every wave-store on a 64- / 128-byte boundary,
    the columns in front of the first boundary as the row's head): every pixel written exactly once, same bytes as
    the oracle, for row pitches and base addresses that put the rows anywhere on the 8-byte grid."""
    rng = np.random.default_rng(13)
    cases = [(4, 3), (6, 4), (10, 7), (14, 5), (18, 4), (30, 6), (34, 9), (66, 5), (130, 7), (258, 4), (270, 5)]
    for (w, h) in cases:
        stride = (w + 3) & ~3
        src = rng.integers(0, 256, (h, stride), dtype=np.uint8)
        for align in (64, 128):
            for dst_stride, base in ((4 * w, 0), (4 * w, 8), (4 * w + 8, 40), (4 * w + 24, 56), (4 * w + 64, 120)):
                for pat, fmt in (("bggr", "RGBx"), ("gbrg", "xBGR"), ("grbg", "BGRx"), ("rggb", "xRGB")):
                    r, g, b = pkg.FORMATS[fmt]
                    want = oracle.bayer2rgb(src, w, pat, r, g, b)
                    got = bayer2rgb_model_aligned(pkg, src, w, pat, fmt, dst_stride, base, align)
                    assert np.array_equal(got, want), (w, h, align, dst_stride, base, pat, fmt)


def test_lane_model_on_golden_fixtures(pkg, golden):
    for name in [k for k in golden.files if k.startswith("in_")]:
        dims = name[3:]
        w, h = (int(v) for v in dims.split("x"))
        for pat in PATTERNS:
            for lay in ("RGBx", "BGRx", "xRGB", "xBGR"):
                got = bayer2rgb_model(pkg, golden[name], w, pat, lay)
                assert np.array_equal(got, golden["out_%s_%s_%s" % (dims, pat, lay)]), (dims, pat, lay)


def test_selectors_are_rejected_for_bad_cfg(pkg):
    import ctypes
    sel = (ctypes.c_uint32 * 4)()
    swap = ctypes.c_int()
    f = pkg.lib().mibayer_plan_selectors
    assert f(ctypes.byref(pkg.make_cfg(64, 48, 0, (0, 2, 1))), ctypes.byref(sel), ctypes.byref(swap)) == pkg.ERR_LAYOUT
    assert f(ctypes.byref(pkg.make_cfg(63, 48, 0, (0, 1, 2))), ctypes.byref(sel), ctypes.byref(swap)) == pkg.ERR_GEOMETRY
    assert f(ctypes.byref(pkg.make_cfg(64, 48, 0, (1, 2, 3), flags=pkg.FLAG_RGB2BAYER)), ctypes.byref(sel),
             ctypes.byref(swap)) == pkg.ERR_ARG
    assert f(ctypes.byref(pkg.make_cfg(64, 48, 3, (2, 1, 0))), ctypes.byref(sel), ctypes.byref(swap)) == pkg.OK
    assert swap.value == 0 and all(0 < v for v in sel)
