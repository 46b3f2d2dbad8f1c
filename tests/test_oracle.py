"""CPU tests of the oracle itself: it must reproduce the reference before it may judge the HIP path.

Pins (see oracle/README.md): the reference's own frame-level functions and row kernels compiled here (oracle/_ref,
when present), the whole-element md5 known answers, the hand-checkable 4x4 frame, committed golden fixtures (generated
through the reference's frame functions)."""
import hashlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATTERNS = ("bggr", "gbrg", "grbg", "rggb")
LAYOUTS = ("RGBx", "BGRx", "xRGB", "xBGR")


def md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_prng_check_value(oracle):
    # SURVEY.md Appendix C check value
    row = oracle.fill_synthetic(7680, 1, 1, seed=3)[0, 0]
    assert row[:8].tolist() == [204, 225, 36, 209, 124, 122, 38, 104]
    assert np.array_equal(row, oracle.synthetic_frames(7680, 1, 1, 3)[0, 0])
    a = oracle.fill_synthetic(34, 5, 3, seed=9, first_frame=2, stride=36)
    b = oracle.synthetic_frames(34, 5, 3, 9, first_frame=2, stride=36)
    assert np.array_equal(a, b)


def test_hand_checkable_4x4(oracle):
    # SURVEY.md Appendix B.4 (oracle = compiled reference element)
    S = np.array([[10, 200, 30, 180], [90, 250, 70, 5], [50, 120, 255, 0], [33, 77, 141, 222]], np.uint8)
    bggr = [[(250, 145, 10), (250, 200, 20), (250, 135, 30), (5, 180, 30)],
            [(250, 90, 30), (250, 120, 87), (250, 70, 143), (5, 80, 143)],
            [(164, 91, 50), (164, 120, 153), (164, 113, 255), (114, 0, 255)],
            [(77, 33, 30), (77, 124, 87), (77, 141, 143), (222, 116, 143)]]
    gbrg = [[(90, 10, 200), (80, 135, 200), (70, 30, 200), (70, 18, 180)],
            [(90, 140, 160), (80, 250, 160), (70, 197, 160), (70, 5, 90)],
            [(62, 50, 120), (84, 159, 120), (106, 255, 120), (106, 185, 0)],
            [(33, 54, 160), (87, 77, 160), (141, 110, 160), (141, 222, 90)]]
    for pat, want in (("bggr", bggr), ("gbrg", gbrg)):
        for impl in ("c", "np"):
            if impl == "c":
                out = oracle.bayer2rgb(S, 4, pat, 0, 1, 2).reshape(4, 4, 4)
            else:
                out = oracle.np_oracle.bayer2rgb(S, pat, 0, 1, 2)
            assert out[..., :3].tolist() == [[list(p) for p in row] for row in want], (pat, impl)
            assert (out[..., 3] == 255).all()


def test_known_md5_answers(oracle):
    with open(os.path.join(ROOT, "tests", "golden", "known_md5.json")) as f:
        entries = json.load(f)["entries"]
    assert len(entries) >= 7
    for e in entries:
        src = oracle.fill_synthetic(e["width"], e["height"], 1, e["seed"])[0]
        assert md5(src) == e["md5_input"]
        r, g, b = oracle.LAYOUTS[e["format"]]
        out = oracle.bayer2rgb(src, e["width"], e["pattern"], r, g, b)
        assert md5(out) == e["md5_output"], e


def test_golden_fixtures_all_implementations(oracle, golden):
    names = [k for k in golden.files if k.startswith("in_")]
    assert len(names) >= 17
    for name in names:
        dims = name[3:]
        w, h = (int(v) for v in dims.split("x"))
        src = golden[name]
        for pat in PATTERNS:
            for lay in LAYOUTS:
                want = golden["out_%s_%s_%s" % (dims, pat, lay)]
                r, g, b = oracle.LAYOUTS[lay]
                got_c = oracle.bayer2rgb(src, w, pat, r, g, b)
                got_np = oracle.np_oracle.bayer2rgb(src[:, :w], pat, r, g, b).reshape(h, 4 * w)
                assert np.array_equal(got_c, want), (dims, pat, lay)
                assert np.array_equal(got_np, want), (dims, pat, lay)


def r2b_key_dims(name):
    """'in_5x7p8' -> ('5x7p8', 5, 7, 8)"""
    key = name[3:]
    dims, _, pad = key.partition("p")
    w, h = (int(v) for v in dims.split("x"))
    return key, w, h, int(pad or 0)


def test_oracle_equals_reference_frame_driver(oracle):
    """The restated frame driver (ring priming, up(0) = 1, dn(H-1) = H-4, the four edge columns, pattern symmetry)
    against the REFERENCE'S OWN gst_bayer2rgb_process (gstbayer2rgb.c:387-451) compiled here: golden size list,
    camera widths off every grid, padded source and destination rows; C and NumPy forms."""
    if not oracle.have_ref_frame():
        pytest.skip("oracle/_ref/libbayer_frame_ref.so not built (needs /root/reference at build time)")
    rng = np.random.default_rng(55)
    sizes = [(w, h) for w in (4, 6, 8, 66) for h in (3, 4, 5, 6, 7)] + [
        (64, 48), (16, 3), (258, 9), (130, 33), (1366, 11), (3838, 6), (4056, 5), (2590, 7), (640, 480), (1920, 31)]
    n = 0
    for (w, h) in sizes:
        src = rng.integers(0, 256, (h, (w + 3) & ~3), dtype=np.uint8)
        if w == 66:
            src[::2] = 255
            src[1::2] &= 1
        for pat in PATTERNS:
            for lay in sorted(oracle.LAYOUTS):
                r, g, b = oracle.LAYOUTS[lay]
                want = oracle.ref_frame_bayer2rgb(src, w, pat, r, g, b)
                assert np.array_equal(oracle.bayer2rgb(src, w, pat, r, g, b), want), (w, h, pat, lay)
                if w * h <= 70 * 70:
                    got_np = oracle.np_oracle.bayer2rgb(src[:, :w], pat, r, g, b).reshape(h, 4 * w)
                    assert np.array_equal(got_np, want), (w, h, pat, lay)
                n += 1
    assert n == len(sizes) * 4 * len(oracle.LAYOUTS)
    # padded rows on both sides: the reference honours the strides it is given (:476-477)
    src = rng.integers(0, 256, (9, 140), dtype=np.uint8)
    a = oracle.ref_frame_bayer2rgb(src, 130, "gbrg", 3, 2, 1, dst_stride=4 * 130 + 40)
    assert np.array_equal(a, oracle.bayer2rgb(src, 130, "gbrg", 3, 2, 1, dst_stride=4 * 130 + 40))
    assert (a[:, 4 * 130:] == 0xA5).all()


def test_known_md5_answers_reproduce_through_the_reference_frame_driver(oracle):
    if not oracle.have_ref_frame():
        pytest.skip("oracle/_ref/libbayer_frame_ref.so not built")
    with open(os.path.join(ROOT, "tests", "golden", "known_md5.json")) as f:
        entries = json.load(f)["entries"]
    for e in entries:
        if e["width"] > 3840:
            continue        # 8K: covered by make_golden.py itself; keep the CPU suite short
        src = oracle.fill_synthetic(e["width"], e["height"], 1, e["seed"])[0]
        r, g, b = oracle.LAYOUTS[e["format"]]
        assert md5(oracle.ref_frame_bayer2rgb(src, e["width"], e["pattern"], r, g, b)) == e["md5_output"], e


def test_rgb2bayer_oracle_equals_reference_transform(oracle):
    """oracle.rgb2bayer (C and NumPy) against the REFERENCE'S OWN gst_rgb2bayer_transform (gstrgb2bayer.c:229-278)
    compiled here: odd and even sizes, padded source rows (GstVideoMeta stride), all four orders."""
    if not oracle.have_ref_frame():
        pytest.skip("oracle/_ref/libbayer_frame_ref.so not built (needs /root/reference at build time)")
    rng = np.random.default_rng(56)
    for (w, h, pad) in [(1, 1, 0), (2, 3, 0), (3, 2, 0), (5, 7, 0), (5, 7, 8), (16, 9, 0), (66, 50, 0), (101, 33, 0),
                        (130, 21, 24), (257, 5, 0), (1030, 17, 0), (1920, 31, 0), (3840, 9, 0)]:
        src = rng.integers(0, 256, (h, 4 * w + pad), dtype=np.uint8)
        for pat in PATTERNS:
            want = oracle.ref_frame_rgb2bayer(src, w, pat)
            got = oracle.rgb2bayer(src, w, pat, 1, 2, 3)
            assert np.array_equal(got, want), (w, h, pad, pat)      # incl. the untouched padding columns
            got_np = oracle.np_oracle.rgb2bayer(src[:, :4 * w].reshape(h, w, 4), pat, 1, 2, 3)
            assert np.array_equal(got_np, want[:, :w]), (w, h, pad, pat)


def test_rgb2bayer_golden_fixtures_and_known_md5(oracle, golden_r2b):
    """Committed outputs of the reference's gst_rgb2bayer_transform: the oracle reproduces every one (runs on boxes
    without oracle/_ref too)."""
    names = [k for k in golden_r2b.files if k.startswith("in_")]
    assert len(names) >= 12
    for name in names:
        key, w, h, pad = r2b_key_dims(name)
        src = golden_r2b[name]
        assert src.shape == (h, 4 * w + pad)
        for pat in PATTERNS:
            want = golden_r2b["out_%s_%s" % (key, pat)]
            assert np.array_equal(oracle.rgb2bayer(src, w, pat, 1, 2, 3)[:, :w], want), (key, pat)
    with open(os.path.join(ROOT, "tests", "golden", "known_md5.json")) as f:
        entries = json.load(f)["rgb2bayer_entries"]
    assert len(entries) >= 6
    for e in entries:
        w, h = e["width"], e["height"]
        src = oracle.fill_synthetic(4 * w, h, 1, e["seed"])[0]
        assert md5(src) == e["md5_input"]
        assert md5(oracle.rgb2bayer(src, w, e["pattern"], 1, 2, 3)[:, :w]) == e["md5_output"], e


def test_own_rows_equal_reference_rows(oracle):
    if not oracle.have_ref_rows():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    rng = np.random.default_rng(5)
    for (w, h) in [(4, 3), (6, 4), (66, 50), (64, 48), (130, 33), (1920, 8)]:
        stride = (w + 3) & ~3
        src = rng.integers(0, 256, (h, stride), dtype=np.uint8)
        for pat in PATTERNS:
            for lay in LAYOUTS:
                r, g, b = oracle.LAYOUTS[lay]
                a = oracle.bayer2rgb(src, w, pat, r, g, b)
                b_ = oracle.bayer2rgb(src, w, pat, r, g, b, ref_rows=True)
                assert np.array_equal(a, b_), (w, h, pat, lay)


def test_simd_row_kernels_equal_reference_rows(oracle):
    """The "ORC-equivalent" SSE2 / AVX2 row kernels (oracle/bayer2rgb_simd.c, the SIMD leg of bench.py's
    cpu_baseline) against the reference's own compiled row kernels (oracle/_ref) and the scalar restatement:
    widths around every vector-loop boundary (8 / 16 pixel pairs per iteration + scalar tails)."""
    have_ref = oracle.have_ref_rows()
    rng = np.random.default_rng(15)
    sizes = [(4, 3), (6, 4), (18, 5), (20, 5), (22, 7), (34, 6), (36, 4), (38, 9), (66, 50), (130, 33),
             (256, 8), (258, 8), (1920, 6), (3840, 5)]
    for (w, h) in sizes:
        stride = (w + 3) & ~3
        src = rng.integers(0, 256, (h, stride), dtype=np.uint8)
        if w == 66:
            src[::2] = 255      # rounding stress: 255/254 pairs and zeros
            src[1::2] &= 1
        for pat in PATTERNS:
            for lay in LAYOUTS:
                r, g, b = oracle.LAYOUTS[lay]
                want = oracle.bayer2rgb(src, w, pat, r, g, b, ref_rows=have_ref)
                for isa in oracle.simd_isas():
                    got = oracle.bayer2rgb_mode(src, w, pat, r, g, b, mode=isa)
                    assert np.array_equal(got, want), (w, h, pat, lay, isa)


def test_row_bands_equal_whole_frame(oracle):
    """Band-parallel driver of the all-cores baseline: any cut of a frame into horizontal bands reproduces the
    whole-frame loop, including the last row's pairing with row H-4 when the last band is shorter than 4 rows."""
    rng = np.random.default_rng(16)
    for (w, h) in [(16, 3), (16, 4), (18, 5), (34, 9), (64, 48), (130, 33)]:
        src = rng.integers(0, 256, (h, (w + 3) & ~3), dtype=np.uint8)
        for pat in ("bggr", "grbg"):
            want = oracle.bayer2rgb(src, w, pat, 2, 1, 0)
            for mode in ["own"] + oracle.simd_isas():
                for cut in range(1, h):
                    dst = np.full((h, 4 * w), 0xA5, np.uint8)
                    oracle.bayer2rgb_mode(src, w, pat, 2, 1, 0, mode=mode, y0=0, y1=cut, dst=dst)
                    assert (dst[cut:] == 0xA5).all()
                    oracle.bayer2rgb_mode(src, w, pat, 2, 1, 0, mode=mode, y0=cut, y1=h, dst=dst)
                    assert np.array_equal(dst, want), (w, h, pat, mode, cut)
    src = oracle.fill_synthetic(64, 48, 5, seed=12)
    one = np.stack([oracle.bayer2rgb(f, 64, "gbrg", 0, 1, 2) for f in src])
    for mode in ["own"] + oracle.simd_isas():
        for nbands, nt in ((1, 3), (7, 4), (48, 16), (100, 5)):
            got = oracle.bayer2rgb_batch_bands(src, 64, "gbrg", 0, 1, 2, nbands=nbands, nthreads=nt, mode=mode)
            assert np.array_equal(got, one), (mode, nbands, nt)


def test_simd_known_md5_answers(oracle):
    """The SIMD forms reproduce the whole-element md5 known answers of the compiled reference element too."""
    with open(os.path.join(ROOT, "tests", "golden", "known_md5.json")) as f:
        entries = json.load(f)["entries"]
    for e in entries:
        if e["width"] * e["height"] > 3840 * 2160:
            continue
        src = oracle.fill_synthetic(e["width"], e["height"], 1, e["seed"])[0]
        r, g, b = oracle.LAYOUTS[e["format"]]
        for isa in oracle.simd_isas():
            assert md5(oracle.bayer2rgb_mode(src, e["width"], e["pattern"], r, g, b, mode=isa)) == e["md5_output"], (e, isa)


def test_extreme_inputs_closed_form_vs_ring(oracle):
    # rounding stress: all-0, all-255, 0/255 checkerboards, gradients
    w, h = 34, 12
    yy, xx = np.mgrid[0:h, 0:w]
    cases = [np.zeros((h, w)), np.full((h, w), 255), ((yy + xx) & 1) * 255, (xx & 1) * 255,
             (yy & 1) * 255, (xx * 7 + yy * 13) & 255, 255 - ((xx + yy) & 1)]
    for S in cases:
        S = S.astype(np.uint8)
        src = np.zeros((h, 36), np.uint8)
        src[:, :w] = S
        for pat in PATTERNS:
            a = oracle.bayer2rgb(src, w, pat, 2, 1, 0).reshape(h, w, 4)
            b = oracle.np_oracle.bayer2rgb(S, pat, 2, 1, 0)
            assert np.array_equal(a, b)


def test_bottom_row_pairs_with_row_h_minus_4(oracle):
    # the parity trap of SURVEY.md A.4(3): changing row H-4 changes the last output row,
    # changing row H-2 changes it too (via the row above), but a "mirror" implementation
    # would not depend on row H-4 at all
    rng = np.random.default_rng(1)
    S = rng.integers(0, 256, (12, 16), dtype=np.uint8)
    base = oracle.bayer2rgb(S, 16, "bggr", 0, 1, 2)
    S2 = S.copy()
    S2[12 - 4] ^= 0xFF
    mod = oracle.bayer2rgb(S2, 16, "bggr", 0, 1, 2)
    assert not np.array_equal(base[11], mod[11])


def test_padding_and_dst_stride(oracle):
    rng = np.random.default_rng(2)
    w, h = 10, 6
    src = rng.integers(0, 256, (h, 12), dtype=np.uint8)
    a = oracle.bayer2rgb(src, w, "grbg", 1, 2, 3)
    src2 = src.copy()
    src2[:, w:] ^= 0x5A          # padding bytes must not matter
    assert np.array_equal(a, oracle.bayer2rgb(src2, w, "grbg", 1, 2, 3))
    wide = oracle.bayer2rgb(src, w, "grbg", 1, 2, 3, dst_stride=4 * w + 24)
    assert np.array_equal(wide[:, :4 * w], a)
    assert (wide[:, 4 * w:] == 0xA5).all()      # row padding untouched


def test_batch_threads_match_single(oracle):
    src = oracle.fill_synthetic(64, 48, 5, seed=11)
    one = np.stack([oracle.bayer2rgb(f, 64, "rggb", 2, 1, 0) for f in src])
    for nt in (1, 2, 4):
        assert np.array_equal(oracle.bayer2rgb_batch(src, 64, "rggb", 2, 1, 0, nthreads=nt), one)


def test_domain_is_rejected(oracle):
    ok = np.zeros((4, 4), np.uint8)
    for (w, h) in [(2, 4), (5, 4), (4, 2), (4, 1)]:
        src = np.zeros((h, 8), np.uint8)
        with pytest.raises(ValueError):
            oracle.bayer2rgb(src, w, "bggr", 0, 1, 2)
    with pytest.raises(ValueError):
        oracle.bayer2rgb(ok, 4, "bggr", 0, 2, 1)    # not one of the reference's 4 layouts


def test_rgb2bayer_oracle_forms_agree_and_invert_bayer2rgb(oracle):
    """rgb2bayer (gstrgb2bayer.c:254-268) in C and NumPy agree for every 4-byte layout (the reference's ARGB and the
    three this build adds), and it is the exact left
    inverse of bayer2rgb: every output pixel of bayer2rgb carries the original sample at its own CFA site."""
    rng = np.random.default_rng(8)
    for (w, h) in [(1, 1), (3, 2), (5, 7), (16, 9), (66, 50)]:
        P = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        for pat in PATTERNS:
            for (r, g, b) in [(1, 2, 3), (0, 1, 2), (2, 1, 0), (3, 2, 1)]:
                a = oracle.rgb2bayer(P.reshape(h, 4 * w), w, pat, r, g, b)
                assert np.array_equal(a[:, :w], oracle.np_oracle.rgb2bayer(P, pat, r, g, b))
                assert (a[:, w:] == 0xA5).all()
    for (w, h) in [(4, 3), (66, 50), (130, 21)]:
        S = rng.integers(0, 256, (h, w), dtype=np.uint8)
        src = np.zeros((h, (w + 3) & ~3), np.uint8)
        src[:, :w] = S
        for pat in PATTERNS:
            for lay in LAYOUTS:
                r, g, b = oracle.LAYOUTS[lay]
                rgb = oracle.bayer2rgb(src, w, pat, r, g, b)
                back = oracle.rgb2bayer(rgb, w, pat, r, g, b)
                assert np.array_equal(back[:, :w], S), (w, h, pat, lay)


def test_rgb2bayer_oracle_matches_videotestsrc_bayer_writer(oracle, tmp_path):
    """An independent pin for the CFA-site -> channel mapping of the rgb2bayer oracle: gst-plugins-base's videotestsrc
    (the GStreamer 1.14 binary of this image, not part of the reference tree) paints every pattern as ARGB and, for
    video/x-bayer caps, writes the mosaic itself.  For the same pattern and size that mosaic must be exactly
    rgb2bayer(ARGB frame) for all four orders.  (A second, independent pin next to the reference's own transform
    function: test_rgb2bayer_oracle_equals_reference_transform.)"""
    import subprocess
    from test_gst_element import GST_LAUNCH, GST_PREFIX
    if not os.path.exists(GST_LAUNCH):
        pytest.skip("no GStreamer installation")
    env = dict(os.environ, GST_PLUGIN_SYSTEM_PATH_1_0=os.path.join(GST_PREFIX, "lib", "gstreamer-1.0"),
               GST_PLUGIN_SCANNER=os.path.join(GST_PREFIX, "libexec", "gstreamer-1.0", "gst-plugin-scanner"),
               GST_REGISTRY=str(tmp_path / "registry.bin"))
    env.pop("GST_PLUGIN_PATH_1_0", None)

    def shoot(pattern, caps, path):
        res = subprocess.run([GST_LAUNCH, "-q", "videotestsrc", "num-buffers=1", "pattern=" + pattern, "!", caps, "!",
                              "filesink", "location=" + str(path)], capture_output=True, text=True, env=env, timeout=120)
        assert res.returncode == 0, res.stderr[-1000:]
        return np.fromfile(path, np.uint8)

    for (w, h) in ((322, 241), (130, 37), (64, 48)):
        for pattern in ("smpte", "snow", "pinwheel"):
            argb = shoot(pattern, "video/x-raw,format=ARGB,width=%d,height=%d" % (w, h), tmp_path / "a.raw")
            argb = argb.reshape(h, 4 * w)
            assert len(np.unique(argb)) > 8
            for order in ("bggr", "gbrg", "grbg", "rggb"):
                mosaic = shoot(pattern, "video/x-bayer,format=%s,width=%d,height=%d" % (order, w, h),
                               tmp_path / "m.raw").reshape(h, -1)
                want = oracle.rgb2bayer(argb, w, order, 1, 2, 3)
                assert np.array_equal(mosaic[:, :w], want[:, :w]), (w, h, pattern, order)
