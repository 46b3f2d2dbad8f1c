"""GPU tests of the inverse direction (MIBAYER_FLAG_RGB2BAYER; reference gst/bayer/gstrgb2bayer.c:230-278):
bit-exact against the committed outputs of the reference's own gst_rgb2bayer_transform (tests/golden/
rgb2bayer_small.npz, the rgb2bayer md5 table), against the oracle pinned to it, exact left inverse of bayer2rgb at full
size, and through the element."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from test_gst_element import GST_LAUNCH, gst_env, needs_gst, plugin  # noqa: F401

pytestmark = pytest.mark.gpu
PATTERNS = ("bggr", "gbrg", "grbg", "rggb")


def test_rgb2bayer_golden_fixtures_of_the_reference_transform(gpu_pkg, golden_r2b):
    """HIP path vs the committed outputs of the reference's gst_rgb2bayer_transform (no oracle in between): every
    fixture (odd sizes, padded source rows) x 4 orders, host path and device-resident path."""
    names = [k for k in golden_r2b.files if k.startswith("in_")]
    assert len(names) >= 12
    for name in names:
        key = name[3:]
        dims, _, pad = key.partition("p")
        w, h = (int(v) for v in dims.split("x"))
        pad = int(pad or 0)
        src = golden_r2b[name]
        for pat in PATTERNS:
            want = golden_r2b["out_%s_%s" % (key, pat)]
            with gpu_pkg.Context(w, h, pat, (1, 2, 3), src_stride=4 * w + pad, flags=gpu_pkg.FLAG_RGB2BAYER) as ctx:
                got_h = ctx.process_host(src)
                got_d = ctx.process_batch_via_device(src[None])[0]
            assert np.array_equal(got_h[:, :w], want), (key, pat)
            assert np.array_equal(got_d[:, :w], want), (key, pat)


def test_rgb2bayer_known_md5_answers_full_size(gpu_pkg):
    """1080p / 4K / an odd 3841x2161 frame: md5 of the HIP mosaic == md5 of the reference transform's mosaic
    (tests/golden/known_md5.json: rgb2bayer_entries; inputs regenerated on the device's host side by the Appendix C
    generator, so no oracle arithmetic takes part)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "known_md5.json")) as f:
        entries = json.load(f)["rgb2bayer_entries"]
    assert len(entries) >= 6
    for e in entries:
        w, h = e["width"], e["height"]
        # the ARGB frame is the generator's 4W x H byte image
        with gpu_pkg.Context(4 * w, h, "bggr", "RGBx") as gen:
            d = gen.device_alloc(4 * w * h)
            gen.fill_synthetic(d, 1, seed=e["seed"])
            gen.sync()
            src = gen.from_device(d, 4 * w * h).reshape(h, 4 * w)
            gen.device_free(d)
        assert hashlib.md5(src.tobytes()).hexdigest() == e["md5_input"]
        with gpu_pkg.Context(w, h, e["pattern"], (1, 2, 3), flags=gpu_pkg.FLAG_RGB2BAYER) as ctx:
            got = ctx.process_batch_via_device(src[None])[0]
        assert hashlib.md5(np.ascontiguousarray(got[:, :w]).tobytes()).hexdigest() == e["md5_output"], e


def test_rgb2bayer_equals_the_reference_transform_directly(gpu_pkg, oracle):
    """HIP mosaic against the REFERENCE'S OWN gst_rgb2bayer_transform (oracle/_ref/libbayer_frame_ref.so on this box),
    no restatement in between: 4K, 1080p, odd and padded geometries."""
    if not oracle.have_ref_frame():
        pytest.skip("oracle/_ref/libbayer_frame_ref.so not on this box")
    rng = np.random.default_rng(78)
    for (w, h, pad, pat) in [(3840, 2160, 0, "rggb"), (1920, 1080, 0, "gbrg"), (3841, 2161, 0, "grbg"),
                             (1367, 769, 40, "bggr"), (5, 7, 8, "gbrg"), (1, 1, 0, "rggb")]:
        src = rng.integers(0, 256, (h, 4 * w + pad), dtype=np.uint8)
        want = oracle.ref_frame_rgb2bayer(src, w, pat)
        with gpu_pkg.Context(w, h, pat, (1, 2, 3), src_stride=4 * w + pad, flags=gpu_pkg.FLAG_RGB2BAYER) as ctx:
            got = ctx.process_batch_via_device(src[None])[0]
        assert np.array_equal(got[:, :w], want[:, :w]), (w, h, pad, pat)


def test_rgb2bayer_matches_oracle_all_sizes(gpu_pkg, oracle):
    rng = np.random.default_rng(9)
    for (w, h) in [(1, 1), (2, 3), (3, 2), (5, 7), (16, 9), (66, 50), (257, 5), (1024, 9), (1030, 17), (1920, 31)]:
        src = rng.integers(0, 256, (h, 4 * w), dtype=np.uint8)
        for pat in PATTERNS:
            for (r, g, b) in [(1, 2, 3), (0, 1, 2), (2, 1, 0), (3, 2, 1)]:
                want = oracle.rgb2bayer(src, w, pat, r, g, b)
                with gpu_pkg.Context(w, h, pat, (r, g, b), flags=gpu_pkg.FLAG_RGB2BAYER) as ctx:
                    assert ctx.src_stride == 4 * w and ctx.dst_stride == (w + 3) & ~3
                    got_h = ctx.process_host(src)
                    got_d = ctx.process_batch_via_device(src[None])[0]
                for got in (got_h, got_d):
                    assert np.array_equal(got[:, :w], want[:, :w]), (w, h, pat, (r, g, b))
                assert (got_d[:, w:] == 0).all()         # padding columns are written as zero on the device


@pytest.mark.parametrize("shape", ["0:2", "0:8", "1:4", "2:8", "4:4", "4:8", "8:8"])
@pytest.mark.parametrize("ldnt", ["0", "1"])
def test_rgb2bayer_every_launch_shape_is_bit_exact(gpu_lab_pkg, oracle, shape, ldnt, monkeypatch):
    """(Lab build: the launch shapes are MIBAYER_R2B_* knobs; the product build carries the measured one.)
    The tile kernel (flat = 0, rows per block) and every flat-kernel shape (groups per thread : pixels per group,
    with and without the nt hint on the loads), in both block orders, on sizes that exercise row-straddling waves,
    odd dword counts per row (8-pixel groups fall back to 4), padded strides, partial last dwords and batches."""
    gpu_pkg = gpu_lab_pkg
    flat, px = shape.split(":")
    monkeypatch.setenv("MIBAYER_R2B_FLAT", flat)
    monkeypatch.setenv("MIBAYER_R2B_ROWS" if flat == "0" else "MIBAYER_R2B_PX", px)
    monkeypatch.setenv("MIBAYER_R2B_LDNT", ldnt)
    rng = np.random.default_rng(19)
    for band in ("-1", "0"):
        monkeypatch.setenv("MIBAYER_XCD_BAND", band)
        for (w, h, n, pad) in [(1, 1, 1, 0), (5, 7, 2, 0), (12, 5, 3, 0), (66, 50, 2, 0), (130, 21, 5, 24),
                               (1000, 33, 2, 0), (1024, 64, 3, 0), (2056, 19, 2, 16), (3840, 37, 2, 0)]:
            src = rng.integers(0, 256, (n, h, 4 * w + pad), dtype=np.uint8)
            with gpu_pkg.Context(w, h, "gbrg", (1, 2, 3), src_stride=4 * w + pad,
                                 flags=gpu_pkg.FLAG_RGB2BAYER) as ctx:
                got = ctx.process_batch_via_device(src)
                one = ctx.process_host(src[0])
            for f in range(n):
                want = oracle.rgb2bayer(src[f], w, "gbrg", 1, 2, 3)
                assert np.array_equal(got[f][:, :w], want[:, :w]), (shape, ldnt, band, w, h, f)
                assert (got[f][:, w:] == 0).all()
            assert np.array_equal(one[:, :w], oracle.rgb2bayer(src[0], w, "gbrg", 1, 2, 3)[:, :w])


@pytest.mark.parametrize("bands", ["", "1", "3", "8"], ids=["default", "1", "3", "8"])
def test_rgb2bayer_banded_synchronous_host_path(gpu_lab_pkg, oracle, bands, monkeypatch):
    """Frames of 16 MB input and more go through the synchronous host path in horizontal bands (16-row units), like
    bayer2rgb's: same bytes for every band count, odd heights, padded source rows; queued use stays unbanded."""
    gpu_pkg = gpu_lab_pkg         # MIBAYER_HOST_BANDS is a knob of the lab build
    if bands:
        monkeypatch.setenv("MIBAYER_HOST_BANDS", bands)
    else:
        monkeypatch.delenv("MIBAYER_HOST_BANDS", raising=False)
    rng = np.random.default_rng(41)
    for (w, h, pat, pad) in ((3840, 2160, "rggb", 0), (2050, 2053, "gbrg", 0), (2560, 1667, "grbg", 64)):
        src = rng.integers(0, 256, (h, 4 * w + pad), dtype=np.uint8)
        want = oracle.rgb2bayer(src, w, pat, 1, 2, 3)
        with gpu_pkg.Context(w, h, pat, (1, 2, 3), src_stride=4 * w + pad, inflight=2,
                             flags=gpu_pkg.FLAG_RGB2BAYER) as ctx:
            for _ in range(2):
                assert np.array_equal(ctx.process_host(src)[:, :w], want[:, :w]), (w, h, pat, bands)
            outs = [np.zeros((h, ctx.dst_stride), np.uint8) for _ in range(2)]
            for i, o in enumerate(outs):
                ctx.submit(src, o, tag=i + 1)
            assert [ctx.wait(), ctx.wait()] == [1, 2]
            for o in outs:
                assert np.array_equal(o[:, :w], want[:, :w])


def test_rgb2bayer_padded_source_rows_and_batch(gpu_pkg, oracle):
    rng = np.random.default_rng(10)
    w, h, n = 130, 21, 5
    src = rng.integers(0, 256, (n, h, 4 * w + 24), dtype=np.uint8)       # GstVideoMeta-style padded rows
    with gpu_pkg.Context(w, h, "grbg", (1, 2, 3), src_stride=4 * w + 24, flags=gpu_pkg.FLAG_RGB2BAYER) as ctx:
        got = ctx.process_batch_via_device(src)
    for f in range(n):
        want = oracle.rgb2bayer(src[f], w, "grbg", 1, 2, 3)
        assert np.array_equal(got[f][:, :w], want[:, :w])


def test_round_trip_4k_batch_is_identity(gpu_pkg):
    """Size-independent property at BASELINE size: rgb2bayer(bayer2rgb(S)) == S for every order and layout,
    because bayer2rgb keeps each original sample at its own CFA site (4K x 8 frames, device-resident)."""
    w, h, n = 3840, 2160, 8
    for pat, fmt in (("bggr", "RGBx"), ("rggb", "BGRx"), ("grbg", "xRGB"), ("gbrg", "xBGR")):
        with gpu_pkg.Context(w, h, pat, fmt) as fwd, \
                gpu_pkg.Context(w, h, pat, gpu_pkg.FORMATS[fmt], flags=gpu_pkg.FLAG_RGB2BAYER) as inv:
            d_src = fwd.device_alloc(n * fwd.src_bytes)
            d_rgb = fwd.device_alloc(n * fwd.dst_bytes)
            d_back = fwd.device_alloc(n * inv.dst_bytes)
            fwd.fill_synthetic(d_src, n, seed=2)
            fwd.process_device(d_src, d_rgb, n)
            fwd.sync()
            inv.process_device(d_rgb, d_back, n)
            inv.sync()
            a = fwd.from_device(d_src, n * fwd.src_bytes)
            b = inv.from_device(d_back, n * inv.dst_bytes)
            assert np.array_equal(a, b), (pat, fmt)
            for p in (d_src, d_rgb, d_back):
                fwd.device_free(p)


@needs_gst
@pytest.mark.parametrize("props", ["", "inflight=3", "inflight=2 devices=0,0 pinned-pool=false"])
def test_rgb2bayer_element_and_plugin_round_trip(plugin, gpu_pkg, oracle, tmp_path, props):
    """videotestsrc ARGB -> rgb2bayer -> bayer2rgb through one pipeline; the mosaic equals the oracle's
    rgb2bayer of the ARGB frames and the final RGBx equals the oracle's bayer2rgb of that mosaic
    (the reference's own pipeline test for this element only asserts EOS:
    tests/check/elements/autovideoconvert.c:98-110).  With `inflight` > 1 / `devices` the element runs the same
    queued mode as bayer2rgb: every frame arrives, in order, and the tail is drained at EOS."""
    w, h, n = 320, 240, 7 if props else 3
    argb, mosaic, rgb = (str(tmp_path / f) for f in ("argb.raw", "mosaic.raw", "rgb.raw"))
    pipeline = ("videotestsrc num-buffers=%d ! video/x-raw,format=ARGB,width=%d,height=%d,framerate=30/1 "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! rgb2bayer %s "
                "! video/x-bayer,format=grbg ! tee name=u u. ! queue ! filesink location=%s "
                "u. ! queue ! bayer2rgb ! video/x-raw,format=RGBx ! filesink location=%s"
                % (n, w, h, argb, props, mosaic, rgb))
    res = subprocess.run([GST_LAUNCH, "-q"] + pipeline.split(), capture_output=True, text=True,
                         env=gst_env(tmp_path), timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    A = np.fromfile(argb, np.uint8).reshape(n, h, 4 * w)
    M = np.fromfile(mosaic, np.uint8).reshape(n, h, w)
    R = np.fromfile(rgb, np.uint8).reshape(n, h, 4 * w)
    for f in range(n):
        assert np.array_equal(M[f], oracle.rgb2bayer(A[f], w, "grbg", 1, 2, 3)[:, :w])
        assert np.array_equal(R[f], oracle.bayer2rgb(M[f], w, "grbg", 0, 1, 2))
