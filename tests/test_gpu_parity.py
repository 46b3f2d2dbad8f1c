"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI of
include/mibayer.h, must be bit-exact (uint8) with the CPU oracle -- and through it with the
reference -- for every Bayer order x byte layout, on edge geometries, on the committed golden
fixtures, on the survey's whole-element md5 known answers at 1080p/4K/8K, and for every kernel
variant."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATTERNS = ("bggr", "gbrg", "grbg", "rggb")
LAYOUTS = ("RGBx", "BGRx", "xRGB", "xBGR")
ALL_FORMATS = ("RGBx", "xRGB", "BGRx", "xBGR", "RGBA", "ARGB", "BGRA", "ABGR")


def md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


def check(pkg, oracle, src, w, pat, fmt, variant=0, dst_stride=0, via="device"):
    h, stride = src.shape
    r, g, b = pkg.FORMATS[fmt]
    want = oracle.bayer2rgb(src, w, pat, r, g, b, dst_stride=dst_stride or None)
    with pkg.Context(w, h, pat, fmt, src_stride=stride, dst_stride=dst_stride, variant=variant) as ctx:
        if via == "host":
            got = ctx.process_host(src)
        else:
            got = ctx.process_batch_via_device(src[None])[0]
    if via == "host" and dst_stride:
        # padding bytes of the destination rows keep their guard fill, as with the reference
        assert (got[:, 4 * w:] == 0xA5).all()
    else:
        want = want.copy()
    n_bad = int((got[:, :4 * w] != want[:, :4 * w]).sum())
    assert n_bad == 0, "%dx%d %s->%s variant %d via %s: %d differing bytes, first at %s" % (
        w, h, pat, fmt, variant, via, n_bad, np.argwhere(got[:, :4 * w] != want[:, :4 * w])[:4].tolist())


def test_golden_fixtures(gpu_pkg, golden):
    for name in [k for k in golden.files if k.startswith("in_")]:
        dims = name[3:]
        w, h = (int(v) for v in dims.split("x"))
        src = golden[name]
        for pat in PATTERNS:
            for lay in LAYOUTS:
                want = golden["out_%s_%s_%s" % (dims, pat, lay)]
                with gpu_pkg.Context(w, h, pat, lay, src_stride=src.shape[1]) as ctx:
                    got = ctx.process_batch_via_device(src[None])[0]
                assert np.array_equal(got, want), (dims, pat, lay)


def test_all_orders_all_formats_fast_and_generic_paths(gpu_pkg, oracle):
    rng = np.random.default_rng(3)
    # 512x70: W%16==0 -> 16-byte fast path, two tile rows; 258x37 / 1030x9: generic path,
    # W%4==2 tail lane, tiles crossing the 256/1024-pixel strip boundaries
    for (w, h) in [(512, 70), (258, 37), (1030, 9), (2048, 33)]:
        stride = (w + 3) & ~3
        src = rng.integers(0, 256, (h, stride), dtype=np.uint8)
        for pat in PATTERNS:
            for fmt in ALL_FORMATS:
                check(gpu_pkg, oracle, src, w, pat, fmt)


def test_edge_geometries(gpu_pkg, oracle):
    rng = np.random.default_rng(4)
    sizes = [(4, 3), (4, 4), (6, 4), (8, 5), (16, 3), (16, 4), (254, 4), (256, 5), (258, 6), (260, 7),
             (1024, 8), (1026, 31), (1028, 32), (1280, 33), (66, 34), (4, 65), (3840, 3)]
    for (w, h) in sizes:
        stride = (w + 3) & ~3
        src = rng.integers(0, 256, (h, stride), dtype=np.uint8)
        for pat in ("bggr", "gbrg"):
            check(gpu_pkg, oracle, src, w, pat, "BGRx")
            check(gpu_pkg, oracle, src, w, pat, "xRGB", via="host")


def test_every_variant_is_bit_exact(gpu_pkg, oracle):
    rng = np.random.default_rng(5)
    names = gpu_pkg.variant_names()
    for (w, h) in [(1296, 75), (262, 19), (3840, 40)]:
        stride = (w + 3) & ~3
        src = rng.integers(0, 256, (h, stride), dtype=np.uint8)
        for v in range(len(names)):
            for pat, fmt in (("rggb", "BGRx"), ("grbg", "xBGR")):
                check(gpu_pkg, oracle, src, w, pat, fmt, variant=v)


def test_rounding_stress_inputs(gpu_pkg, oracle):
    w, h = 528, 20
    yy, xx = np.mgrid[0:h, 0:w]
    cases = [np.zeros((h, w)), np.full((h, w), 255), ((yy + xx) & 1) * 255, (xx & 1) * 255,
             (yy & 1) * 255, (xx * 7 + yy * 13) & 255, 255 - ((xx + yy) & 1), (xx & 3) * 85]
    for S in cases:
        S = np.ascontiguousarray(S.astype(np.uint8))
        for pat in PATTERNS:
            check(gpu_pkg, oracle, S, w, pat, "RGBx")


def test_strides_and_padding(gpu_pkg, oracle):
    rng = np.random.default_rng(6)
    w, h = 130, 21
    src = rng.integers(0, 256, (h, 144), dtype=np.uint8)      # padded source rows
    check(gpu_pkg, oracle, src, w, "grbg", "BGRx")
    check(gpu_pkg, oracle, src, w, "grbg", "BGRx", dst_stride=4 * w + 40)
    check(gpu_pkg, oracle, src, w, "grbg", "BGRx", dst_stride=4 * w + 40, via="host")
    src2 = src.copy()
    src2[:, w:] ^= 0xFF          # padding bytes never influence the result
    r, g, b = gpu_pkg.FORMATS["BGRx"]
    with gpu_pkg.Context(w, h, "grbg", "BGRx", src_stride=144) as ctx:
        assert np.array_equal(ctx.process_host(src), ctx.process_host(src2))


@pytest.mark.parametrize("bands", ["", "1", "3", "8"], ids=["default", "1", "3", "8"])
def test_banded_synchronous_host_path(gpu_lab_pkg, oracle, bands, monkeypatch):
    """Frames of 16 MB and more go through the synchronous host path in horizontal bands (upload of band b+1, kernel
    of band b and download of band b-1 overlap inside one frame).  Every band count gives the reference's bytes: the
    halo rows across band boundaries, the top row (up(0) = 1) and the bottom rows (dn(H-1) = H-4) included; heights
    that are no multiple of the tile height; padded destination rows (2-D download); the queued use of the same
    context (which does not band) in between.  (MIBAYER_HOST_BANDS is a knob of the lab build; the product build runs
    the "default" arm, covered by the full-size tests.)"""
    gpu_pkg = gpu_lab_pkg
    if bands:
        monkeypatch.setenv("MIBAYER_HOST_BANDS", bands)
    else:
        monkeypatch.delenv("MIBAYER_HOST_BANDS", raising=False)
    rng = np.random.default_rng(40)
    for (w, h, pat, fmt, pad) in ((3840, 2160, "rggb", "BGRx", 0), (2048, 2053, "gbrg", "xRGB", 0),
                                  (2560, 1666, "grbg", "RGBx", 64)):
        r, g, b = gpu_pkg.FORMATS[fmt]
        src = rng.integers(0, 256, (h, (w + 3) & ~3), dtype=np.uint8)
        want = oracle.bayer2rgb(src, w, pat, r, g, b)
        with gpu_pkg.Context(w, h, pat, fmt, dst_stride=4 * w + pad, inflight=2) as ctx:
            for _ in range(2):
                got = ctx.process_host(src)
                assert np.array_equal(got[:, :4 * w], want), (w, h, pat, fmt, bands)
                if pad:
                    assert (got[:, 4 * w:] == 0xA5).all()     # padding never written
            # two frames in flight through the same context: unbanded path, same bytes
            outs = [np.full((h, 4 * w + pad), 0xA5, np.uint8) for _ in range(2)]
            for i, o in enumerate(outs):
                ctx.submit(src, o, tag=i + 1)
            assert [ctx.wait(), ctx.wait()] == [1, 2]
            for o in outs:
                assert np.array_equal(o[:, :4 * w], want)
            assert np.array_equal(ctx.process_host(src)[:, :4 * w], want)


def test_known_md5_answers_full_size(gpu_pkg):
    """BASELINE.json configs 2-4 geometries: md5 of the HIP output equals the md5 the compiled
    reference element produced.  Input is generated ON the device."""
    with open(os.path.join(ROOT, "tests", "golden", "known_md5.json")) as f:
        entries = json.load(f)["entries"]
    for e in entries:
        w, h = e["width"], e["height"]
        with gpu_pkg.Context(w, h, e["pattern"], e["format"]) as ctx:
            d_src = ctx.device_alloc(ctx.src_bytes)
            d_dst = ctx.device_alloc(ctx.dst_bytes)
            ctx.fill_synthetic(d_src, 1, e["seed"])
            ctx.process_device(d_src, d_dst, 1)
            ctx.sync()
            assert md5(ctx.from_device(d_src, ctx.src_bytes)) == e["md5_input"], e
            assert md5(ctx.from_device(d_dst, ctx.dst_bytes)) == e["md5_output"], e
            ctx.device_free(d_src)
            ctx.device_free(d_dst)


def test_hip_equals_the_reference_frame_driver_directly(gpu_pkg, oracle):
    """HIP output against the REFERENCE'S OWN gst_bayer2rgb_process (gstbayer2rgb.c:387-451, compiled into
    oracle/_ref/libbayer_frame_ref.so, which travels to the GPU box as a binary) with no restatement in between:
    1080p, 4K, camera geometries off every grid, one order x layout each way round."""
    if not oracle.have_ref_frame():
        pytest.skip("oracle/_ref/libbayer_frame_ref.so not on this box")
    rng = np.random.default_rng(77)
    cases = [(1920, 1080, "rggb", "BGRx"), (3840, 2160, "bggr", "RGBx"), (3840, 2160, "grbg", "xBGR"),
             (3838, 2160, "gbrg", "ARGB"), (4056, 3040, "rggb", "RGBA"), (1366, 768, "grbg", "BGRA"),
             (2590, 1942, "bggr", "xRGB"), (640, 480, "bggr", "RGBx"), (4, 3, "gbrg", "ABGR"), (66, 7, "rggb", "BGRx")]
    for (w, h, pat, fmt) in cases:
        src = rng.integers(0, 256, (h, (w + 3) & ~3), dtype=np.uint8)
        r, g, b = gpu_pkg.FORMATS[fmt]
        want = oracle.ref_frame_bayer2rgb(src, w, pat, r, g, b)
        with gpu_pkg.Context(w, h, pat, fmt, src_stride=src.shape[1]) as ctx:
            got_d = ctx.process_batch_via_device(src[None])[0]
            got_h = ctx.process_host(src)
        assert np.array_equal(got_d, want), (w, h, pat, fmt)
        assert np.array_equal(got_h, want), (w, h, pat, fmt)


def test_batch_launch_equals_per_frame(gpu_pkg, oracle):
    """One launch over N frames (blockIdx decodes frame, tile) == N oracle frames; also with a
    frame pitch larger than the frame (gaps must stay untouched)."""
    w, h, n = 640, 50, 9
    src = oracle.fill_synthetic(w, h, n, seed=21)
    want = oracle.bayer2rgb_batch(src, w, "gbrg", 2, 1, 0, nthreads=2)
    with gpu_pkg.Context(w, h, "gbrg", "BGRx") as ctx:
        got = ctx.process_batch_via_device(src)
        assert np.array_equal(got, want)
        gap_s, gap_d = ctx.src_bytes + 256, ctx.dst_bytes + 512
        d_src = ctx.device_alloc(n * gap_s)
        d_dst = ctx.device_alloc(n * gap_d)
        host = np.full((n, gap_s), 0x11, np.uint8)
        host[:, :ctx.src_bytes] = src.reshape(n, -1)
        ctx.to_device(d_src, host)
        ctx.to_device(d_dst, np.full(n * gap_d, 0xEE, np.uint8))
        ctx.process_device(d_src, d_dst, n, src_frame_bytes=gap_s, dst_frame_bytes=gap_d)
        ctx.sync()
        out = ctx.from_device(d_dst, n * gap_d).reshape(n, gap_d)
        assert np.array_equal(out[:, :ctx.dst_bytes].reshape(want.shape), want)
        assert (out[:, ctx.dst_bytes:] == 0xEE).all()
        ctx.device_free(d_src)
        ctx.device_free(d_dst)


def test_linearity_free_property_4k_batch(gpu_pkg, oracle):
    """Full-size property check (config 3 geometry): frames of a 4K batch are independent and the
    kernel is deterministic -- frame f of a batch launch equals the single-frame launch of the same
    input, and frame 0 equals the oracle."""
    w, h, n = 3840, 2160, 6
    with gpu_pkg.Context(w, h, "rggb", "BGRx") as ctx:
        d_src = ctx.device_alloc(n * ctx.src_bytes)
        d_dst = ctx.device_alloc(n * ctx.dst_bytes)
        d_one = ctx.device_alloc(ctx.dst_bytes)
        ctx.fill_synthetic(d_src, n, seed=2)
        ctx.process_device(d_src, d_dst, n)
        ctx.sync()
        batch = ctx.from_device(d_dst, n * ctx.dst_bytes).reshape(n, -1)
        for f in (0, 3, 5):
            ctx.process_device(d_src + f * ctx.src_bytes, d_one, 1)
            ctx.sync()
            assert np.array_equal(ctx.from_device(d_one, ctx.dst_bytes), batch[f])
        src0 = oracle.fill_synthetic(w, h, 1, seed=2)[0]
        assert np.array_equal(oracle.bayer2rgb(src0, w, "rggb", 2, 1, 0).reshape(-1), batch[0])
        # frame 5 against the oracle too (frame index enters the generator)
        src5 = oracle.fill_synthetic(w, h, 1, seed=2, first_frame=5)[0]
        assert np.array_equal(oracle.bayer2rgb(src5, w, "rggb", 2, 1, 0).reshape(-1), batch[5])
        for p in (d_src, d_dst, d_one):
            ctx.device_free(p)


def test_async_ring_order_and_tags(gpu_pkg, oracle):
    w, h, n = 320, 64, 7
    src = oracle.fill_synthetic(w, h, n, seed=31)
    want = oracle.bayer2rgb_batch(src, w, "bggr", 0, 1, 2)
    with gpu_pkg.Context(w, h, "bggr", "RGBx", inflight=3) as ctx:
        outs = [np.zeros((h, 4 * w), np.uint8) for _ in range(n)]
        done = []
        for i in range(n):
            if ctx.pending() == 3:
                done.append(ctx.wait())
            ctx.submit(src[i], outs[i], tag=1000 + i)
        with pytest.raises(gpu_pkg.MibayerError):
            while True:                     # ring never holds more than `inflight`
                ctx.submit(src[0], np.zeros((h, 4 * w), np.uint8), tag=1)
        # drain the extra submissions too
        while ctx.pending():
            done.append(ctx.wait())
        assert done[:n] == [1000 + i for i in range(n)]
        with pytest.raises(gpu_pkg.MibayerError) as e:
            ctx.wait()
        assert e.value.status == gpu_pkg.ERR_EMPTY
        for i in range(n):
            assert np.array_equal(outs[i], want[i])


def test_pinned_host_memory_path(gpu_pkg, oracle):
    import ctypes
    w, h = 1920, 1080            # BASELINE.json configs[1]
    L = gpu_pkg.lib()
    with gpu_pkg.Context(w, h, "rggb", "BGRx") as ctx:
        p_src = L.mibayer_host_alloc(ctx.src_bytes)
        p_dst = L.mibayer_host_alloc(ctx.dst_bytes)
        assert p_src and p_dst
        src = np.ctypeslib.as_array(ctypes.cast(p_src, ctypes.POINTER(ctypes.c_uint8)), (h, w))
        dst = np.ctypeslib.as_array(ctypes.cast(p_dst, ctypes.POINTER(ctypes.c_uint8)), (h, 4 * w))
        src[:] = oracle.fill_synthetic(w, h, 1, seed=1)[0]
        ctx.process_host(src, dst)
        assert md5(dst) == "f14f6ad248ef0bac0f28546db6d14813"     # SURVEY.md B.3, 1080p rggb->BGRx
        L.mibayer_host_free(p_src)
        L.mibayer_host_free(p_dst)


@pytest.mark.parametrize("w,h,n", [(1280, 96, 12), (1000, 60, 6), (250, 40, 8), (4056, 24, 2), (1366, 30, 4)],
                         ids=["1280", "1000_one_tile_off_sector", "250", "4056", "1366"])
def test_autotune_keeps_results_bit_exact(gpu_pkg, oracle, w, h, n):
    """mibayer_autotune() runs every candidate plan on the caller's buffers: whichever wins, d_dst
    holds the oracle's bytes afterwards and so does every later launch.  1000 px: rows that fit one tile and sit
    off the sector grid -- the context keeps its nt shape while the other candidates used to be looked up as
    plain-store twins, four shapes for a three-entry table."""
    src = np.random.default_rng(41).integers(0, 256, (n, h, (w + 3) & ~3), dtype=np.uint8)
    want = np.stack([oracle.bayer2rgb(src[f], w, "grbg", 1, 2, 3) for f in range(n)])
    with gpu_pkg.Context(w, h, "grbg", "xRGB") as ctx:
        d_src = ctx.device_alloc(n * ctx.src_bytes)
        d_dst = ctx.device_alloc(n * ctx.dst_bytes)
        ctx.to_device(d_src, src)
        report = ctx.autotune(d_src, d_dst, n)
        assert "band" in report and "ms" in report
        assert np.array_equal(ctx.from_device(d_dst, n * ctx.dst_bytes).reshape(want.shape), want)
        ctx.to_device(d_dst, np.zeros(n * ctx.dst_bytes, np.uint8))
        ctx.process_device(d_src, d_dst, n)
        ctx.sync()
        assert np.array_equal(ctx.from_device(d_dst, n * ctx.dst_bytes).reshape(want.shape), want)
        assert ctx.launch_geometry(n)["band"] in (0, 1, -(-ctx.launch_geometry(n)["tile_rows"] // 8))
        ctx.device_free(d_src)
        ctx.device_free(d_dst)


def _frames_equal_oracle(ctx, oracle, d_dst, frames, w, h, pattern, fmt, seed, nthreads=8):
    """Compare the listed frames of a device-resident batch with the oracle (inputs re-generated on the host
    with the same counter-based generator, frame index = position in the batch)."""
    r, g, b = oracle.LAYOUTS[fmt]
    for f in frames:
        src = oracle.fill_synthetic(w, h, 1, seed, first_frame=f)[0]
        want = oracle.bayer2rgb(src, w, pattern, r, g, b).reshape(-1)
        got = ctx.from_device(d_dst + f * ctx.dst_bytes, ctx.dst_bytes)
        assert np.array_equal(got, want), "frame %d of the batch differs (%s->%s)" % (f, pattern, fmt)


def test_config3_4k_batch64_all_four_orders(gpu_pkg, oracle):
    """BASELINE.json configs[2] at full size: 3840x2160, batch = 64 frames, each of bggr/rggb/grbg/gbrg -> BGRx
    in ONE launch per order.  Every 5th frame plus the first and last are compared byte for byte with the oracle;
    all 64 are covered by the size-independent property that frame f of the batch equals the single-frame launch
    hash chain (md5 of per-frame md5s is identical between the batch launch and 64 single launches)."""
    w, h, n = 3840, 2160, 64
    sample = sorted(set(list(range(0, n, 5)) + [n - 1]))
    with gpu_pkg.Context(w, h, "bggr", "BGRx") as c0:
        d_src = c0.device_alloc(n * c0.src_bytes)
        d_dst = c0.device_alloc(n * c0.dst_bytes)
        d_one = c0.device_alloc(c0.dst_bytes)
        c0.fill_synthetic(d_src, n, seed=2)
        c0.sync()
        assert md5(c0.from_device(d_src, c0.src_bytes)) == "cd74cbcc694cec2ee316a4120eeca8f5"   # SURVEY B.3 input
        known_frame0 = {"bggr": "09c1a1414adc6f66966a028e891cddfa", "rggb": "2fc0dc5ff3a92760b0943a34abeffe66",
                        "grbg": "ec15421e44cff4f399842deec374378b", "gbrg": "d3d3c99d165672895f2ae0b642158928"}
        for pattern in PATTERNS:
            with gpu_pkg.Context(w, h, pattern, "BGRx") as ctx:
                ctx.process_device(d_src, d_dst, n)
                ctx.sync()
                assert md5(ctx.from_device(d_dst, ctx.dst_bytes)) == known_frame0[pattern]
                _frames_equal_oracle(ctx, oracle, d_dst, sample, w, h, pattern, "BGRx", 2)
                batch_chain = hashlib.md5()
                single_chain = hashlib.md5()
                for f in range(n):
                    batch_chain.update(hashlib.md5(ctx.from_device(d_dst + f * ctx.dst_bytes, ctx.dst_bytes)).digest())
                for f in range(n):
                    ctx.process_device(d_src + f * ctx.src_bytes, d_one, 1)
                    ctx.sync()
                    single_chain.update(hashlib.md5(ctx.from_device(d_one, ctx.dst_bytes)).digest())
                assert batch_chain.hexdigest() == single_chain.hexdigest(), pattern
        for p in (d_src, d_dst, d_one):
            c0.device_free(p)


def test_config4_8k_batch64_per_gpu_share(gpu_pkg, oracle):
    """BASELINE.json configs[3]: 7680x4320, 512 frames sharded round-robin over 8 GPUs = 64 frames per GPU.
    This is rank 0's share on one GPU (global frames 0, 8, 16, ...): 2.1 GB in, 8.5 GB out, one launch."""
    w, h, n, world = 7680, 4320, 64, 8
    with gpu_pkg.Context(w, h, "bggr", "RGBx") as ctx:
        d_src = ctx.device_alloc(n * ctx.src_bytes)
        d_dst = ctx.device_alloc(n * ctx.dst_bytes)
        for i in range(n):                       # rank 0 of 8: local frame i is global frame 8*i
            ctx.fill_synthetic(d_src + i * ctx.src_bytes, 1, seed=3, first_frame=i * world)
        ctx.process_device(d_src, d_dst, n)
        ctx.sync()
        assert md5(ctx.from_device(d_src, ctx.src_bytes)) == "a2ba091ae9ecc96f05430032ce80b506"
        assert md5(ctx.from_device(d_dst, ctx.dst_bytes)) == "82665f3caa7a0df5435db7a86150b8dd"
        r, g, b = oracle.LAYOUTS["RGBx"]
        for i in (1, 31, 63):
            src = oracle.fill_synthetic(w, h, 1, 3, first_frame=i * world)[0]
            want = oracle.bayer2rgb(src, w, "bggr", r, g, b).reshape(-1)
            assert np.array_equal(ctx.from_device(d_dst + i * ctx.dst_bytes, ctx.dst_bytes), want), i
        # idempotence: a second launch over the same input reproduces every byte of the 8.5 GB output
        before = [hashlib.md5(ctx.from_device(d_dst + i * ctx.dst_bytes, ctx.dst_bytes)).hexdigest()
                  for i in range(0, n, 7)]
        ctx.process_device(d_src, d_dst, n)
        ctx.sync()
        after = [hashlib.md5(ctx.from_device(d_dst + i * ctx.dst_bytes, ctx.dst_bytes)).hexdigest()
                 for i in range(0, n, 7)]
        assert before == after
        ctx.device_free(d_src)
        ctx.device_free(d_dst)


def test_config4_and_5_every_ranks_share(gpu_pkg, oracle):
    """not only rank 0's share of the two 8-GPU configs.  configs[3] (7680x4320, 512 frames over 8
    GPUs): for every rank 1..7 the first and the last frame of its share -- global frames r and r + 8*63, seed 3 --
    are generated on the device the way bench.py does (frame-index-addressed, so a wrong g -> rank map shows) and
    converted, and compared with the oracle.  configs[4] (1000 frames over 8 GPUs): shard_frames() gives every rank
    its 125 frames, g % 8 == r, and one frame per rank (its last) goes through the hipGraph host path."""
    import bench
    w, h, world = 7680, 4320, 8
    r_, g_, b_ = oracle.LAYOUTS["RGBx"]
    with gpu_pkg.Context(w, h, "bggr", "RGBx") as ctx:
        d_src = ctx.device_alloc(2 * ctx.src_bytes)
        d_dst = ctx.device_alloc(2 * ctx.dst_bytes)
        for rank in range(1, world):
            share = bench.shard_frames(512, world, rank)
            assert len(share) == 64 and share[0] == rank and share[-1] == rank + 8 * 63
            for k, gframe in enumerate((share[0], share[-1])):
                ctx.fill_synthetic(d_src + k * ctx.src_bytes, 1, seed=3, first_frame=gframe)
            ctx.process_device(d_src, d_dst, 2)
            ctx.sync()
            for k, gframe in enumerate((share[0], share[-1])):
                src = oracle.fill_synthetic(w, h, 1, 3, first_frame=gframe)[0]
                assert np.array_equal(ctx.from_device(d_src + k * ctx.src_bytes, ctx.src_bytes), src.reshape(-1))
                want = oracle.bayer2rgb(src, w, "bggr", r_, g_, b_).reshape(-1)
                assert np.array_equal(ctx.from_device(d_dst + k * ctx.dst_bytes, ctx.dst_bytes), want), (rank, gframe)
        ctx.device_free(d_src)
        ctx.device_free(d_dst)
    w, h = 3840, 2160
    r_, g_, b_ = oracle.LAYOUTS["BGRx"]
    shares = [bench.shard_frames(1000, world, rank) for rank in range(world)]
    assert [len(s_) for s_ in shares] == [125] * 8 and sorted(f for s_ in shares for f in s_) == list(range(1000))
    with gpu_pkg.Context(w, h, "rggb", "BGRx", inflight=2, flags=gpu_pkg.FLAG_HIPGRAPH) as ctx:
        for rank, share in enumerate(shares):
            assert all(f % world == rank for f in share)
            src = oracle.fill_synthetic(w, h, 1, 2, first_frame=share[-1])[0]
            assert np.array_equal(ctx.process_host(src), oracle.bayer2rgb(src, w, "rggb", r_, g_, b_)), rank


def test_plan_cache_hands_the_measured_plan_to_later_contexts(gpu_pkg, oracle):
    """what mibayer_autotune measured is kept per (device, geometry, LAUNCH CLASS) for the process; the
    next context of that geometry starts from it (source CACHED) without measuring, other geometries do not, the bytes
    are the oracle's before and after, a context created earlier can pick the plan up (mibayer_plan_from_cache), and
    mibayer_plan_cache_clear() / MIBAYER_PLAN_CACHE=0 bring the default back.  ADVICE r04: a plan measured on one
    launch class (here: launches of 8 1080p frames, the frame class) never becomes the default of the other."""
    w, h, n = 1920, 1080, 8
    big = 64                    # a launch over 64 frames falls into the batch class
    L = gpu_pkg.lib()
    L.mibayer_plan_cache_clear()
    src = oracle.fill_synthetic(w, h, n, seed=5)
    want = oracle.bayer2rgb_batch(src, w, "rggb", 2, 1, 0, nthreads=4)
    D, M, C = gpu_pkg.PLAN_DEFAULT, gpu_pkg.PLAN_MEASURED, gpu_pkg.PLAN_CACHED
    with gpu_pkg.Context(w, h, "rggb", "BGRx") as first, gpu_pkg.Context(w, h, "gbrg", "BGRx") as early:
        assert first.get_plan_for(n)[3] == D and early.get_plan_for(n)[3] == D and first.plan_source == D
        default_plan, default_big = first.get_plan_for(n)[:3], first.get_plan_for(big)
        assert first.get_plan() == default_big[:3]          # mibayer_get_plan describes the batch class
        assert np.array_equal(first.process_batch_via_device(src), want)
        d_src = first.device_alloc(n * first.src_bytes)
        d_dst = first.device_alloc(n * first.dst_bytes)
        first.to_device(d_src, src)
        report = first.autotune(d_src, d_dst, n)
        assert first.get_plan_for(n)[3] == M and "band" in report
        assert first.get_plan_for(big) == default_big and first.plan_source == D     # the other class: untouched
        measured = first.get_plan_for(n)[:3]
        assert np.array_equal(first.from_device(d_dst, n * first.dst_bytes).reshape(want.shape), want)
        # a later context of the same geometry (another Bayer order: same kernel): the measured plan, not measured again
        with gpu_pkg.Context(w, h, "bggr", "RGBx") as second:
            assert second.get_plan_for(n) == measured + (C,) and second.get_plan_for(1) == measured + (C,)
            assert second.get_plan_for(big) == default_big and second.plan_source == D
            assert np.array_equal(second.process_batch_via_device(src[:2]),
                                  oracle.bayer2rgb_batch(src[:2], w, "bggr", 0, 1, 2, nthreads=2))
        # one created before the measurement picks it up on request
        assert early.get_plan_for(n)[:3] == default_plan and early.plan_from_cache()
        assert early.get_plan_for(n) == measured + (C,) and early.get_plan_for(big) == default_big
        # another geometry, another stride: not this entry
        with gpu_pkg.Context(w, h + 2, "rggb", "BGRx") as other, \
                gpu_pkg.Context(w, h, "rggb", "BGRx", dst_stride=4 * w + 64) as padded:
            assert other.get_plan_for(n)[3] == D and padded.get_plan_for(n)[3] == D
            assert not other.plan_from_cache()
        # an explicit variant is never overridden
        with gpu_pkg.Context(w, h, "rggb", "BGRx", variant=3) as pinned:
            assert pinned.get_plan_for(n)[3] == D and pinned.get_plan_for(n)[0] == 3
            assert pinned.variant_name == "lds_1x8_r4_dpp_nt"
        # the list form measures over separate allocations and records too
        L.mibayer_plan_cache_clear()
        with gpu_pkg.Context(w, h, "rggb", "BGRx") as third:
            assert third.get_plan_for(4)[3] == D
            srcs = [d_src + f * third.src_bytes for f in range(4)]
            dsts = [d_dst + f * third.dst_bytes for f in range(4)]
            third.autotune_list(srcs, dsts)
            assert third.get_plan_for(4)[3] == M
            assert np.array_equal(third.from_device(d_dst, 4 * third.dst_bytes).reshape(want[:4].shape), want[:4])
            with gpu_pkg.Context(w, h, "rggb", "BGRx") as fourth:
                assert fourth.get_plan_for(4) == third.get_plan_for(4)[:3] + (C,)
        # an explicit pin (mibayer_set_plan) holds for every class; mibayer_set_plan_for for one
        with gpu_pkg.Context(w, h, "rggb", "BGRx") as pin:
            pin.set_plan(2, 0)
            assert pin.get_plan_for(1) == (2, 0, 0, gpu_pkg.PLAN_SET) == pin.get_plan_for(big)
            pin.set_plan_for(1, 3, 1)
            assert pin.get_plan_for(1)[:2] == (3, 1) and pin.get_plan_for(big)[:2] == (2, 0)
            assert pin.launch_geometry(1)["tile_w"] == 256 and pin.launch_geometry(big)["tile_w"] == 512
            assert np.array_equal(pin.process_batch_via_device(src[:3]), want[:3])
        L.mibayer_plan_cache_clear()
        with gpu_pkg.Context(w, h, "rggb", "BGRx") as again:
            assert again.get_plan_for(n) == default_plan + (D,)
        first.device_free(d_src)
        first.device_free(d_dst)
    code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as e; p = e.load_package(); "
            "c = p.Context(640, 480, 'rggb', 'BGRx'); d = c.device_alloc(c.src_bytes); o = c.device_alloc(c.dst_bytes); "
            "c.fill_synthetic(d, 1, 1); c.autotune(d, o, 1); c2 = p.Context(640, 480, 'rggb', 'BGRx'); "
            "print('source', c.get_plan_for(1)[3], c2.get_plan_for(1)[3])" % ROOT)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, MIBAYER_PLAN_CACHE="0"))
    assert res.returncode == 0 and "source 1 0" in res.stdout, res.stdout + res.stderr[-1500:]


def test_frame_class_plan_and_the_second_compute_queue(gpu_pkg, oracle):
    """Round 5: (1) a launch over ONE frame runs in the production shape whose grid needs the fewest rounds of the
    device's workgroup slots (4K: 256x32 tiles = 1020 workgroups on 1024 slots, where 1024x8 tiles need 1080), batch
    launches keep the batch plan; the two classes give the same bytes.  (2) mibayer_ctx_frame_queue: independent frames
    dealt round-robin over the device's four frame queues (hardware queues of their own, shared by the contexts of the
    device) -- what hipbayer2rgb does -- come out bit-exact, and mibayer_sync covers every queue the context used."""
    w, h = 3840, 2160
    with gpu_pkg.Context(w, h, "rggb", "BGRx") as ctx:
        slots = 4 * 256
        g1, g64 = ctx.launch_geometry(1), ctx.launch_geometry(64)
        assert g64["tile_w"] == 1024 and ctx.get_plan_for(64)[0] == 1
        if g1["tile_w"] == 256:       # a 256-CU device (the rule uses the device's own CU count)
            assert g1["grid_blocks"] == 1020 <= slots and ctx.get_plan_for(1)[0] == 3
        assert ctx.get_plan_for(4)[0] == ctx.get_plan_for(1)[0] and ctx.get_plan_for(5)[0] == 1
        n = 6
        src = oracle.fill_synthetic(w, h, n, seed=77)
        want = oracle.bayer2rgb_batch(src, w, "rggb", 2, 1, 0, nthreads=6)
        srcs = [ctx.device_alloc(ctx.src_bytes) for _ in range(n)]
        dsts = [ctx.device_alloc(ctx.dst_bytes) for _ in range(n)]
        for f in range(n):
            ctx.to_device(srcs[f], src[f])
        fq = ctx.frame_queues
        assert len(set(fq)) == gpu_pkg.FRAME_QUEUES == 4 and ctx.stream not in fq and ctx.frame_queues == fq
        with gpu_pkg.Context(w, h, "bggr", "RGBx") as other:        # per device, not per context
            assert other.frame_queues == fq
        with pytest.raises(gpu_pkg.MibayerError):
            ctx.frame_queue(4)
        for rep in range(3):
            for f in range(n):
                ctx.process_device(srcs[f], dsts[f], 1, stream=fq[f % 4])
        ctx.sync()                    # every frame queue this context launched on
        for f in range(n):
            assert np.array_equal(ctx.from_device(dsts[f], ctx.dst_bytes).reshape(want[f].shape), want[f]), f
        # the batch-class plan on the same single frames: same bytes
        ctx.set_plan_for(1, *ctx.get_plan_for(64)[:3])
        assert ctx.launch_geometry(1)["tile_w"] == 1024
        for f in range(n):
            ctx.to_device(dsts[f], np.zeros(16, np.uint8))
            ctx.process_device(srcs[f], dsts[f], 1, stream=fq[(f + 1) % 4])
        ctx.sync()
        for f in range(n):
            assert np.array_equal(ctx.from_device(dsts[f], ctx.dst_bytes).reshape(want[f].shape), want[f]), f
        for p in srcs + dsts:
            ctx.device_free(p)
    # batch-class defaults of common sensor widths with a measured winner (known_width_plan, mibayer_kernels.hip);
    # the frame class keeps its own rule, generic geometries and explicit variants are not touched
    for (gw, gh, variant, band) in ((2688, 1520, 1, 0), (4096, 2160, 3, 1), (2304, 1296, 2, 0), (4112, 3008, 2, 1),
                                    (3840, 2160, 1, -2 ** 31), (2690, 1520, None, None)):
        with gpu_pkg.Context(gw, gh, "grbg", "BGRx") as c3:
            if variant is not None:
                assert c3.get_plan_for(4096)[:2] == (variant, band), (gw, c3.get_plan_for(4096))
                assert c3.get_plan_for(1)[1] == -2 ** 31
            else:
                assert gpu_pkg.variant_names()[c3.get_plan_for(4096)[0]].endswith("_dpp")      # write-back twin
            src = np.random.default_rng(gw).integers(0, 256, (5, gh, (gw + 3) & ~3), dtype=np.uint8)
            assert np.array_equal(c3.process_batch_via_device(src),
                                  oracle.bayer2rgb_batch(src, gw, "grbg", 2, 1, 0, nthreads=5)), gw
    with gpu_pkg.Context(2688, 1520, "grbg", "BGRx", variant=3) as c4:
        assert c4.get_plan_for(4096)[:2] == (3, -2 ** 31)
    # the rule across geometries: fewest rounds, widest tile among equals; rows of one tile keep the narrow-tile rule
    for (gw, gh, tile_w) in ((3264, 2448, 256), (7680, 4320, 512), (2592, 1944, 1024), (4096, 2160, 1024),
                             (640, 480, 1024), (1920, 1080, 1024), (500, 300, 512)):
        with gpu_pkg.Context(gw, gh, "bggr", "RGBx") as c2:     # (MI355X: 256 CUs x 4 workgroups)
            assert c2.launch_geometry(1)["tile_w"] == tile_w, (gw, gh, c2.launch_geometry(1))


def test_plan_change_on_a_host_path_context_recuts_the_bands_and_rebuilds_the_graphs(gpu_pkg, oracle):
    """A frame-class plan that changes after the host path has run -- mibayer_set_plan[_for], the plan cache picked up
    later, mibayer_autotune -- changes the tile height the synchronous path cuts its bands in and the kernel node of the
    graphs captured per slot: bands are cut again and graphs of an older plan epoch rebuilt; the bytes stay the
    oracle's in the synchronous, the queued and both hipGraph forms."""
    w, h = 3840, 2160
    src = oracle.fill_synthetic(w, h, 1, seed=31)[0]
    want = oracle.bayer2rgb(src, w, "gbrg", 2, 1, 0)
    for flags in (0, gpu_pkg.FLAG_HIPGRAPH, gpu_pkg.FLAG_HIPGRAPH | gpu_pkg.FLAG_HIPGRAPH_CHAIN):
        with gpu_pkg.Context(w, h, "gbrg", "BGRx", inflight=2, flags=flags) as ctx:
            assert np.array_equal(ctx.process_host(src), want), flags
            for variant in (1, 2, 3, 1):
                ctx.set_plan_for(1, variant, -2 ** 31)
                assert ctx.launch_geometry(1)["tile_h"] == {1: 8, 2: 16, 3: 32}[variant]
                assert np.array_equal(ctx.process_host(src), want), (flags, variant)
                outs = [np.zeros_like(want) for _ in range(3)]
                for i, o in enumerate(outs):
                    if ctx.pending() == 2:
                        ctx.wait()
                    ctx.submit(src, o, tag=i + 1)
                while ctx.pending():
                    ctx.wait()
                assert all(np.array_equal(o, want) for o in outs), (flags, variant)
    # ... and through the plan cache: a context created BEFORE a measurement picks the plan up on request
    gpu_pkg.lib().mibayer_plan_cache_clear()
    with gpu_pkg.Context(w, h, "gbrg", "BGRx", inflight=2) as early, gpu_pkg.Context(w, h, "rggb", "BGRx") as tuner:
        assert np.array_equal(early.process_host(src), want)
        d_src, d_dst = tuner.device_alloc(tuner.src_bytes), tuner.device_alloc(tuner.dst_bytes)
        tuner.to_device(d_src, src)
        tuner.autotune(d_src, d_dst, 1)                 # the frame class of this geometry is in the cache now
        assert early.plan_from_cache() and early.get_plan_for(1)[3] == gpu_pkg.PLAN_CACHED
        assert np.array_equal(early.process_host(src), want)
        tuner.device_free(d_src)
        tuner.device_free(d_dst)
    gpu_pkg.lib().mibayer_plan_cache_clear()


def test_host_waits_nap_when_other_frames_are_queued(gpu_pkg, oracle):
    """VERDICT r03 #4 / ADVICE r03: a wait spins only while the frame waited for is alone in flight; with other frames
    queued behind it the thread naps (a few wake-ups per frame), so the CPU spent waiting per frame falls far below the
    frame time, at an unchanged frame rate; mibayer_set_wait_spin() pins either behaviour; results stay the oracle's."""
    import time
    w, h, n = 3840, 2160, 60
    L = gpu_pkg.lib()
    src = oracle.fill_synthetic(w, h, 1, seed=8)[0]
    want = oracle.bayer2rgb(src, w, "rggb", 2, 1, 0)
    bufs = [(_pinned(L, w * h, (h, w)), _pinned(L, 4 * w * h, (h, 4 * w))) for _ in range(3)]
    for (ps, s_), _ in bufs:
        s_[...] = src

    def stream(inflight, spin):
        with gpu_pkg.Context(w, h, "rggb", "BGRx", inflight=inflight) as ctx:
            ctx.set_wait_spin(spin)
            for phase in range(2):              # the first pass warms the ring and the link
                before = ctx.host_stats()
                t0 = time.perf_counter()
                for i in range(n):
                    if ctx.pending() == inflight:
                        ctx.wait()
                    ctx.submit(bufs[i % inflight][0][1], bufs[i % inflight][1][1], tag=i + 1)
                while ctx.pending():
                    ctx.wait()
                dt = time.perf_counter() - t0
            after = ctx.host_stats()
        assert all(np.array_equal(bufs[k][1][1], want) for k in range(inflight))
        d = {k: after[k] - before[k] for k in after}
        return n / dt, d["wait_cpu_ms"] * 1e3 / n, d["wait_wall_ms"] * 1e3 / n, d["polls"] / n, d["naps"] / n

    def best_of(inflight, spin, tries=3):       # frame rates of 40-ms runs scatter by several per cent: the best of three
        runs = [stream(inflight, spin) for _ in range(tries)]
        return max(runs, key=lambda r: r[0])

    fps_spin, cpu_spin, wall_spin, polls_spin, _ = best_of(3, 1000000)    # every wait spins (what round 3 did)
    fps_auto, cpu_auto, wall_auto, polls_auto, naps_auto = best_of(3, -1)  # automatic: other frames are queued -> naps
    fps_sync, cpu_sync, wall_sync, _, naps_sync = best_of(1, -1, 1)        # synchronous use: alone in flight -> spins
    assert cpu_spin > 0.7 * wall_spin                       # spinning: the wait costs its wall time in CPU
    assert cpu_auto < 0.5 * cpu_spin and naps_auto >= 1.0, (cpu_auto, cpu_spin, naps_auto)
    assert polls_auto < 0.2 * polls_spin
    assert fps_auto > 0.90 * fps_spin, (fps_auto, fps_spin)
    assert naps_sync < 0.5 and cpu_sync > 0.5 * wall_sync   # latency first when nothing else is in flight
    for (ps, _), (pd, _) in bufs:
        L.mibayer_host_free(ps)
        L.mibayer_host_free(pd)


def _pinned(L, nbytes, shape):
    import ctypes
    p = L.mibayer_host_alloc(nbytes)
    assert p
    return p, np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape)


@pytest.mark.parametrize("flags,threads", [(0, "0"), (1, "0"), (0, "1"), (1, "1")],
                         ids=["streams", "hipgraph", "streams_thread_per_shard", "hipgraph_thread_per_shard"])
def test_pool_round_robin_shards_keep_order(gpu_pkg, oracle, flags, threads, monkeypatch):
    """Multi-GPU logic without 8 GPUs: N logical shards on the visible device(s) (ordinals repeat), frames
    round-robin, results in submission order and bit-exact; with and without MIBAYER_FLAG_HIPGRAPH; driven by the
    calling thread alone or by a submit thread per shard (MIBAYER_POOL_THREADS=1, pinned frames)."""
    monkeypatch.setenv("MIBAYER_POOL_THREADS", threads)
    w, h, n = 1920, 1080, 23
    ndev = gpu_pkg.device_count()
    devices = [i % ndev for i in range(4)]
    src = oracle.fill_synthetic(w, h, n, seed=51)
    want = oracle.bayer2rgb_batch(src, w, "rggb", 2, 1, 0, nthreads=4)
    L = gpu_pkg.lib()
    with gpu_pkg.Pool(devices, w, h, "rggb", "BGRx", inflight=2, flags=flags) as pool:
        assert pool.capacity == 8
        bufs = [(_pinned(L, w * h, (h, w)), _pinned(L, 4 * w * h, (h, 4 * w))) for _ in range(pool.capacity)]
        outs, order = {}, []
        for i in range(n):
            if pool.pending() == pool.capacity:
                t = pool.wait()
                order.append(t)
                outs[t - 1] = bufs[(t - 1) % pool.capacity][1][1].copy()
            (_, s_arr), (_, d_arr) = bufs[i % pool.capacity]
            s_arr[:] = src[i]
            pool.submit(s_arr, d_arr, tag=i + 1)
        while pool.pending():
            t = pool.wait()
            order.append(t)
            outs[t - 1] = bufs[(t - 1) % pool.capacity][1][1].copy()
        assert order == list(range(1, n + 1))
        for i in range(n):
            assert np.array_equal(outs[i], want[i]), i
        with pytest.raises(gpu_pkg.MibayerError) as e:
            pool.wait()
        assert e.value.status == gpu_pkg.ERR_EMPTY
        for (ps, _), (pd, _) in bufs:
            L.mibayer_host_free(ps)
            L.mibayer_host_free(pd)


@pytest.mark.parametrize("memory", ["pinned", "pageable"])
def test_pool_drops_a_failed_device_and_redoes_its_frames(gpu_pkg, oracle, memory):
    """Failure drill on real hardware (SURVEY.md section 5 "a failed device is dropped from the round-robin set"): four
    shards on the visible GPU(s); shard 1 reports a device error after 3 frames, shard 3 after 7.  Every frame still
    comes back once, in submission order, bit-exact; the pool reports each drop once and shrinks its capacity; the
    stream fails only when the last shard is gone.  `pageable` buffers take the per-shard helper-thread path."""
    w, h, n = 1920, 1080, 41
    ndev = gpu_pkg.device_count()
    devices = [i % ndev for i in range(4)]
    src = oracle.fill_synthetic(w, h, n, seed=53)
    want = oracle.bayer2rgb_batch(src, w, "grbg", 0, 1, 2, nthreads=4)
    L = gpu_pkg.lib()
    with gpu_pkg.Pool(devices, w, h, "grbg", "RGBx", inflight=2) as pool:
        cap0 = pool.capacity
        if memory == "pinned":
            bufs = [(_pinned(L, w * h, (h, w)), _pinned(L, 4 * w * h, (h, 4 * w))) for _ in range(cap0)]
        else:
            bufs = [((0, np.empty((h, w), np.uint8)), (0, np.empty((h, 4 * w), np.uint8))) for _ in range(cap0)]
        pool.inject_fault(1, 3)
        pool.inject_fault(3, 7)
        outs, order, notes = {}, [], []

        def collect():
            t = pool.wait()
            order.append(t)
            outs[t - 1] = bufs[(t - 1) % cap0][1][1].copy()
            nf, dev, alive, msg = pool.take_failure()
            if nf:
                notes.append((nf, dev, alive, msg))

        for i in range(n):
            (_, s_arr), (_, d_arr) = bufs[i % cap0]
            while True:
                if pool.pending() >= min(cap0, lib_capacity(gpu_pkg, pool)):
                    collect()
                    continue
                s_arr[:] = src[i]
                d_arr[:] = 0
                try:
                    pool.submit(s_arr, d_arr, tag=i + 1)
                    break
                except gpu_pkg.MibayerError as e:
                    assert e.status == gpu_pkg.ERR_BUSY
                    collect()
        while pool.pending():
            collect()
        assert order == list(range(1, n + 1))
        for i in range(n):
            assert np.array_equal(outs[i], want[i]), i
        assert sum(x[0] for x in notes) == 2 and pool.alive() == 2 and lib_capacity(gpu_pkg, pool) == 4
        assert all("dropped from the rotation" in x[3] for x in notes)
        # the last two go as well: now, and only now, the stream is dead
        pool.inject_fault(0, 0)
        pool.inject_fault(2, 0)
        (_, s_arr), (_, d_arr) = bufs[0]
        with pytest.raises(gpu_pkg.MibayerError) as e:
            for k in range(6):
                pool.submit(s_arr, d_arr, tag=100 + k)
                pool.wait()
        assert e.value.status == gpu_pkg.ERR_HIP and pool.alive() == 0
        if memory == "pinned":
            for (ps, _), (pd, _) in bufs:
                L.mibayer_host_free(ps)
                L.mibayer_host_free(pd)


def test_pool_failover_randomised_drills(gpu_pkg, oracle):
    """Seeded drills on real hardware: 2-6 logical shards, 1-3 frames in flight each, pinned or pageable frames (or a
    mix: shards switch to their helper thread one by one), one to all-but-one shards failing after random frame counts.
    Every frame comes back once, in order, bit-exact; one report per dropped shard; the survivors carry on."""
    rng = np.random.default_rng(int(os.environ.get("MIBAYER_FUZZ_SEED", "4242")))
    w, h = 642, 50                      # small frames (and the generic kernel): the drill is about the pool
    L = gpu_pkg.lib()
    ndev = gpu_pkg.device_count()
    for drill in range(int(os.environ.get("MIBAYER_FUZZ_DRILLS", "12"))):
        nshards = int(rng.integers(2, 7))
        inflight = int(rng.integers(1, 4))
        n = int(rng.integers(20, 60))
        memory = ("pinned", "pageable", "mixed")[int(rng.integers(0, 3))]
        nfail = int(rng.integers(1, nshards))
        failing = sorted(rng.choice(nshards, size=nfail, replace=False).tolist())
        after = {s_: int(rng.integers(0, 8)) for s_ in failing}
        order_name = PATTERNS[int(rng.integers(0, 4))]
        stride = (w + 3) & ~3
        src = oracle.fill_synthetic(w, h, n, seed=200 + drill, stride=stride)
        want = oracle.bayer2rgb_batch(src, w, order_name, 2, 1, 0, nthreads=2)
        what = (drill, nshards, inflight, n, memory, after)
        with gpu_pkg.Pool([i % ndev for i in range(nshards)], w, h, order_name, "BGRx", inflight=inflight) as pool:
            cap0 = pool.capacity
            bufs, pinned = [], []
            for k in range(cap0):
                if memory == "pinned" or (memory == "mixed" and rng.integers(0, 2)):
                    a, b = _pinned(L, stride * h, (h, stride)), _pinned(L, 4 * w * h, (h, 4 * w))
                    pinned += [a[0], b[0]]
                    bufs.append((a[1], b[1]))
                else:
                    bufs.append((np.empty((h, stride), np.uint8), np.empty((h, 4 * w), np.uint8)))
            for s_, k in after.items():
                pool.inject_fault(s_, k)
            outs, order, reports = {}, [], 0

            def collect():
                nonlocal reports
                t = pool.wait()
                order.append(t)
                outs[t - 1] = bufs[(t - 1) % cap0][1].copy()
                reports += pool.take_failure()[0]

            for i in range(n):
                s_arr, d_arr = bufs[i % cap0]
                while True:
                    if pool.pending() >= min(cap0, lib_capacity(gpu_pkg, pool)):
                        collect()
                        continue
                    s_arr[:] = src[i]
                    d_arr[:] = 0
                    try:
                        pool.submit(s_arr, d_arr, tag=i + 1)
                        break
                    except gpu_pkg.MibayerError as e:
                        assert e.status == gpu_pkg.ERR_BUSY, what
                        collect()
            while pool.pending():
                collect()
            reports += pool.take_failure()[0]
            assert order == list(range(1, n + 1)), what
            for i in range(n):
                assert np.array_equal(outs[i], want[i]), (what, i)
            # a shard that never reached its failure point is still alive; every drop was reported exactly once
            assert pool.alive() == nshards - reports and 0 <= reports <= nfail, what
            assert pool.alive() >= nshards - nfail >= 1, what
        for p_ in pinned:
            L.mibayer_host_free(p_)


def lib_capacity(gpu_pkg, pool):
    return gpu_pkg.lib().mibayer_pool_capacity(pool._h)


def test_pageable_frames_go_through_helper_threads_and_match(gpu_pkg, oracle, monkeypatch):
    """Pageable numpy buffers: one helper thread per shard runs the blocking copies (MIBAYER_POOL_HELPERS=0 = the old
    behaviour, everything on the calling thread).  Same bytes, same order, both ways, with pinned frames (the direct,
    enqueue-only path) interleaved on the same shards."""
    w, h, n = 1280, 720, 19
    ndev = gpu_pkg.device_count()
    src = oracle.fill_synthetic(w, h, n, seed=57)
    want = oracle.bayer2rgb_batch(src, w, "bggr", 3, 2, 1, nthreads=4)
    L = gpu_pkg.lib()
    pinned = {i: (_pinned(L, w * h, (h, w)), _pinned(L, 4 * w * h, (h, 4 * w))) for i in range(2, n, 3)}
    for helpers in ("1", "0"):
        monkeypatch.setenv("MIBAYER_POOL_HELPERS", helpers)
        with gpu_pkg.Pool([i % ndev for i in range(3)], w, h, "bggr", "xBGR", inflight=2) as pool:
            dsts, order = [], []
            for i in range(n):
                if pool.pending() == pool.capacity:
                    order.append(pool.wait())
                if i in pinned:
                    s_arr, d_arr = pinned[i][0][1], pinned[i][1][1]
                    s_arr[:] = src[i]
                    d_arr[:] = 0
                else:
                    s_arr, d_arr = src[i], np.zeros((h, 4 * w), np.uint8)
                dsts.append(d_arr)
                pool.submit(s_arr, d_arr, tag=i + 1)
            while pool.pending():
                order.append(pool.wait())
            assert order == list(range(1, n + 1))
            for i in range(n):
                assert np.array_equal(dsts[i], want[i]), (helpers, i)
    for (ps, _), (pd, _) in pinned.values():
        L.mibayer_host_free(ps)
        L.mibayer_host_free(pd)


def test_numa_local_pinned_allocation(gpu_pkg, oracle):
    """mibayer_host_alloc_near(device): pinned memory on the NUMA node next to the GPU (a no-op on one-node hosts);
    the block works like any other pinned block and is freed by mibayer_host_free."""
    import ctypes
    L = gpu_pkg.lib()
    node = L.mibayer_device_numa_node(0)
    assert node >= -1
    w, h = 1920, 1080
    ps, pd = L.mibayer_host_alloc_near(0, w * h), L.mibayer_host_alloc_near(0, 4 * w * h)
    assert ps and pd
    s = np.ctypeslib.as_array(ctypes.cast(ps, ctypes.POINTER(ctypes.c_uint8)), (w * h,)).reshape(h, w)
    d = np.ctypeslib.as_array(ctypes.cast(pd, ctypes.POINTER(ctypes.c_uint8)), (4 * w * h,)).reshape(h, 4 * w)
    s[:] = oracle.fill_synthetic(w, h, 1, seed=59)[0]
    got_node = L.mibayer_host_numa_node(ctypes.c_void_p(pd))
    if node >= 0 and got_node >= 0:
        assert got_node == node, (got_node, node)          # the pages sit next to the GPU
    with gpu_pkg.Context(w, h, "rggb", "BGRx", device=0) as ctx:
        ctx.process_host(s, d)
    assert np.array_equal(d, oracle.bayer2rgb(s, w, "rggb", 2, 1, 0))
    L.mibayer_host_free(ps)
    L.mibayer_host_free(pd)
    assert L.mibayer_device_numa_node(99) == -1
    p = L.mibayer_host_alloc_near(99, 4096)                 # unknown device: plain pinned memory
    assert p
    L.mibayer_host_free(p)


@pytest.mark.parametrize("mechanism", ["streams", "hipgraph", "hipgraph_chain"])
def test_config5_4k_stream_pinned_double_buffered_ring(gpu_pkg, oracle, mechanism, monkeypatch):
    """BASELINE.json configs[4] at full size on one GPU's share: a 3840x2160 stream of 72 frames through TWO pinned
    source and TWO pinned destination buffers (double-buffered H2D/D2H, two frames in flight), every output frame
    compared with the oracle -- for the streams+events chain, for the hipGraph-captured launch (the compute-queue
    segment of each slot as a graph) and for the whole-chain-per-slot graph arm."""
    w, h, n, chunk = 3840, 2160, 72, 12
    flags = 0 if mechanism == "streams" else gpu_pkg.FLAG_HIPGRAPH
    if mechanism == "hipgraph_chain":
        flags |= gpu_pkg.FLAG_HIPGRAPH_CHAIN
    L = gpu_pkg.lib()
    r, g, b = gpu_pkg.FORMATS["BGRx"]
    srcs = [_pinned(L, w * h, (h, w)) for _ in range(2)]
    dsts = [_pinned(L, 4 * w * h, (h, 4 * w)) for _ in range(2)]
    nthreads = min(os.cpu_count() or 1, 64)
    with gpu_pkg.Context(w, h, "gbrg", "BGRx", inflight=2, flags=flags) as ctx:
        checked = 0
        for c0 in range(0, n, chunk):
            frames = oracle.fill_synthetic(w, h, chunk, seed=4, first_frame=c0)
            want = oracle.bayer2rgb_batch_bands(frames, w, "gbrg", r, g, b, nbands=4, nthreads=nthreads,
                                                mode=oracle.simd_isas()[-1])
            for i in range(chunk):
                if ctx.pending() == 2:
                    t = ctx.wait()
                    assert np.array_equal(dsts[t % 2][1], want[t - c0]), (mechanism, t)
                    checked += 1
                srcs[i % 2][1][:] = frames[i]
                ctx.submit(srcs[i % 2][1], dsts[i % 2][1], tag=c0 + i)
            while ctx.pending():        # the chunk's last two frames, before `want` is replaced
                t = ctx.wait()
                assert np.array_equal(dsts[t % 2][1], want[t - c0]), (mechanism, t)
                checked += 1
        assert checked == n
    for p, _ in srcs + dsts:
        L.mibayer_host_free(p)


def test_list_launch_over_separately_allocated_frames(gpu_pkg, gpu_lab_pkg, oracle):
    """mibayer_process_device_list: one launch per 16 frames that are separate device allocations (hipbayer2rgb
    batch=N).  21 frames (split 16 + 5), every Bayer order; then with one pointer only 4-byte aligned (the generic
    kernel takes the launch) and for a width that always needs the generic kernel."""
    for (w, h, n, pat, fmt, misalign) in ((1920, 1080, 21, "rggb", "BGRx", 0), (1920, 1080, 5, "gbrg", "xRGB", 4),
                                          (642, 50, 7, "grbg", "RGBx", 0), (3840, 2160, 3, "bggr", "xBGR", 0)):
        stride = (w + 3) & ~3
        src = oracle.fill_synthetic(w, h, n, seed=63, stride=stride)
        r, g, b = gpu_pkg.FORMATS[fmt]
        with gpu_pkg.Context(w, h, pat, fmt) as ctx:
            d_srcs = [ctx.device_alloc(ctx.src_bytes + 64) for _ in range(n)]
            d_dsts = [ctx.device_alloc(ctx.dst_bytes + 64) for _ in range(n)]
            offs = [misalign if f == n // 2 else 0 for f in range(n)]
            for f in range(n):
                ctx.to_device(d_srcs[f] + offs[f], src[f])
            ctx.process_device_list([p + o for p, o in zip(d_srcs, offs)], [p + o for p, o in zip(d_dsts, offs)])
            ctx.sync()
            for f in range(n):
                got = ctx.from_device(d_dsts[f] + offs[f], ctx.dst_bytes).reshape(h, 4 * w)
                assert np.array_equal(got, oracle.bayer2rgb(src[f], w, pat, r, g, b)), (w, h, pat, f)
            for p in d_srcs + d_dsts:
                ctx.device_free(p)
    # the inverse direction: the same entry point, up to 16 separately allocated frames per launch of the flat kernel
    # (21 = 16 + 5), widths that are not a multiple of 4 (partial last item, mosaic rows padded to 4), one frame at a
    # 4-byte-aligned address (the dword-wise arm), every Bayer order; MIBAYER_R2B_FLAT=0: frame by frame (tile kernel)
    cases = ((640, 480, 21, "rggb", 0, None), (1918, 50, 5, "gbrg", 4, None), (3840, 2160, 3, "bggr", 0, None),
             (130, 21, 17, "grbg", 8, None), (641, 37, 4, "rggb", 0, None), (640, 480, 3, "gbrg", 0, "0"))
    for (w, h, n, pat, misalign, flat) in cases:
        rgb = np.random.default_rng(64).integers(0, 256, (n, h, 4 * w), dtype=np.uint8)
        if flat is not None:
            os.environ["MIBAYER_R2B_FLAT"] = flat
        try:                # (the tile-kernel arm is a knob of the lab build)
            ctx_cm = (gpu_lab_pkg if flat is not None else gpu_pkg).Context(w, h, pat, (1, 2, 3),
                                                                             flags=gpu_pkg.FLAG_RGB2BAYER)
        finally:
            os.environ.pop("MIBAYER_R2B_FLAT", None)
        with ctx_cm as ctx:
            d_srcs = [ctx.device_alloc(ctx.src_bytes + 64) for _ in range(n)]
            d_dsts = [ctx.device_alloc(ctx.dst_bytes + 64) for _ in range(n)]
            offs = [misalign if f == n // 2 else 0 for f in range(n)]
            for f in range(n):
                ctx.to_device(d_srcs[f] + offs[f], rgb[f])
                ctx.to_device(d_dsts[f], np.full(ctx.dst_bytes + 64, 0xC3, np.uint8))
            ctx.process_device_list([p + o for p, o in zip(d_srcs, offs)], [p + o for p, o in zip(d_dsts, offs)])
            ctx.sync()
            for f in range(n):
                raw = ctx.from_device(d_dsts[f], ctx.dst_bytes + 64)
                got = raw[offs[f]:offs[f] + ctx.dst_bytes].reshape(h, ctx.dst_stride)
                assert np.array_equal(got[:, :w], oracle.rgb2bayer(rgb[f], w, pat, 1, 2, 3)[:, :w]), (w, h, pat, f)
                assert (raw[:offs[f]] == 0xC3).all() and (raw[offs[f] + ctx.dst_bytes:] == 0xC3).all(), (w, h, f)
            for p in d_srcs + d_dsts:
                ctx.device_free(p)


def test_hipgraph_host_path_with_changing_pointers(gpu_pkg, oracle):
    """Graph mode patches the two host pointers into the instantiated graph per frame: pageable and pinned,
    fresh and recycled pointers must all give the oracle's bytes."""
    w, h = 642, 50          # W % 4 == 2 -> generic kernel inside the graph
    src = oracle.fill_synthetic(w, h, 6, seed=61, stride=644)
    want = [oracle.bayer2rgb(f, w, "gbrg", 0, 1, 2) for f in src]
    with gpu_pkg.Context(w, h, "gbrg", "RGBx", src_stride=644, inflight=2, flags=gpu_pkg.FLAG_HIPGRAPH) as ctx:
        for i in range(6):
            got = ctx.process_host(src[i].copy())           # new pageable buffers every frame
            assert np.array_equal(got, want[i]), i
        outs = [np.zeros((h, 4 * w), np.uint8) for _ in range(6)]
        for i in range(6):
            if ctx.pending() == 2:
                ctx.wait()
            ctx.submit(src[i], outs[i], tag=i + 1)
        while ctx.pending():
            ctx.wait()
        for i in range(6):
            assert np.array_equal(outs[i], want[i]), i


def test_randomised_geometries_strides_variants(gpu_pkg, oracle):
    """Seeded fuzz over the whole configuration space of the ABI: even widths 4..4100, heights 3..130, padded
    source and destination strides, all orders and formats, every kernel variant, host and device paths."""
    rng = np.random.default_rng(int(os.environ.get("MIBAYER_FUZZ_SEED", "20260926")))
    nvar = len(gpu_pkg.variant_names())
    for case in range(int(os.environ.get("MIBAYER_FUZZ_CASES", "60"))):      # soak runs: tools/README.md
        w = int(rng.integers(2, 2051)) * 2
        h = int(rng.integers(3, 131))
        if case % 10 == 0:
            w = int(rng.choice([4, 254, 256, 258, 1022, 1024, 1026, 4096, 4100]))
        pat = PATTERNS[int(rng.integers(0, 4))]
        fmt = ALL_FORMATS[int(rng.integers(0, 8))]
        sstride = ((w + 3) & ~3) + 4 * int(rng.integers(0, 5)) * int(rng.integers(0, 2))
        dstride = 4 * w + 4 * int(rng.integers(0, 9)) * int(rng.integers(0, 2))
        variant = int(rng.integers(0, nvar))
        via = "host" if rng.integers(0, 2) else "device"
        src = rng.integers(0, 256, (h, sstride), dtype=np.uint8)
        check(gpu_pkg, oracle, src, w, pat, fmt, variant=variant, dst_stride=dstride, via=via)


def test_guard_bands_around_the_destination_stay_intact(gpu_pkg, oracle):
    """No out-of-bounds stores: 4 KiB guard bands before and after the device destination (and the row padding
    inside it) keep their fill for edge geometries, both kernel paths, every production shape and the A/B arms."""
    rng = np.random.default_rng(77)
    guard = 4096
    names = gpu_pkg.variant_names()
    cases = [(4, 3), (6, 5), (254, 9), (258, 33), (1022, 8), (1026, 17), (2050, 12), (3840, 10), (512, 64)]
    for (w, h) in cases:
        sstride = (w + 3) & ~3
        dstride = 4 * w + 32
        src = rng.integers(0, 256, (h, sstride), dtype=np.uint8)
        want = oracle.bayer2rgb(src, w, "grbg", 2, 1, 0)
        for v in range(len(names)):
            with gpu_pkg.Context(w, h, "grbg", "BGRx", src_stride=sstride, dst_stride=dstride, variant=v) as ctx:
                d_src = ctx.device_alloc(ctx.src_bytes)
                d_all = ctx.device_alloc(ctx.dst_bytes + 2 * guard)
                ctx.to_device(d_src, src)
                ctx.to_device(d_all, np.full(ctx.dst_bytes + 2 * guard, 0xC3, np.uint8))
                ctx.process_device(d_src, d_all + guard, 1)
                ctx.sync()
                out = ctx.from_device(d_all, ctx.dst_bytes + 2 * guard)
                ctx.device_free(d_src)
                ctx.device_free(d_all)
            assert (out[:guard] == 0xC3).all() and (out[-guard:] == 0xC3).all(), (w, h, names[v])
            body = out[guard:-guard].reshape(h, dstride)
            assert np.array_equal(body[:, :4 * w], want), (w, h, names[v])
            assert (body[:, 4 * w:] == 0xC3).all(), (w, h, names[v])


@pytest.mark.gpu
@pytest.mark.parametrize("align", ["0", "64", "128"])
def test_sector_aligned_store_arm(gpu_lab_pkg, oracle, align, monkeypatch):
    """Generic geometries whose output rows start off the 64-byte sector grid (width % 16 != 0, padded pitches,
    destinations at odd offsets): the per-row lane shift of bayer2rgb_lds_aligned_kernel -- every wave-store on a
    64- / 128-byte boundary, the columns in front of the first boundary as the row's head -- against the oracle, with
    guard bands, for every production shape and its plain-store twin, single frames, batches whose frame pitch moves
    the phase from frame to frame, and list launches.  MIBAYER_ALIGN_STORES=0 is the unshifted generic arm.  (Lab build:
    the forced alignment is a tuning knob; the product build reaches the 128-byte arm through mibayer_set_plan and
    the autotuner, test_shifted_arm_through_set_plan below.)"""
    gpu_pkg = gpu_lab_pkg
    monkeypatch.setenv("MIBAYER_ALIGN_STORES", align)
    rng = np.random.default_rng(314)
    guard = 4096
    names = gpu_pkg.variant_names()
    shapes = [0] + [names.index(n) for n in ("lds_4x2_r4_dpp_nt", "lds_2x4_r4_dpp_nt", "lds_1x8_r4_dpp_nt",
                                             "lds_1x8_r4_dpp", "lds_4x2_r4_dpp", "lds_2x4_r4_dpp",
                                             "lds_4x2_r4_dpp_hy", "lds_2x4_r4_dpp_hy", "lds_1x8_r4_dpp_hy")]
    # (w, h, extra destination pitch, destination offset inside the allocation, frames)
    cases = [(4056, 9, 0, 0, 1), (3838, 10, 0, 8, 1), (1366, 11, 0, 0, 3), (30, 7, 0, 40, 2), (6, 4, 0, 8, 3),
             (4, 3, 8, 24, 2), (270, 5, 24, 56, 1), (1026, 17, 8, 0, 2), (2050, 12, 40, 16, 1), (1030, 33, 0, 120, 2),
             (258, 35, 0, 0, 4), (1290, 8, 104, 72, 1), (14, 40, 0, 8, 1), (1024 + 18, 9, 0, 0, 1),
             # rows of four-column lanes at an 8-byte phase, a wave of four columns
             (4056, 9, 8, 8, 2), (1024, 6, 8, 8, 1), (260, 7, 0, 8, 2), (516, 5, 24, 0, 1), (8, 6, 0, 8, 1)]
    for ci, (w, h, pad, off, n) in enumerate(cases):
        sstride = (w + 3) & ~3
        dstride = 4 * w + pad
        src = rng.integers(0, 256, (n, h, sstride), dtype=np.uint8)
        pat = PATTERNS[ci % 4]
        fmt = LAYOUTS[(ci // 2) % 4]
        r, g, b = gpu_pkg.FORMATS[fmt]
        want = np.stack([oracle.bayer2rgb(src[f], w, pat, r, g, b) for f in range(n)])
        for v in shapes:
            with gpu_pkg.Context(w, h, pat, fmt, src_stride=sstride, dst_stride=dstride, variant=v) as ctx:
                fb = ctx.dst_bytes + (8 if (ci & 1) else 0)      # frame pitch: moves the row phase per frame
                total = n * fb
                d_src = ctx.device_alloc(n * ctx.src_bytes)
                d_all = ctx.device_alloc(total + 2 * guard + 128)
                ctx.to_device(d_src, src)
                ctx.to_device(d_all, np.full(total + 2 * guard + 128, 0xC3, np.uint8))
                ctx.process_device(d_src, d_all + guard + off, n, dst_frame_bytes=fb)
                ctx.sync()
                out = ctx.from_device(d_all, total + 2 * guard + 128)
                key = (w, h, pad, off, n, names[v], align)
                assert (out[:guard + off] == 0xC3).all() and (out[guard + off + total:] == 0xC3).all(), key
                for f in range(n):
                    body = out[guard + off + f * fb: guard + off + f * fb + ctx.dst_bytes].reshape(h, dstride)
                    bad = np.argwhere(body[:, :4 * w] != want[f])
                    assert bad.size == 0, (key, f, bad[:4].tolist())
                    assert (body[:, 4 * w:] == 0xC3).all(), key
                    assert (out[guard + off + f * fb + ctx.dst_bytes: guard + off + (f + 1) * fb] == 0xC3).all(), key
                if n > 1 and v in (0, shapes[1]):
                    # the same frames as separate allocations at different phases: one list launch
                    ctx.to_device(d_all, np.full(total + 2 * guard + 128, 0xC3, np.uint8))
                    srcs = [d_src + f * ctx.src_bytes for f in range(n)]
                    dsts = [d_all + guard + off + f * fb for f in range(n)]
                    ctx.process_device_list(srcs, dsts)
                    ctx.sync()
                    out = ctx.from_device(d_all, total + 2 * guard + 128)
                    for f in range(n):
                        body = out[guard + off + f * fb: guard + off + f * fb + ctx.dst_bytes].reshape(h, dstride)
                        assert np.array_equal(body[:, :4 * w], want[f]), (key, "list", f)
                    assert (out[:guard + off] == 0xC3).all() and (out[guard + off + total:] == 0xC3).all(), key
                ctx.device_free(d_src)
                ctx.device_free(d_all)


@pytest.mark.gpu
def test_shifted_arm_through_set_plan(gpu_pkg, oracle):
    """The product build's way to the sector-aligned arm (no tuning environment): mibayer_set_plan (variant, band,
    128) -- what mibayer_autotune picks among and what the plan cache hands to later contexts -- on generic geometries
    with guard bands; plans that do not exist (64-byte flavour, shapes without the arm) are refused."""
    rng = np.random.default_rng(2718)
    names = gpu_pkg.variant_names()
    guard = 2048
    for (w, h, pad, off, n) in ((4056, 9, 0, 0, 2), (3838, 10, 0, 8, 1), (1366, 11, 0, 0, 3), (1030, 33, 0, 120, 2),
                                (258, 35, 0, 0, 4)):
        sstride, dstride = (w + 3) & ~3, 4 * w + pad
        src = rng.integers(0, 256, (n, h, sstride), dtype=np.uint8)
        want = np.stack([oracle.bayer2rgb(src[f], w, "grbg", 2, 1, 0) for f in range(n)])
        for vname in ("lds_2x4_r4_dpp_nt", "lds_1x8_r4_dpp_nt"):
            for band in (0, 1):
                with gpu_pkg.Context(w, h, "grbg", "BGRx", src_stride=sstride, dst_stride=dstride) as ctx:
                    ctx.set_plan(names.index(vname), band, 128)
                    assert ctx.get_plan() == (names.index(vname), band, 128) and ctx.plan_source == gpu_pkg.PLAN_SET
                    total = n * ctx.dst_bytes
                    d_src = ctx.device_alloc(n * ctx.src_bytes)
                    d_all = ctx.device_alloc(total + 2 * guard + 128)
                    ctx.to_device(d_src, src)
                    ctx.to_device(d_all, np.full(total + 2 * guard + 128, 0xC3, np.uint8))
                    ctx.process_device(d_src, d_all + guard + off, n)
                    ctx.sync()
                    out = ctx.from_device(d_all, total + 2 * guard + 128)
                    assert (out[:guard + off] == 0xC3).all() and (out[guard + off + total:] == 0xC3).all()
                    body = out[guard + off: guard + off + total].reshape(n, h, dstride)
                    assert np.array_equal(body[:, :, :4 * w], want), (w, h, vname, band)
                    ctx.device_free(d_src)
                    ctx.device_free(d_all)
    with gpu_pkg.Context(4056, 16, "grbg", "BGRx") as ctx:
        refused = [(names.index("lds_1x8_r4_dpp_hy"), 0, 128), (0, 0, 0), (len(names), 0, 0), (1, 0, 32)]
        if not gpu_pkg.lib().mibayer_is_lab_build():        # the 64-byte flavour exists in the lab build only
            refused.append((names.index("lds_1x8_r4_dpp_nt"), 0, 64))
        for bad in refused:
            with pytest.raises(gpu_pkg.MibayerError):
                ctx.set_plan(*bad)


GRAPH_SCRIPT = r"""
import sys
import numpy as np
import torch                      # first: the process then runs on torch's bundled HIP runtime (as bench.py does)
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as entry
pkg, oracle = entry.load_package(), entry.load_oracle()
w, h, n = 1280, 720, 4
src = oracle.fill_synthetic(w, h, n, seed=81)
want = oracle.bayer2rgb_batch(src, w, "rggb", 2, 1, 0, nthreads=2)
d_src = torch.from_numpy(src.reshape(-1)).cuda()
d_dst = torch.zeros(n * h * 4 * w, dtype=torch.uint8, device="cuda")
with pkg.Context(w, h, "rggb", "BGRx", device=0) as ctx:
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        ctx.process_device(d_src.data_ptr(), d_dst.data_ptr(), n, stream=side.cuda_stream)      # warm-up
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for f in range(n):                   # four single-frame launches inside one graph
                ctx.process_device(d_src.data_ptr() + f * ctx.src_bytes, d_dst.data_ptr() + f * ctx.dst_bytes, 1,
                                   stream=side.cuda_stream)
    d_dst.zero_()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(d_dst.cpu().numpy().reshape(want.shape), want), "first replay"
    src2 = oracle.fill_synthetic(w, h, n, seed=82)             # new input, same graph
    d_src.copy_(torch.from_numpy(src2.reshape(-1)))
    graph.replay()
    torch.cuda.synchronize()
    want2 = oracle.bayer2rgb_batch(src2, w, "rggb", 2, 1, 0, nthreads=2)
    assert np.array_equal(d_dst.cpu().numpy().reshape(want2.shape), want2), "second replay"
print("graph replay ok")
"""


def test_device_launch_can_be_captured_in_a_hipgraph(gpu_pkg):
    """mibayer_process_device only enqueues a kernel on the caller's stream, so a caller may capture it in a
    hipGraph (here through torch.cuda.graph) and replay it: BASELINE.json configs[4] "hipGraph-captured launch"
    for device-resident consumers.  Runs in its own process with torch imported first, the way bench.py runs
    (one HIP runtime per process: whichever of torch's bundled libamdhip64 / the system one is loaded first)."""
    import subprocess
    import sys
    res = subprocess.run([sys.executable, "-c", GRAPH_SCRIPT, ROOT], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout + res.stderr)[-2000:]
    assert "graph replay ok" in res.stdout


@pytest.mark.parametrize("flags", [0, 1], ids=["streams", "hipgraph"])
def test_contexts_on_several_threads_share_the_device_queues(gpu_pkg, oracle, flags):
    """The reference's threading contract: several element instances run concurrently on
    different streaming threads.  Here every context of a device shares that device's three queues
    (csrc/mibayer_abi.hip, DeviceQueues), so: four threads, each with its own context (own geometry, order and layout),
    stream frames through submit / wait at the same time while a fifth thread creates, uses and destroys short-lived
    contexts (the queue set's reference count goes up and down under load).  Every frame of every thread is
    bit-exact and in order."""
    import threading
    jobs = [(640, 480, "bggr", "RGBx", (0, 1, 2), 40), (1920, 1080, "rggb", "BGRx", (2, 1, 0), 24),
            (1282, 722, "grbg", "xRGB", (1, 2, 3), 24), (3840, 2160, "gbrg", "xBGR", (3, 2, 1), 10)]
    inputs = []
    for k, (w, h, order, fmt, offs, n) in enumerate(jobs):
        src = oracle.fill_synthetic(w, h, n, seed=70 + k)
        want = oracle.bayer2rgb_batch(src, w, order, *offs, nthreads=4)
        inputs.append((src, want))
    errors, stop = [], threading.Event()

    def stream(k):
        try:
            w, h, order, fmt, _, n = jobs[k]
            src, want = inputs[k]
            stride = (w + 3) & ~3
            with gpu_pkg.Context(w, h, order, fmt, inflight=3, flags=flags) as c:
                outs = [np.zeros((h, 4 * w), np.uint8) for _ in range(3)]
                padded = [np.zeros((h, stride), np.uint8) for _ in range(3)]
                done = 0
                for i in range(n):
                    if c.pending() == 3:
                        t = c.wait()
                        assert t == done + 1
                        assert np.array_equal(outs[done % 3], want[done]), (k, done)
                        done += 1
                    padded[i % 3][:, :w] = src[i].reshape(h, -1)[:, :w]
                    c.submit(padded[i % 3], outs[i % 3], tag=i + 1)
                while c.pending():
                    t = c.wait()
                    assert t == done + 1
                    assert np.array_equal(outs[done % 3], want[done]), (k, done)
                    done += 1
                assert done == n
        except Exception as exc:            # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(exc)))

    def churn():
        try:
            src, want = inputs[0]
            w, h, order, fmt = jobs[0][:4]
            while not stop.is_set():
                with gpu_pkg.Context(w, h, order, fmt) as c:
                    assert np.array_equal(c.process_host(src[0]), want[0])
        except Exception as exc:            # noqa: BLE001
            errors.append(("churn", repr(exc)))

    threads = [threading.Thread(target=stream, args=(k,)) for k in range(len(jobs))]
    extra = threading.Thread(target=churn)
    extra.start()
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    stop.set()
    extra.join(60)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads) and not extra.is_alive()


WAIT_DEADLINE_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as entry
pkg, oracle = entry.load_package(), entry.load_oracle()
def pinned(nbytes, shape):
    import ctypes
    p = L.mibayer_host_alloc(nbytes)
    assert p
    return p, np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape)
def status_of(call, *args):
    try:
        call(*args)
    except pkg.MibayerError as e:
        return e.status
    return pkg.OK
import time
w, h = 640, 480
src = oracle.fill_synthetic(w, h, 1, seed=91)[0]
want = oracle.bayer2rgb(src, w, "rggb", 2, 1, 0)
L = pkg.lib()
ctx = pkg.Context(w, h, "rggb", "BGRx", inflight=2)
assert np.array_equal(ctx.process_host(src), want)         # ring allocated, device warm
ctx.set_wait_timeout(120)
ctx.stall(1500)
p_src, host_src = pinned(src.size, src.shape)
host_src[...] = src
p_dst, host_dst = pinned(want.size, want.shape)
t0 = time.monotonic()
ctx.submit(host_src, host_dst, 1)
st = status_of(ctx.wait)
dt = time.monotonic() - t0
assert st == pkg.ERR_TIMEOUT and 0.1 <= dt < 0.8, (st, dt)
t0 = time.monotonic()
assert status_of(ctx.wait) == pkg.ERR_TIMEOUT
assert status_of(ctx.submit, host_src, host_dst, 2) == pkg.ERR_TIMEOUT
ctx.close()
assert time.monotonic() - t0 < 0.3                          # nothing waited for the stalled device
# per-device, not for ever (ADVICE r03): while the device has not caught up, ONE wedge is outstanding and a pinned
# block handed back is parked on the deferred list instead of hipHostFree (which would wait for that device)
assert L.mibayer_wedged_contexts() == 1
p_tmp, _ = pinned(4096, (4096,))
t0 = time.monotonic()
L.mibayer_host_free(p_tmp)
assert time.monotonic() - t0 < 0.2 and L.mibayer_deferred_frees() == 1
time.sleep(1.6)                                             # the drill ends; the late DMA lands in host_dst
# the device has caught up: the wedge settles by itself, the orphaned ring of the destroyed context and the parked
# block are released (polled from here, from create / destroy / host_free; nothing blocks)
assert L.mibayer_wedged_contexts() == 0 and L.mibayer_deferred_frees() == 0
with pkg.Context(w, h, "rggb", "BGRx") as ctx2:
    assert np.array_equal(ctx2.process_host(src), want)
# destroy itself is what finds the device stalled (a frame in flight, nobody waited for it): bounded, and the
# frames the device may still write are NOT handed to the next context of this geometry
ctx3 = pkg.Context(w, h, "rggb", "BGRx", inflight=2)
assert np.array_equal(ctx3.process_host(src), want)
ctx3.set_wait_timeout(100)
ctx3.stall(1200)
host_dst[...] = 0
ctx3.submit(host_src, host_dst, 3)
t0 = time.monotonic()
ctx3.close()
assert 0.08 <= time.monotonic() - t0 < 0.6
src_b = oracle.fill_synthetic(w, h, 1, seed=93)[0]
with pkg.Context(w, h, "rggb", "BGRx", inflight=2) as ctx4:     # queued behind the drill on the shared queues
    ctx4.set_wait_timeout(5000)
    assert np.array_equal(ctx4.process_host(src_b), oracle.bayer2rgb(src_b, w, "rggb", 2, 1, 0))
assert np.array_equal(host_dst, want)                       # the abandoned frame did land, late, where it belonged
assert L.mibayer_wedged_contexts() == 0
# a spurious timeout heals: the context that ran into the deadline works again once the device has caught up, and the
# frame it still holds is collected, in order, with the right bytes
ctx5 = pkg.Context(w, h, "rggb", "BGRx", inflight=2)
assert np.array_equal(ctx5.process_host(src), want)
ctx5.set_wait_timeout(100)
ctx5.stall(700)
host_dst[...] = 0
ctx5.submit(host_src, host_dst, 7)
assert status_of(ctx5.wait) == pkg.ERR_TIMEOUT and status_of(ctx5.wait) == pkg.ERR_TIMEOUT
assert L.mibayer_wedged_contexts() == 1
time.sleep(0.8)
ctx5.set_wait_timeout(5000)
assert ctx5.wait() == 7 and np.array_equal(host_dst, want) and L.mibayer_wedged_contexts() == 0
host_dst[...] = 0
ctx5.submit(host_src, host_dst, 8)
assert ctx5.wait() == 8 and np.array_equal(host_dst, want)
ctx5.close()
# the same for a DEVICE-RESIDENT context (hipbayer2rgb's mibayer_sync, Context.sync(); ADVICE r04): nothing is pending
# in its ring, so mibayer_sync itself must find out that the device has caught up
ctx6 = pkg.Context(w, h, "rggb", "BGRx")
d_src, d_dst = ctx6.device_alloc(ctx6.src_bytes), ctx6.device_alloc(ctx6.dst_bytes)
ctx6.to_device(d_src, src)
ctx6.process_device(d_src, d_dst, 1)
ctx6.sync()
ctx6.set_wait_timeout(100)
ctx6.stall(700)
ctx6.process_device(d_src, d_dst, 1)
assert status_of(ctx6.sync) == pkg.ERR_TIMEOUT and status_of(ctx6.sync) == pkg.ERR_TIMEOUT
assert L.mibayer_wedged_contexts() == 1
time.sleep(0.8)
assert status_of(ctx6.sync) == pkg.OK and L.mibayer_wedged_contexts() == 0     # healed, within the 100 ms deadline
ctx6.process_device(d_src, d_dst, 1)
ctx6.sync()
assert np.array_equal(ctx6.from_device(d_dst, ctx6.dst_bytes).reshape(want.shape), want)
ctx6.device_free(d_src); ctx6.device_free(d_dst)
ctx6.close()
L.mibayer_host_free(p_src)
L.mibayer_host_free(p_dst)
assert L.mibayer_deferred_frees() == 0                      # really freed: no wedge is outstanding any more
print("wait deadline drill ok")
"""


def test_wait_deadline_on_a_stalled_device(gpu_pkg):
    """A device that does not hand a frame back within the deadline: mibayer_wait returns MIBAYER_ERR_TIMEOUT (it
    does not hang), the context is wedged from then on -- every later call returns at once -- and destroying it
    does not block either.  The stall drill (csrc/mibayer_hooks.h) ends by itself; afterwards the device converts
    bit-exactly again."""
    # in a process of its own: a context that is still wedged when the process ends leaves its ring to the wedge
    # registry, which the other tests of this run should not inherit
    # (the stall drill is part of the pool's seam, csrc/mibayer_hooks.h: exported by the lab build only)
    res = subprocess.run([sys.executable, "-c", WAIT_DEADLINE_SCRIPT, ROOT], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, MIBAYER_LIB_PATH=gpu_pkg.LAB_LIB_PATH))
    assert res.returncode == 0 and "wait deadline drill ok" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]


POOL_STALL_SCRIPT = r"""
import sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as entry
pkg, oracle = entry.load_package(), entry.load_oracle()
L = pkg.lib()
w, h, n = 1280, 720, 24
src = oracle.fill_synthetic(w, h, n, seed=92)
want = oracle.bayer2rgb_batch(src, w, "grbg", 0, 1, 2, nthreads=2)
def pinned(nbytes, shape):
    import ctypes
    p = L.mibayer_host_alloc(nbytes)
    return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape)
hs = pinned(src.size, src.shape); hs[...] = src
hd = pinned(want.size, want.shape); hd[...] = 0
def stream(stall_at):
    sub = got = 0
    status = pkg.OK
    lost = []
    with pkg.Pool([0, 0], w, h, "grbg", "RGBx", inflight=2) as pool:
        pool.set_wait_timeout(150)
        t0 = time.monotonic()
        try:
            while got < n:
                if sub == stall_at:
                    pool.inject_stall(0, 2500)
                    stall_at = -1
                if sub < n:
                    try:
                        pool.submit(hs[sub], hd[sub], sub + 1)
                        sub += 1
                        continue
                    except pkg.MibayerError as e:
                        if e.status != pkg.ERR_BUSY:
                            raise
                try:
                    tag = pool.wait()
                except pkg.FrameLost as e:      # in flight on the stalled shard: dropped, in order, buffers quarantined
                    tag = e.tag
                    lost.append(tag)
                    if pool.alive() == 0:       # ... and that was the last device: the stream has failed
                        got += 1
                        raise
                assert tag == got + 1, (tag, got)
                got += 1
        except pkg.MibayerError as e:
            status = e.status
        dt = time.monotonic() - t0
        nf, dev, alive, msg = pool.take_failure()
        still_lost = pool.lost()
        t1 = time.monotonic()
    return got, status, dt, time.monotonic() - t1, nf, alive, msg, lost, still_lost
got, status, dt, dt_destroy, nf, alive, msg, lost, still_lost = stream(6)
# ONE device: the stalled shard's copies sit at the head of the device's DMA engines, so its twin stalls with it --
# what this run must show is that nothing hangs: the stream ends with a status, in bounded time
assert dt < 1.5 and dt_destroy < 0.5, (dt, dt_destroy)
assert status in (pkg.OK, pkg.ERR_HIP, pkg.ERR_TIMEOUT), status
assert nf >= 1 and ("did not" in msg or "deadline" in msg), (nf, msg)
delivered = [f for f in range(got) if f + 1 not in lost]
assert all(np.array_equal(hd[f], want[f]) for f in delivered)
assert still_lost == len(lost)                      # nothing was handed back while the device held it
assert L.mibayer_wedged_contexts() >= 1
time.sleep(2.7)                                     # the drill ends; late DMAs land in hd, which is still ours
assert L.mibayer_wedged_contexts() == 0             # ... and the wedges settle by themselves
# ADVICE r03: frames that were in flight on the stalled device were NOT converted again behind its back and handed
# back; their quarantined destinations now hold what the late download wrote -- their own frame
for tag in lost:
    assert np.array_equal(hd[tag - 1], want[tag - 1]), tag
hd[...] = 0
got2, status2, dt2, _, nf2, alive2, _, lost2, _ = stream(-1)
assert got2 == n and status2 == pkg.OK and nf2 == 0 and not lost2
assert np.array_equal(hd, want)
print("pool stall drill ok: %d frames handed back (%d lost) before the stall took the device, status %d after %.3f s (%s)"
      % (got, len(lost), status, dt, msg))
"""


def test_pool_never_hangs_on_a_shard_that_stops_answering(gpu_pkg):
    """devices=0,0, shard 0's compute queue occupied for 2.5 s, wait deadline 150 ms.  With ONE physical device the
    second shard cannot survive -- the stalled shard's download waits at the head of the device's D2H engine, so
    its twin's copies queue behind it (two logical shards meet again in the DMA engines and hardware queues; two
    real GPUs do not, and the survivor logic is exercised on the context double under ASan / TSan,
    tests/test_pool_logic.py).  What must hold on real HIP calls: every wait is bounded, the stream ends with a
    status instead of hanging, the frames delivered before are bit-exact and in order, the frames that were in flight
    on the stalled device come back LOST instead of being converted again behind its back (their late downloads land in
    the quarantined buffers), mibayer_pool_destroy does not block on the wedged contexts, the wedges settle by
    themselves once the stall is over, and a new pool then converts everything."""
    res = subprocess.run([sys.executable, "-c", POOL_STALL_SCRIPT, ROOT], capture_output=True, text=True,
                         timeout=120, env=dict(os.environ, MIBAYER_SHARED_QUEUES="0",
                                               MIBAYER_LIB_PATH=gpu_pkg.LAB_LIB_PATH))     # (a knob of the lab build)
    assert res.returncode == 0 and "pool stall drill ok" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]


def test_destroying_a_context_does_not_wait_for_its_neighbours_on_the_shared_queues(gpu_lab_pkg, oracle):
    """Two contexts on one device share its three queues.  While one of them has a long batch (here: a 400 ms
    stall + a 64-frame batch) on the compute queue, destroying -- or syncing -- the other returns at once: it waits
    for its own frames' events only and hands its device frames to the per-device cache instead of hipFree (which
    drains the device).  The batch still comes out bit-exact."""
    import time
    w, h, n = 1920, 1080, 64
    a = gpu_lab_pkg.Context(w, h, "rggb", "BGRx")
    b = gpu_lab_pkg.Context(w, h, "bggr", "RGBx")
    one = oracle.fill_synthetic(w, h, 1, seed=93)[0]
    assert np.array_equal(b.process_host(one), oracle.bayer2rgb(one, w, "bggr", 0, 1, 2))     # b owns a ring now
    d_src = a.device_alloc(n * a.src_bytes)
    d_dst = a.device_alloc(n * a.dst_bytes)
    a.fill_synthetic(d_src, n, 5)
    a.sync()
    a.stall(400)
    a.process_device(d_src, d_dst, n)
    t0 = time.monotonic()
    b.sync()
    b.close()
    dt = time.monotonic() - t0
    a.sync()
    total = time.monotonic() - t0
    assert dt < 0.1 and total > 0.3, (dt, total)
    _frames_equal_oracle(a, oracle, d_dst, [0, 31, 63], w, h, "rggb", "BGRx", 5)
    a.device_free(d_src)
    a.device_free(d_dst)
    a.close()


def _bench_two_ranks(extra, port):
    """`bench.py --gpus 2` the way the driver launches it for a SCALE run (torch.distributed.run, one rank per GPU),
    except that both ranks share the one GPU of this box and the control plane is gloo."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--share-gpu", "--backend", "gloo"] + extra
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert res.returncode == 0 and len(lines) == 1, res.stdout[-1500:] + res.stderr[-3000:]
    return json.loads(lines[0])


def test_bench_two_ranks_batch_mode():
    """the N > 1 flow of bench.py -- process-group set-up, frame sharding g -> rank g % N, barriers,
    max-over-ranks, the per-GPU gather -- runs every round, not for the first time in a SCALE run."""
    j = _bench_two_ranks(["--steps", "5", "--warmup", "2", "--no-cpu", "--no-host-path"], 29611)
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["warmup"] == 2 and j["scaling"] == "weak"
    assert j["unit"] == "Mpix/s" and j["dtype"] == "u8" and j["higher_is_better"] is True
    assert len(j["per_gpu"]) == 2 and sorted(p["rank"] for p in j["per_gpu"]) == [0, 1]
    assert all("bit-exact" in p["parity"] for p in j["per_gpu"])
    # value = the pixels ALL ranks converted / the max-over-ranks time
    want = 3840 * 2160 * 64 * 2 * 5 / (j["ms_per_step"] * 5 * 1e-3) / 1e6
    assert abs(j["value"] - want) / want < 0.01, (j["value"], want)
    assert j["config"]["control_plane"] == "gloo" and j["roofline"]["bound"] == "hbm"
    assert 0.2 < j["roofline"]["frac"] < 1.0                    # two ranks share one GPU: about half each
    # the line says what it ran on -- here two ranks on ONE card (--share-gpu), no RCCL
    assert j["distinct_gpus"] == 1 and j["rccl_nranks"] is None
    for p in j["per_gpu"]:
        assert p["pci_bus_id"] and len(p["pci_bus_id"].split(":")) == 3 and p["numa_node"] >= -1
        assert p["plan_source"] in ("measured", "cached") and p["kernel_variant"] and "compute_units" in p
    assert len(set(p["pci_bus_id"] for p in j["per_gpu"])) == 1


def test_bench_refuses_two_ranks_on_one_card_unless_told():
    """Two ranks on one GPU without --share-gpu: refused before anything is timed, no JSON line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29613", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-host-path"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HIP_VISIBLE_DEVICES="0,0")
    env.pop("WORLD_SIZE", None)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    out = res.stdout + res.stderr
    if "distinct GPUs" not in out:
        # the runtime may refuse the duplicated ordinal itself (one visible device, LOCAL_RANK 1 has no GPU)
        assert res.returncode != 0 and not [ln for ln in res.stdout.splitlines() if ln.startswith("{")], out[-3000:]
        return
    assert res.returncode != 0 and "--share-gpu" in out
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_bench_two_ranks_stream_mode():
    """BASELINE.json configs[4] through two ranks: 1000 frames round-robin, per-rank breakdown and parity."""
    j = _bench_two_ranks(["--mode", "stream"], 29612)
    assert j["n_gpus"] == 2 and j["steps"] == 1000 and j["scaling"] == "strong"
    assert len(j["per_gpu"]) == 2 and sorted(p["frames"] for p in j["per_gpu"]) == [500, 500]
    assert all("bit-exact" in p["parity"] for p in j["per_gpu"])
    want = 3840 * 2160 * 1000 / (j["ms_per_step"] * 1000 * 1e-3) / 1e6
    assert abs(j["value"] - want) / want < 0.01
    assert j["roofline"]["bound"] == "pcie" and 0 < j["roofline"]["frac"] < 1.2


@pytest.mark.gpu
def test_frame_class_plan_of_rows_at_an_8_byte_phase(gpu_pkg, oracle):
    """Round 6: one frame per launch of a geometry whose output rows sit at an 8-byte phase
    (width % 4 == 2) no longer takes the batch class's write-back / chunk-per-XCD plan: the streaming-store shape with
    the fewest rounds, and with 256-px tiles the store arm whose wave-stores start on 128-byte boundaries (3838x2160:
    39 -> 48 % of HBM peak per frame).  Batch launches keep their plan; rows that are 16-byte aligned keep the
    hybrid-store plan.  Bit-exact in either class, frame by frame and through a list launch."""
    names = gpu_pkg.variant_names()
    for (w, h, frame_plan) in ((3838, 2160, ("lds_1x8_r4_dpp_nt", 128)), (2046, 1080, ("lds_4x2_r4_dpp_nt", 0)),
                               (1366, 768, ("lds_4x2_r4_dpp_nt", 0)), (7678, 4320, ("lds_2x4_r4_dpp_nt", 128)),
                               (3838, 2158, ("lds_1x8_r4_dpp_nt", 128))):
        with gpu_pkg.Context(w, h, "gbrg", "xRGB") as ctx:
            v1, b1, a1, src1 = ctx.get_plan_for(1)
            vb = ctx.get_plan_for(4096)[0]
            if ctx.launch_geometry(1)["grid_blocks"] and gpu_pkg.lib().mibayer_frame_class_variant(w, h, 256) == \
                    names.index(frame_plan[0]):       # a 256-CU device: the rule uses the device's own CU count
                assert (names[v1], a1) == frame_plan and b1 == -2 ** 31, (w, names[v1], b1, a1)
            assert names[vb].endswith("_dpp"), names[vb]              # the batch class: write-back twin, untouched
            n = 3
            stride = (w + 3) & ~3
            src = np.random.default_rng(w).integers(0, 256, (n, h, stride), dtype=np.uint8)
            want = oracle.bayer2rgb_batch(src, w, "gbrg", 1, 2, 3, nthreads=3)
            srcs = [ctx.device_alloc(ctx.src_bytes) for _ in range(n)]
            dsts = [ctx.device_alloc(ctx.dst_bytes) for _ in range(n)]
            for f in range(n):
                ctx.to_device(srcs[f], src[f])
                ctx.process_device(srcs[f], dsts[f], 1)
            ctx.sync()
            for f in range(n):
                assert np.array_equal(ctx.from_device(dsts[f], ctx.dst_bytes).reshape(want[f].shape), want[f]), (w, f)
                ctx.to_device(dsts[f], np.zeros(64, np.uint8))
            ctx.process_device_list(srcs, dsts)
            ctx.sync()
            for f in range(n):
                assert np.array_equal(ctx.from_device(dsts[f], ctx.dst_bytes).reshape(want[f].shape), want[f]), (w, f)
            for p in srcs + dsts:
                ctx.device_free(p)
    with gpu_pkg.Context(4056, 3040, "rggb", "BGRx") as ctx:          # 16-byte aligned rows: hybrid stores, band 1
        assert names[ctx.get_plan_for(1)[0]].endswith("_hy") and ctx.get_plan_for(1)[1:3] == (1, 0)


@pytest.mark.gpu
def test_synthetic_frame_generator_is_the_oracles(gpu_pkg, oracle):
    """mibayer_fill_synthetic (the frames of bench.py and of hipbayersrc) byte for byte against the
    oracle's generator: widths on and off the 16-column groups a thread writes, padded strides (padding columns are 0),
    several frames per call with a first-frame offset and a frame pitch larger than the frame (round 6: the kernel was
    rewritten without its per-dword 64-bit divisions, 9.2 -> ~2 us per 4K frame)."""
    for (w, h, n, first, seed, pad, pitch_extra) in ((3840, 2160, 3, 0, 2, 0, 0), (1282, 37, 4, 5, 7, 0, 4096),
                                                     (18, 3, 2, 1, 9, 12, 0), (4, 5, 1, 0, 1, 0, 0),
                                                     (2050, 61, 3, 1000, 11, 4, 64), (640, 480, 65, 2, 3, 0, 0)):
        stride = ((w + 3) & ~3) + pad
        with gpu_pkg.Context(w, h, "bggr", "RGBx", src_stride=stride) as ctx:
            pitch = ctx.src_bytes + pitch_extra
            d = ctx.device_alloc(n * pitch)
            ctx.to_device(d, np.full(n * pitch, 0xEE, np.uint8))
            ctx.fill_synthetic(d, n, seed=seed, first_frame=first, src_frame_bytes=pitch)
            ctx.sync()
            got = ctx.from_device(d, n * pitch).reshape(n, pitch)
            want = oracle.fill_synthetic(w, h, n, seed=seed, first_frame=first, stride=stride).reshape(n, -1)
            assert np.array_equal(got[:, :ctx.src_bytes], want), (w, h, n)
            assert (got[:, ctx.src_bytes:] == 0xEE).all(), (w, h)      # nothing written between the frames
            ctx.device_free(d)
