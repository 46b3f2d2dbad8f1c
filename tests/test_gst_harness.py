"""Element-level tests through GstHarness (tests/check/element_harness.c, compiled on the fly): hand-built
buffers in, byte-for-byte comparison out, flush and EOS behaviour of the queued mode, state cycling, two
element instances running concurrently.  Conventions of the reference's own element tests
(tests/check/elements/vkcolorconvert.c:61-113, tests/check/generic/states.c:106-216)."""
import os
import subprocess

import numpy as np
import pytest

from test_gst_element import GST_PREFIX, gst_env, needs_gst, plugin  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, needs_gst]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bin") / "element_harness")
    inc = ["-I%s/include/gstreamer-1.0" % GST_PREFIX, "-I%s/include/glib-2.0" % GST_PREFIX,
           "-I%s/lib/glib-2.0/include" % GST_PREFIX]
    cmd = ["gcc", "-O1", "-Wall"] + inc + [os.path.join(ROOT, "tests", "check", "element_harness.c"), "-o", exe,
                                           "-L%s/lib" % GST_PREFIX, "-lgstcheck-1.0", "-lgstvideo-1.0", "-lgstreamer-1.0",
                                           "-lgobject-2.0", "-lglib-2.0", "-Wl,-rpath,%s/lib" % GST_PREFIX]
    res = subprocess.run(cmd, capture_output=True, text=True)
    # GStreamer's development files are there (needs_gst): a driver that does not build is a defect, not a skip
    assert res.returncode == 0, "cannot build the GstHarness driver: " + res.stderr[-1500:]
    return exe


def run(exe, tmp, *args):
    res = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, env=gst_env(tmp), timeout=300)
    assert res.returncode == 0, (res.stdout + res.stderr)[-1500:]
    return dict(kv.split("=") for kv in res.stdout.split() if "=" in kv)


CAPS = "video/x-bayer,format=%s,width=%d,height=%d,framerate=30/1"


@pytest.mark.parametrize("launch", ["bayer2rgb", "bayer2rgb inflight=3", "bayer2rgb inflight=2 devices=0,0",
                                    "bayer2rgb inflight=4 hipgraph=true pinned-pool=false"])
def test_harness_convert_every_frame_in_order(plugin, gpu_pkg, oracle, harness, tmp_path, launch):
    w, h, n = 258, 36, 11                # W % 4 == 2: padded source rows, generic kernel
    src = oracle.fill_synthetic(w, h, n, seed=71, stride=260)
    inp, outp = tmp_path / "in.raw", tmp_path / "out.raw"
    src.tofile(inp)
    kv = run(harness, tmp_path, "convert", launch, CAPS % ("gbrg", w, h), inp, 260 * h, outp)
    assert kv["pushed"] == str(n) and kv["pulled"] == str(n)
    got = np.fromfile(outp, np.uint8).reshape(n, h, 4 * w)
    for i in range(n):
        assert np.array_equal(got[i], oracle.bayer2rgb(src[i], w, "gbrg", 0, 1, 2)), (launch, i)     # default RGBx


@pytest.mark.parametrize("launch", ["bayer2rgb", "bayer2rgb inflight=3", "bayer2rgb inflight=2 devices=0,0"])
def test_harness_caps_change_mid_stream(plugin, gpu_pkg, oracle, harness, tmp_path, launch):
    """Resolution AND Bayer order change while the stream runs: the frames still in flight under the old caps come
    out first (queued mode drains on the CAPS event), the GPU pool is rebuilt for the new geometry, every frame
    of both halves is converted, in order."""
    (w1, h1, p1, n1), (w2, h2, p2, n2) = (64, 48, "bggr", 7), (130, 22, "grbg", 6)
    a = oracle.fill_synthetic(w1, h1, n1, seed=81)
    b = oracle.fill_synthetic(w2, h2, n2, seed=82, stride=132)
    fa, fb, outp = tmp_path / "a.raw", tmp_path / "b.raw", tmp_path / "out.raw"
    a.tofile(fa)
    b.tofile(fb)
    kv = run(harness, tmp_path, "renegotiate", launch, CAPS % (p1, w1, h1), fa, w1 * h1,
             CAPS % (p2, w2, h2), fb, 132 * h2, outp)
    assert kv["pushed"] == str(n1 + n2) and kv["pulled"] == str(n1 + n2)
    got = np.fromfile(outp, np.uint8)
    first, second = got[:n1 * h1 * 4 * w1].reshape(n1, h1, 4 * w1), got[n1 * h1 * 4 * w1:].reshape(n2, h2, 4 * w2)
    for i in range(n1):
        assert np.array_equal(first[i], oracle.bayer2rgb(a[i], w1, p1, 0, 1, 2)), (launch, "first", i)
    for i in range(n2):
        assert np.array_equal(second[i], oracle.bayer2rgb(b[i], w2, p2, 0, 1, 2)), (launch, "second", i)


def test_harness_flush_drops_frames_in_flight(plugin, gpu_pkg, oracle, harness, tmp_path):
    """Queued mode, capacity 4: the first buffer leaves at once (sinks preroll on it), the next two are held;
    FLUSH_START/STOP must drop exactly those two; after the flush the first buffer again leaves at once and the
    rest come out as the pool fills / at EOS, and nothing else."""
    w, h, n = 64, 48, 8
    src = oracle.fill_synthetic(w, h, n, seed=72)
    inp, outp = tmp_path / "in.raw", tmp_path / "out.raw"
    src.tofile(inp)
    kv = run(harness, tmp_path, "flush", "bayer2rgb inflight=4", CAPS % ("bggr", w, h), inp, w * h, outp, 3)
    assert kv["before_flush_pulled"] == "0"           # frame 0 had already been pulled right after its push
    assert kv["pushed"] == "8" and kv["pulled"] == "6"
    got = np.fromfile(outp, np.uint8).reshape(6, h, 4 * w)
    for k, i in enumerate([0, 3, 4, 5, 6, 7]):
        assert np.array_equal(got[k], oracle.bayer2rgb(src[i], w, "bggr", 0, 1, 2)), i


def test_state_cycles_one_pipeline_instance(plugin, gpu_pkg, harness, tmp_path):
    """NULL <-> PLAYING ten times on one pipeline: contexts, streams, pinned pools and device rings are
    released and re-created without error (reference pattern: tests/check/generic/states.c)."""
    desc = ("videotestsrc num-buffers=6 ! video/x-bayer,format=rggb,width=640,height=480 "
            "! bayer2rgb inflight=2 devices=0,0 ! video/x-raw,format=BGRx ! fakesink")
    kv = run(harness, tmp_path, "states", desc, 10)
    assert kv["cycles_ok"] == "10"


def test_two_instances_in_parallel_branches(plugin, gpu_pkg, harness, tmp_path):
    """Two bayer2rgb instances (different orders/formats, one synchronous, one queued) and one rgb2bayer run
    concurrently on different streaming threads of one process."""
    desc = ("videotestsrc num-buffers=12 ! video/x-raw,format=ARGB,width=320,height=240 ! rgb2bayer "
            "! video/x-bayer,format=bggr ! tee name=t "
            "t. ! queue ! bayer2rgb ! video/x-raw,format=RGBx ! fakesink "
            "t. ! queue ! bayer2rgb inflight=3 ! video/x-raw,format=xBGR ! fakesink")
    kv = run(harness, tmp_path, "states", desc, 3)
    assert kv["cycles_ok"] == "3"
