#!/bin/bash
set +e
export TMPDIR=/tmp
echo "== pytest rgb2bayer"; timeout 900 python -m pytest tests/test_gpu_rgb2bayer.py -m gpu -x -q 2>&1 | tail -5
for band in -1 0 1 8; do
MIBAYER_XCD_BAND=$band python - <<'PY'
import os
import __graft_entry__ as g
pkg = g.load_package()
W, H, N = 3840, 2160, 64
with pkg.Context(W, H, "rggb", (1, 2, 3), flags=pkg.FLAG_RGB2BAYER) as c:
    d_src = c.device_alloc(N * c.src_bytes); d_dst = c.device_alloc(N * c.dst_bytes)
    ts = [c.time_device(d_src, d_dst, N, warmup=3, reps=20) for _ in range(5)]
    t = sorted(ts)[2]
    print("band %s rgb2bayer 4K x 64: median %.4f ms  %.1f GB/s = %.1f%% of 8 TB/s" % (os.environ["MIBAYER_XCD_BAND"], t, 5.0 * W * H * N / t / 1e6, 5.0 * W * H * N / t / 1e6 / 80))
PY
done
