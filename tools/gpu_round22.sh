#!/bin/bash
set +e
export TMPDIR=/tmp
rocm-smi --showserial 2>/dev/null | grep Serial
timeout 600 python -m pytest tests/test_gpu_rgb2bayer.py -m gpu -q -k "matches_oracle or round_trip" 2>&1 | tail -2
for rows in 2 4 8 16; do for band in -1 0; do
MIBAYER_R2B_ROWS=$rows MIBAYER_XCD_BAND=$band python - <<'PY'
import os
import __graft_entry__ as g
pkg = g.load_package()
W, H, N = 3840, 2160, 64
with pkg.Context(W, H, "rggb", (1, 2, 3), flags=pkg.FLAG_RGB2BAYER) as c:
    d_src = c.device_alloc(N * c.src_bytes); d_dst = c.device_alloc(N * c.dst_bytes)
    ts = [c.time_device(d_src, d_dst, N, warmup=3, reps=20) for _ in range(5)]
    t = sorted(ts)[2]
    print("rows %2s band %2s rgb2bayer 4K x 64: median %.4f ms  %.1f GB/s = %.1f%% of 8 TB/s" % (os.environ["MIBAYER_R2B_ROWS"], os.environ["MIBAYER_XCD_BAND"], t, 5.0 * W * H * N / t / 1e6, 5.0 * W * H * N / t / 1e6 / 80))
PY
done; done 2>&1 | tee gpurun_out/r2b_rows.log
