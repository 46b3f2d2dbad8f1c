#!/bin/bash
set +e
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q --timeout 240 2>&1 | tail -4
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as e
pkg = e.load_package()
print("# default plan (variant 0, no autotune) and autotuned plan, device-resident batches")
for (w, h, n) in ((4056, 3040, 32), (3838, 2160, 64), (1366, 768, 512), (2592, 1944, 64), (3840, 2160, 64), (1920, 1080, 256)):
    with pkg.Context(w, h, "rggb", "BGRx") as ctx:
        d_src = ctx.device_alloc(n * ctx.src_bytes); d_dst = ctx.device_alloc(n * ctx.dst_bytes)
        ctx.fill_synthetic(d_src, n, 2); ctx.sync()
        for _ in range(4): ctx.time_device(d_src, d_dst, n, warmup=0, reps=40)
        t = sorted(ctx.time_device(d_src, d_dst, n, warmup=2, reps=10) for _ in range(7))[3]
        name, g = ctx.variant_name, ctx.launch_geometry(n)
        rep = ctx.autotune(d_src, d_dst, n)
        t2 = sorted(ctx.time_device(d_src, d_dst, n, warmup=2, reps=10) for _ in range(7))[3]
        print("%4dx%-4d x %3d  default %-18s band %5d  %.4f ms %5.1f %%   autotuned %-18s band %5d  %.4f ms %5.1f %%" % (
            w, h, n, name, g["band"], t, 5.0 * w * h * n / t / 1e6 / 80, ctx.variant_name, ctx.launch_geometry(n)["band"], t2, 5.0 * w * h * n / t2 / 1e6 / 80))
        ctx.device_free(d_src); ctx.device_free(d_dst)
PY
