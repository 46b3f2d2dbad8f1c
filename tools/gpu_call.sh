#!/bin/bash
# One parameterised entry for the GPU passes of a round (replaces the one-shot r02_call*.sh scripts):
#   bash tools/gpu_call.sh TAG step [step ...]        e.g.  gpurun -- 'bash tools/gpu_call.sh r03 tests_generic sweep_generic counters'
# Every step writes its log to gpurun_out/TAG_<step>.log (merged back by gpurun); copy what should be judged to profiles/.
set +e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp

step_box() {
  rocm-smi --showmemorypartition --showcomputepartition --showclocks --showpower --showserial 2>&1 | grep -v "^$" | head -40
}
step_tests_generic() {   # the parity tests that touch the generic / store-alignment arms, the autotune candidates
  timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 600 \
    -k "sector_aligned or autotune or edge_geometries or randomised or guard_bands or all_orders or strides or every_variant or list_launch or golden" 2>&1 | tail -15
}
step_tests() {
  timeout 2400 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -15
}
step_fuzz() {            # soak of the seeded fuzz, both store-alignment arms
  for al in 64 128 0; do
    echo "== MIBAYER_ALIGN_STORES=$al"
    MIBAYER_ALIGN_STORES=$al MIBAYER_FUZZ_SEED=${FUZZ_SEED:-303} MIBAYER_FUZZ_CASES=${FUZZ_CASES:-1500} timeout 1500 \
      python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomised 2>&1 | tail -3
  done
}
step_sweep_generic() {   # interleaved A/B: shapes x store policy x block order x store alignment, sensor geometries
  for geo in "4056 3040 32" "3838 2160 64" "1366 768 512" "2592 1944 64"; do
    arms=""
    for sh in 4x2 2x4 1x8; do
      for al in 0 64 128; do
        for b in 1 -1 0; do arms="$arms lds_${sh}_r4_dpp_nt:$b::$al"; done
      done
      for al in 0 64; do
        for b in -1 1; do arms="$arms lds_${sh}_r4_dpp:$b::$al"; done
      done
    done
    echo "== $geo"
    timeout 600 python tools/sweep2.py $geo 5 $arms 2>&1 | grep -v "^d_src"
  done
}
step_counters() {
  bash tools/geometry_counters.sh $TAG
}
step_defaults() {        # default plan vs autotuned plan per geometry (what a user gets without tuning)
  python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as e
pkg = e.load_package()
print("# default plan (variant 0, no autotune) and autotuned plan, device-resident batches")
for (w, h, n) in ((4056, 3040, 32), (3838, 2160, 64), (1366, 768, 512), (2592, 1944, 64), (4048, 3040, 32), (4064, 3040, 32),
                  (3840, 2160, 64), (1920, 1080, 256), (7680, 4320, 16)):
    with pkg.Context(w, h, "rggb", "BGRx") as ctx:
        d_src = ctx.device_alloc(n * ctx.src_bytes); d_dst = ctx.device_alloc(n * ctx.dst_bytes)
        ctx.fill_synthetic(d_src, n, 2); ctx.sync()
        for _ in range(4): ctx.time_device(d_src, d_dst, n, warmup=0, reps=40)
        t = sorted(ctx.time_device(d_src, d_dst, n, warmup=2, reps=10) for _ in range(7))[3]
        name, g = ctx.variant_name, ctx.launch_geometry(n)
        rep = ctx.autotune(d_src, d_dst, n)
        t2 = sorted(ctx.time_device(d_src, d_dst, n, warmup=2, reps=10) for _ in range(7))[3]
        print("%4dx%-4d x %3d  default %-18s band %5d  %.4f ms %5.1f %%   autotuned %-18s band %5d  %.4f ms %5.1f %%" % (
            w, h, n, name, g["band"], t, 5.0 * w * h * n / t / 1e6 / 80, ctx.variant_name, ctx.launch_geometry(n)["band"], t2, 5.0 * w * h * n / t2 / 1e6 / 80), flush=True)
        ctx.device_free(d_src); ctx.device_free(d_dst)
PY
}
for s in "$@"; do
  echo "######## $s"
  step_$s 2>&1 | tee gpurun_out/${TAG}_$s.log
done
