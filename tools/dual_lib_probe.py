#!/usr/bin/env python3
"""Probe: two builds of libmibayer (different files, same kernel names) loaded into ONE process and used alternately."""
import importlib.util, os, shutil, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
oracle = entry.load_oracle()
copy = "/tmp/libmibayer_second.so"
shutil.copy(pkg.LIB_PATH, copy)
spec = importlib.util.spec_from_file_location("gst_plugins_bad_amd_second", os.path.join(entry.PKG_DIR, "__init__.py"),
                                              submodule_search_locations=[entry.PKG_DIR])
second = importlib.util.module_from_spec(spec); sys.modules["gst_plugins_bad_amd_second"] = second
spec.loader.exec_module(second); second.LIB_PATH = copy
W, H = 640, 480
src = oracle.fill_synthetic(W, H, 1, seed=7)[0]
want = oracle.bayer2rgb(src, W, "rggb", 2, 1, 0)
for rnd in range(3):
    for name, p in (("first", pkg), ("second", second)):
        with p.Context(W, H, "rggb", "BGRx", device=0) as ctx:
            got = ctx.process_host(src); got2 = ctx.process_batch_via_device(src[None])[0]
        assert np.array_equal(got, want) and np.array_equal(got2, want), (rnd, name)
print("dual-lib ok:", pkg.lib()._name, second.lib()._name)
