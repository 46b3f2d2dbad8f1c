"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol of
include/mibayer.h, validates configurations like the reference's set_caps/process would, and has no
CPU compute path behind it (no GPU here => every compute entry point must fail loudly)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mibayer.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mibayer_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(pkg):
    syms = declared_symbols()
    assert len(syms) >= 24
    raw = ctypes.CDLL(pkg.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), "libmibayer.so does not export %s" % s
    assert sorted(pkg.ABI) == syms, "python harness and header disagree"


def exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


def test_the_dynamic_symbol_table_is_exactly_the_header(pkg):
    """what libmibayer.so exports IS include/mibayer.h -- no C++ internals (_ZN7mibayer...), no kernel
    handles, no pool seam (csrc/mibayer_hooks.h: mibayer_internal_*).  The list is generated from the header into a
    linker version script (Makefile); the lab build adds the seam (drills, tools) and nothing else."""
    syms = declared_symbols()
    assert exported(pkg.LIB_PATH) == syms
    hooks = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "gst-plugins-bad_amd", "csrc", "mibayer_hooks.h")).read(), flags=re.S)
    seam = sorted(set(re.findall(r"\b(mibayer_internal_[a-z0-9_]+)\s*\(", hooks)))
    assert len(seam) == 6
    if os.path.exists(pkg.LAB_LIB_PATH):
        assert exported(pkg.LAB_LIB_PATH) == sorted(syms + seam)
    # the header says which entry points are the drop-in and which are tuning / diagnostics: every function is in
    # exactly one group, and the core is the boundary SURVEY.md section 8(b) proposed
    groups = header_groups()
    assert sorted(sum(groups.values(), [])) == syms
    assert set(groups["core"]) >= {"mibayer_create", "mibayer_destroy", "mibayer_process_host", "mibayer_submit",
                                   "mibayer_wait", "mibayer_process_device", "mibayer_host_alloc", "mibayer_host_free",
                                   "mibayer_device_count", "mibayer_strerror"}
    assert len(groups["core"]) <= 16, groups["core"]


def header_groups():
    """{group: [function, ...]} from the `MIBAYER ABI GROUPS` comment of include/mibayer.h"""
    text = open(os.path.join(ROOT, "include", "mibayer.h")).read()
    block = text[text.index("MIBAYER ABI GROUPS"):text.index("END OF MIBAYER ABI GROUPS")]
    parts = re.split(r"\* group ([a-z-]+):", block)
    return {name: sorted(set(re.findall(r"\b(mibayer_[a-z0-9_]+)\b", body))) for name, body in zip(parts[1::2], parts[2::2])}


def test_version_strerror_variants(pkg):
    L = pkg.lib()
    assert L.mibayer_abi_version() == 5
    for code in range(-8, 1):
        assert L.mibayer_strerror(code)
    names = pkg.variant_names()
    assert len(names) == len(set(names)) >= 1
    assert L.mibayer_variant_name(len(names)) is None


def make_cfg(pkg, **kw):
    d = dict(width=64, height=48, src_stride=0, dst_stride=0, pattern=0, r_off=0, g_off=1, b_off=2,
             device=-1, inflight=0, variant=0, flags=0)
    d.update(kw)
    return pkg.Cfg(ctypes.sizeof(pkg.Cfg), d["width"], d["height"], d["src_stride"], d["dst_stride"],
                   d["pattern"], d["r_off"], d["g_off"], d["b_off"], d["device"], d["inflight"],
                   d["variant"], d["flags"])


def create(pkg, cfg):
    h = ctypes.c_void_p()
    rc = pkg.lib().mibayer_create(ctypes.byref(cfg), ctypes.byref(h))
    if rc == 0:
        pkg.lib().mibayer_destroy(h)
    return rc


def test_cfg_validation_matches_reference_domain(pkg):
    ok_or_nodev = (pkg.OK, pkg.ERR_NO_DEVICE)
    assert create(pkg, make_cfg(pkg)) in ok_or_nodev
    # geometry: odd / tiny widths and heights < 3 are where the reference reads uninitialised memory
    for kw in (dict(width=5), dict(width=2), dict(width=0), dict(height=2), dict(height=0),
               dict(src_stride=62), dict(src_stride=66), dict(dst_stride=252), dict(dst_stride=258)):
        assert create(pkg, make_cfg(pkg, **kw)) == pkg.ERR_GEOMETRY, kw
    # the 4 byte layouts of gstbayer2rgb.c:409-421 and nothing else
    for (r, g, b) in [(0, 1, 2), (2, 1, 0), (1, 2, 3), (3, 2, 1)]:
        assert create(pkg, make_cfg(pkg, r_off=r, g_off=g, b_off=b)) in ok_or_nodev
    for (r, g, b) in [(0, 2, 1), (1, 1, 1), (0, 1, 3), (4, 1, 0)]:
        assert create(pkg, make_cfg(pkg, r_off=r, g_off=g, b_off=b)) == pkg.ERR_LAYOUT
    for kw in (dict(pattern=4), dict(pattern=-1), dict(flags=4), dict(flags=0x80000000), dict(variant=10 ** 6), dict(inflight=-1)):
        assert create(pkg, make_cfg(pkg, **kw)) == pkg.ERR_ARG, kw
    bad = make_cfg(pkg)
    bad.struct_size = 8
    assert create(pkg, bad) == pkg.ERR_ARG
    assert pkg.lib().mibayer_create(None, None) == pkg.ERR_ARG


def test_no_cpu_fallback(pkg):
    """Without a device the product must refuse, not compute on the host."""
    if pkg.device_count() > 0:
        pytest.skip("a GPU is visible; the no-device behaviour is checked on CPU boxes")
    with pytest.raises(pkg.MibayerError) as e:
        pkg.Context(64, 48)
    assert e.value.status == pkg.ERR_NO_DEVICE
    assert pkg.lib().mibayer_host_alloc(64) is None


def test_product_does_not_link_or_mention_the_oracle(pkg):
    out = subprocess.run(["ldd", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    pdir = os.path.join(ROOT, "gst-plugins-bad_amd")
    for dirpath, _, files in os.walk(pdir):
        for f in files:
            if f.endswith((".hip", ".h", ".c", ".cpp", ".py")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in text and "import oracle" not in text, f
                assert "bayer2rgb_oracle" not in text, f


def test_xcd_block_to_tile_is_a_bijection(pkg):
    """Every tile is processed by exactly one block, for the identity map and for every band width;
    with a band map, XCD k (blocks k, k+8, ...) owns whole tile rows, walks each left to right and
    keeps vertically adjacent rows of a band on the same XCD."""
    f = pkg.lib().mibayer_block_to_tile
    for tiles_x, tile_rows in [(1, 1), (15, 68), (4, 270), (3, 7), (8, 8), (2, 65), (30, 4320 // 32)]:
        ntiles = tiles_x * tile_rows
        for band in (0, 1, 2, 3, 8, (tile_rows + 7) // 8):
            if band == 0:
                grid = ntiles
            else:
                groups = -(-tile_rows // (8 * band))
                grid = groups * 8 * band * tiles_x
            tiles = [f(b, tiles_x, tile_rows, band) for b in range(grid)]
            live = [t for t in tiles if t >= 0]
            assert sorted(live) == list(range(ntiles)), (tiles_x, tile_rows, band)
            assert f(grid + 8 * band * tiles_x, tiles_x, tile_rows, band) == -1 or band == 0
            if band == 0:
                assert tiles == list(range(ntiles))
                continue
            for k in range(8):
                mine = [t for t in tiles[k::8] if t >= 0]
                rows = [t // tiles_x for t in mine]
                assert all((r // band) % 8 == k for r in rows)          # whole rows per XCD
                for r in set(rows):                                       # each row left to right
                    xs = [t % tiles_x for t in mine if t // tiles_x == r]
                    assert xs == list(range(tiles_x))
                assert rows == sorted(rows)
    assert f(-1, 4, 10, 1) == -1 and f(0, 0, 10, 1) == -1 and f(0, 4, 0, 1) == -1


def test_launch_geometry_needs_no_device_math(pkg):
    # pure host arithmetic check of the tile grid through the variant table
    names = pkg.variant_names()
    assert names[0] == "auto" and all(n.startswith(("lds_", "direct_", "persist_")) for n in names[1:])


def test_pool_cfg_validation(pkg):
    L = pkg.lib()
    h = ctypes.c_void_p()
    pc = pkg.PoolCfg()
    pc.struct_size = ctypes.sizeof(pkg.PoolCfg)
    pc.stream = pkg.make_cfg(64, 48)
    for n in (0, -1, pkg.MAX_SHARDS + 1):
        pc.ndevices = n
        assert L.mibayer_pool_create(ctypes.byref(pc), ctypes.byref(h)) == pkg.ERR_ARG
    pc.ndevices = 2
    pc.devices[0], pc.devices[1] = 0, -3
    assert L.mibayer_pool_create(ctypes.byref(pc), ctypes.byref(h)) in (pkg.ERR_NO_DEVICE,)
    pc.devices[1] = 0
    pc.stream = pkg.make_cfg(63, 48)          # geometry errors of the shards surface unchanged
    assert L.mibayer_pool_create(ctypes.byref(pc), ctypes.byref(h)) == pkg.ERR_GEOMETRY
    pc.struct_size = 4
    assert L.mibayer_pool_create(ctypes.byref(pc), ctypes.byref(h)) == pkg.ERR_ARG
    assert L.mibayer_pool_pending(None) == pkg.ERR_ARG


def test_rgb2bayer_cfg_validation(pkg):
    """MIBAYER_FLAG_RGB2BAYER: the reference loop has no neighbourhood, so every size >= 1 is in the domain;
    strides default to 4*W in / ROUND_UP_4(W) out (gstrgb2bayer.c:179, :255)."""
    ok_or_nodev = (pkg.OK, pkg.ERR_NO_DEVICE)
    F = pkg.FLAG_RGB2BAYER
    for (w, h) in [(1, 1), (3, 2), (5, 7), (640, 480)]:
        assert create(pkg, make_cfg(pkg, width=w, height=h, r_off=1, g_off=2, b_off=3, flags=F)) in ok_or_nodev
    for kw in (dict(width=0), dict(height=0), dict(src_stride=4 * 64 - 4), dict(src_stride=4 * 64 + 2),
               dict(dst_stride=60), dict(dst_stride=66)):
        assert create(pkg, make_cfg(pkg, r_off=1, g_off=2, b_off=3, flags=F, **kw)) == pkg.ERR_GEOMETRY, kw
    for (r, g, b) in [(1, 1, 3), (4, 2, 3), (-1, 2, 3)]:
        assert create(pkg, make_cfg(pkg, r_off=r, g_off=g, b_off=b, flags=F)) == pkg.ERR_LAYOUT
    assert create(pkg, make_cfg(pkg, r_off=1, g_off=2, b_off=3, flags=F, variant=1)) == pkg.ERR_ARG


def test_auto_variant_minimises_wasted_lanes(pkg):
    """variant 0: rows that fit one tile take the narrowest tile that covers them (256 / 512 / 1024 px); wider rows
    take 1024-px tiles when they pad the width <= 7 % beyond the best shape, else the least padding."""
    names = pkg.variant_names()
    f = pkg.lib().mibayer_auto_variant
    want = {3840: "lds_4x2", 7680: "lds_4x2", 1920: "lds_4x2", 640: "lds_4x2", 800: "lds_4x2", 1280: "lds_1x8",
            300: "lds_2x4", 256: "lds_1x8", 258: "lds_2x4", 200: "lds_1x8", 4: "lds_1x8", 1024: "lds_4x2",
            1026: "lds_1x8", 2048: "lds_4x2", 512: "lds_2x4", 514: "lds_4x2", 4000: "lds_4x2", 3072: "lds_4x2",
            2600: "lds_1x8"}
    for w, prefix in want.items():
        assert names[f(w)].startswith(prefix), (w, names[f(w)])


def test_frame_class_rule_and_known_widths(pkg):
    """Round 5, pure host arithmetic (no device): a launch of ONE frame takes the production shape whose grid needs the
    fewest rounds of the device's workgroup slots (256 CUs x 4), the widest tile among equals; rows that fit one tile
    keep the narrow-tile rule; thirteen common sensor widths carry a measured batch-class default."""
    import ctypes
    L = pkg.lib()
    names = pkg.variant_names()

    def tiles(w, h, v):
        tw, th = {1: (1024, 8), 2: (512, 16), 3: (256, 32)}[v]
        return -(-w // tw) * -(-h // th)
    want = {(3840, 2160): 3, (3264, 2448): 3, (7680, 4320): 2, (2592, 1944): 1, (4096, 2160): 1, (1920, 1080): 1,
            (4056, 3040): 1, (640, 480): 1, (500, 300): 2, (200, 100): 3, (5120, 2880): 1}
    for (w, h), v in want.items():
        got = L.mibayer_frame_class_variant(w, h, 256)
        assert got == v, (w, h, names[got])
        if w > 1024:        # it IS the minimum number of rounds, and no wider tile reaches it
            rounds = {k: -(-tiles(w, h, k) // 1024) for k in (1, 2, 3)}
            assert rounds[got] == min(rounds.values()) and all(rounds[k] > rounds[got] for k in range(1, got))
    assert tiles(3840, 2160, 1) == 1080 and tiles(3840, 2160, 3) == 1020           # the 4K case of DESIGN.md section 5
    # another device size changes the answer: 4K on 304 CUs (1216 slots) fits 1024x8 tiles in one round
    assert L.mibayer_frame_class_variant(3840, 2160, 304) == 1
    assert L.mibayer_frame_class_variant(0, 10, 256) == pkg.ERR_ARG and L.mibayer_frame_class_variant(64, 48, 0) == pkg.ERR_ARG
    v, b = ctypes.c_int(), ctypes.c_int()
    known = {2048: (3, 1), 4096: (3, 1), 8192: (3, 1), 2304: (2, 0), 3264: (2, 0), 2448: (1, 0), 2560: (1, 0),
             2592: (1, 0), 2688: (1, 0), 4608: (1, 0), 4112: (2, 1), 4208: (2, 1), 6000: (2, 1)}
    for w, plan in known.items():
        assert L.mibayer_known_width_plan(w, ctypes.byref(v), ctypes.byref(b)) == 1 and (v.value, b.value) == plan, w
        assert L.mibayer_known_width_plan(w, None, None) == 1
    for w in (3840, 1920, 7680, 1280, 4000, 2050):
        assert L.mibayer_known_width_plan(w, ctypes.byref(v), ctypes.byref(b)) == 0


def test_every_environment_knob_of_the_library_is_documented():
    """Every MIBAYER_* variable the native sources read is named in DESIGN.md, INTEGRATION.md or the ABI header: a knob
    nobody can find is not a knob.  Two lists: what the PRODUCT library reads -- operational
    variables only -- and the tuning knobs that exist in the lab build alone (LAB_GETENV compiles to nothing without
    -DMIBAYER_LAB)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "gst-plugins-bad_amd", "csrc")
    product, lab = set(), set()
    for name in os.listdir(csrc):
        with open(os.path.join(csrc, name)) as f:
            text = f.read()
        product |= set(re.findall(r'(?<!LAB_)\bgetenv \("(MIBAYER_[A-Z0-9_]+)"\)', text))
        lab |= set(re.findall(r'LAB_GETENV \("(MIBAYER_[A-Z0-9_]+)"\)', text))
    docs = ""
    for name in ("DESIGN.md", "INTEGRATION.md", os.path.join("include", "mibayer.h")):
        with open(os.path.join(root, name)) as f:
            docs += f.read()
    assert product == {"MIBAYER_ROCTX", "MIBAYER_WAIT_TIMEOUT_MS", "MIBAYER_WAIT_SPIN_US", "MIBAYER_PLAN_CACHE",
                       "MIBAYER_POOL_THREADS", "MIBAYER_POOL_HELPERS", "MIBAYER_POOL_NUMA", "MIBAYER_POOL_PIN_THREADS",
                       "MIBAYER_INJECT_FAULT", "MIBAYER_INJECT_STALL"}, sorted(product)
    assert len(lab) >= 12 and not (lab & product), sorted(lab & product)
    missing = sorted(v for v in product | lab if v not in docs)
    assert not missing, missing


def test_product_and_lab_builds(pkg, lab_pkg):
    """the default library is the lean one -- the three production tile shapes with their plain-store
    and hybrid-store twins, under 300 KB -- and `make lab` adds the experiment arms behind the same ABI; the first ten
    variant ids mean the same in both."""
    L, LL = pkg.lib(), lab_pkg.lib()
    if os.environ.get("MIBAYER_LIB_PATH"):
        pytest.skip("MIBAYER_LIB_PATH overrides the product library")
    assert L.mibayer_is_lab_build() == 0 and LL.mibayer_is_lab_build() == 1
    names, lab_names = pkg.variant_names(), lab_pkg.variant_names()
    assert len(names) == 10 and len(lab_names) > 20 and lab_names[:10] == names
    assert names[1:4] == ["lds_4x2_r4_dpp_nt", "lds_2x4_r4_dpp_nt", "lds_1x8_r4_dpp_nt"]
    assert names[4:7] == ["lds_4x2_r4_dpp", "lds_2x4_r4_dpp", "lds_1x8_r4_dpp"]
    assert names[7:10] == ["lds_4x2_r4_dpp_hy", "lds_2x4_r4_dpp_hy", "lds_1x8_r4_dpp_hy"]
    assert os.path.getsize(pkg.LIB_PATH) < 300 * 1024, os.path.getsize(pkg.LIB_PATH)
    for name in pkg.ABI:
        assert hasattr(LL, name), name
