#!/usr/bin/env python3
"""Generates tests/golden/bayer2rgb_small.npz and tests/golden/known_md5.json.

Run in the build container, where /root/reference is mounted:  python tests/golden/make_golden.py

Expected outputs come from oracle_bayer2rgb_refrows(), i.e. every row is computed by the
REFERENCE's own compiled row kernels (oracle/_ref/libbayerorc_ref.so = gst/bayer/gstbayerorc-dist.c
built with the reference's -DDISABLE_ORC mode); inputs come from the counter-based generator of
SURVEY.md Appendix C.  known_md5.json records the whole-element known answers that the survey
session obtained from the compiled reference element; this script
re-derives each of them through the reference row kernels and refuses to write the file if any
differs.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

SMALL = [(w, h) for w in (4, 6, 8, 66) for h in (4, 5, 6, 7)] + [(64, 48), (16, 3), (258, 9)]
PATTERNS = ("bggr", "gbrg", "grbg", "rggb")
LAYOUTS = ("RGBx", "BGRx", "xRGB", "xBGR")   # the A-variants are byte-identical

# (W, H, seed, pattern, format, md5(input), md5(output))
KNOWN = [
    (64, 48, 7, "bggr", "RGBx", "1cd2c8032637ebe12cf364733239d02d", "5e213c796b18997f2a81d54aee9afcd8"),
    (1920, 1080, 1, "rggb", "BGRx", "981a421ca7fc7fd8ff834b8173c3200f", "f14f6ad248ef0bac0f28546db6d14813"),
    (3840, 2160, 2, "bggr", "BGRx", "cd74cbcc694cec2ee316a4120eeca8f5", "09c1a1414adc6f66966a028e891cddfa"),
    (3840, 2160, 2, "rggb", "BGRx", "cd74cbcc694cec2ee316a4120eeca8f5", "2fc0dc5ff3a92760b0943a34abeffe66"),
    (3840, 2160, 2, "grbg", "BGRx", "cd74cbcc694cec2ee316a4120eeca8f5", "ec15421e44cff4f399842deec374378b"),
    (3840, 2160, 2, "gbrg", "BGRx", "cd74cbcc694cec2ee316a4120eeca8f5", "d3d3c99d165672895f2ae0b642158928"),
    (7680, 4320, 3, "bggr", "RGBx", "a2ba091ae9ecc96f05430032ce80b506", "82665f3caa7a0df5435db7a86150b8dd"),
]


def md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    if not oracle.have_ref_rows():
        sys.exit("oracle/_ref/libbayerorc_ref.so missing: run `make -C oracle` where /root/reference exists")
    arrays = {}
    for i, (w, h) in enumerate(SMALL):
        stride = (w + 3) & ~3
        src = oracle.fill_synthetic(w, h, 1, seed=100 + i, stride=stride)[0]
        arrays["in_%dx%d" % (w, h)] = src
        for pat in PATTERNS:
            for lay in LAYOUTS:
                r, g, b = oracle.LAYOUTS[lay]
                arrays["out_%dx%d_%s_%s" % (w, h, pat, lay)] = oracle.bayer2rgb(
                    src, w, pat, r, g, b, ref_rows=True)
    np.savez_compressed(os.path.join(HERE, "bayer2rgb_small.npz"), **arrays)

    table = []
    for (w, h, seed, pat, lay, md5_in, md5_out) in KNOWN:
        src = oracle.fill_synthetic(w, h, 1, seed=seed)[0]
        r, g, b = oracle.LAYOUTS[lay]
        out = oracle.bayer2rgb(src, w, pat, r, g, b, ref_rows=True)
        if md5(src) != md5_in or md5(out) != md5_out:
            sys.exit("known answer %s does not reproduce" % ((w, h, seed, pat, lay),))
        table.append(dict(width=w, height=h, seed=seed, pattern=pat, format=lay,
                          md5_input=md5_in, md5_output=md5_out))
    with open(os.path.join(HERE, "known_md5.json"), "w") as f:
        json.dump({"source": "SURVEY.md Appendix B.3 (compiled reference element), re-derived through "
                             "oracle/_ref reference row kernels by tests/golden/make_golden.py",
                   "entries": table}, f, indent=1)
    print("wrote %d arrays, %d known answers" % (len(arrays), len(table)))


if __name__ == "__main__":
    main()
