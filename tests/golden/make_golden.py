#!/usr/bin/env python3
"""Generates tests/golden/bayer2rgb_small.npz, tests/golden/rgb2bayer_small.npz and tests/golden/known_md5.json.

Run in the build container, where /root/reference is mounted:  python tests/golden/make_golden.py

Every expected output is computed by the REFERENCE'S OWN FRAME-LEVEL FUNCTIONS compiled here
(oracle/_ref/libbayer_frame_ref.so, `make -C oracle ref_frame`): gst_bayer2rgb_process
(gst/bayer/gstbayer2rgb.c:387-451, with its row helper :354-381 and the reference's -DDISABLE_ORC row
kernels) and gst_rgb2bayer_transform (gst/bayer/gstrgb2bayer.c:229-278).  The restated frame driver over
the reference's row kernels (oracle_bayer2rgb_refrows) must agree byte for byte, or nothing is written.
Inputs come from the counter-based generator of SURVEY.md Appendix C.  known_md5.json holds the
whole-element known answers the survey session recorded -- re-derived here through
the reference's frame driver, refused if any differs -- plus rgb2bayer known answers at full size from the
reference's transform.
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


# rgb2bayer: (W, H, extra bytes per source row); odd widths and heights included (the reference accepts them)
R2B_SMALL = [(1, 1, 0), (2, 3, 0), (3, 2, 0), (5, 7, 0), (5, 7, 8), (16, 9, 0), (66, 50, 0), (101, 33, 0), (64, 48, 0),
             (130, 21, 24), (257, 5, 0), (1030, 17, 0)]
# rgb2bayer at full size: (W, H, seed, pattern) -> md5 of the reference's mosaic
R2B_KNOWN = [(1920, 1080, 1, "rggb"), (3840, 2160, 2, "bggr"), (3840, 2160, 2, "gbrg"), (3840, 2160, 2, "grbg"),
             (3840, 2160, 2, "rggb"), (3841, 2161, 4, "grbg")]


def argb_frame(w, h, seed, pad=0):
    """An ARGB frame (H, 4W + pad) from the Appendix C generator (a 4W x H byte image)."""
    return oracle.fill_synthetic(4 * w + pad, h, 1, seed=seed)[0]


def main():
    if not oracle.have_ref_rows() or not oracle.have_ref_frame():
        sys.exit("oracle/_ref/*.so missing: run `make -C oracle` where /root/reference exists")
    arrays = {}
    for i, (w, h) in enumerate(SMALL):
        stride = (w + 3) & ~3
        src = oracle.fill_synthetic(w, h, 1, seed=100 + i, stride=stride)[0]
        arrays["in_%dx%d" % (w, h)] = src
        for pat in PATTERNS:
            for lay in LAYOUTS:
                r, g, b = oracle.LAYOUTS[lay]
                out = oracle.ref_frame_bayer2rgb(src, w, pat, r, g, b)
                if not np.array_equal(out, oracle.bayer2rgb(src, w, pat, r, g, b, ref_rows=True)):
                    sys.exit("reference frame driver and restated driver over reference rows differ: %s"
                             % ((w, h, pat, lay),))
                arrays["out_%dx%d_%s_%s" % (w, h, pat, lay)] = out
    np.savez_compressed(os.path.join(HERE, "bayer2rgb_small.npz"), **arrays)

    inv = {}
    for i, (w, h, pad) in enumerate(R2B_SMALL):
        key = "%dx%d%s" % (w, h, "p%d" % pad if pad else "")
        src = argb_frame(w, h, 200 + i, pad)
        inv["in_" + key] = src
        for pat in PATTERNS:
            inv["out_%s_%s" % (key, pat)] = oracle.ref_frame_rgb2bayer(src, w, pat)[:, :w]
    np.savez_compressed(os.path.join(HERE, "rgb2bayer_small.npz"), **inv)

    table = []
    for (w, h, seed, pat, lay, md5_in, md5_out) in KNOWN:
        src = oracle.fill_synthetic(w, h, 1, seed=seed)[0]
        r, g, b = oracle.LAYOUTS[lay]
        out = oracle.ref_frame_bayer2rgb(src, w, pat, r, g, b)
        if md5(src) != md5_in or md5(out) != md5_out:
            sys.exit("known answer %s does not reproduce" % ((w, h, seed, pat, lay),))
        table.append(dict(width=w, height=h, seed=seed, pattern=pat, format=lay,
                          md5_input=md5_in, md5_output=md5_out))
    inv_table = []
    for (w, h, seed, pat) in R2B_KNOWN:
        src = argb_frame(w, h, seed)
        out = oracle.ref_frame_rgb2bayer(src, w, pat)[:, :w]
        inv_table.append(dict(width=w, height=h, seed=seed, pattern=pat, format="ARGB",
                              md5_input=md5(src), md5_output=md5(out)))
    with open(os.path.join(HERE, "known_md5.json"), "w") as f:
        json.dump({"source": "SURVEY.md Appendix B.3 (compiled reference element), re-derived through the "
                             "reference's own gst_bayer2rgb_process (oracle/_ref/libbayer_frame_ref.so) by "
                             "tests/golden/make_golden.py",
                   "entries": table,
                   "rgb2bayer_source": "the reference's own gst_rgb2bayer_transform (gstrgb2bayer.c:229-278) compiled "
                                       "here; input = Appendix C generator as a 4W x H byte image (ARGB), output = the "
                                       "W valid columns of each mosaic row",
                   "rgb2bayer_entries": inv_table}, f, indent=1)
    print("wrote %d + %d arrays, %d + %d known answers" % (len(arrays), len(inv), len(table), len(inv_table)))


if __name__ == "__main__":
    main()
