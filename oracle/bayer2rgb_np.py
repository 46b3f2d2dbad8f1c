"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Closed-form NumPy restatement of gst-plugins-bad's bayer2rgb, independent of
the ring-buffer form in bayer2rgb_oracle.c: every output pixel is written as a
function of the 3x3 source neighbourhood.

Reference (relative to /root/reference):
  gst/bayer/gstbayer2rgb.c:354-381  horizontal lines E/O and their edge columns
  gst/bayer/gstbayer2rgb.c:429-447  ring of 4 row slots -> up()/dn() row maps
  gst/bayer/gstbayer2rgb.c:400-427  pattern symmetry (swap r/b, swap row types)
  gst/bayer/gstbayerorc.orc:43-92   merge_bg / merge_gr arithmetic
"""
import numpy as np

PATTERNS = {"bggr": 0, "gbrg": 1, "grbg": 2, "rggb": 3}  # gstbayer2rgb.c:95-101
# byte offsets (r, g, b) per output format, gstbayer2rgb.c:268-271
LAYOUTS = {
    "RGBx": (0, 1, 2), "RGBA": (0, 1, 2),
    "BGRx": (2, 1, 0), "BGRA": (2, 1, 0),
    "xRGB": (1, 2, 3), "ARGB": (1, 2, 3),
    "xBGR": (3, 2, 1), "ABGR": (3, 2, 1),
}


def avg(a, b):
    """ORC avgub: (a + b + 1) >> 1, gstbayerorc-dist.c:225-226."""
    return ((a.astype(np.uint16) + b.astype(np.uint16) + 1) >> 1).astype(np.uint8)


def horizontal_lines(S):
    """E, O for all rows at once (Appendix A.1; gstbayer2rgb.c:354-381)."""
    H, W = S.shape
    left = np.empty_like(S)
    right = np.empty_like(S)
    left[:, 1:] = S[:, :-1]
    left[:, 0] = S[:, 1]
    right[:, :-1] = S[:, 1:]
    right[:, -1] = S[:, -2]
    A = avg(left, right)
    E = S.copy()
    O = S.copy()
    E[:, 1::2] = A[:, 1::2]
    O[:, 0::2] = A[:, 0::2]
    E[:, W - 1] = S[:, W - 2]   # :372-380 tail
    O[:, 0] = S[:, 1]           # :361 head
    O[:, W - 2] = S[:, W - 3]   # :372-380 tail copies the LEFT neighbour
    return E, O


def row_maps(H):
    """up(j), dn(j) (Appendix A.2; ring slots of gstbayer2rgb.c:430-447)."""
    up = np.arange(H) - 1
    up[0] = 1
    dn = np.arange(H) + 1
    dn[H - 1] = H - 4 if H >= 4 else 1
    return up, dn


def bayer2rgb(S, pattern, r_off, g_off, b_off):
    """S: (H, W) uint8 -> (H, W, 4) uint8."""
    if isinstance(pattern, str):
        pattern = PATTERNS[pattern]
    S = np.ascontiguousarray(S, dtype=np.uint8)
    H, W = S.shape
    if W < 4 or W % 2 or H < 3:
        raise ValueError("outside the reference's defined domain")
    swap_rb = pattern in (PATTERNS["rggb"], PATTERNS["gbrg"])     # :403-407
    swap_rows = pattern in (PATTERNS["grbg"], PATTERNS["gbrg"])   # :422-427
    rp, bp = (b_off, r_off) if swap_rb else (r_off, b_off)
    gp = g_off
    ap = 6 - rp - gp - bp
    E, O = horizontal_lines(S)
    up, dn = row_maps(H)
    Eu, Ed, Ou, Od = E[up], E[dn], O[up], O[dn]
    VE, VO = avg(Eu, Ed), avg(Ou, Od)
    even_x = (np.arange(W) % 2 == 0)[None, :]
    # T = 0 rows (merge_bg, orc:43-66)
    g_bg = np.where(even_x, avg(VE, O), O)
    # T = 1 rows (merge_gr, orc:69-92)
    g_gr = np.where(even_x, E, avg(VO, E))
    T = ((np.arange(H) & 1) ^ int(swap_rows)).astype(bool)[:, None]
    out = np.empty((H, W, 4), np.uint8)
    out[..., bp] = np.where(T, VE, E)
    out[..., rp] = np.where(T, O, VO)
    out[..., gp] = np.where(T, g_gr, g_bg)
    out[..., ap] = 255
    return out


def rgb2bayer(P, pattern, r_off=1, g_off=2, b_off=3):
    """Inverse element (gst/bayer/gstrgb2bayer.c:254-268).  P: (H, W, 4) uint8 -> (H, W) uint8.
    Pinned against the reference's compiled gst_rgb2bayer_transform (see bayer2rgb_oracle.h)."""
    if isinstance(pattern, str):
        pattern = PATTERNS[pattern]
    P = np.asarray(P, dtype=np.uint8)
    H, W, _ = P.shape
    site = ((np.arange(H) & 1)[:, None] << 1) | (np.arange(W) & 1)[None, :]
    return np.where(site == pattern, P[..., b_off], np.where((site ^ 3) == pattern, P[..., r_off], P[..., g_off]))


def synthetic_frames(W, H, nframes, seed, first_frame=0, stride=None):
    """SURVEY.md Appendix C counter-based generator -> (nframes, H, stride)."""
    stride = W if stride is None else stride
    f = (np.arange(nframes, dtype=np.uint64) + np.uint64(first_frame))[:, None, None]
    y = np.arange(H, dtype=np.uint64)[None, :, None]
    x = np.arange(W, dtype=np.uint64)[None, None, :]
    idx = (f * np.uint64(H * W) + y * np.uint64(W) + x) & np.uint64(0xFFFFFFFF)
    z = (idx * np.uint64(2654435761) + np.uint64((seed * 0x9E3779B9) & 0xFFFFFFFF)) & np.uint64(0xFFFFFFFF)
    z ^= z >> np.uint64(16)
    z = (z * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    z ^= z >> np.uint64(13)
    z = (z * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    z ^= z >> np.uint64(16)
    out = np.zeros((nframes, H, stride), np.uint8)
    out[:, :, :W] = (z & np.uint64(0xFF)).astype(np.uint8)
    return out
