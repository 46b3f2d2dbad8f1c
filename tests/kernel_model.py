"""TEST INFRASTRUCTURE: a NumPy model of the packed-byte arithmetic of the HIP kernel
(gst-plugins-bad_amd/csrc/mibayer_kernels.hip: row_lines, merge_rows, store_pixels), one uint32 per lane exactly as
on the device, driven by the REAL v_perm selectors that the library computes (mibayer_plan_selectors).  It lets the
kernel's algorithmic restatement -- neighbour dwords, edge-lane overrides, double-rounded green, byte interleave --
be checked against the oracle on a CPU, where the kernel itself cannot run."""
import ctypes

import numpy as np

U32 = np.uint32


def lerp_u8(a, b):
    """v_lerp_u8 (a, b, 0x01010101): per byte (a + b + 1) >> 1."""
    out = np.zeros_like(a)
    for k in range(4):
        x = (a >> U32(8 * k)) & U32(0xFF)
        y = (b >> U32(8 * k)) & U32(0xFF)
        out |= (((x + y + U32(1)) >> U32(1)) & U32(0xFF)) << U32(8 * k)
    return out


def alignbit(hi, lo, nbytes):
    """v_alignbit_b32: ({hi,lo} >> 8*nbytes) & 0xffffffff."""
    v = (hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)
    return ((v >> np.uint64(8 * nbytes)) & np.uint64(0xFFFFFFFF)).astype(U32)


def perm(s0, s1, sel):
    """v_perm_b32: byte i = byte sel[i] of {s0,s1} (0-3 = s1, 4-7 = s0), 12 = 0x00, >= 13 = 0xff."""
    out = np.zeros_like(s0)
    for i in range(4):
        idx = (int(sel) >> (8 * i)) & 0xFF
        if idx < 4:
            b = (s1 >> U32(8 * idx)) & U32(0xFF)
        elif idx < 8:
            b = (s0 >> U32(8 * (idx - 4))) & U32(0xFF)
        elif idx == 12:
            b = np.zeros_like(s0)
        else:
            b = np.full_like(s0, 0xFF)
        out |= b << U32(8 * i)
    return out


def bsel(m, a, b):
    m = U32(m)
    return (a & m) | (b & ~m)


EVEN = 0x00FF00FF


def row_lines(C, width):
    """C: (rows, nlanes) uint32 = the source dwords, lane l holds columns 4l..4l+3.
    Returns (E, O) per lane, with the device's edge-lane overrides."""
    rows, nl = C.shape
    cl = np.zeros_like(C)
    cr = np.zeros_like(C)
    cl[:, 1:] = C[:, :-1]              # DPP wave_shr:1 (lane 0 keeps the halo: unused, overridden below)
    cr[:, :-1] = C[:, 1:]              # DPP wave_shl:1
    lsh = alignbit(C, cl, 3)           # [cl.3, c0, c1, c2]
    rsh = alignbit(cr, C, 1)           # [c1, c2, c3, cr.0]
    # first lane: O[0] = S[1]
    lsh[:, 0] = (lsh[:, 0] & U32(0xFFFFFF00)) | ((C[:, 0] >> U32(8)) & U32(0xFF))
    last = (width + 3) // 4 - 1
    if width % 4 == 0:                 # lastmode 1: columns W-4..W-1
        t = lsh[:, last] >> U32(16)
        rsh[:, last] = t | (t << U32(16))
    else:                              # lastmode 2: columns W-2..W-1 only
        rsh[:, last] = lsh[:, last]
    a = lerp_u8(lsh, rsh)
    return bsel(EVEN, C, a), bsel(EVEN, a, C)


def bayer2rgb_model(pkg, src, width, pattern, fmt):
    """src: (H, stride) uint8 -> (H, 4*width) uint8 through the lane-level model."""
    H = src.shape[0]
    nl = (width + 3) // 4
    padded = np.zeros((H, 4 * nl), np.uint8)
    padded[:, :min(src.shape[1], 4 * nl)] = src[:, :4 * nl]
    C = padded.view("<u4").astype(U32)
    r, g, b = pkg.FORMATS[fmt] if isinstance(fmt, str) else fmt
    pat = pkg.PATTERNS[pattern] if isinstance(pattern, str) else pattern
    cfg = pkg.make_cfg(width, H, pat, (r, g, b))
    sel = (ctypes.c_uint32 * 4)()
    swap = ctypes.c_int()
    rc = pkg.lib().mibayer_plan_selectors(ctypes.byref(cfg), ctypes.byref(sel), ctypes.byref(swap))
    assert rc == 0, rc
    E, O = row_lines(C, width)
    up = np.arange(H) - 1
    up[0] = 1                                          # map_row (y < 0) = 1
    dn = np.arange(H) + 1
    dn[H - 1] = H - 4 if H >= 4 else 1                 # dn_last
    ve, vo = lerp_u8(E[up], E[dn]), lerp_u8(O[up], O[dn])
    T = ((np.arange(H) & 1) ^ swap.value).astype(bool)[:, None]
    bq = np.where(T, ve, E)
    rq = np.where(T, O, vo)
    g_bg = bsel(EVEN, lerp_u8(ve, O), O)
    g_gr = bsel(EVEN, E, lerp_u8(vo, E))
    G = np.where(T, g_gr, g_bg)
    m_lo = perm(rq, bq, 0x01050004)
    m_hi = perm(rq, bq, 0x03070206)
    px = np.stack([perm(m_lo, G, sel[0]), perm(m_lo, G, sel[1]), perm(m_hi, G, sel[2]), perm(m_hi, G, sel[3])],
                  axis=-1)                             # (H, nl, 4) dwords = 4 output pixels per lane
    out = np.ascontiguousarray(px.astype("<u4")).view(np.uint8).reshape(H, 16 * nl)
    return out[:, :4 * width]


def _lines_at(P, xs, width):
    """Lane-level (E, O) for lanes whose four columns start at xs (even, any alignment) -- the per-row shifted lane
    map of bayer2rgb_lds_aligned_kernel.  P: (rows, 4 + padded width + margin) uint8, column x of the frame at
    P[:, 4 + x], zero outside the readable columns (what the LDS tile holds)."""
    def dword(off):
        cols = (xs + off + 4)[:, None] + np.arange(4)[None, :]
        b = P[:, cols].astype(U32)                     # (rows, nl, 4)
        return b[..., 0] | (b[..., 1] << U32(8)) | (b[..., 2] << U32(16)) | (b[..., 3] << U32(24))
    c, cl, cr = dword(0), dword(-4), dword(4)
    lsh = alignbit(c, cl, 3)
    rsh = alignbit(cr, c, 1)
    first = xs == 0
    lsh[:, first] = (lsh[:, first] & U32(0xFFFFFF00)) | ((c[:, first] >> U32(8)) & U32(0xFF))
    m1 = xs + 4 == width
    t = lsh[:, m1] >> U32(16)
    rsh[:, m1] = t | (t << U32(16))
    m2 = xs + 2 == width
    rsh[:, m2] = lsh[:, m2]
    a = lerp_u8(lsh, rsh)
    return bsel(EVEN, c, a), bsel(EVEN, a, c)


def bayer2rgb_model_aligned(pkg, src, width, pattern, fmt, dst_stride, base_mod, align=64):
    """The sector-aligned arm (bayer2rgb_lds_aligned_kernel): row j of the output starts at address
    base_mod + j * dst_stride (mod align); its lanes are shifted by s_j = ((-addr) mod align) / 4 columns, the s_j
    columns in front are the row's head.  Returns (H, 4*width) uint8 and checks that every pixel is written once."""
    H = src.shape[0]
    r, g, b = pkg.FORMATS[fmt] if isinstance(fmt, str) else fmt
    pat = pkg.PATTERNS[pattern] if isinstance(pattern, str) else pattern
    cfg = pkg.make_cfg(width, H, pat, (r, g, b))
    sel = (ctypes.c_uint32 * 4)()
    swap = ctypes.c_int()
    assert pkg.lib().mibayer_plan_selectors(ctypes.byref(cfg), ctypes.byref(sel), ctypes.byref(swap)) == 0
    wl4 = (width + 3) & ~3
    P = np.zeros((H, 4 + wl4 + 48 + align), np.uint8)
    P[:, 4:4 + min(src.shape[1], wl4)] = src[:, :wl4]
    up = np.arange(H) - 1
    up[0] = 1
    dn = np.arange(H) + 1
    dn[H - 1] = H - 4 if H >= 4 else 1
    out = np.zeros((H, 4 * width), np.uint8)
    written = np.zeros((H, width), np.int32)
    for j in range(H):
        addr = base_mod + j * dst_stride
        assert addr % 8 == 0
        s = ((-addr) % align) // 4
        T = (j & 1) ^ swap.value
        passes = [(s + 4 * np.arange(max(0, (width - s + 3) // 4)), width)]            # main lanes
        if s > 0:
            passes.append((4 * np.arange((s + 3) // 4), min(s, width)))                  # head lanes
        for xs, lim in passes:
            if xs.size == 0:
                continue
            rows = P[[up[j], j, dn[j]]]
            E, O = _lines_at(rows, xs, width)
            ve, vo = lerp_u8(E[0], E[2]), lerp_u8(O[0], O[2])
            if T:
                bq, rq, G = ve, O[1], bsel(EVEN, E[1], lerp_u8(vo, E[1]))
            else:
                bq, rq, G = E[1], vo, bsel(EVEN, lerp_u8(ve, O[1]), O[1])
            m_lo = perm(rq, bq, 0x01050004)
            m_hi = perm(rq, bq, 0x03070206)
            px = np.stack([perm(m_lo, G, sel[0]), perm(m_lo, G, sel[1]), perm(m_hi, G, sel[2]),
                           perm(m_hi, G, sel[3])], axis=-1).astype("<u4")
            pxb = np.ascontiguousarray(px).view(np.uint8).reshape(-1, 16)
            for i, x in enumerate(xs):
                n = 4 if x + 4 <= lim else (2 if x + 2 == lim else 0)
                out[j, 4 * x:4 * (x + n)] = pxb[i, :4 * n]
                written[j, x:x + n] += 1
    assert (written == 1).all(), "aligned lane map: a pixel written %s times" % np.unique(written)
    return out
