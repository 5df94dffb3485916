"""Oracle: scan-order index tables (numpy, integer work).  TEST INFRASTRUCTURE ONLY.

A *table* is an int64 array (K, L) of flat pixel indices ``p = row * W + col``:
sequence position ``l`` of direction ``k`` reads pixel ``table[k, l]``.

Families (reference direction order is preserved):
  raster   K=4  Models/SS2D/csms6s.py:13-22     CrossScan
  line     K=4  Models/SS2D/SpiralLine.py:27-82 generate_indices (the Helix half;
                the full Helix scan = raster(4) + line(4), csms6s.py:161-172)
  window   K=4  Models/SS2D/Window.py:3-35      generate_window_indices
  dilation K=4  Models/SS2D/Dilation.py:3-45    generate_dilation_indices

Size rule.  The reference only ships tables for H in {12,24,48,96} (csms6s.py:58-61,
107-110, 157) with window sizes 4/8/12/16 and dilation rate 4.  Other sizes have NO
reference behaviour; ``default_window_size`` documents the rule this project defines
for them (also DESIGN.md).
"""
import hashlib

import numpy as np

REFERENCE_WINDOW = {12: 4, 24: 8, 48: 12, 96: 16}  # csms6s.py:107-108
REFERENCE_DILATION = 4  # csms6s.py:59


def default_window_size(h: int) -> int:
    """Window side for feature size h.  Reference values where they exist."""
    if h in REFERENCE_WINDOW:
        return REFERENCE_WINDOW[h]
    if h >= 64 and h % 16 == 0:
        return 16
    if h % 8 == 0 and h > 8:
        return 8
    if h % 4 == 0 and h > 4:
        return 4
    if h % 2 == 0 and h > 2:
        return 2
    return 1


def raster_table(h: int, w: int) -> np.ndarray:
    """csms6s.py:18-22: row-major, column-major, and the flips of both."""
    rowmaj = np.arange(h * w, dtype=np.int64)
    # direction 1 walks x.transpose(2,3) row-major: position w_*H + h_ reads pixel (h_, w_)
    colmaj = rowmaj.reshape(h, w).T.reshape(-1)
    return np.stack([rowmaj, colmaj, rowmaj[::-1], colmaj[::-1]])


def _bresenham(x0: int, y0: int, x1: int, y1: int):
    """Integer line from (x0,y0) to (x1,y1) inclusive, error-accumulator form used by
    SpiralLine.py:3-24 (err = dx - dy; x steps when 2err > -dy; y steps when 2err < dx)."""
    dx, dy = abs(x1 - x0), abs(y1 - y0)
    step_x = 1 if x1 > x0 else -1
    step_y = 1 if y1 > y0 else -1
    err = dx - dy
    x, y = x0, y0
    pts = [(x, y)]
    while (x, y) != (x1, y1):
        twice = 2 * err
        move_x = twice > -dy
        move_y = twice < dx
        if move_x:
            err -= dy
            x += step_x
        if move_y:
            err += dx
            y += step_y
        pts.append((x, y))
    return pts


def line_table(h: int, w: int) -> np.ndarray:
    """SpiralLine.py:27-82.  Two interleaved fans of Bresenham lines through the image:
    set A = lines leaving the left edge at even rows then the bottom edge at even columns,
    set B = the odd ones.  Directions: [A, A with every line reversed, B, B reversed].
    Flat index is ``x + y*H`` (SpiralLine.py:103), i.e. x is the column, y the row."""
    assert h == w, "reference asserts square maps (modules.py:212)"

    def fan(parity):
        lines = []
        for r in range(parity, h, 2):
            lines.append(_bresenham(0, r, h - 1, w - 1 - r))
        if parity == 0:
            c0 = 0 if h % 2 == 0 else 2
        else:
            c0 = 1
            if h % 2 != 0:
                lines.append(_bresenham(0, w - 1, h - 1, 0))
        for c in range(c0, w, 2):
            lines.append(_bresenham(c, w - 1, h - 1 - c, 0))
        return lines

    out = []
    for parity in (0, 1):
        lines = fan(parity)
        fwd = [x + y * h for ln in lines for (x, y) in ln]
        rev = [x + y * h for ln in lines for (x, y) in ln[::-1]]
        out += [fwd, rev]
    return np.asarray(out, dtype=np.int64)


def window_table(h: int, w: int, ws: int = None) -> np.ndarray:
    """Window.py:3-35.  Non-overlapping ws x ws windows.
    dir0: windows row-major, pixels row-major inside; dir1 = reverse(dir0);
    dir2: windows column-major, pixels column-major inside; dir3 = reverse(dir2)."""
    assert h == w
    ws = default_window_size(h) if ws is None else ws
    assert h % ws == 0
    n = h // ws
    pix = np.arange(h * w, dtype=np.int64).reshape(n, ws, n, ws)  # [wr, r, wc, c]
    d0 = pix.transpose(0, 2, 1, 3).reshape(-1)  # wr, wc, r, c
    d2 = pix.transpose(2, 0, 3, 1).reshape(-1)  # wc, wr, c, r
    return np.stack([d0, d0[::-1], d2, d2[::-1]])


def dilation_table(h: int, w: int, rate: int = REFERENCE_DILATION) -> np.ndarray:
    """Dilation.py:3-45.  The four raster orders [row-major, column-major, reversed
    row-major, reversed column-major], each regrouped by ``position mod rate``."""
    assert h == w
    base = raster_table(h, w)
    order = np.concatenate([np.arange(m, h * w, rate) for m in range(rate)])
    return base[:, order]


def helix_table(h: int, w: int) -> np.ndarray:
    """csms6s.py:161-172: K=8 = raster(4) followed by line(4)."""
    return np.concatenate([raster_table(h, w), line_table(h, w)], axis=0)


FAMILIES = {
    "raster": raster_table,
    "line": line_table,
    "helix": helix_table,
    "window": window_table,
    "dilation": dilation_table,
}


def table(family: str, h: int, w: int = None) -> np.ndarray:
    return FAMILIES[family](h, h if w is None else w)


def table_hash(row: np.ndarray) -> str:
    """First 16 hex of SHA-256 over the little-endian int64 flat gather index of ONE
    direction -- the digest SURVEY.md section 8c tabulates for the reference tables."""
    return hashlib.sha256(np.ascontiguousarray(row, dtype="<i8").tobytes()).hexdigest()[:16]
