#!/usr/bin/env python3
"""Regenerate tests/golden/screenshot_colours.npz (build container only: python tests/golden/make_screenshot_colours.py).

The reference holds no test vectors; the only outputs of its colour path under /root/reference are four screenshots that
README.md links: screenshots/colorscheme-{viridis,magma,plasma,cool}.png.  They show the plotters widget filled with
`background()` = gradient.eval_continuous(0.0), axes drawn in `foreground()` = eval_continuous(1.0)
(src/colorscheme.rs:41-53) and pixels from `color_for` (:55-71), scaled by the toolkit (so many pixels are blends, but
many are verbatim gradient colours).

Stored per image -- DATA, not the picture:
  <name>_background   the most frequent opaque colour
  <name>_axis         the colour of the longest horizontal single-colour run that is not the background, the window's
                      one-pixel border (rows within 4 px of the first / last opaque row) left out: the x axis line
  <name>_colours      every distinct opaque RGB triple, [n][3] u8
  <name>_counts       how many opaque pixels carry it, [n] u32
  <name>_axis_geometry  [y_top, y_bottom] of the vertical (frequency) axis line and the centre rows of its tick marks, top to bottom,
                      in screenshot pixels: the log-frequency axis of simple_spectrogram.rs:107 (32 .. 22030 Hz, ticks at 32 * 2^k) as drawn
tests/test_host_logic.py reads the file; nothing on the GPU box needs /root/reference.
"""
import os
import sys

import numpy as np

GOLD = os.path.dirname(os.path.abspath(__file__))
SHOTS = "/root/reference/screenshots"
NAMES = ("viridis", "magma", "plasma", "cool")


def longest_run_colour(rgb, opaque, background):
    """colour of the longest horizontal run of one opaque non-background colour"""
    best, colour = 0, None
    key = (rgb[..., 0].astype(np.uint32) << 16) | (rgb[..., 1].astype(np.uint32) << 8) | rgb[..., 2]
    bg = (int(background[0]) << 16) | (int(background[1]) << 8) | int(background[2])
    rows = np.flatnonzero(opaque.any(axis=1))
    for y in range(rows[0] + 5, rows[-1] - 4):
        row = np.where(opaque[y], key[y], 0xFFFFFFFF)
        edges = np.flatnonzero(np.diff(row) != 0)
        starts = np.concatenate(([0], edges + 1))
        ends = np.concatenate((edges + 1, [row.size]))
        for s, e in zip(starts, ends):
            if e - s > best and row[s] != 0xFFFFFFFF and row[s] != bg:
                best, colour = e - s, rgb[y, s].copy()
    return colour, best


def axis_geometry(rgb, opaque, axis):
    """[y_top, y_bottom, tick centres ...] of the vertical axis: the column with the most axis-coloured pixels above the x axis line;
    ticks = runs of axis colour just left of it"""
    m = (rgb == axis).all(axis=2) & opaque
    y_axis = int(np.argmax(m.sum(axis=1)))                      # the horizontal (time) axis line
    x_axis = int(np.argmax(m[:y_axis].sum(axis=0)))             # the vertical (frequency) axis line
    ys = np.flatnonzero(m[:y_axis + 1, x_axis])
    left = np.flatnonzero(m[:y_axis + 1, x_axis - 4])
    runs = np.split(left, np.flatnonzero(np.diff(left) > 1) + 1)
    ticks = [float(r.mean()) for r in runs if len(r)]
    return np.array([float(ys.min()), float(y_axis)] + ticks, np.float64)


def main():
    import matplotlib.image as mi

    out = {}
    for name in NAMES:
        im = mi.imread(os.path.join(SHOTS, "colorscheme-%s.png" % name))
        u = np.rint(im * 255.0).astype(np.uint8)
        opaque = u[..., 3] == 255
        rgb = u[..., :3]
        cols, cnt = np.unique(rgb[opaque], axis=0, return_counts=True)
        background = cols[np.argmax(cnt)]
        axis, run = longest_run_colour(rgb, opaque, background)
        out[name + "_background"] = background
        out[name + "_axis"] = axis
        out[name + "_axis_geometry"] = axis_geometry(rgb, opaque, axis)
        out[name + "_colours"] = cols.astype(np.uint8)
        out[name + "_counts"] = cnt.astype(np.uint32)
        print(name, "opaque pixels", int(opaque.sum()), "distinct", len(cols), "background", tuple(int(c) for c in background),
              "axis", tuple(int(c) for c in axis), "run", run)
    np.savez_compressed(os.path.join(GOLD, "screenshot_colours.npz"), **out)


if __name__ == "__main__":
    if not os.path.isdir(SHOTS):
        sys.exit("the reference's screenshots are only present in the build container")
    main()
