"""oracle/gradients.py -- TEST INFRASTRUCTURE (the checker's half of the colour-scheme tests; nothing under
spectrogram_rs_amd/ imports this, and this imports nothing from there).

The continuous gradients of the reference's scheme list (src/colorscheme.rs:125-151) are colorous 1.0.12 constants
(Cargo.lock:530-532; un-vendored), ports of d3-scale-chromatic.  Restated here from the PUBLISHED d3 formulas, for the
oracle's pixel stage to evaluate (oracle.set_gradient_fn):

  interpolateRgbBasis(scheme)   uniform cubic B-spline per channel over the ColorBrewer anchors, end anchors reflected,
                                bytes by Math.round, clamped                         RED_YELLOW_BLUE ... ORANGES  (:130-148)
  interpolateTurbo / Cividis    per channel a quintic in t                           TURBO, CIVIDIS               (:140,142)
  interpolateCubehelixLong      (h, s, l) linear in t, no shortest arc, gamma 1,
                                then d3-color's Cubehelix -> sRGB matrix, bytes by
                                TRUNCATION (the reference's Cool screenshot decides)  CUBEHELIX, COOL (and WARM)   (:141,143)

PARITY UNPINNED against colorous itself except where the reference's own screenshots decide (tests/test_host_logic.py:
the four 256-entry tables' values and Cool's curve + byte rule; it also says which others have an independent pin: the
default cube helix against matplotlib's `cubehelix`, +-1 LSB; the spline's end colours against ColorBrewer's; Turbo /
Cividis only in shape).  Anchors: tests/golden/brewer_anchors.npz (make_brewer_anchors.py).
"""
import math
import os

import numpy as np

_ANCHORS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "brewer_anchors.npz")


def _byte(v):
    r = math.floor(v + 0.5)            # Math.round: half up
    return int(0.0 if r < 0.0 else (255.0 if r > 255.0 else r))


def _byte_trunc(v):
    """Rust `as u8` on a float: toward zero, saturating, NaN -> 0 -- what colorous does for the cubehelix family, as the
    reference's own screenshot of Cool shows (tests/golden/screenshot_colours.npz: (109, 63, 169) for (109.70, 63.81, 169.91))"""
    return int(0.0 if not (v > 0.0) else (255.0 if v >= 255.0 else math.floor(v)))


def rgb_basis(anchors):
    """d3-interpolate: basis(t1, v0, v1, v2, v3) = ((1 - 3 t1 + 3 t2 - t3) v0 + (4 - 6 t2 + 3 t3) v1 + (1 + 3 t1 + 3 t2 - 3 t3) v2 + t3 v3) / 6
    with i = t <= 0 ? (t = 0) : t >= 1 ? (t = 1, n - 1) : floor(t n), v0 = i > 0 ? v[i-1] : 2 v1 - v2, v3 = i < n - 1 ? v[i+2] : 2 v2 - v1"""
    a = [tuple(float(c) for c in rgb) for rgb in anchors]
    n = len(a) - 1

    def fn(t):
        if not (t > 0.0):              # also NaN
            t, i = 0.0, 0
        elif t >= 1.0:
            t, i = 1.0, n - 1
        else:
            i = int(math.floor(t * float(n)))
        t1 = (t - float(i) / float(n)) * float(n)
        t2 = t1 * t1
        t3 = t2 * t1
        out = []
        for ch in range(3):
            v1, v2 = a[i][ch], a[i + 1][ch]
            v0 = a[i - 1][ch] if i > 0 else 2.0 * v1 - v2
            v3 = a[i + 2][ch] if i < n - 1 else 2.0 * v2 - v1
            out.append(_byte(((1.0 - 3.0 * t1 + 3.0 * t2 - t3) * v0 + (4.0 - 6.0 * t2 + 3.0 * t3) * v1
                              + (1.0 + 3.0 * t1 + 3.0 * t2 - 3.0 * t3) * v2 + t3 * v3) / 6.0))
        return tuple(out)
    return fn


def quintic(cr, cg, cb):
    """d3-scale-chromatic's closed forms: c0 + t (c1 + t (c2 + t (c3 + t (c4 + t c5)))) per channel, t clamped to [0, 1]"""
    def fn(t):
        t = 0.0 if t != t else max(0.0, min(1.0, t))
        out = []
        for c in (cr, cg, cb):
            v = c[5]
            for k in (4, 3, 2, 1, 0):
                v = c[k] + t * v
            out.append(_byte(v))
        return tuple(out)
    return fn


def cubehelix_long(h0, s0, l0, h1, s1, l1):
    """d3-interpolate's interpolateCubehelixLong between two cubehelix colours + d3-color's Cubehelix.rgb()"""
    def fn(t):
        t = 0.0 if t != t else max(0.0, min(1.0, t))
        h = (h0 + t * (h1 - h0) + 120.0) * (math.pi / 180.0)
        s, l = s0 + t * (s1 - s0), l0 + t * (l1 - l0)
        a, ch, sh = s * l * (1.0 - l), math.cos(h), math.sin(h)
        return (_byte_trunc(255.0 * (l + a * (-0.14861 * ch + 1.78277 * sh))),
                _byte_trunc(255.0 * (l + a * (-0.29227 * ch + -0.90649 * sh))),
                _byte_trunc(255.0 * (l + a * (1.97294 * ch))))
    return fn


CLOSED_FORM = {
    "turbo": quintic((34.61, 1172.33, -10793.56, 33300.12, -38394.49, 14825.05),
                     (23.31, 557.33, 1225.33, -3574.96, 1073.77, 707.56),
                     (27.2, 3211.1, -15327.97, 27814.0, -22569.18, 6838.66)),
    "cividis": quintic((-4.54, -35.34, 2381.73, -6402.7, 7024.72, -2710.57),
                       (32.49, 170.73, 52.82, -131.46, 176.58, -67.37),
                       (81.24, 442.36, -2482.43, 6167.24, -6614.94, 2475.67)),
    "cubehelix": cubehelix_long(300.0, 0.5, 0.0, -240.0, 0.5, 1.0),     # interpolateCubehelixDefault
    "cool": cubehelix_long(260.0, 0.75, 0.35, 80.0, 1.5, 0.8),          # interpolateCool
    "warm": cubehelix_long(-100.0, 0.75, 0.35, 80.0, 1.5, 0.8),         # interpolateWarm
}
ANCHORS = {k: v for k, v in np.load(_ANCHORS).items()}
BREWER = {name: rgb_basis(a) for name, a in ANCHORS.items()}
CONTINUOUS = dict(CLOSED_FORM, **BREWER)
