"""Host-side mirror of src/colorscheme.rs (ColorScheme).

A gradient is a [n][3] uint8 table standing in for a colorous `Gradient` (the crate is not
vendored in the reference; Viridis/Magma/Inferno/Plasma are 256-entry ramps there).  Colour
evaluation happens on the GPU: `color_for` and whole pixel columns go through the engine.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from .engine import SpectrogramEngine, builtin_gradient

MIN_DB = -70.0  # colorscheme.rs:16
MAX_DB = -10.0  # colorscheme.rs:17


class ColorScheme:
    """`gradient` is a [n][3] uint8 ramp (one of the 256-entry ramps) or a callable t -> (r, g, b)
    standing in for a continuous colorous gradient (spline ColorBrewer ramps, Turbo, Cividis, ...)."""

    def __init__(self, gradient, name: str, background: Optional[Tuple[int, int, int]] = None, builtin: Optional[str] = None):
        self.builtin = builtin   # a name sgx_set_builtin_scheme knows: the engine then evaluates the gradient itself
        self.gradient_fn = gradient if callable(gradient) else None
        self.gradient = None if callable(gradient) else np.ascontiguousarray(gradient, np.uint8).reshape(-1, 3)
        self.name = name
        self._background = background

    @classmethod
    def new_mono(cls, gradient, name: str) -> "ColorScheme":
        """colorscheme.rs:24-30"""
        return cls(_resolve(gradient), name, None, _builtin_name(gradient))

    @classmethod
    def new_stereo(cls, gradient, background: Sequence[int], name: str) -> "ColorScheme":
        """colorscheme.rs:32-39"""
        return cls(_resolve(gradient), name, tuple(int(x) for x in background), _builtin_name(gradient))

    @property
    def is_stereo(self) -> bool:
        return self._background is not None

    def _eval(self, t: float) -> Tuple[int, int, int]:
        if self.gradient_fn is not None:
            return tuple(int(c) for c in self.gradient_fn(t))
        n = len(self.gradient)
        x = np.floor(t * n) if t == t else 0.0
        i = int(min(max(x, 0.0), n - 1))
        return tuple(int(c) for c in self.gradient[i])

    def background(self) -> Tuple[int, int, int]:
        # colorscheme.rs:41-44
        return self._background if self._background is not None else self._eval(0.0)

    def foreground(self) -> Tuple[int, int, int]:
        # colorscheme.rs:46-53
        return self._eval(1.0) if self._background is None else self._eval(0.5)

    def apply(self, engine: SpectrogramEngine) -> None:
        if self.builtin is not None:
            engine.set_builtin_scheme(self.builtin, stereo=self.is_stereo)
        elif self.gradient_fn is not None:
            engine.set_gradient_fn(self.gradient_fn, stereo=self.is_stereo)
        else:
            engine.set_gradient(self.gradient, stereo=self.is_stereo)

    def lookup_table(self, resolution: int, engine: SpectrogramEngine) -> np.ndarray:
        """colorscheme.rs:73-92 -- [res][res][4] float32"""
        self.apply(engine)
        return engine.lookup_table(resolution)


def _builtin_name(gradient) -> Optional[str]:
    return gradient if isinstance(gradient, str) and (gradient in CONTINUOUS or gradient in ("viridis", "magma", "inferno", "plasma")) else None


def _resolve(gradient):
    if isinstance(gradient, str):
        return CONTINUOUS[gradient] if gradient in CONTINUOUS else builtin_gradient(gradient)
    if callable(gradient):
        return gradient
    return np.asarray(gradient, np.uint8)


def _poly_gradient(cr, cg, cb):
    """d3-scale-chromatic style closed forms: per channel a quintic in t (Horner, alternating signs as
    published), clamped to [0, 255] and rounded.  PARITY UNPINNED against colorous (crate not vendored)."""
    import math

    def fn(t):
        t = 0.0 if t != t else max(0.0, min(1.0, t))
        out = []
        for c in (cr, cg, cb):
            v = c[5]
            for k in (4, 3, 2, 1, 0):
                v = c[k] + t * v
            out.append(int(max(0.0, min(255.0, math.floor(v + 0.5)))))   # Math.round: half up
        return tuple(out)
    return fn


def _cubehelix_long(h0, s0, l0, h1, s1, l1):
    """d3-interpolate's interpolateCubehelixLong((h0, s0, l0), (h1, s1, l1)) -- hue, saturation and lightness each linear
    in t, no shortest-arc on the hue, gamma 1 -- then d3-color's Cubehelix -> sRGB matrix, bytes by rounding, clamped:
    what d3-scale-chromatic, which colorous ports, evaluates for CUBEHELIX / COOL / WARM (colorscheme.rs:141,143).  The
    same arithmetic, in the same order, as helix_eval in csrc/sgx_api.hip.  PARITY UNPINNED against colorous."""
    import math

    def fn(t):
        t = 0.0 if t != t else max(0.0, min(1.0, t))
        h = (h0 + t * (h1 - h0) + 120.0) * (math.pi / 180.0)
        s, l = s0 + t * (s1 - s0), l0 + t * (l1 - l0)
        a, ch, sh = s * l * (1.0 - l), math.cos(h), math.sin(h)
        vals = (255.0 * (l + a * (-0.14861 * ch + 1.78277 * sh)),
                255.0 * (l + a * (-0.29227 * ch + -0.90649 * sh)),
                255.0 * (l + a * (1.97294 * ch)))
        return tuple(int(max(0.0, min(255.0, math.floor(v + 0.5)))) for v in vals)
    return fn


def _basis_gradient(anchors):
    """d3-interpolate's interpolateRgbBasis over ColorBrewer anchors -- what d3-scale-chromatic's ramp(scheme), which
    colorous ports, evaluates for RED_YELLOW_BLUE ... ORANGES (colorscheme.rs:130-148): a uniform cubic B-spline per
    channel, end anchors reflected, bytes by rounding to nearest.  The same arithmetic, in the same order, as
    brewer_eval in csrc/sgx_api.hip (the engine's built-in); anchors generated from matplotlib's ColorBrewer data
    (tools/gen_gradients.py -> _brewer.py).  PARITY UNPINNED against colorous (crate not vendored)."""
    import math

    a = [tuple(float(c) for c in rgb) for rgb in anchors]
    n = len(a) - 1

    def fn(t):
        if not (t > 0.0):
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
            v = ((1.0 - 3.0 * t1 + 3.0 * t2 - t3) * v0 + (4.0 - 6.0 * t2 + 3.0 * t3) * v1
                 + (1.0 + 3.0 * t1 + 3.0 * t2 - 3.0 * t3) * v2 + t3 * v3) / 6.0
            r = math.floor(v + 0.5)
            out.append(int(0.0 if r < 0.0 else (255.0 if r > 255.0 else r)))
        return tuple(out)
    return fn


# coefficient lists c0..c5 of  c0 + t (c1 + t (c2 + t (c3 + t (c4 + t c5))))
CONTINUOUS = {
    # interpolateTurbo
    "turbo": _poly_gradient((34.61, 1172.33, -10793.56, 33300.12, -38394.49, 14825.05),
                            (23.31, 557.33, 1225.33, -3574.96, 1073.77, 707.56),
                            (27.2, 3211.1, -15327.97, 27814.0, -22569.18, 6838.66)),
    # interpolateCividis
    "cividis": _poly_gradient((-4.54, -35.34, 2381.73, -6402.7, 7024.72, -2710.57),
                              (32.49, 170.73, 52.82, -131.46, 176.58, -67.37),
                              (81.24, 442.36, -2482.43, 6167.24, -6614.94, 2475.67)),
    # interpolateCubehelixDefault = cubehelixLong(cubehelix(300, 0.5, 0.0), cubehelix(-240, 0.5, 1.0)); interpolateCool / Warm
    "cubehelix": _cubehelix_long(300.0, 0.5, 0.0, -240.0, 0.5, 1.0),
    "cool": _cubehelix_long(260.0, 0.75, 0.35, 80.0, 1.5, 0.8),
    "warm": _cubehelix_long(-100.0, 0.75, 0.35, 80.0, 1.5, 0.8),
}
CLOSED_FORM = tuple(CONTINUOUS)   # evaluated by the engine itself too (poly_eval / helix_eval in csrc/sgx_api.hip)


from ._brewer import ANCHORS as _BREWER_ANCHORS  # noqa: E402  (generated)

CONTINUOUS.update({name: _basis_gradient(anchors) for name, anchors in _BREWER_ANCHORS.items()})
BREWER = tuple(_BREWER_ANCHORS)   # names the engine evaluates itself (sgx_set_builtin_scheme)


def default_color_schemes() -> List[ColorScheme]:
    """colorscheme.rs:125-151: the reference's 19 entries in the reference's order, every one evaluated by the engine
    itself (sgx_set_builtin_scheme): the four 256-entry ramps (tables), the ColorBrewer B-spline gradients (anchors from
    ColorBrewer via matplotlib, d3's interpolateRgbBasis), the closed-form Turbo / Cividis polynomials and the
    cubehelix-space interpolations CUBEHELIX and COOL (d3's interpolateCubehelixLong; the default helix is the curve
    matplotlib's `cubehelix` traces).  All [third-party, unpinned]: colorous is not vendored in the reference."""
    black = (0, 0, 0)
    return [
        ColorScheme.new_stereo("red_yellow_blue", black, "Blue-Yellow-Red (Stereo)"),
        ColorScheme.new_mono("magma", "Magma"),
        ColorScheme.new_mono("viridis", "Viridis"),
        ColorScheme.new_stereo("red_blue", black, "Blue-Red (Stereo)"),
        ColorScheme.new_stereo("spectral", black, "Spectral (Stereo)"),
        ColorScheme.new_stereo("red_yellow_green", black, "Green-Yellow-Red (Stereo)"),
        ColorScheme.new_stereo("pink_green", black, "Green-Pink (Stereo)"),
        ColorScheme.new_stereo("purple_orange", black, "Orange-Purple (Stereo)"),
        ColorScheme.new_mono("inferno", "Inferno"),
        ColorScheme.new_mono("plasma", "Plasma"),
        ColorScheme.new_mono("cividis", "Cividis"),
        ColorScheme.new_mono("cubehelix", "Cube-helix"),
        ColorScheme.new_mono("turbo", "Turbo"),
        ColorScheme.new_mono("cool", "Cool"),
        ColorScheme.new_mono("reds", "Reds"),
        ColorScheme.new_mono("blues", "Blues"),
        ColorScheme.new_mono("greens", "Greens"),
        ColorScheme.new_mono("greys", "Greys"),
        ColorScheme.new_mono("oranges", "Oranges"),
    ]
