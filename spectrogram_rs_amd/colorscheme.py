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

    def __init__(self, gradient, name: str, background: Optional[Tuple[int, int, int]] = None):
        self.gradient_fn = gradient if callable(gradient) else None
        self.gradient = None if callable(gradient) else np.ascontiguousarray(gradient, np.uint8).reshape(-1, 3)
        self.name = name
        self._background = background

    @classmethod
    def new_mono(cls, gradient, name: str) -> "ColorScheme":
        """colorscheme.rs:24-30"""
        return cls(_resolve(gradient), name, None)

    @classmethod
    def new_stereo(cls, gradient, background: Sequence[int], name: str) -> "ColorScheme":
        """colorscheme.rs:32-39"""
        return cls(_resolve(gradient), name, tuple(int(x) for x in background))

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
        if self.gradient_fn is not None:
            engine.set_gradient_fn(self.gradient_fn, stereo=self.is_stereo)
        else:
            engine.set_gradient(self.gradient, stereo=self.is_stereo)

    def lookup_table(self, resolution: int, engine: SpectrogramEngine) -> np.ndarray:
        """colorscheme.rs:73-92 -- [res][res][4] float32"""
        self.apply(engine)
        return engine.lookup_table(resolution)


def _resolve(gradient):
    if isinstance(gradient, str):
        return CONTINUOUS[gradient] if gradient in CONTINUOUS else builtin_gradient(gradient)
    if callable(gradient):
        return gradient
    return np.asarray(gradient, np.uint8)


def _poly_gradient(cr, cg, cb):
    """d3-scale-chromatic style closed forms: per channel a quintic in t (Horner, alternating signs as
    published), clamped to [0, 255] and rounded.  PARITY UNPINNED against colorous (crate not vendored)."""
    def fn(t):
        t = 0.0 if t != t else max(0.0, min(1.0, t))
        out = []
        for c in (cr, cg, cb):
            v = c[5]
            for k in (4, 3, 2, 1, 0):
                v = c[k] + t * v
            out.append(int(max(0, min(255, round(v)))))
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
}


def default_color_schemes() -> List[ColorScheme]:
    """colorscheme.rs:125-151, restricted to what this package can evaluate itself: the four 256-entry
    ramps (tables) and the closed-form Turbo / Cividis polynomials (continuous gradients through the
    callback route).  Any other colorous gradient -- the spline-interpolated ColorBrewer ramps, Cubehelix,
    Cool -- is rendered by handing the engine colorous' own eval_continuous as the callback."""
    return [
        ColorScheme.new_mono("magma", "Magma"),
        ColorScheme.new_mono("viridis", "Viridis"),
        ColorScheme.new_mono("inferno", "Inferno"),
        ColorScheme.new_mono("plasma", "Plasma"),
        ColorScheme.new_mono("cividis", "Cividis"),
        ColorScheme.new_mono("turbo", "Turbo"),
    ]
