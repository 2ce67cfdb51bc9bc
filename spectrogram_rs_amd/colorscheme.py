"""Host-side mirror of src/colorscheme.rs (ColorScheme).

A gradient is a [n][3] uint8 table standing in for a colorous `Gradient` (the crate is not
vendored in the reference; Viridis/Magma/Inferno/Plasma are 256-entry ramps there).  Colour
evaluation happens on the GPU: `color_for` and whole pixel columns go through the engine.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from .engine import SpectrogramEngine, builtin_gradient, builtin_gradient_eval

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


RAMPS = ("viridis", "magma", "inferno", "plasma")     # 256-entry tables (colorscheme.rs:131-139)


def _is_builtin(name: str) -> bool:
    """does the library know a gradient of this name (sgx_builtin_gradient_eval answers for tables and continuous ones alike)"""
    try:
        builtin_gradient_eval(name, 0.0)
        return True
    except KeyError:
        return False


def _builtin_name(gradient) -> Optional[str]:
    return gradient if isinstance(gradient, str) and _is_builtin(gradient) else None


def _resolve(gradient):
    """a gradient name -> what ColorScheme evaluates on the host side (foreground / background, colorscheme.rs:41-53): the table
    of a 256-entry ramp, or for a continuous built-in (the ColorBrewer splines, Turbo, Cividis, Cube-helix, Cool) a callable that
    asks the LIBRARY (sgx_builtin_gradient_eval) -- this package holds no colour arithmetic of its own; the checker's restatement
    of the published formulas lives in oracle/gradients.py"""
    if isinstance(gradient, str):
        if gradient in RAMPS:
            return builtin_gradient(gradient)
        if not _is_builtin(gradient):
            raise KeyError(gradient)
        return lambda t, _name=gradient: builtin_gradient_eval(_name, t)
    if callable(gradient):
        return gradient
    return np.asarray(gradient, np.uint8)


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
