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
    def __init__(self, gradient: np.ndarray, name: str, background: Optional[Tuple[int, int, int]] = None):
        self.gradient = np.ascontiguousarray(gradient, np.uint8).reshape(-1, 3)
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
        engine.set_gradient(self.gradient, stereo=self.is_stereo)

    def lookup_table(self, resolution: int, engine: SpectrogramEngine) -> np.ndarray:
        """colorscheme.rs:73-92 -- [res][res][4] float32"""
        self.apply(engine)
        return engine.lookup_table(resolution)


def _resolve(gradient) -> np.ndarray:
    if isinstance(gradient, str):
        return builtin_gradient(gradient)
    return np.asarray(gradient, np.uint8)


def default_color_schemes() -> List[ColorScheme]:
    """colorscheme.rs:125-151, restricted to the gradients this build carries tables for (the
    four 256-entry ramps).  The spline-interpolated ColorBrewer gradients and the closed-form
    Turbo/Cividis/Cubehelix/Cool ramps are listed in DESIGN.md as not yet built."""
    return [
        ColorScheme.new_mono("magma", "Magma"),
        ColorScheme.new_mono("viridis", "Viridis"),
        ColorScheme.new_mono("inferno", "Inferno"),
        ColorScheme.new_mono("plasma", "Plasma"),
    ]
