#!/usr/bin/env python3
"""Regenerate tests/golden/brewer_anchors.npz (run from the repo root: python tests/golden/make_brewer_anchors.py).

The ColorBrewer anchor colours of the spline gradients the reference lists (colorscheme.rs:130-148: colorous'
RED_YELLOW_BLUE ... ORANGES, ports of d3-scale-chromatic's scheme arrays): 11-class diverging and 9-class sequential
schemes, read here from matplotlib's copy of ColorBrewer (matplotlib._cm._<Name>_data).  They are the CHECKER's data
(oracle/gradients.py); the product carries its own copy inside csrc/sgx_gradients.inc (tools/gen_gradients.py), and
tests/test_host_logic.py compares what the library evaluates from its copy with what the oracle evaluates from this one.
"""
import os

import numpy as np

GOLD = os.path.dirname(os.path.abspath(__file__))
SCHEMES = {"red_yellow_blue": "RdYlBu", "red_blue": "RdBu", "spectral": "Spectral", "red_yellow_green": "RdYlGn", "pink_green": "PiYG",
           "purple_orange": "PuOr", "purple_green": "PRGn", "brown_green": "BrBG", "red_grey": "RdGy", "reds": "Reds", "blues": "Blues",
           "greens": "Greens", "greys": "Greys", "oranges": "Oranges", "purples": "Purples"}


def main():
    import matplotlib
    import matplotlib._cm as _cm

    out = {}
    for name, mpl in SCHEMES.items():
        data = np.asarray(getattr(_cm, "_%s_data" % mpl), np.float64)
        a = np.rint(data * 255.0).astype(np.uint8)
        assert a.shape in ((11, 3), (9, 3)) and np.abs(a / 255.0 - data).max() < 1e-6     # ColorBrewer's colours ARE 8-bit
        out[name] = a
    # two colours ColorBrewer publishes (RdYlBu 11-class: #a50026 ... #313695)
    assert tuple(out["red_yellow_blue"][0]) == (0xa5, 0x00, 0x26) and tuple(out["red_yellow_blue"][-1]) == (0x31, 0x36, 0x95)
    np.savez(os.path.join(GOLD, "brewer_anchors.npz"), **out)
    print("wrote brewer_anchors.npz from matplotlib", matplotlib.__version__)


if __name__ == "__main__":
    main()
