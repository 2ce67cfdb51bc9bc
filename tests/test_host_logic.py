"""Host-side mirror of the reference interface: ring buffer, hop loop, log axis.  No GPU: the
transform is replaced by a recording fake that implements the AudioTransform interface."""
import math

import numpy as np
import pytest

import oracle
from spectrogram_rs_amd.fourier import AudioStreamTransform, AudioTransform, RingBuffer
from spectrogram_rs_amd.log_scaling import LogCoordf64


class FakeTransform(AudioTransform):
    def __init__(self, sr, W):
        self.sr, self.W, self.calls = sr, W, []

    def sample_rate(self):
        return self.sr

    def num_input_samples(self):
        return self.W

    def process(self, samples):
        s = np.asarray(samples).reshape(-1, 2)[:self.W]
        if len(s) < self.W:
            return None
        self.calls.append(s.copy())
        return s[:1]


def test_ring_buffer_semantics():
    rb = RingBuffer(8)
    assert rb.push_iter([(i, -i) for i in range(6)]) == 6
    assert rb.push_mono([9, 9, 9, 9]) == 2  # overflow is dropped (audio_input_list_model.rs:70)
    assert len(rb) == 8 and rb.iter()[6].tolist() == [9, 9]
    assert rb.skip(3) == 3 and rb.iter()[0].tolist() == [3, -3]
    assert rb.skip(100) == 5 and len(rb) == 0


def test_hop_loop_yields_reference_frames_and_skips_like_the_reference():
    sr, W, stride = 1000.0, 16, 0.004  # H = 4
    rb = RingBuffer(4096)
    data = np.arange(2 * 45, dtype=np.float32).reshape(45, 2)
    rb.push_iter(data)
    tr = FakeTransform(sr, W)
    st = AudioStreamTransform(rb, tr, stride)
    assert st.stride_samples() == 4 == oracle.hop_samples(sr, stride)
    frames = list(st.process())
    n = oracle.num_frames(45, W, 4)
    assert len(frames) == n == 8
    for t, call in enumerate(tr.calls):
        assert np.array_equal(call, data[t * 4:t * 4 + W])  # frame t covers [tH, tH+W)
    # the terminating short read also skips H (audio_transform.rs:37-41, quirk Q1)
    assert len(rb) == 45 - (n + 1) * 4
    # nothing more to yield
    assert list(st.process()) == [] and len(rb) == 45 - (n + 2) * 4
    with pytest.raises(ValueError):
        AudioStreamTransform(rb, tr, 1e-9).process().__next__()


def test_public_fields_are_reassignable():
    rb = RingBuffer(64)
    st = AudioStreamTransform(rb, FakeTransform(100.0, 4), 0.02)
    st.transform = FakeTransform(200.0, 8)  # set_sample_rate replaces the transform wholesale
    st.stride = 0.01
    assert st.stride_samples() == 2
    rb.push_iter(np.zeros((12, 2), np.float32))
    assert len(list(st.process())) == 3


def test_log_axis_matches_oracle_and_roundtrips():
    ax = LogCoordf64.reversible_log_scale(32.0, 22030.0).with_base(2.0).with_zero_point(0.0)  # simple_spectrogram.rs:107
    assert ax.linear == (math.log(32.0), math.log(22030.0)) and not ax.negative
    for p in (0, 1, 511, 1023, 1024):
        assert ax.unmap(p, (0, 1024)) == oracle.log_unmap(32.0, 22030.0, p, 0, 1024)
    assert ax.unmap(-1, (0, 1024)) is None and ax.unmap(1025, (0, 1024)) is None
    assert ax.unmap(0, (0, 1024)) == pytest.approx(32.0) and ax.unmap(1024, (0, 1024)) == pytest.approx(22030.0)
    for p in (3, 400, 1000):
        assert ax.map(ax.unmap(p, (0, 1024)), (0, 1024)) == p
    e = np.array([np.float32(ax.unmap(p, (0, 1024))) for p in range(1025)], np.float32)
    assert np.array_equal(e, oracle.bin_edges(1024))


def mixed_radix_plan(P):
    """csrc/stft_mixed.hip make_plan restated: P's factors 7, 5, 4 (pairs of twos), 3 and a last 2, grouped into stages of
    one or two factors with a product <= 28 -- fewest stages, then the smallest largest radix, then the smallest sum."""
    n, factors = P, []
    for f in (7, 5):
        while n % f == 0:
            factors.append(f)
            n //= f
    threes = []
    while n % 3 == 0:
        threes.append(3)
        n //= 3
    while n % 4 == 0:
        factors.append(4)
        n //= 4
    factors += threes
    if n % 2 == 0:
        factors.append(2)
        n //= 2
    if n != 1:
        return None
    factors.sort(reverse=True)
    best = [None, None]

    def search(rest, cur):
        if not rest:
            prods = [a * b for a, b in cur]
            key = (len(cur), max(prods), sum(prods))
            if best[0] is None or key < best[0]:
                best[0], best[1] = key, list(cur)
            return
        f, rest = rest[-1], rest[:-1]
        search(rest, cur + [(f, 1)])
        for i, g in enumerate(rest):
            if (i > 0 and rest[i] == rest[i - 1]) or f * g > 28:
                continue
            search(rest[:i] + rest[i + 1:], cur + [(max(f, g), min(f, g))])

    search(factors, [])
    odd = lambda g: (g[0] * g[1]) & (g[0] * g[1] - 1) != 0  # noqa: E731
    return sorted(best[1], key=lambda g: (not odd(g), -(g[0] * g[1]) if odd(g) else g[0] * g[1]))


def test_mixed_radix_plan_and_block_index_arithmetic_for_every_supported_length():
    # the two sizes the application produces (0.05 s at 48 and 44.1 kHz) make three trips through LDS
    assert [a * b for a, b in mixed_radix_plan(4800)] == [20, 15, 16]
    assert sorted(a * b for a, b in mixed_radix_plan(4410)) == [14, 15, 21]
    assert [a * b for a, b in mixed_radix_plan(2048)] == [8, 16, 16] and [a * b for a, b in mixed_radix_plan(8192)] == [4, 8, 16, 16]
    # csrc/stft_mixed.hip splits a butterfly number b into (block, offset) with one float multiply,
    # block = uint((b + 0.5f) * (1.0f / m)), instead of an integer division.  float32 arithmetic is the same on the
    # host: every stage of every supported length (2W <= 20480 with prime factors 2, 3, 5, 7) is checked here,
    # and that the kernel has a stage for every (RA, RB) the plan can ask for.
    have = {(7, 4), (7, 3), (7, 2), (5, 5), (5, 4), (5, 3), (5, 2), (4, 4), (4, 3), (4, 2), (3, 3), (3, 2), (7, 1), (5, 1), (4, 1), (3, 1), (2, 1)}
    # real-input mode (a mono stream, every frame its own transform) runs the W-point plan: 2400 and 2205 points at 48 and 44.1 kHz,
    # odd lengths included -- so every length from 8 on is checked, not only the even 2W
    assert [a * b for a, b in mixed_radix_plan(2400)] == [15, 10, 16] and [a * b for a, b in mixed_radix_plan(2205)] == [21, 15, 7]
    lengths = 0
    for P in range(8, 20481):
        plan = mixed_radix_plan(P)
        if plan is None:
            continue
        lengths += 1
        assert len(plan) <= 8 and set(plan) <= have, (P, plan)
        ns = P
        for ra, rb in plan:
            r = ra * rb
            m = ns // r
            b = np.arange(P // r, dtype=np.float32)
            got = ((b + np.float32(0.5)) * (np.float32(1.0) / np.float32(m))).astype(np.uint32)
            assert np.array_equal(got, (np.arange(P // r) // m).astype(np.uint32)), (P, r, m)
            ns = m
        assert ns == 1
        # the padded LDS position adds j / R_last by the same kind of multiply (j < m of the first stage at most)
        r_last = plan[-1][0] * plan[-1][1]
        j = np.arange(P // (plan[0][0] * plan[0][1]), dtype=np.float32)
        got = ((j + np.float32(0.5)) * (np.float32(1.0) / np.float32(r_last))).astype(np.uint32)
        assert np.array_equal(got, (np.arange(len(j)) // r_last).astype(np.uint32)), (P, r_last)
    assert lengths > 400


def test_default_color_schemes_follow_the_reference_list():
    # colorscheme.rs:125-151: the 19 entries in the reference's order, every one a built-in of the engine, the diverging
    # ones as stereo schemes on a black background
    from oracle.gradients import CONTINUOUS
    from spectrogram_rs_amd.colorscheme import default_color_schemes
    ds = default_color_schemes()
    assert [d.name for d in ds] == ["Blue-Yellow-Red (Stereo)", "Magma", "Viridis", "Blue-Red (Stereo)", "Spectral (Stereo)",
                                    "Green-Yellow-Red (Stereo)", "Green-Pink (Stereo)", "Orange-Purple (Stereo)", "Inferno", "Plasma",
                                    "Cividis", "Cube-helix", "Turbo", "Cool", "Reds", "Blues", "Greens", "Greys", "Oranges"]
    assert [d.is_stereo for d in ds] == [True, False, False, True, True, True, True, True] + [False] * 11
    assert all(d.builtin is not None for d in ds)
    assert all(d.background() == (0, 0, 0) for d in ds if d.is_stereo)
    # the spline passes near (not through) its interior anchors and exactly through the reflected ends
    ryb = CONTINUOUS["red_yellow_blue"]
    assert ryb(0.0) == (165, 0, 38) and ryb(1.0) == (49, 54, 149) and ryb(0.5) == (250, 248, 193)
    assert ryb(-3.0) == ryb(0.0) and ryb(7.0) == ryb(1.0) and ryb(float("nan")) == ryb(0.0)
    greys = CONTINUOUS["greys"]
    assert greys(0.0) == (255, 255, 255) and greys(1.0) == (0, 0, 0)
    vals = [greys(i / 512.0)[0] for i in range(513)]
    assert all(a >= b for a, b in zip(vals, vals[1:]))        # monotone ramp


def test_builtin_gradient_eval_matches_the_oracles_restatement():
    # the product's half: the C++ built-ins (csrc/sgx_api.hip: brewer_eval / poly_eval / helix_eval over the anchors of
    # csrc/sgx_gradients.inc), asked through the C ABI.  The checker's half: oracle/gradients.py over tests/golden/brewer_anchors.npz.
    # The product package holds no colour arithmetic in Python (its ColorScheme mirror asks the library).
    import numpy as np
    from oracle.gradients import BREWER, CLOSED_FORM, CONTINUOUS
    from spectrogram_rs_amd import colorscheme
    from spectrogram_rs_amd.engine import builtin_gradient_eval
    assert set(CLOSED_FORM) == {"turbo", "cividis", "cubehelix", "cool", "warm"} and len(BREWER) == 15
    assert not any(hasattr(colorscheme, n) for n in ("CONTINUOUS", "_basis_gradient", "_poly_gradient", "_cubehelix_long"))
    for name in CONTINUOUS:
        for t in list(np.linspace(-0.05, 1.05, 1501)) + [float("nan")]:
            assert builtin_gradient_eval(name, t) == CONTINUOUS[name](t), (name, t)
    # the mirror's host-side colours come from the library too
    for d in colorscheme.default_color_schemes():
        if d.gradient_fn is not None:
            assert d.foreground() == builtin_gradient_eval(d.builtin, 0.5 if d.is_stereo else 1.0)


def test_independent_pins_of_the_colour_schemes_and_which_stay_unpinned():
    # colorous (un-vendored) is the only authority on the reference's 19 gradients (colorscheme.rs:125-151).  What this image can
    # pin INDEPENDENTLY of the build's own restatements, and how tightly:
    #   Cube-helix                      matplotlib's `cubehelix` map (Green 2011: start 0.5, -1.5 rotations, hue 1, gamma 1), the same
    #                                   published curve d3's interpolateCubehelixDefault traces: +-1 LSB everywhere
    #   Viridis Magma Inferno Plasma    the tables ARE matplotlib's (d3 and colorous port the same 256 colours: the first five and the
    #                                   last Viridis entries d3 publishes are checked by tools/gen_gradients.py)
    #   13 ColorBrewer splines          only the two END colours (a reflected B-spline passes exactly through its end anchors): equal
    #                                   to ColorBrewer's, read through matplotlib's colormap of the same scheme; the interior is d3's
    #                                   spline against matplotlib's piecewise-linear ramp over the same anchors -- close, not a pin
    #   Turbo, Cividis                  d3 publishes POLYNOMIAL FITS of these maps; matplotlib ships the maps themselves (256 entries):
    #                                   the fit is within 26 / 19 LSB of the map -- the right shape, not a pin
    #   Cool                            the reference's own screenshot (test_reference_screenshots_pin_...): curve and byte rule
    # => UNPINNED against anything but this build's reading of the d3 formulas: the 13 splines (interior), Turbo, Cividis.
    import matplotlib
    import numpy as np
    from spectrogram_rs_amd.engine import builtin_gradient, builtin_gradient_eval
    ts = (np.arange(256) + 0.5) / 256.0
    mpl = lambda name: np.rint(np.asarray(matplotlib.colormaps[name](ts))[:, :3] * 255.0).astype(int)
    lib = lambda name: np.array([builtin_gradient_eval(name, float(t)) for t in ts], int)
    assert np.abs(lib("cubehelix") - mpl("cubehelix")).max() <= 2      # (truncated bytes against matplotlib's rounded 256-entry samples; the curve itself: next test)
    for name in ("viridis", "magma", "inferno", "plasma"):
        assert np.array_equal(builtin_gradient(name), np.rint(np.asarray(matplotlib.colormaps[name].colors) * 255.0).astype(np.uint8))
    brewer = {"red_yellow_blue": "RdYlBu", "red_blue": "RdBu", "spectral": "Spectral", "red_yellow_green": "RdYlGn", "pink_green": "PiYG",
              "purple_orange": "PuOr", "reds": "Reds", "blues": "Blues", "greens": "Greens", "greys": "Greys", "oranges": "Oranges"}
    for name, m in brewer.items():
        cm = matplotlib.colormaps[m]
        for t in (0.0, 1.0):
            assert builtin_gradient_eval(name, t) == tuple(int(v) for v in np.rint(np.asarray(cm(t))[:3] * 255.0)), (name, t)
        assert np.abs(lib(name) - mpl(m)).max() <= 16, name          # spline vs piecewise-linear over the same anchors (worst: 15, Spectral)
    assert np.abs(lib("turbo") - mpl("turbo")).max() <= 26 and np.abs(lib("cividis") - mpl("cividis")).max() <= 19


def test_cubehelix_default_is_the_curve_matplotlib_traces():
    # colorscheme.rs:141 CUBEHELIX = d3's interpolateCubehelixDefault: (h, s, l) from (300, 0.5, 0) to (-240, 0.5, 1),
    # hue linear without shortest-arc.  That is Green's (2011) helix with start 0.5, -1.5 rotations, hue 1, gamma 1 --
    # the function matplotlib's `cubehelix` colormap samples -- an independent source for the default parameters.
    import matplotlib._cm as _cm
    import numpy as np
    from spectrogram_rs_amd.engine import builtin_gradient_eval
    fns = _cm.cubehelix(gamma=1.0, s=0.5, r=-1.5, h=1.0)
    ts = np.linspace(0.0, 1.0, 2049)
    # bytes by truncation (Rust's `as u8`): the rule the reference's Cool screenshot shows for this family (test below)
    want = np.stack([np.clip(np.floor(255.0 * np.asarray(fns[k](ts), np.float64)), 0, 255) for k in ("red", "green", "blue")], 1)
    got = np.array([builtin_gradient_eval("cubehelix", float(t)) for t in ts], np.float64)
    d = np.abs(got - want)
    assert d.max() <= 1 and (d > 0).mean() < 2e-3, (d.max(), (d > 0).mean())   # same curve; a byte may sit on an integer
    assert builtin_gradient_eval("cubehelix", 0.0) == (0, 0, 0) and builtin_gradient_eval("cubehelix", 1.0) == (255, 255, 255)
    # Cool / Warm meet at t = 1 (d3's rainbow is warm(2t) then cool(2 - 2t)): (h, s, l) = (80, 1.5, 0.8)
    assert builtin_gradient_eval("cool", 1.0) == builtin_gradient_eval("warm", 1.0)
    assert builtin_gradient_eval("cool", 0.0) == (109, 63, 169) and builtin_gradient_eval("warm", 0.0) == (109, 63, 169)


def test_reference_screenshots_pin_the_table_values_and_cools_curve():
    # The ONLY outputs of the colour path that the reference itself holds: screenshots/colorscheme-{viridis,magma,plasma,cool}.png
    # (README.md).  tests/golden/screenshot_colours.npz (make_screenshot_colours.py, build container) keeps their data: the window's
    # fill = ColorScheme::background() = gradient.eval_continuous(0.0), the axes = foreground() = eval_continuous(1.0)
    # (colorscheme.rs:41-53), and the set of distinct opaque colours (pixels of color_for, :55-71, scaled by the toolkit: many blends,
    # many verbatim).  What this pins: the VALUES of the three 256-entry tables (entry 0, entry 255, and >= 150 further entries found
    # verbatim) and, for Cool -- a continuous curve with no table -- the curve AND the byte rule (truncation, not d3's rounding).
    # What it does not pin: the index rule t -> entry (no pixel's t is known), Inferno and the other 14 gradients.
    import os
    import numpy as np
    from oracle.gradients import CONTINUOUS
    from spectrogram_rs_amd.engine import builtin_gradient, builtin_gradient_eval
    shots = np.load(os.path.join(os.path.dirname(__file__), "golden", "screenshot_colours.npz"))
    for name in ("viridis", "magma", "plasma"):
        table = builtin_gradient(name)
        assert tuple(shots[name + "_background"]) == tuple(table[0]), name
        assert tuple(shots[name + "_axis"]) == tuple(table[255]), name
        seen = set(map(tuple, shots[name + "_colours"].tolist()))
        verbatim = sum(tuple(int(v) for v in e) in seen for e in table)
        assert verbatim >= 150, (name, verbatim)                      # measured: viridis 243, magma 171, plasma 236
    # Cool: background / axes are the curve's ends under truncation -- (109.70, 63.81, 169.91) and (175.23, 239.78, 90.54)
    assert tuple(shots["cool_background"]) == builtin_gradient_eval("cool", 0.0) == (109, 63, 169)
    assert tuple(shots["cool_axis"]) == builtin_gradient_eval("cool", 1.0) == (175, 239, 90)
    cols, cnt = shots["cool_colours"], shots["cool_counts"]
    seen = set(map(tuple, cols.tolist()))
    curve = {builtin_gradient_eval("cool", float(t)) for t in np.linspace(0.0, 1.0, 20001)}
    assert curve == {CONTINUOUS["cool"](float(t)) for t in np.linspace(0.0, 1.0, 20001)}
    hit = len(curve & seen)
    assert len(curve) > 600 and hit >= 0.95 * len(curve), (len(curve), hit)          # measured: 599 of 623 (d3's rounding: 364 of 619)
    # pixel-weighted: more than half of ALL opaque pixels carry a colour of the curve verbatim, three quarters are within 1 LSB of it
    # (the rest: the toolkit's scaling blends, the border, the axis labels' anti-aliasing)
    arr = np.array(sorted(curve), np.int32)
    dist = np.array([int(np.abs(arr - c).max(axis=1).min()) for c in cols.astype(np.int32)])
    exact, near = cnt[dist == 0].sum() / cnt.sum(), cnt[dist <= 1].sum() / cnt.sum()
    assert exact >= 0.5 and near >= 0.75, (exact, near)                                # measured: 0.540, 0.768


def test_reference_screenshots_pin_the_log_frequency_axis():
    # The same four screenshots show the frequency axis of simple_spectrogram.rs:107 -- (32.0..22030.0).reversible_log_scale() -- as
    # drawn: a vertical axis line whose top end is the range's upper end, tick marks at 32 * 2^k Hz (labels 32.0 ... 16384.0).
    # tests/golden/screenshot_colours.npz keeps the geometry (top / bottom row of the line, the ticks' centre rows).  What this pins,
    # independently of this build: rows a18 / a19 -- the axis is linear in log f and runs from 32 Hz to 22030 Hz (LogCoordf64::map,
    # the inverse of the `unmap` the pixel path calls, log_scaling.rs:47-51,114-119).  Resolution: 1.5 screenshot pixels of 501 = 2 % in
    # frequency: a range end of 20 kHz or 24 kHz would be 7 pixels off.
    import os
    import numpy as np
    import oracle
    shots = np.load(os.path.join(os.path.dirname(__file__), "golden", "screenshot_colours.npz"))
    for name in ("viridis", "magma", "plasma", "cool"):
        g = shots[name + "_axis_geometry"]
        y_top, ticks = g[0], g[2:]
        assert len(ticks) == 10                                    # 32, 64, ..., 16384 Hz, top to bottom: 16384 first
        k = np.arange(9, -1, -1, dtype=np.float64)                 # octaves above 32 Hz
        b, c = np.polyfit(k, ticks, 1)                             # y = c + b k: b < 0 pixels per octave, c = the row of 32 Hz
        assert np.abs(ticks - (c + b * k)).max() <= 1.5            # linear in log f (the ticks are snapped to device pixels)
        octaves = np.log2(22030.0 / 32.0)
        assert abs((c + b * octaves) - y_top) <= 1.5               # ... and the line ends where 22030 Hz falls
        assert abs((c + b * np.log2(20000.0 / 32.0)) - y_top) > 5 and abs((c + b * np.log2(24000.0 / 32.0)) - y_top) > 5
        # the oracle's (and through test_gpu_parity the engine's) row edges are that axis: unmap at a tick's height returns its frequency
        N = 1_000_000
        for kk, y in zip(k, ticks):
            pfrac = (c - y) / (c - (c + b * octaves))
            f = oracle.log_unmap(32.0, 22030.0, int(round(pfrac * N)), 0, N)
            assert abs(f / (32.0 * 2.0 ** kk) - 1.0) <= 0.02, (name, kk, f)


def test_integration_md_shows_the_binding_files_verbatim():
    # INTEGRATION.md's Rust blocks are generated from bindings/rust/*.rs (tools/sync_integration.py): no drift, no stubs
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, os.path.join(root, "tools", "sync_integration.py"), "--check"]).returncode == 0
    for name in os.listdir(os.path.join(root, "bindings", "rust")):
        text = open(os.path.join(root, "bindings", "rust", name)).read()
        assert "todo!" not in text and "unimplemented!" not in text, name


def test_plane_transposes_read_conflict_free_at_the_strides_the_kernels_use():
    # the 16-byte read sides of the LDS plane transposes (stft4096_wg.hip, stft4096_real.hip: `TR`) against the bank model of
    # tools/lds_b128_conflicts.py: one lane per bank in every lane group -- and the strides the model assumes are the kernels'
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lds_b128_conflicts", os.path.join(root, "tools", "lds_b128_conflicts.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name, w in mod.shipped():
        assert w == 1, name
    real = open(os.path.join(root, "spectrogram_rs_amd", "csrc", "stft4096_real.hip")).read()
    wg = open(os.path.join(root, "spectrogram_rs_amd", "csrc", "stft4096_wg.hip")).read()
    assert "addtid_rows<272>" in real and "addtid_rows<280>" in real
    assert "plane + 272 * (tid >> 4) + 68 * ((tid & 15) >> 2) + 16 * (tid & 3)" in real and "plane + 272 * (tid >> 4) + 68 * ((tid & 15) >> 2) + 16 * (tid & 3)" in wg
    assert "plane + 280 * ((tid & 127) >> 3) + 68 * ((8 * (tid >> 7) + (tid & 7)) >> 2) + 16 * (tid & 3)" in real
