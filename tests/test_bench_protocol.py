"""bench.py's measurement protocol on a simulated device (no GPU): place_output keeps the first allocation unless another
candidate is faster in BOTH interleaved passes, reports both passes, raises a clear error when nothing can be allocated;
measure_leg's figure is the steady part of the sustained window.  """
import sys
import os
import types

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


class FakeDevice:
    """a virtual clock: launch(buf) advances it by the buffer's cost times a clock-ramp factor that decays with work done"""

    def __init__(self, ramp=0.0):
        self.now_ms, self.ramp, self.launches = 0.0, ramp, 0

    def launch(self, buf):
        cold = 1.0 + self.ramp * max(0.0, 1.0 - self.launches / 50.0)     # the first 50 launches run on a ramping clock
        self.now_ms += buf.cost * cold
        self.launches += 1


class FakeBuf:
    def __init__(self, cost):
        self.cost = cost

    def zero_(self):
        return self


def fake_torch(dev):
    class Event:
        def __init__(self, enable_timing=True):
            self.t = None

        def record(self):
            self.t = dev.now_ms

        def elapsed_time(self, other):
            return other.t - self.t
    cuda = types.SimpleNamespace(Event=Event, synchronize=lambda: None, empty_cache=lambda: None)
    return types.SimpleNamespace(cuda=cuda)


def run_place(costs, ramp=0.0, n=None, monkeypatch=None):
    dev = FakeDevice(ramp)
    it = iter(costs)

    def alloc():
        try:
            return FakeBuf(next(it))
        except StopIteration:
            raise RuntimeError("out of memory")
    # the heat phase runs on wall time: make it a fixed number of rounds instead
    ticks = iter(range(10**6))
    monkeypatch.setattr(bench.time, "perf_counter", lambda: next(ticks) * 0.05)
    return bench.place_output(fake_torch(dev), n or len(costs), alloc, dev.launch, 17.4e9), dev


def test_equal_candidates_keep_the_first_allocation_even_on_a_ramping_clock(monkeypatch):
    # round 3's protocol (each candidate cold, right after its allocation) would have ranked these by clock ramp
    (buf, first, rep), _ = run_place([3.30, 3.30, 3.30, 3.30], ramp=0.15, monkeypatch=monkeypatch)
    assert rep["chosen"] == 0 and first is None and not rep["kept_selection"]
    assert rep["spread"] < 0.01 and all(rep["stable_within_2pct"])
    assert len(rep["pass1_ms_per_launch"]) == len(rep["pass2_ms_per_launch"]) == 4


def test_a_stable_class_difference_is_selected_and_the_first_buffer_survives(monkeypatch):
    (buf, first, rep), _ = run_place([3.72, 3.53, 3.30, 3.53], monkeypatch=monkeypatch)
    assert rep["chosen"] == 2 and rep["kept_selection"] and buf.cost == 3.30
    assert first is not None and first.cost == 3.72          # kept alive for roofline.first_allocation
    assert rep["frac_of_candidate_0"] < rep["frac_of_fastest"]


def test_a_difference_inside_the_margin_is_not_a_selection(monkeypatch):
    (buf, first, rep), _ = run_place([3.30, 3.26, 3.29], monkeypatch=monkeypatch)
    assert rep["chosen"] == 0 and rep["fastest"] == 1 and not rep["kept_selection"] and first is None


def test_one_placement_is_the_first_allocation_and_no_allocation_is_a_clear_error(monkeypatch):
    (buf, first, rep), dev = run_place([3.5, 3.3], n=1, monkeypatch=monkeypatch)
    assert buf.cost == 3.5 and first is None and rep["chosen"] == 0 and dev.launches == 1
    with pytest.raises(RuntimeError, match="could be allocated"):
        run_place([], n=4, monkeypatch=monkeypatch)
    # out of memory after two candidates: the study runs on what there is
    (buf, first, rep), _ = run_place([3.7, 3.3], n=4, monkeypatch=monkeypatch)
    assert rep["candidates"] == 2 and rep["chosen"] == 1


def test_measure_leg_reports_burst_and_the_steady_part_of_the_sustained_window(monkeypatch):
    dev = FakeDevice(ramp=0.2)
    buf = FakeBuf(2.0)
    ticks = iter(range(10**6))
    monkeypatch.setattr(bench.time, "perf_counter", lambda: next(ticks) * 0.1)
    m = bench.measure_leg(fake_torch(dev), lambda: dev.launch(buf), sustain_s=1.0)
    assert m["launch_ms_burst"]["n"] == 5 and m["launch_ms_burst"]["mean"] > 2.3      # still ramping
    assert m["launch_ms_sustained"]["n"] >= 32 and abs(m["mean_ms"] - 2.0) < 0.02     # steady
    t = bench.leg_times(m)
    assert t["launch_ms"] == t["launch_ms_sustained"] and t["sustain_s"] > 0
    m0 = bench.measure_leg(fake_torch(FakeDevice()), lambda: None, sustain_s=0.0)
    assert m0["launch_ms_sustained"]["n"] == 5                                           # no sustain window: the burst is the figure
