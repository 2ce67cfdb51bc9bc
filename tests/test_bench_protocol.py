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


# ---- the ONE stdout line (the 21 KB line of that round was not parsed by the driver) --------------------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def full_record_of_round_5():
    import json
    with open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")) as f:      # the real 21 041-byte line of round 5
        return json.load(f)


def test_the_stdout_line_fits_its_budget_and_keeps_every_contract_key():
    import json
    full = full_record_of_round_5()
    assert len(json.dumps(full)) > 20_000
    line = bench.compact_line(full)
    txt = json.dumps(line)
    assert len(txt) <= bench.LINE_BUDGET == 4096
    for k in bench.CONTRACT_KEYS:
        assert k in line, k
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[k] == full[k]                                              # the contract's scalars: verbatim, never rounded
    assert line["config"]["workload"].startswith("configs[1]")
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6
    assert "traffic" in rf and abs(rf["frac"] - full["roofline"]["frac"]) < 1e-6
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] == 16 and cb["unit"] == "frames/s" and "sample" in cb and cb["kind"] in ("port", "library")
    assert abs(line["config3_frac"] - full["config3"]["roofline"]["frac"]) < 1e-6
    assert abs(line["config4_frac"] - full["config4"]["roofline"]["frac"]) < 1e-6
    assert abs(line["stereo_frac"] - full["stereo4096"]["roofline"]["frac"]) < 1e-6
    assert line["legs_file"] == bench.LEGS_FILE


def test_an_oversized_record_loses_extras_never_contract_keys():
    import json
    full = full_record_of_round_5()
    full["error"] = "RuntimeError: " + "x" * 5000                             # a long error text
    full["config"]["workload"] += " " + "y" * 3000
    full["config5"] = {"backend": "nccl", "ranks_seen": list(range(8)), "frames_total": 10**8, "frames_per_gpu": [12_500_000] * 8,
                       "chunk_columns": 65536, "rounds": 191, "frames_per_s": 1.9e9, "gathered_bytes": 358_400_000_000,
                       "checksum_all_columns": 2**63 + 12345, "sharded_equals_single_gpu_on_first_chunk_of_every_rank": True,
                       "GBps_per_source_alone": [{"source": r, "bytes": 1, "GBps": 47.123456789} for r in range(1, 8)],
                       "workload": "z" * 500}
    line = bench.compact_line(full)
    assert len(json.dumps(line)) <= 4096
    for k in bench.CONTRACT_KEYS:
        assert k in line, k
    assert line["error"].startswith("RuntimeError") and line["config5"]["ranks_seen"] == list(range(8))
    assert line["config5"]["checksum_all_columns"] == 2**63 + 12345           # integers are never rounded


def test_emit_line_puts_the_full_record_on_stderr_and_in_the_legs_file(tmp_path, monkeypatch, capsys):
    import json
    full = full_record_of_round_5()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit_line(full)
    out, err = capsys.readouterr()
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) <= 4096 and json.loads(lines[0])["metric"] == full["metric"]
    legs = [ln for ln in err.splitlines() if ln.startswith("bench_legs ")]
    assert len(legs) == 1 and json.loads(legs[0][len("bench_legs "):]) == full
    assert json.load(open(tmp_path / bench.LEGS_FILE)) == full


def test_fp32_fraction_is_counted_flops_times_rate_over_the_vector_peak(monkeypatch):
    monkeypatch.setattr(bench, "load_profile_json", lambda name: {"flop_per_unit": {"stereo4096": 300_000.0}} if name == "fp32_flops" else None)
    f = bench.fp32_fracs("stereo4096", bench.NOMINAL_FLOP_4096, 200e6)
    assert abs(f["fp32_frac"] - 300_000.0 * 200e6 / 157.3e12) < 1e-12
    assert abs(f["fp32_frac_nominal"] - 245_760 * 200e6 / 157.3e12) < 1e-12
    assert bench.fp32_fracs("config4", 4 * bench.NOMINAL_FLOP_16384, 8.4e6)["fp32_frac"] is None      # no counter pass for that leg: said, not guessed
