"""bench.py's N = 1 run on the GPU, at reduced sizes, inside the suite the driver runs: ONE stdout line that parses, is at most 4 096 bytes
(the 21 KB line of that round was not parsed) and carries the contract keys with `roofline` and `cpu_baseline`; the full
record on stderr and in bench_legs.json.  A fresh child process (this pytest process has touched the GPU and is never replaced)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_the_default_run_prints_one_parsable_line_within_its_budget(tmp_path):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "BENCH_LAUNCH_ONLY", "BENCH_BACKEND", "BENCH_SINGLE_DEVICE", "BENCH_TEST_HOOKS"):
        env.pop(k, None)
    argv = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--frames", "65536", "--placements", "2", "--sustain-s", "0.2", "--leg-sustain-s", "0.1",
            "--pixel-frames", "65536", "--config4-hops", "2048", "--stereo-frames", "65536", "--paired-frames", "65536", "--complex-frames", "65536",
            "--app-frames", "16384", "--cpu-frames", "8192"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-3000:]                       # nothing else on stdout
    assert len(lines[0].encode()) <= 4096
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["metric"].startswith("STFT frames/sec") and line["unit"] == "frames/s" and line["n_gpus"] == 1 and line["steps"] == 3
    assert line["dtype"] == "f32" and line["vs_baseline"] is None and line["scaling"] == "weak" and line["higher_is_better"] is True
    assert line["config"]["workload"].startswith("configs[1]") and "model" not in line["config"]
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5
    assert "traffic" in rf and rf["kernel"].startswith("sgx::")
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("port", "reference") and cb["unit"] == "frames/s" and cb["sample"] and cb["parity_on_sample"] is True
    for k in ("config3_frac", "config4_frac", "stereo_frac"):
        assert 0 < line[k] < 1, k
    assert line["config3_rgba"]["stage_wise_bit_exact_on_128_frames"] is True and line["config3_rgba"]["max_lut_step"] <= 1
    # the full record: on stderr, and beside bench.py
    legs = [ln for ln in p.stderr.splitlines() if ln.startswith("bench_legs ")]
    assert len(legs) == 1
    full = json.loads(legs[0][len("bench_legs "):])
    assert full["value"] == line["value"] and "config4" in full and "placement" in full["roofline"]
    assert json.load(open(os.path.join(ROOT, "bench_legs.json")))["value"] == line["value"]
