"""BASELINE configs[4] on hardware, inside the suite the driver runs: bench.py's config-5 leg -- PCM generated on the device chunk by
chunk, the fused PCM -> RGBA kernel, the two-slot ring, the gather of pixel columns to rank 0, sgx_checksum_add over every piece --
run as TWO FRESH rank processes on the one MI355X of the box (BENCH_BACKEND=gloo BENCH_SINGLE_DEVICE=1: every rank on cuda:0, pieces
staged through the host; RCCL refuses two ranks on one device).  The ranks are ordinary children of a bare `python bench.py --gpus 2`
(subprocess; this pytest process has touched the GPU and is never replaced).  What the first 8-GPU run then adds is RCCL alone.

Frame t depends only on samples [tH, tH + W) (reference: src/fourier/audio_transform.rs:34-42), so the sharded bytes must be the
single process's bytes: the root's checksum over all gathered columns equals the one computed here in-process."""
import importlib.util
import json
import os
import socket
import subprocess
import sys

import numpy as np

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

TOTAL, CHUNK = 300_003, 32_768          # rank 0: 150 002 frames, rank 1: 150 001 -> 5 rounds each, uneven tails (18 930 / 18 929)


def run_bench(extra_env, timeout=900):
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_SINGLE_DEVICE="1", **extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "BENCH_LAUNCH_ONLY"):
        env.pop(k, None)
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "65536", "--placements", "1", "--sustain-s", "0",
            "--config5-frames", str(TOTAL), "--config5-chunk", str(CHUNK), "--leg-timeout", "120"]
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def single_process_checksum(total, chunk):
    spec = importlib.util.spec_from_file_location("config5_single", os.path.join(ROOT, "tools", "config5_single.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.single_process_checksum(total, chunk)


def test_config5_leg_two_ranks_on_one_gpu_equals_the_single_process():
    p = run_bench({})
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert "error" not in line, line.get("error")
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    c5 = line["config5"]
    assert c5["backend"] == "gloo" and c5["ranks_seen"] == [0, 1]
    assert c5["frames_total"] == TOTAL and c5["frames_per_gpu"] == [150_002, 150_001]
    assert c5["chunk_columns"] == CHUNK and c5["rounds"] == 5
    assert c5["gathered_bytes"] == 150_001 * 4096                      # rank 1's columns, every one, once
    assert c5["sharded_equals_single_gpu_on_first_chunk_of_every_rank"] is True
    # another chunk size on purpose: the checksum carries global word indices, so it cannot depend on how the stream was cut
    assert c5["checksum_all_columns"] == single_process_checksum(TOTAL, 50_000)
    assert c5["frames_per_s"] > 0 and c5["overlapped_s"] > 0


def test_config5_leg_a_rank_dying_inside_the_leg_fails_the_run():
    p = run_bench({"BENCH_FAIL_RANK": "1", "BENCH_TEST_HOOKS": "1"})
    assert p.returncode != 0, p.stdout[-2000:]
    for ln in p.stdout.splitlines():
        if ln.startswith("{"):                 # rank 0 may still report: then the line says what happened
            assert "error" in json.loads(ln)


def one_rank_over_rccl(total, extra_argv=(), timeout=900):
    # RCCL itself on the box's one MI355X: a single launched rank (RANK / WORLD_SIZE as torch.distributed.run sets them) takes the
    # multi-rank path of bench.py with the default backend -- process group init with device_id, barriers, the MAX all-reduce of the
    # timings, the all-gather of the ranks -- and the config-5 leg with itself as the root.  (Two ranks on one device: RCCL answers
    # "invalid usage", tools/rccl_probe.py.)  What an 8-GPU node adds is the point-to-point gather between devices.
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               BENCH_GROUP_OF_ONE="1", BENCH_TEST_HOOKS="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("BENCH_BACKEND", "BENCH_SINGLE_DEVICE", "BENCH_LAUNCH_ONLY"):
        env.pop(k, None)
    argv = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--frames", "65536", "--placements", "1", "--sustain-s", "0",
            "--config5-frames", str(total), "--leg-timeout", "600", *extra_argv]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    assert len(lines[0]) <= 4096                 # the stdout line's budget (bench.compact_line)
    line = json.loads(lines[0])
    assert "error" not in line, line.get("error")
    return line


def test_config5_leg_one_rank_over_rccl():
    total = 100_001
    c5 = one_rank_over_rccl(total, ["--config5-chunk", str(CHUNK)])["config5"]
    assert c5["backend"] == "nccl" and c5["ranks_seen"] == [0]
    assert c5["frames_total"] == total and c5["frames_per_gpu"] == [total] and c5["gathered_bytes"] == 0
    assert c5["checksum_all_columns"] == single_process_checksum(total, 50_000)


def test_config5_at_its_own_1e8_frames_and_its_pixels_at_far_offsets(tmp_path):
    """BASELINE configs[4] at its own length on the one GPU: 1e8 frames = 1 526 rounds of 65 536 columns through the leg's ring (one rank,
    RCCL group of one), sample offsets up to 99 999 999 * 256 + 2048 = 2.56e10 (> 2^34).  Eight pieces spread from the first frame to
    the last leave 8 columns each behind; every one of those 64 columns is checked here against the oracle, which regenerates the
    frame's samples on the host from the ABSOLUTE sample index (reference hop loop: src/fourier/audio_transform.rs:34-42, frame t =
    samples [tH, tH + W)):  stage-wise bit-exact (the engine's own magnitudes of that frame -> the oracle's pixel stage == the bytes
    the leg produced) and end to end at most one LUT step on < 2e-3 of the pixels."""
    import torch

    import oracle
    from spectrogram_rs_amd import SpectrogramEngine, builtin_gradient

    total, W, H, R = 100_000_000, 2048, 256, 1024
    probes = str(tmp_path / "probes.npz")
    line = one_rank_over_rccl(total, ["--config5-probe", "8", "--config5-probe-file", probes], timeout=1500)
    c5 = line["config5"]
    assert c5["frames_total"] == total and c5["frames_per_gpu"] == [total] and c5["rounds"] == 1526 and c5["chunk_columns"] == 65_536
    assert c5["probes"] == 64
    d = np.load(probes)
    frames, rgba = d["frames"], d["rgba"]
    assert frames.shape == (64,) and rgba.shape == (64, R, 4)
    assert frames[0] == 0 and frames[-1] == total - 1 and (np.diff(frames) > 0).all()
    assert int(frames[-1]) * H + W > 2**34                  # the far end of the stream really is beyond 2^34 samples

    eng = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=1, interp=1, gradient="viridis")
    lut = builtin_gradient("viridis")
    level = {tuple(int(x) for x in c): i for i, c in enumerate(lut)}
    bad = 0
    for t, col in zip(frames, rgba):
        t = int(t)
        host = oracle.white_noise(W, first=t * H)
        dev = eng.white_noise(W, first=t * H)
        assert np.array_equal(dev.cpu().numpy(), host), t                                   # the generator at the far offset
        own = eng.stft_batch(dev).cpu().numpy()[:, 0]                                         # [1][M][2]
        assert np.array_equal(oracle.render_columns(own, 48000, lut, interp=1)[0], col), t   # stage-wise: bit for bit
        ref = oracle.render_columns(oracle.fft_process(np.stack([host, host], 1), W)[None], 48000, lut, interp=1)[0]
        diff = np.argwhere((ref != col).any(axis=1))[:, 0]
        bad += len(diff)
        for r in diff:
            assert abs(level[tuple(int(x) for x in col[r, :3])] - level[tuple(int(x) for x in ref[r, :3])]) <= 1, (t, r)
    assert bad <= 2e-3 * rgba.shape[0] * R, bad
    eng.close()
    torch.cuda.synchronize()
    # and the whole run's checksum == ONE process rendering the 1e8 frames with another chunk size
    assert c5["checksum_all_columns"] == single_process_checksum(total, 50_000)
