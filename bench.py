#!/usr/bin/env python3
"""bench.py -- STFT frames/s (4096-point, hop 256) on N MI355X, with roofline and CPU baseline.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
  N > 1: either launched by the driver as
      python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
  or BARE (`python bench.py --gpus N`): this process then touches no GPU at all, starts exactly that command as a
  child (one fresh rank process per GPU), relays rank 0's JSON line and exits with the child's status.

A "step" is one pass of the hot path over one batch of synthetic PCM that is already resident in HBM: BASELINE
config 2 -- 1e6 frames of mono white noise (256 001 792 samples), W 2048 / P 4096 / H 256, output [1e6][2047][2]
float32 magnitudes.  With N GPUs every rank transforms its own 1e6-frame shard of one long stream (contiguous
frame ranges, sample offset rank * 1e6 * H; no data-path collective: the path shards by frame) => weak scaling;
value = N * 1e6 * K / t.

Order of a run (every rank): the output buffer is placed (see place_output: candidates are timed HOT and INTERLEAVED, two
passes in opposite order; a candidate other than the first allocation is taken only where both passes agree) -> W warm-up
steps -> K "burst" steps (timed per launch, NOT the headline) -> back-to-back steps for --sustain-s seconds (per-launch
times kept) -> barrier, EXACTLY K timed steps, barrier: `value`, `ms_per_step` and `roofline.frac` come from these last K
steps, i.e. from a device that has been under load for seconds.  `roofline.first_allocation` is the first allocation
under the same protocol (K timed steps on the same hot device) -- what a caller who allocates once gets; it IS `value`
whenever the first allocation was kept, and always with --placements 1.  Every side leg (config 3 / 4, stereo,
independent mono frames, the application's operating point) is timed the same way: 5 burst launches, then back-to-back
launches for --leg-sustain-s seconds; its figure is the mean of the last two thirds of that window.

Extra objects on the same line:
  roofline     -- dominant kernel (the STFT kernel): algorithmic bytes per launch / launch duration (HIP events on the
                  launch stream) against the 8 TB/s HBM peak; frac (timed region), frac_burst, frac_sustained
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference's algorithm; the reference itself is Rust + FFTW and
                  cannot be built in this image) timed on this host's cores, bounded sample (rank 0, N = 1 only)
  config3      -- (N = 1) BASELINE config 3: the same stream -> RGBA pixel columns, with its own roofline
  config4      -- (N = 1) BASELINE config 4 at its own size: 16384-point, hop 512, 8 interleaved channels, 1e5 hop positions
                  (26 GB of output), with its own roofline
  mono_paired_frames -- (N = 1) the headline stream with two frames per transform (SGX_FLAG_PAIRED_FRAMES, opt-in), the headline mode
                  of rounds 1-3.  The headline itself runs the library's default, the reference's dataflow -- every mono frame its own transform
                  (audio_input_list_model.rs:67-69), as a real-input 2048-point transform (csrc/stft4096_real.hip) -- so that
                  north_star's tolerance holds against every frame's OWN peak on any input; this leg says what that costs
  mono_complex_frames -- (N = 1) the same stream with every frame as the literal (s, s) 4096-point transform (SGX_FLAG_COMPLEX_MONO)
  config5      -- (N > 1) BASELINE config 5: 1e8 frames frame-sharded over the ranks, PCM generated on device chunk by
                  chunk, pixel columns gathered to rank 0 over RCCL / xGMI, every piece consumed (checksummed) by the root
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
W, H, M, R = 2048, 256, 2047, 1024
ALGO_BYTES_STFT = H * 1 * 4 + M * 2 * 4  # 17 400 B / frame: each input sample once, each output byte once
ALGO_BYTES_PIXEL = H * 1 * 4 + R * 4     # 5 120 B / frame
W4, H4, C4 = 8192, 512, 8
ALGO_BYTES_CFG4 = H4 * C4 * 4 + (C4 // 2) * (W4 - 1) * 8  # 278 496 B / hop position
PROFILE_ROUND = "r06"
W_APP, H_APP = 2400, 93     # the application's own operating point: 48 kHz x 0.05 s (gpu_spectrogram.rs:323), hop (2/1024) s (simple_spectrogram.rs:102)
ALGO_BYTES_STEREO = H * 2 * 4 + M * 2 * 4                     # 18 424 B / frame: an (l, r) stream, what the reference feeds
ALGO_BYTES_APP = H_APP * 2 * 4 + (W_APP - 1) * 2 * 4          # 19 936 B / frame
ALGO_BYTES_APP_PIXEL = H_APP * 2 * 4 + R * 4                  # 4 840 B / frame
FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 vector rate (spec)
NOMINAL_FLOP_4096 = 5 * 4096 * 12      # 245 760: the 5 P log2 P convention of SURVEY 8(d), one complex 4096-point transform
NOMINAL_FLOP_16384 = 5 * 16384 * 14    # 1 146 880
LINE_BUDGET = 4096           # bytes of the ONE stdout line (round 5's 21 KB line was not parsed by the driver)
LEGS_FILE = "bench_legs.json"
KERNEL_NAMES = {
    0: ("generic power-of-two (workgroup per frame, LDS radix-4)", "sgx::stft_generic_kernel"),
    2: ("stft4096 workgroup-per-transform (256 threads x 16 points, radix-16 x3, mono frame pairs), scalar codelets",
        "sgx::wg::stft4096_wg_kernel<true, 0, false, 0>"),
    "real": ("stft4096 real-input: every mono frame its own transform, a 2048-point complex transform of the real frame + one butterfly per bin "
             "(256 threads x 2 frames x 8 points, radix 8 x 16 x 16, sliding half-row window)", "sgx::wgr::stft4096_real_kernel<0, 0, true>"),
    6: ("mixed radix at the window's own length (compile-time plan)", "sgx::mix::stft_mixed_fixed_kernel"),
    10: ("stft16384 as 32 x 32 x 16 in one 512-thread workgroup (32 points per thread, two LDS exchanges, three barriers; hop 512: the window slides in registers)",
         "sgx::w16k::stft16384_w_kernel<false, true, true>"),   # (more than two channels: the pairs read where they lie; hop 512: runs of hop positions per workgroup)
    9: ("stft4800 workgroup-per-transform (320 threads, 16 x 20 x 15, resident twiddles, mono frame pairs)", "sgx::w48::stft4800_wg_kernel<0, false>"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=1_000_000, help="frames per GPU per step (config 2: 1e6)")
    ap.add_argument("--placements", type=int, default=6, help="candidate allocations of the output buffer, timed hot and interleaved; another than the first is kept "
                                                              "only if it is faster by > 2 %% in BOTH passes (1 = take the first, no study)")
    ap.add_argument("--leg-sustain-s", type=float, default=1.0, help="seconds of back-to-back launches behind the burst of every side leg")
    ap.add_argument("--paired-frames", type=int, default=1_000_000, help="N = 1: frames of the two-frames-per-transform leg (0 = skip)")
    ap.add_argument("--complex-frames", type=int, default=1_000_000, help="N = 1: frames of the (s, s)-transform-per-frame leg (0 = skip)")
    ap.add_argument("--sustain-s", type=float, default=3.0, help="seconds of back-to-back steps before the timed region (0 = skip)")
    ap.add_argument("--pixel-frames", type=int, default=1_000_000, help="N = 1: frames of the config-3 leg (0 = skip)")
    ap.add_argument("--config4-hops", type=int, default=100_000, help="N = 1: hop positions of the config-4 leg (BASELINE: 1e5 = 26 GB of output; 0 = skip)")
    ap.add_argument("--stereo-frames", type=int, default=1_000_000, help="N = 1: frames of the stereo 4096-point leg (0 = skip)")
    ap.add_argument("--app-frames", type=int, default=262_144, help="N = 1: frames of the leg at the application's operating point, W 2400 / hop 93 stereo (0 = skip)")
    ap.add_argument("--config5-frames", type=int, default=100_000_000, help="N > 1: total frames of the config-5 leg (0 = skip)")
    ap.add_argument("--config5-chunk", type=int, default=65_536, help="columns per rank per gather round")
    ap.add_argument("--config5-probe", type=int, default=0, help="config-5 leg: pieces (spread from the first to the last frame) of which 8 columns each are "
                                                                 "kept and written to --config5-probe-file, for a host-side check of far stream offsets (0 = none)")
    ap.add_argument("--config5-probe-file", default=None, help="where the probed columns go (.npz: frames [n] int64, rgba [n][R][4] uint8); default gpurun_out/ or the repo root")
    ap.add_argument("--leg-timeout", type=float, default=600.0, help="seconds after which a stalled collective leg is given up (exit 3)")
    ap.add_argument("--cpu-frames", type=int, default=262_144, help="frames of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--generic", action="store_true", help="force the generic power-of-two kernel")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------
# bare `python bench.py --gpus N`: no GPU call in this process; one fresh child per rank
# ---------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_children(args, argv):
    """Start `python -m torch.distributed.run --nproc-per-node N bench.py <argv>` and relay its output.  Nothing in
    this process has initialised a GPU (no torch import, no HIP call): the ranks are ordinary child processes."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["BENCH_LAUNCHED_BY_PARENT"] = "1"
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out          # rank 0's JSON line: printed last, alone
        else:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("bench.py: the ranks exited 0 without printing a JSON line", file=sys.stderr)
        rc = 4
    return rc


def test_hook(name):
    """BENCH_FAIL_RANK (a rank that dies: fault injection) and BENCH_GROUP_OF_ONE (one launched rank takes the multi-rank path) are
    test hooks: they read as unset unless BENCH_TEST_HOOKS=1 is set beside them, so that a stray variable cannot alter a real run."""
    return os.environ.get(name) if os.environ.get("BENCH_TEST_HOOKS") == "1" else None


def host_info():
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    quota = None   # a cgroup CPU quota (cpu.max "<quota> <period>" / cfs_quota_us): CPUs' worth of time this container may use
    for path, v2 in (("/sys/fs/cgroup/cpu.max", True), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", False)):
        try:
            with open(path) as f:
                txt = f.read().split()
            if v2 and txt and txt[0] != "max":
                quota = float(txt[0]) / float(txt[1])
            elif not v2 and txt and int(txt[0]) > 0:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    quota = int(txt[0]) / float(f.read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    return {"nproc": os.cpu_count(), "affinity": aff, "cgroup_cpu_quota": quota, "cpu_model": model}


def csrc_sha16():
    """sha256 over the kernel sources (csrc/*.hip, *.hpp, *.inc, *.cpp, sorted by name): what a committed counter summary is stamped
    with (tools/pmc_round.py) -- counters of an older kernel are not quoted for a newer one"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "spectrogram_rs_amd", "csrc", "*.*"))):
        if path.endswith((".hip", ".hpp", ".inc", ".cpp")) and not os.path.basename(path).startswith("ab_"):   # (ab_*: patched copies of A/B tools)
            with open(path, "rb") as f:
                h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def load_profile_json(name):
    """A committed summary of rocprofv3 --pmc passes (profiles/<round>_<name>.json: this round's, else the newest earlier round's), or
    None -- also None (with the reason on stderr) when every summary was taken from other kernel sources than the ones in the tree."""
    sha = csrc_sha16()
    n = int(PROFILE_ROUND[1:])
    for rnd in (f"r{k:02d}" for k in range(n, 0, -1)):
        path = os.path.join(ROOT, "profiles", f"{rnd}_{name}.json")
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:  # noqa: BLE001 -- absent or unreadable: try the round before
            continue
        if d.get("csrc_sha16") != sha:
            print(f"bench.py: {os.path.relpath(path, ROOT)} was measured on other kernel sources "
                  f"({d.get('csrc_sha16')} != {sha}): not quoted", file=sys.stderr)
            return None
        d["source"] = os.path.relpath(path, ROOT)
        return d
    return None


def r6(x):
    """a float at 6 significant digits (the stdout line is budgeted; the full figures are in bench_legs.json)"""
    return float(f"{x:.6g}") if isinstance(x, float) else x


def pick(d, *keys):
    return {k: r6(d[k]) for k in keys if isinstance(d, dict) and d.get(k) is not None}


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")


def compact_line(full, budget=LINE_BUDGET):
    """The ONE stdout line: the contract keys, `config`, a `roofline` and a `cpu_baseline` cut down to their figures, and one-number
    summaries of the side legs -- never more than `budget` bytes (the 21 KB line of that round was not parsed).
    Everything else (per-launch statistics, placement passes, the prose) goes to stderr and to bench_legs.json (emit_line).
    Optional keys are dropped from the end of DROP_ORDER until the line fits; the contract keys are never dropped."""
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data") if k in full}
    cfg = full.get("config") or {}
    line["config"] = pick(cfg, "workload", "window", "fft_length", "hop", "channels", "frames_per_gpu", "kernel", "sharding")
    roof = full.get("roofline") or {}
    rl = pick(roof, "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launch_ms", "bytes_per_frame", "frames_per_launch",
              "frac_burst", "frac_sustained", "fp32_frac", "fp32_frac_nominal", "valu_issue_frac", "traffic_source", "sclk_mhz_after_run")
    rl.setdefault("traffic", None)
    if roof.get("first_allocation"):
        rl["first_allocation"] = pick(roof["first_allocation"], "launch_ms", "frac", "is_value")
    if roof.get("placement"):
        rl["placement"] = pick(roof["placement"], "candidates", "chosen", "spread")
    if roof.get("measured_device"):
        rl["measured_device"] = pick(roof["measured_device"], "fill_GBps", "copy_GBps", "frac_of_fill", "frac_of_copy")
    line["roofline"] = rl
    cpu = full.get("cpu_baseline")
    if isinstance(cpu, dict):
        cb = pick(cpu, "value", "unit", "cores", "kind", "impl", "sample", "nproc", "cpu_model", "parity_on_sample")
        if isinstance(cpu.get("port"), dict):
            cb["port"] = pick(cpu["port"], "value", "impl")
        if isinstance(cpu.get("library"), dict):
            cb["library"] = pick(cpu["library"], "value", "impl", "matches_oracle")
        if isinstance(cpu.get("single_thread"), dict):
            cb["single_thread"] = pick(cpu["single_thread"], "value")
        cb["fftw"] = cpu["fftw"] if isinstance(cpu.get("fftw"), dict) and "as_written" in cpu["fftw"] else "absent"
        if isinstance(cb["fftw"], dict):
            cb["fftw"] = pick(cb["fftw"], "as_written", "hoisted")
        line["cpu_baseline"] = cb
    for k in ("error", "checksum_first_4096_frames", "achieved_GBps_algorithmic"):
        if k in full:
            line[k] = r6(full[k])
    c3 = full.get("config3")
    if isinstance(c3, dict):
        rf = c3.get("roofline") or {}
        line["config3_frac"] = r6(rf.get("frac"))
        line["config3_fp32_frac"] = r6(rf.get("fp32_frac"))
        line["config3_fp32_frac_nominal"] = r6(rf.get("fp32_frac_nominal"))
        line["config3_valu_issue_frac"] = r6(rf.get("valu_issue_frac"))
        line["config3_frames_per_s"] = r6(c3.get("frames_per_s"))
        if isinstance(c3.get("cubic"), dict) and "frames_per_s" in c3["cubic"]:
            line["config3_cubic_frames_per_s"] = r6(c3["cubic"]["frames_per_s"])
            line["config3_cubic_fp32_frac"] = r6(c3["cubic"].get("fp32_frac"))
        par = (c3.get("rgba_vs_oracle") or {}).get("cosine") or {}
        line["config3_rgba"] = pick(par, "mismatch_rate", "max_lut_step", "stage_wise_bit_exact_on_128_frames")
    c4 = full.get("config4")
    if isinstance(c4, dict):
        rf = c4.get("roofline") or {}
        line["config4_frac"] = r6(rf.get("frac"))
        line["config4_fp32_frac"] = r6(rf.get("fp32_frac"))
        line["config4_fp32_frac_nominal"] = r6(rf.get("fp32_frac_nominal"))
        line["config4_valu_issue_frac"] = r6(rf.get("valu_issue_frac"))
        line["config4_hop_positions_per_s"] = r6(c4.get("hop_positions_per_s"))
        line["config4_traffic_over_algorithmic"] = r6(rf.get("traffic_over_algorithmic"))
    st = full.get("stereo4096")
    if isinstance(st, dict):
        rf = st.get("roofline") or {}
        line["stereo_frac"] = r6(rf.get("frac"))
        line["stereo_fp32_frac"] = r6(rf.get("fp32_frac"))
        line["stereo_fp32_frac_nominal"] = r6(rf.get("fp32_frac_nominal"))
        line["stereo_valu_issue_frac"] = r6(rf.get("valu_issue_frac"))
        line["stereo_frames_per_s"] = r6(st.get("frames_per_s"))
    for key, short in (("mono_paired_frames", "mono_paired_frac"), ("mono_complex_frames", "mono_complex_frac")):
        if isinstance(full.get(key), dict):
            line[short] = r6((full[key].get("roofline") or {}).get("frac"))
    ap = full.get("app_point")
    if isinstance(ap, dict):
        line["app_point"] = {k: r6(ap[k]["frames_per_s"]) for k in ("rows_f32", "pcm_to_rgba", "mono_rows_f32", "mono_pcm_to_rgba")
                             if isinstance(ap.get(k), dict) and "frames_per_s" in ap[k]}
    c5 = full.get("config5")
    if isinstance(c5, dict):
        d = pick(c5, "backend", "ranks_seen", "frames_total", "frames_per_gpu", "chunk_columns", "rounds", "frames_per_s",
                 "algorithmic_GBps", "overlapped_s", "render_only_s", "gather_only_s", "overlap_ratio", "gathered_bytes",
                 "GBps_into_root", "GBps_into_root_gather_only", "checksum_all_columns",
                 "sharded_equals_single_gpu_on_first_chunk_of_every_rank", "probes_file", "probes")
        if isinstance(c5.get("GBps_per_source_alone"), list):
            d["GBps_per_source_alone"] = [r6(q.get("GBps")) for q in c5["GBps_per_source_alone"]]
        line["config5"] = d
    line["legs_file"] = LEGS_FILE
    DROP_ORDER = ("app_point", "mono_complex_frac", "mono_paired_frac", "achieved_GBps_algorithmic", "config3_rgba",
                  "config4_traffic_over_algorithmic", "config3_cubic_fp32_frac", "config3_cubic_frames_per_s", "stereo_frames_per_s",
                  "config4_hop_positions_per_s", "config3_frames_per_s", "checksum_first_4096_frames", "legs_file")
    size = lambda: len(json.dumps(line, separators=(", ", ": ")))
    for k in DROP_ORDER:
        if size() <= budget:
            break
        line.pop(k, None)
    # still too long (a very long error text, many ranks): shorten what is free text, then the optional sub-objects
    for obj, key, keep in ((line, "error", 300), (line.get("config", {}), "workload", 160), (line.get("cpu_baseline", {}), "sample", 120),
                           (line.get("config", {}), "kernel", 80)):
        if size() > budget and isinstance(obj.get(key), str) and len(obj[key]) > keep:
            obj[key] = obj[key][:keep - 3] + "..."
    for obj, key in ((line.get("roofline", {}), "measured_device"), (line.get("roofline", {}), "placement"),
                     (line.get("config5", {}), "GBps_per_source_alone"), (line.get("cpu_baseline", {}), "library"),
                     (line.get("cpu_baseline", {}), "single_thread"), (line.get("roofline", {}), "first_allocation")):
        if size() > budget:
            obj.pop(key, None)
    assert size() <= budget, f"bench.py: the stdout line is {size()} bytes (> {budget})"
    return line


def emit_line(full, stream=None):
    """rank 0: the full record as ONE line on stderr (prefix `bench_legs `) and in bench_legs.json next to bench.py (and in gpurun_out/
    when that directory exists), then the compact line -- alone -- on stdout."""
    txt = json.dumps(full)
    print("bench_legs " + txt, file=sys.stderr, flush=True)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, LEGS_FILE), "w") as f:
                    f.write(txt + "\n")
            except OSError as e:      # a read-only tree: the line on stderr is the copy
                print(f"bench.py: {LEGS_FILE} not written in {d}: {e}", file=sys.stderr)
    print(json.dumps(compact_line(full)), file=stream or sys.stdout, flush=True)


def kernel_name(eng):
    """(description, profiler name) of the transform kernel an engine's float rows come from"""
    if eng.info.stft_kernel == 2 and eng.channels == 1 and eng.info.render_path & 8:
        return KERNEL_NAMES["real"]
    return KERNEL_NAMES.get(eng.info.stft_kernel, ("?", "?"))


def fp32_fracs(leg, nominal_flop_per_unit, units_per_s):
    """SURVEY 8(d) asks for the FP32-vector fraction beside the HBM fraction.  `fp32_frac_nominal`: the 5 P log2 P convention (one
    complex transform of the reference's length per frame, whatever the kernel really does) x rate / 157.3 TFLOP/s; `fp32_frac`: the
    f32 operations the kernel EXECUTES per unit -- SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 x 64 lanes (FMA twice), a rocprofv3 --pmc
    pass of this round committed as profiles/<round>_fp32_flops.json (tools/pmc_flops.py) -- x rate / peak; None without that file."""
    prof = load_profile_json("fp32_flops") or {}
    counted = (prof.get("flop_per_unit") or {}).get(leg)
    peak = FP32_PEAK_TFLOPS * 1e12
    out = {"fp32_frac_nominal": nominal_flop_per_unit * units_per_s / peak,
           "fp32_frac": (counted * units_per_s / peak) if counted else None,
           "fp32_flop_per_unit_counted": counted, "fp32_flop_per_unit_nominal": nominal_flop_per_unit}
    # ... and the share of the vector pipes' ISSUE slots the kernel fills: every wave-instruction of the vector ALU (SQ_INSTS_VALU: adds,
    # multiplies, moves, address arithmetic alike) holds its SIMD for 2 clocks; 1 024 SIMDs at the 2.4 GHz the peak is quoted at.  An
    # FFT is adds and multiplies (1 flop per lane and slot) more than fused multiply-adds (2): fp32_frac can at best reach about half.
    lg = (prof.get("legs") or {}).get(leg) or {}
    if lg.get("valu_wave_instructions_all") and lg.get("units_per_launch"):
        out["valu_issue_frac"] = lg["valu_wave_instructions_all"] / lg["units_per_launch"] * units_per_s * 2.0 / (1024 * 2.4e9)
    return out


def stats_ms(v):
    if not v:
        return None
    s = sorted(v)
    return {"n": len(s), "min": s[0], "median": s[len(s) // 2], "mean": sum(s) / len(s), "max": s[-1]}


def library_fft_rate(np, oracle, W, H, frames, cores, ref):
    """frames/s of the FFT call of fft.rs:77 alone, done with scipy.fft (pocketfft, complex64) on `cores` host threads
    over a bounded sample (no FFTW in this image or on the GPU box); window / pack / split are checked against the
    oracle's first frames but NOT part of the figure (numpy would do them on one thread)."""
    import scipy.fft

    win = oracle.hann_window(W)
    batch, done, dt_fft, first = 4096, 0, 0.0, None
    z = np.zeros((batch, 2 * W), np.complex64)                    # the padding half stays zero (out-of-place FFT)
    while done < frames:
        m = min(batch, frames - done)
        host = oracle.white_noise((m - 1) * H + W, first=done * H)
        fr = np.lib.stride_tricks.as_strided(host, shape=(m, W), strides=(host.strides[0] * H, host.strides[0]))
        sw = fr * win
        z.real[:m, :W] = sw                                       # mono -> (s, s): l + i r
        z.imag[:m, :W] = sw
        t1 = time.perf_counter()
        F = scipy.fft.fft(z[:m], axis=1, workers=cores)
        dt_fft += time.perf_counter() - t1
        if first is None:
            a, b = F[:8, 1:W], F[:8, 2 * W - 1:W:-1]              # F[k], F[P - k], k = 1 .. W-1
            first = np.stack([np.abs(a + np.conj(b)), np.abs(a - np.conj(b))], axis=2) * np.float32(1.0 / W)
        done += m
    peak = np.abs(ref[:8, 0]).max(axis=(1, 2), keepdims=True)
    ok = bool((np.abs(first - ref[:8, 0]) <= 2e-5 * np.maximum(np.abs(ref[:8, 0]), 0.02 * peak)).all())
    return {"fft_call_only": frames / dt_fft, "unit": "frames/s", "cores": cores, "frames": frames,
            "what": "scipy.fft (pocketfft) complex64 4096-point c2c, the FFT call alone, all host threads", "matches_oracle": ok}


def library_loop_rate(np, oracle, W, H, frames, cores, ref):
    """frames/s of the WHOLE reference frame loop (fft.rs:47-99: Hann, (s, s) pack, zero pad, c2c FFT, L/R split, hypot, 2 / W) with an
    optimised library FFT -- scipy.fft (pocketfft) complex64 -- every one of `cores` host threads running the loop end to end on its own
    chunks of frames (numpy and pocketfft release the GIL).  The nearest thing to the reference's FFTW-backed CPU path this image
    holds; checked against the oracle on the first 64 frames."""
    from concurrent.futures import ThreadPoolExecutor

    import scipy.fft

    win = oracle.hann_window(W)
    host = oracle.white_noise((frames - 1) * H + W)
    scale = np.float32(0.5) * (np.float32(2.0) / np.float32(W))        # hypot / 2, then 2 / W (fft.rs:87-88,92-95)
    chunk = 256

    import threading
    tls = threading.local()

    def work(f0):
        m = min(chunk, frames - f0)
        if not hasattr(tls, "z"):                                       # per-thread scratch, allocated once (the reference allocates per frame)
            tls.z = np.zeros((chunk, 2 * W), np.complex64)              # :65-69: the padding half stays zero
            tls.t = np.empty((chunk, W - 1), np.complex64)
        z, t = tls.z[:m], tls.t[:m]
        seg = host[f0 * H:(f0 + m - 1) * H + W]
        fr = np.lib.stride_tricks.as_strided(seg, shape=(m, W), strides=(seg.strides[0] * H, seg.strides[0]))
        np.multiply(fr, win, out=z.real[:, :W])                         # :59-63; mono -> (s, s): l + i r (:57, audio_input_list_model.rs:67-69)
        z.imag[:, :W] = z.real[:, :W]
        F = scipy.fft.fft(z, axis=1, workers=1)                         # :77
        a, b = F[:, 1:W], F[:, 2 * W - 1:W:-1]                          # F[k], F[P - k], k = 1 .. W - 1 (:81-82)
        out = np.empty((m, W - 1, 2), np.float32)
        np.conjugate(b, out=t)
        np.abs(a + t, out=out[:, :, 0])                                 # :87
        np.abs(a - t, out=out[:, :, 1])                                 # :88
        out *= scale                                                    # :92-95
        return f0, out

    first = None
    with ThreadPoolExecutor(max_workers=cores) as pool:
        for _ in pool.map(work, [0] * (2 * cores)):                     # plan caches, every thread's scratch, page-in
            pass
        t0 = time.perf_counter()
        for f0, out in pool.map(work, range(0, frames, chunk)):
            if f0 == 0:
                first = out[:64].copy()
            del out
        dt = time.perf_counter() - t0
    peak = np.abs(ref[:, 0]).max(axis=(1, 2), keepdims=True)
    ok = bool((np.abs(first - ref[:, 0]) <= 2e-5 * np.maximum(np.abs(ref[:, 0]), 0.02 * peak)).all())
    return {"value": frames / dt, "unit": "frames/s", "cores": cores, "frames": frames, "matches_oracle": ok, "impl": "numpy + scipy.fft (pocketfft) complex64",
            "what": "the whole frame loop of fft.rs:47-99 in numpy + scipy.fft (pocketfft) complex64, every thread its own chunks of 256 frames"}


def main_rank(args):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if os.environ.get("BENCH_LAUNCH_ONLY") == "1":
        # launcher self-test (tests/test_bench_launcher.py, no GPU): rendezvous over gloo, one all-reduce, a stub line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.tensor([float(rank)], dtype=torch.float64)
            dist.all_reduce(t)
            ranks_sum = float(t[0])
        else:
            ranks_sum = 0.0
        if test_hook("BENCH_FAIL_RANK") == str(rank):
            os._exit(7)
        if rank == 0:
            print(json.dumps({"metric": "launcher self-test", "n_gpus": world, "ranks_sum": ranks_sum,
                              "launched_by_parent": os.environ.get("BENCH_LAUNCHED_BY_PARENT") == "1"}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return 0
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback"
    # rehearsal knobs (one-GPU box): BENCH_BACKEND=gloo BENCH_SINGLE_DEVICE=1 run every rank on cuda:0 so
    # that the N > 1 control flow can be exercised without a second GPU; the driver never sets them
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if os.environ.get("BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # BENCH_GROUP_OF_ONE=1 (tests/test_gpu_config5.py, one-GPU box): a single launched rank takes the multi-rank path -- process group
    # over `backend` (RCCL: init, barrier, all-reduce, all-gather execute for real; it refuses two ranks on one device), config-5 leg
    grouped = world > 1 or (test_hook("BENCH_GROUP_OF_ONE") == "1" and "RANK" in os.environ)
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from spectrogram_rs_amd import SpectrogramEngine
    from spectrogram_rs_amd.sharding import frame_range, sample_range, stream_columns

    red_dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")

    def barrier():
        if grouped:
            dist.barrier()

    def max_over_ranks(vals):
        t = torch.tensor(vals, dtype=torch.float64, device=red_dev)
        if grouped:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t]

    F = args.frames
    eng = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=1, device=local_rank,
                            force_generic=args.generic, interp=1, gradient="viridis")
    # weak scaling: one stream of world*F frames; rank g owns the contiguous range frame_range(g)
    # and generates exactly its samples (with the W-H halo) -- no input exchange
    first_frame, n_own = frame_range(rank, world, world * F)
    assert n_own == F
    first_sample, n_samples = sample_range(first_frame, F, W, H)
    pcm = eng.white_noise(n_samples, first=first_sample)
    # Where the 16.4 GB output buffer lies can decide K1's rate: allocations of the same size read 3.30, 3.53 or 3.72 ms per launch on
    # some devices depending on their physical pages (profiles/r03_k1_slow_box.txt).  The buffer is the caller's, so a caller may
    # choose -- but a choice must rest on evidence: place_output times the candidates hot and interleaved, twice, and keeps the
    # first allocation unless another is faster in both passes.  roofline.first_allocation reports the first allocation under the
    # protocol of `value` either way.
    mags, mags_first, placement = place_output(torch, args.placements,
                                               lambda: torch.empty((F, 1, M, 2), dtype=torch.float32, device=eng.device),
                                               lambda buf: eng.stft_batch(pcm, out=buf), F * ALGO_BYTES_STFT)

    def timed_launches(n, buf=None):
        buf = mags if buf is None else buf
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in evs:
            a.record()
            eng.stft_batch(pcm, out=buf)
            b.record()
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in evs]

    for _ in range(args.warmup):
        eng.stft_batch(pcm, out=mags)
    torch.cuda.synchronize()
    burst = timed_launches(args.steps)          # cool device: the figure round 1 reported
    sustained, t_s0 = [], time.perf_counter()
    while args.sustain_s > 0 and time.perf_counter() - t_s0 < args.sustain_s:
        sustained += timed_launches(32)          # ~0.1 s per batch; the queue is never empty for longer than a sync
    sustain_wall = time.perf_counter() - t_s0
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in evs:
        a.record()
        eng.stft_batch(pcm, out=mags)
        b.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    timed = [a.elapsed_time(b) for a, b in evs]
    kernel_ms = sum(timed) / max(len(timed), 1)
    steady = sustained[len(sustained) // 3:]     # the first third of the window is still heating up
    sustained_ms = (sum(steady) / len(steady)) if steady else kernel_ms
    burst_ms = sum(burst) / max(len(burst), 1)
    # the first allocation under the same protocol, on the same hot device: K timed steps right behind the K that count
    if mags_first is not None:
        first = timed_launches(args.steps, mags_first)
        first_ms = sum(first) / max(len(first), 1)
        del mags_first
    else:
        first, first_ms = timed, kernel_ms
    elapsed, kernel_ms, sustained_ms, burst_ms, first_ms = max_over_ranks([elapsed, kernel_ms, sustained_ms, burst_ms, first_ms])
    checksum = eng.checksum(mags[:4096])
    try:
        sclk = int(torch.cuda.clock_rate())
    except Exception:  # noqa: BLE001 -- not every build exposes it
        sclk = None

    def frac_of(ms):
        return F * ALGO_BYTES_STFT / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS

    # the fraction of a MEASURED streaming rate next to the fraction of the vendor peak -- the same 16 GB
    # output buffer filled (write only) and copied half to half (read + write) by the runtime's own kernels, hot device
    probe = device_streaming_rates(torch, mags) if rank == 0 else None

    def build_line(extra):
        total_frames = world * F * args.steps
        value = total_frames / elapsed
        achieved = F * ALGO_BYTES_STFT / (kernel_ms * 1e-3) / 1e9
        traffic = load_profile_json("hbm_traffic")
        line = {
            "metric": "STFT frames/sec (4096-pt, hop 256)",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"configs[1]: batched 4096-pt Hann STFT, hop 256, {F} frames/GPU of counter-based white-noise mono PCM resident in HBM",
                "window": W, "fft_length": 2 * W, "hop": H, "channels": 1, "frames_per_gpu": F,
                "kernel": kernel_name(eng)[0],
                "mono_mode": "every frame its own transform (the reference's (s, s) dataflow, audio_input_list_model.rs:67-69; default flags)"
                             if eng.info.render_path & 8 else "two frames per transform (frame 2j in the real part, 2j+1 in the imaginary part)",
                "sharding": "contiguous frame ranges per rank, no data-path collective" if world > 1 else "single GPU",
                "timed_region": f"{args.steps} steps after {args.warmup} warm-up steps, {args.steps} burst steps and {sustain_wall:.1f} s of back-to-back steps",
            },
            "achieved_GBps_algorithmic": value * ALGO_BYTES_STFT / 1e9,
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                **fp32_fracs("config2_stft", NOMINAL_FLOP_4096, F / (kernel_ms * 1e-3)),
                "frac_burst": frac_of(burst_ms), "frac_sustained": frac_of(sustained_ms),
                "traffic": ((traffic or {}).get("stft_bytes_per_frame") or 0) * F or None,
                "traffic_source": (traffic or {}).get("source"),
                "kernel": kernel_name(eng)[1],
                "launch_ms": kernel_ms, "bytes_per_frame": ALGO_BYTES_STFT, "frames_per_launch": F,
                "launch_ms_burst": stats_ms(burst), "launch_ms_sustained": stats_ms(steady),
                "first_allocation": {"launch_ms": first_ms, "frac": frac_of(first_ms), "frames_per_s": F / (first_ms * 1e-3),
                                     "launch_ms_stats": stats_ms(first), "is_value": placement["chosen"] == 0,
                                     "what": "the FIRST allocation of the output buffer, K timed steps on the same hot device "
                                             "(the K steps of `value` themselves when the first allocation was kept)"},
                "placement": placement,
                "sustain_s": sustain_wall, "sclk_mhz_after_run": sclk,
                "measured_device": None if probe is None else dict(
                    probe, frac_of_fill=achieved / probe["fill_GBps"], frac_of_copy=achieved / probe["copy_GBps"]),
                "note": "frac / launch_ms: the K timed steps (device hot); *_burst: the first K steps after warm-up; "
                        "*_sustained: launches of the last two thirds of the sustain window; max over ranks",
            },
            "checksum_first_4096_frames": checksum,
        }
        line.update(extra)
        return line

    # ---- everything below is reported beside the headline, which is final ------------------------------------
    import threading

    def on_stall():
        if rank == 0:
            emit_line(build_line({"error": f"a collective leg did not finish within {args.leg_timeout} s"}))
        os._exit(3)   # a stalled exchange is a failed run: never status 0

    extra = {}
    watchdog = threading.Timer(args.leg_timeout, on_stall)
    watchdog.daemon = True
    if grouped:
        watchdog.start()
    try:
        if not grouped:
            del mags
            torch.cuda.empty_cache()
            if args.pixel_frames > 0:
                extra["config3"] = config3_leg(args, torch, eng, pcm, F)
            if args.config4_hops > 0:
                extra["config4"] = config4_leg(args, torch, local_rank)
            del pcm
            torch.cuda.empty_cache()
            pcm = None
            if args.stereo_frames > 0:
                extra["stereo4096"] = stereo_leg(args, torch, local_rank)
            if args.paired_frames > 0:
                extra["mono_paired_frames"] = mono_mode_leg(args, torch, local_rank, args.paired_frames, dict(paired_frames=True),
                                                            "two frames per transform (SGX_FLAG_PAIRED_FRAMES: the headline mode of rounds 1-3)",
                                                            "sgx::wg::stft4096_wg_kernel<true, 0, false, 0>")
            if args.complex_frames > 0:
                extra["mono_complex_frames"] = mono_mode_leg(args, torch, local_rank, args.complex_frames, dict(complex_mono=True),
                                                             "every frame the literal (s, s) 4096-point transform (SGX_FLAG_COMPLEX_MONO; fft.rs:47-57)",
                                                             "sgx::wg::stft4096_wg_kernel<false, 1, false, 0>")
            if args.app_frames > 0:
                extra["app_point"] = app_point_leg(args, torch, local_rank)
        elif args.config5_frames > 0:
            extra["config5"] = config5_leg(args, torch, dist, eng, rank, world, backend, barrier, max_over_ranks,
                                           frame_range, stream_columns)
    except Exception as e:  # noqa: BLE001 -- reported in the line; the headline stands, the status does not
        extra["error"] = f"{type(e).__name__}: {e}"
        if rank == 0:
            emit_line(build_line(extra))
        os._exit(3)       # the other ranks may be waiting in the exchange: nothing collective from here on
    watchdog.cancel()

    # ---- CPU baseline: the oracle on this host's cores (rank 0, N = 1 only) ------------------------
    if rank == 0 and not grouped and args.cpu_frames > 0:
        if pcm is None:
            pcm = eng.white_noise(W + 63 * H)     # the parity check of the leg reads the first 64 frames
        extra["cpu_baseline"] = cpu_baseline_leg(args, eng, pcm)
    if rank == 0:
        emit_line(build_line(extra))
    if grouped:
        dist.destroy_process_group()
    return 0


def group_ms(torch, launch, buf, reps):
    """mean ms per launch of `reps` back-to-back launches into buf (one event pair around the group)"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch(buf)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def place_output(torch, n_candidates, alloc, launch, bytes_per_launch, reps=10, heat_s=1.0, margin=0.02):
    """On some devices the rate of a store-heavy launch depends on the physical pages of its output buffer
    (profiles/r03_k1_slow_box.txt).  Up to n_candidates buffers are allocated (all alive at once, so that they are different
    pages) and timed HOT and INTERLEAVED: after heat_s seconds of launches (clocks ramped, every candidate touched) two passes
    A B C D / D C B A of `reps` launches each.  (Round 3 timed each candidate cold, 1 + 3 launches, right after its allocation:
    on the driver's box that read 3.71-3.76 ms for a buffer that ran 3.27 ms two seconds later -- clock ramp, not placement.)
    A candidate other than the first allocation is kept only if BOTH its passes are faster than BOTH passes of the first
    allocation by more than `margin` and its own two passes agree within `margin`: a stable class difference, not a draw.
    Returns (chosen buffer, first buffer or None when the first IS the chosen one, report)."""
    cands = []
    for _ in range(max(1, n_candidates)):
        try:
            buf = alloc()
        except RuntimeError:     # out of memory: what we have is what we compare
            break
        buf.zero_()              # first touch belongs to the allocation, not to a step
        cands.append(buf)
    if not cands:
        raise RuntimeError(f"bench.py: not even one output buffer ({bytes_per_launch / 1e9:.1f} GB algorithmic per launch) could be allocated on this device")
    if len(cands) == 1:
        launch(cands[0])
        torch.cuda.synchronize()
        return cands[0], None, {"candidates": 1, "requested": max(1, n_candidates), "chosen": 0, "kept_selection": False,
                                "why": "one candidate: the first allocation is the buffer"}
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < heat_s:          # heat: every candidate in turn
        for buf in cands:
            launch(buf)
        torch.cuda.synchronize()
    order = list(range(len(cands)))
    p1 = {i: group_ms(torch, launch, cands[i], reps) for i in order}
    p2 = {i: group_ms(torch, launch, cands[i], reps) for i in reversed(order)}
    pass1, pass2 = [p1[i] for i in order], [p2[i] for i in order]
    mean = [(a + b) / 2 for a, b in zip(pass1, pass2)]
    stable = [abs(a - b) <= margin * m for a, b, m in zip(pass1, pass2, mean)]
    best = min(order, key=lambda i: mean[i])
    keep = (best != 0 and stable[best] and stable[0]
            and max(pass1[best], pass2[best]) < (1.0 - margin) * min(pass1[0], pass2[0]))
    chosen = best if keep else 0
    keep_buf, first_buf = cands[chosen], (cands[0] if chosen != 0 else None)
    del cands, buf
    torch.cuda.empty_cache()
    frac = lambda ms: bytes_per_launch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    return keep_buf, first_buf, {
        "candidates": len(order), "requested": max(1, n_candidates), "protocol": f"{heat_s:.1f} s of launches over all candidates, then two passes in opposite order, {reps} launches per candidate and pass",
        "pass1_ms_per_launch": pass1, "pass2_ms_per_launch": pass2, "stable_within_2pct": stable,
        "spread": (max(mean) - min(mean)) / min(mean), "fastest": best, "chosen": chosen, "kept_selection": bool(keep),
        "frac_of_candidate_0": frac(mean[0]), "frac_of_fastest": frac(mean[best]),
        "why": "physical placement of the output buffer (profiles/r03_k1_slow_box.txt): a candidate other than the first allocation is kept "
               "only if it is > 2 % faster in both passes"}


def measure_leg(torch, launch, sustain_s, burst=5, warm=2):
    """The protocol of every side leg: `warm` untimed launches, `burst` timed launches, then back-to-back launches for
    sustain_s seconds (per-launch HIP-event times).  The leg's figure is the mean of the last two thirds of the sustained
    window (the first third is still heating up); with sustain_s = 0 it is the burst."""
    burst_ms = event_times(torch, launch, reps=burst, warm=warm)
    sustained, t0 = [], time.perf_counter()
    while sustain_s > 0 and time.perf_counter() - t0 < sustain_s:
        n = max(4, min(32, int(0.1 / max(burst_ms[-1] * 1e-3, 1e-6))))      # ~0.1 s per batch: the queue never runs dry for long
        sustained += event_times(torch, launch, reps=n, warm=0)
    steady = sustained[len(sustained) // 3:] or burst_ms
    return {"mean_ms": sum(steady) / len(steady), "launch_ms_sustained": stats_ms(steady), "launch_ms_burst": stats_ms(burst_ms),
            "sustain_s": time.perf_counter() - t0 if sustain_s > 0 else 0.0}


def leg_times(m):
    """the timing fields every leg carries"""
    return {"launch_ms": m["launch_ms_sustained"], "launch_ms_sustained": m["launch_ms_sustained"], "launch_ms_burst": m["launch_ms_burst"],
            "sustain_s": m["sustain_s"]}


def device_streaming_rates(torch, buf):
    flat = buf.view(-1)
    half = flat.numel() // 2
    a, b = flat[:half], flat[half:2 * half]
    keep = flat[:4096 * 2047 * 2].clone()   # the frames the checksum is taken over
    fill = event_times(torch, lambda: flat.zero_(), 5)
    copy = event_times(torch, lambda: a.copy_(b), 5)
    flat[:keep.numel()].copy_(keep)
    torch.cuda.synchronize()
    nbytes = flat.numel() * 4
    med = lambda v: sorted(v)[len(v) // 2]
    return {"fill_GBps": nbytes / (med(fill) * 1e-3) / 1e9, "copy_GBps": 2 * half * 4 / (med(copy) * 1e-3) / 1e9,
            "how": f"torch zero_() over the {nbytes / 1e9:.1f} GB output buffer (bytes written) and copy_() of one half onto the other "
                   "(bytes read + written), median of 5, HIP events, after the timed steps"}


def event_times(torch, fn, reps, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in evs]


def config3_leg(args, torch, eng, pcm, F):
    """BASELINE config 3: the same stream -> 1024 log rows (cosine) -> Viridis RGBA, one fused kernel."""
    Fp = min(args.pixel_frames, F)
    rgba = torch.empty((Fp, 1, R, 4), dtype=torch.uint8, device=eng.device)
    rgba.zero_()
    m = measure_leg(torch, lambda: eng.render_batch(pcm, max_frames=Fp, out=rgba), args.leg_sustain_s)
    mean = m["mean_ms"]
    achieved = Fp * ALGO_BYTES_PIXEL / (mean * 1e-3) / 1e9
    parity = {"cosine": rgba_vs_oracle(torch, eng, pcm, rgba, Fp, interp=1)}
    pipes = load_profile_json("pixel_pipes")
    traffic = load_profile_json("hbm_traffic")
    # the interpolator the reference actually executes (interpolated_frequency_sample.rs:46-48 calls the cubic one; the cosine one
    # BASELINE names is dead code there): same leg, same kernel, cubic taps
    from spectrogram_rs_amd import SpectrogramEngine
    cubic = None
    try:
        eng3 = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=1, device=eng.device.index, interp=0, gradient="viridis")
        m3 = measure_leg(torch, lambda: eng3.render_batch(pcm, max_frames=Fp, out=rgba), args.leg_sustain_s)
        cubic = dict({"frames_per_s": Fp / (m3["mean_ms"] * 1e-3),
                      "what": "the same leg with the cubic interpolator, which is what the reference runs"}, **leg_times(m3),
                     **fp32_fracs("config3_cubic", NOMINAL_FLOP_4096, Fp / (m3["mean_ms"] * 1e-3)))
        parity["cubic"] = rgba_vs_oracle(torch, eng3, pcm, rgba, Fp, interp=0)
        eng3.close()
    except Exception as e:  # noqa: BLE001 -- an extra, never fatal
        cubic = {"error": f"{type(e).__name__}: {e}"}
    paired = None
    try:
        engp = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=1, device=eng.device.index, interp=1, gradient="viridis", paired_frames=True)
        mpf = measure_leg(torch, lambda: engp.render_batch(pcm, max_frames=Fp, out=rgba), args.leg_sustain_s)
        paired = dict({"frames_per_s": Fp / (mpf["mean_ms"] * 1e-3), "what": "the same leg (cosine) with two frames per transform (SGX_FLAG_PAIRED_FRAMES)"},
                      **leg_times(mpf))
        engp.close()
    except Exception as e:  # noqa: BLE001 -- an extra, never fatal
        paired = {"error": f"{type(e).__name__}: {e}"}
    return {
        "workload": f"configs[2]: {Fp} frames of the same stream -> 1024 log rows (cosine interpolation), Viridis RGBA, fused PCM-to-pixel kernel",
        "paired_frames": paired,
        "frames_per_s": Fp / (mean * 1e-3),
        **leg_times(m),
        "cubic": cubic,
        "rgba_vs_oracle": parity,
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            **fp32_fracs("config3_cosine", NOMINAL_FLOP_4096, Fp / (mean * 1e-3)),
            "bytes_per_frame": ALGO_BYTES_PIXEL, "frames_per_launch": Fp,
            "traffic": ((traffic or {}).get("pixel_bytes_per_frame") or 0) * Fp or None,
            "traffic_source": (traffic or {}).get("source"),
            "kernel": "sgx::wgr::stft4096_real_kernel<2, 2, true>" if eng.info.render_path & 8 else "sgx::wg::stft4096_wg_kernel<true, 0, false, 2>",
            "mono_mode": "every frame its own transform" if eng.info.render_path & 8 else "two frames per transform",
            "binding_pipe": pipes,
            "note": "48 flop per algorithmic byte: above the FP32 ridge (19.7), so the HBM fraction is low by construction; "
                    "the binding pipes (VALU issue, LDS) are in binding_pipe, from this round's SQ counter pass",
        },
    }


def rgba_vs_oracle(torch, eng, pcm, rgba, Fp, interp):
    """the end-to-end pixel result is REPORTED -- mismatch rate and largest LUT-step delta of the fused
    kernel's bytes (full-scale noise, the columns just rendered) against the CPU oracle run end to end (its own float32
    transform, then its pixel stage) on 1 024 sampled frames t = i * 977 mod F.  The stage-wise figure (the oracle's pixel
    stage over the engine's own magnitudes) is bit-exact by test; it is recomputed here on 128 of the frames."""
    import numpy as np

    import oracle

    n = min(1024, Fp)
    ts = [(i * 977) % Fp for i in range(n)]
    got = rgba[ts, 0].cpu().numpy()
    host = np.stack([pcm[t * H:t * H + W].cpu().numpy() for t in ts])
    from spectrogram_rs_amd.engine import builtin_gradient
    lut_rgb = builtin_gradient("viridis")
    ref_mags = np.stack([oracle.fft_process(np.stack([h, h], 1), W) for h in host])
    ref = oracle.render_columns(ref_mags, 48000, lut_rgb, interp=interp)
    level = {tuple(int(x) for x in c): i for i, c in enumerate(lut_rgb)}
    bad = np.argwhere((got != ref).any(axis=2))
    steps = [abs(level.get(tuple(int(x) for x in got[a, b, :3]), 10**6) - level[tuple(int(x) for x in ref[a, b, :3])]) for a, b in bad]
    own = np.concatenate([eng.stft_batch(pcm, first_frame=t, max_frames=1).cpu().numpy()[:, 0] for t in ts[:128]])
    stage_ok = bool(np.array_equal(got[:128], oracle.render_columns(own, 48000, lut_rgb, interp=interp)))
    return {"frames": n, "pixels": int(got.shape[0] * got.shape[1]), "mismatched_pixels": int(len(bad)),
            "mismatch_rate": float(len(bad)) / float(got.shape[0] * got.shape[1]), "max_lut_step": int(max(steps) if steps else 0),
            "stage_wise_bit_exact_on_128_frames": stage_ok,
            "input": "full-scale white noise (the bench stream), frames t = i * 977 mod F"}


def stereo_leg(args, torch, device):
    """The 4096-point transform on an (l, r) stream -- what the reference feeds (audio_input_list_model.rs:66-72): one
    frame per transform, 18 424 algorithmic bytes per frame."""
    from spectrogram_rs_amd import SpectrogramEngine

    Fs = args.stereo_frames
    eng = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=2, device=device)
    pcm = eng.white_noise((Fs - 1) * H + W)
    out, first, placement = place_output(torch, args.placements, lambda: torch.empty((Fs, 1, M, 2), dtype=torch.float32, device=eng.device),
                                         lambda buf: eng.stft_batch(pcm, out=buf), Fs * ALGO_BYTES_STEREO)
    m = measure_leg(torch, lambda: eng.stft_batch(pcm, out=out), args.leg_sustain_s)
    mean = m["mean_ms"]
    achieved = Fs * ALGO_BYTES_STEREO / (mean * 1e-3) / 1e9
    res = {
        "workload": f"4096-pt Hann STFT, hop 256, {Fs} frames of an (l, r) white-noise stream (2 channels interleaved), one frame per transform",
        "frames_per_s": Fs / (mean * 1e-3), **leg_times(m),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     **fp32_fracs("stereo4096", NOMINAL_FLOP_4096, Fs / (mean * 1e-3)),
                     "bytes_per_frame": ALGO_BYTES_STEREO, "frames_per_launch": Fs,
                     "kernel": "sgx::wg::stft4096_wg_kernel<false, 0, true, 0>",   # (l, r) at hop 256: the sliding-window instantiation
                     "first_allocation": first_allocation_of(torch, first, m, lambda buf: eng.stft_batch(pcm, out=buf), Fs * ALGO_BYTES_STEREO, args),
                     "placement": placement,
                     "note": "bound by the transform rate of the kernel (LDS exchanges + vector issue, DESIGN section 4 K1), not by HBM"},
    }
    del out, first
    eng.close()
    return res


def first_allocation_of(torch, first, m_chosen, launch, bytes_per_launch, args):
    """roofline.first_allocation of a side leg: the first allocation under the leg's own protocol (the chosen buffer's figures when
    the first allocation was kept)"""
    m = m_chosen if first is None else measure_leg(torch, lambda: launch(first), min(args.leg_sustain_s, 0.5))
    return {"launch_ms": m["mean_ms"], "frac": bytes_per_launch / (m["mean_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "is_value": first is None}


def mono_mode_leg(args, torch, device, Fi, flags, what, kernel):
    """The headline stream (configs[1]) in another mono mode of the library (include/sgx.h, "Mono streams"): same 17 400 algorithmic
    bytes per frame, same protocol as every side leg."""
    from spectrogram_rs_amd import SpectrogramEngine

    eng = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=1, device=device, **flags)
    pcm = eng.white_noise((Fi - 1) * H + W)
    out, first, placement = place_output(torch, args.placements, lambda: torch.empty((Fi, 1, M, 2), dtype=torch.float32, device=eng.device),
                                         lambda buf: eng.stft_batch(pcm, out=buf), Fi * ALGO_BYTES_STFT)
    m = measure_leg(torch, lambda: eng.stft_batch(pcm, out=out), args.leg_sustain_s)
    mean = m["mean_ms"]
    achieved = Fi * ALGO_BYTES_STFT / (mean * 1e-3) / 1e9
    res = {
        "workload": f"configs[1], {Fi} frames of the same mono stream: {what}",
        "frames_per_s": Fi / (mean * 1e-3), **leg_times(m),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "bytes_per_frame": ALGO_BYTES_STFT, "frames_per_launch": Fi, "kernel": kernel,
                     "first_allocation": first_allocation_of(torch, first, m, lambda buf: eng.stft_batch(pcm, out=buf), Fi * ALGO_BYTES_STFT, args),
                     "placement": placement},
    }
    del out, first
    eng.close()
    return res


def app_point_leg(args, torch, device):
    """The application's own operating point: 48 kHz x 0.05 s -> W 2400 (2W = 4800 = 2^6 3 5^2: mixed radix), hop 93
    ((2 / 1024) s, simple_spectrogram.rs:102), stereo; float32 rows and PCM -> RGBA columns (Viridis, cubic = what the
    reference runs)."""
    from spectrogram_rs_amd import SpectrogramEngine

    Fa = args.app_frames
    eng = SpectrogramEngine(48000.0, period=0.05, hop_samples=H_APP, channels=2, device=device, interp=0, gradient="viridis")
    assert eng.W == W_APP and eng.H == H_APP, (eng.W, eng.H)
    pcm = eng.white_noise((Fa - 1) * H_APP + W_APP)
    out = torch.empty((Fa, 1, W_APP - 1, 2), dtype=torch.float32, device=eng.device)
    out.zero_()
    m = measure_leg(torch, lambda: eng.stft_batch(pcm, out=out), args.leg_sustain_s)
    del out
    rgba = torch.empty((Fa, 1, R, 4), dtype=torch.uint8, device=eng.device)
    rgba.zero_()
    mp = measure_leg(torch, lambda: eng.render_batch(pcm, out=rgba), args.leg_sustain_s)
    mean, meanp = m["mean_ms"], mp["mean_ms"]
    ach, achp = Fa * ALGO_BYTES_APP / (mean * 1e-3) / 1e9, Fa * ALGO_BYTES_APP_PIXEL / (meanp * 1e-3) / 1e9
    name = KERNEL_NAMES.get(eng.info.stft_kernel, ("?", "?"))
    # the same point for a mono device (audio_input_list_model.rs:67-69 duplicates the sample into (s, s)): two frames per transform
    del rgba, pcm
    mono = SpectrogramEngine(48000.0, period=0.05, hop_samples=H_APP, channels=1, device=device, interp=0, gradient="viridis")   # default: every frame its own transform
    monop = SpectrogramEngine(48000.0, period=0.05, hop_samples=H_APP, channels=1, device=device, paired_frames=True)   # opt-in: two frames per transform
    pcm1 = mono.white_noise((Fa - 1) * H_APP + W_APP)
    out1 = torch.empty((Fa, 1, W_APP - 1, 2), dtype=torch.float32, device=mono.device)
    out1.zero_()
    m1 = measure_leg(torch, lambda: mono.stft_batch(pcm1, out=out1), args.leg_sustain_s)
    m1p = measure_leg(torch, lambda: monop.stft_batch(pcm1, out=out1), args.leg_sustain_s)
    mean1, mean1p = m1["mean_ms"], m1p["mean_ms"]
    mono_real = bool(mono.info.render_path & 8)
    del out1
    rgba1 = torch.empty((Fa, 1, R, 4), dtype=torch.uint8, device=mono.device)
    rgba1.zero_()
    m1x = measure_leg(torch, lambda: mono.render_batch(pcm1, out=rgba1), args.leg_sustain_s)
    bytes1x = H_APP * 4 + R * 4
    ach1x = Fa * bytes1x / (m1x["mean_ms"] * 1e-3) / 1e9
    mono_fused = bool(mono.info.render_path & 1)
    del rgba1
    out1 = None
    bytes1 = H_APP * 4 + (W_APP - 1) * 8
    ach1, ach1p = Fa * bytes1 / (mean1 * 1e-3) / 1e9, Fa * bytes1 / (mean1p * 1e-3) / 1e9
    del out1, pcm1
    mono.close()
    monop.close()
    res = {
        "workload": f"the application's operating point: 48 kHz x 0.05 s = W 2400 (4800-point transform), hop 93, (l, r) stream, {Fa} frames",
        "kernel": name[0], "real_time_factor": Fa / (mean * 1e-3) * H_APP / 48000.0,
        "rows_f32": {"frames_per_s": Fa / (mean * 1e-3), **leg_times(m),
                     "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                  "bytes_per_frame": ALGO_BYTES_APP, "frames_per_launch": Fa}},
        "mono_rows_f32": {"frames_per_s": Fa / (mean1 * 1e-3), **leg_times(m1), "mono_mode": "every frame its own transform (default): real-input mode of the mixed-radix kernel, 2400 points" if mono_real else "every frame its own (s, s) transform (default)",
                          "roofline": {"bound": "hbm", "achieved": ach1, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach1 / HBM_PEAK_GBS,
                                       "bytes_per_frame": bytes1, "frames_per_launch": Fa}},
        "mono_rows_f32_paired_frames": {"frames_per_s": Fa / (mean1p * 1e-3), **leg_times(m1p), "mono_mode": "two frames per transform (SGX_FLAG_PAIRED_FRAMES)",
                                        "roofline": {"bound": "hbm", "achieved": ach1p, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach1p / HBM_PEAK_GBS,
                                                     "bytes_per_frame": bytes1, "frames_per_launch": Fa}},
        "mono_pcm_to_rgba": {"frames_per_s": Fa / (m1x["mean_ms"] * 1e-3), **leg_times(m1x), "fused_kernel": mono_fused,
                             "mono_mode": "default (real-input mode, two frames per workgroup)" if mono_real else "default ((s, s) transform)",
                             "roofline": {"bound": "hbm", "achieved": ach1x, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach1x / HBM_PEAK_GBS,
                                          "bytes_per_frame": bytes1x, "frames_per_launch": Fa}},
        "pcm_to_rgba": {"frames_per_s": Fa / (meanp * 1e-3), **leg_times(mp), "fused_kernel": bool(eng.info.render_path & 1),
                        "roofline": {"bound": "hbm", "achieved": achp, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achp / HBM_PEAK_GBS,
                                     "bytes_per_frame": ALGO_BYTES_APP_PIXEL, "frames_per_launch": Fa}},
    }
    eng.close()
    return res


def config4_leg(args, torch, device):
    """BASELINE config 4: 16384-point STFT, hop 512, 8 interleaved channels = 4 (l, r) pairs per hop position."""
    from spectrogram_rs_amd import SpectrogramEngine

    hops = args.config4_hops
    eng = SpectrogramEngine(48000.0, window_samples=W4, hop_samples=H4, channels=C4, device=device)
    pcm = eng.white_noise((hops - 1) * H4 + W4)
    out, first, placement = place_output(torch, args.placements,
                                         lambda: torch.empty((hops, C4 // 2, W4 - 1, 2), dtype=torch.float32, device=eng.device),
                                         lambda buf: eng.stft_batch(pcm, out=buf), hops * ALGO_BYTES_CFG4)
    m = measure_leg(torch, lambda: eng.stft_batch(pcm, out=out), args.leg_sustain_s)
    mean = m["mean_ms"]
    achieved = hops * ALGO_BYTES_CFG4 / (mean * 1e-3) / 1e9
    traffic = load_profile_json("hbm_traffic")
    name = KERNEL_NAMES.get(eng.info.stft_kernel, ("?", "?"))
    return {
        "workload": f"configs[3]: 16384-pt Hann STFT, hop 512, 8 interleaved channels, {hops} hop positions ({4 * hops} transforms)",
        "hop_positions_per_s": hops / (mean * 1e-3), "transforms_per_s": 4 * hops / (mean * 1e-3),
        **leg_times(m), "kernel": name[0], "output_bytes": hops * (C4 // 2) * (W4 - 1) * 8,
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            **fp32_fracs("config4", (C4 // 2) * NOMINAL_FLOP_16384, hops / (mean * 1e-3)),
            "traffic_over_algorithmic": ((traffic or {}).get("config4_bytes_per_hop") or 0) / ALGO_BYTES_CFG4 or None,
            "bytes_per_hop_position": ALGO_BYTES_CFG4, "hop_positions_per_launch": hops,
            "traffic": ((traffic or {}).get("config4_bytes_per_hop") or 0) * hops or None,
            "traffic_source": (traffic or {}).get("source"),
            "kernel": name[1],
            "first_allocation": first_allocation_of(torch, first, m, lambda buf: eng.stft_batch(pcm, out=buf), hops * ALGO_BYTES_CFG4, args),
            "placement": placement,
            "note": "launch = ONE kernel: the (l, r) pairs are read where they lie in the 8-channel stream (no de-interleave pass, no workspace)",
        },
    }


def config5_leg(args, torch, dist, eng, rank, world, backend, barrier, max_over_ranks, frame_range, stream_columns):
    """BASELINE config 5: config 3's pipeline over `--config5-frames` frames of one stream, frame-sharded over the
    ranks; every rank generates its own PCM on the device chunk by chunk (nothing is exchanged on the input side),
    renders the chunk and sends the finished 4 KB pixel columns to rank 0 (RCCL send / recv: world - 1 point-to-point
    flows, one per xGMI link into the root); the root checksums every piece and keeps nothing (1e8 columns = 410 GB)."""
    total, chunk = args.config5_frames, args.config5_chunk
    ranges = [frame_range(r, world, total) for r in range(world)]
    counts = [c for _, c in ranges]
    my_first = ranges[rank][0]
    dev = eng.device
    pcm_buf = torch.empty((chunk - 1) * H + W, dtype=torch.float32, device=dev)
    like = torch.empty((0, R, 4), dtype=torch.uint8, device=dev)
    acc = torch.zeros(1, dtype=torch.int64, device=dev)
    first_piece = {r: torch.zeros(1, dtype=torch.int64, device=dev) for r in range(world)}

    def render_range(f0, n, buf):
        ns = (n - 1) * H + W
        eng.white_noise(ns, first=f0 * H, out=pcm_buf)
        eng.render_batch(pcm_buf[:ns], max_frames=n, out=buf.view(-1, 1, R, 4))

    def produce(c0, n, buf):
        render_range(my_first + c0, n, buf)

    # --config5-probe: a few pieces of the run that counts, spread from the first frame of the stream to the last, leave 8 of their
    # columns behind (first, last, six between) for a host-side check against the oracle at far stream offsets
    n_probe = int(getattr(args, "config5_probe", 0) or 0)
    starts = sorted(ranges[r][0] + c0 for r in range(world) for c0 in range(0, counts[r], chunk))
    probe_at = {starts[(j * (len(starts) - 1)) // max(n_probe - 1, 1)] for j in range(n_probe)} if starts else set()
    probed = []      # (global frame index, [R][4] uint8 on the host)

    def consume(g0, piece):
        eng.checksum_add(piece, acc, base_word=g0 * R)
        for r in range(world):
            if g0 == ranges[r][0]:        # the first piece of rank r: kept apart for the sub-range check below
                eng.checksum_add(piece, first_piece[r], base_word=g0 * R)
        if g0 in probe_at:
            n = piece.shape[0]
            for j in sorted({(q * (n - 1)) // 7 for q in range(8)}):
                probed.append((g0 + j, piece[j].cpu().numpy().copy()))

    def run(produce_fn, consume_fn, send=True):
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        got = stream_columns(counts, chunk, produce_fn, consume_fn, like=like, dst=0, send=send)
        torch.cuda.synchronize()
        barrier()
        return time.perf_counter() - t0, got

    # ranks as the communicator sees them
    seen = [torch.zeros(1, dtype=torch.int64, device=dev if backend == "nccl" else "cpu") for _ in range(world)]
    dist.all_gather(seen, torch.tensor([rank], dtype=torch.int64, device=seen[0].device))
    ranks_seen = sorted(int(t[0]) for t in seen)

    # one small untimed round first: RCCL builds its point-to-point connections lazily on first use
    warm = min(1024, chunk)        # (the sample buffer and the rings hold `chunk` frames: never more per round)
    stream_columns([min(warm, c) for c in counts], warm, produce, lambda g0, p: None, like=like, dst=0)
    acc.zero_()
    if test_hook("BENCH_FAIL_RANK") == str(rank):                 # fault injection (tests/test_gpu_config5.py): a rank that dies inside the leg
        os._exit(7)
    t_over, arrived = run(produce, consume)                       # the run that counts: render + gather, overlapped
    t_comp, _ = run(produce, None, send=False)                    # render only
    t_xfer, _ = run(None, lambda g0, p: None)                     # gather only (re-sends the ring's last contents)
    t_over, t_comp, t_xfer = max_over_ranks([t_over, t_comp, t_xfer])
    # every source link on its own: a transfer-only pass in which only rank r sends (a bounded number of chunks), so that a
    # slow link shows up as itself and not inside an average
    per_source = []
    n_alone = min(8 * chunk, min(c for c in counts if c > 0))
    for r in range(1, world):
        solo = [n_alone if q == r and counts[q] > 0 else 0 for q in range(world)]
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        got_r = stream_columns(solo, chunk, None, lambda g0, p: None, like=like, dst=0)
        torch.cuda.synchronize()
        barrier()
        t_r = max_over_ranks([time.perf_counter() - t0])[0]
        per_source.append({"source": r, "bytes": int(n_alone) * R * 4 if solo[r] else 0, "GBps": (n_alone * R * 4 / t_r / 1e9) if solo[r] else None})
    out = None
    if rank == 0:
        # the sharded bytes equal a single GPU's on a sub-range: the root renders the first chunk of every other rank
        # itself and compares checksums with the piece that arrived from that rank
        ok, scratch = True, torch.empty((chunk, R, 4), dtype=torch.uint8, device=dev)
        for r in range(1, world):
            n = min(chunk, counts[r])
            if n == 0:
                continue
            render_range(ranges[r][0], n, scratch[:n])
            ok = ok and eng.checksum(scratch[:n], base_word=ranges[r][0] * R) == (int(first_piece[r][0]) & (2**64 - 1))
        overlap = (t_comp + t_xfer - t_over) / max(min(t_comp, t_xfer), 1e-9)
        out = {
            "workload": f"configs[4]: {total} frames of one white-noise stream frame-sharded over {world} GPUs "
                        f"({counts[0]} per GPU), PCM generated on device per {chunk}-frame chunk, fused PCM-to-RGBA kernel, "
                        f"pixel columns gathered to rank 0 chunk by chunk",
            "backend": backend, "ranks_seen": ranks_seen, "frames_total": total, "frames_per_gpu": counts,
            "chunk_columns": chunk, "rounds": (max(counts) + chunk - 1) // chunk,
            "frames_per_s": total / t_over, "algorithmic_GBps": total * ALGO_BYTES_PIXEL / t_over / 1e9,
            "overlapped_s": t_over, "render_only_s": t_comp, "gather_only_s": t_xfer,
            "overlap_ratio": max(0.0, min(1.0, overlap)),
            "gathered_bytes": arrived,
            "GBps_into_root": arrived / t_over / 1e9, "GBps_into_root_gather_only": arrived / t_xfer / 1e9,
            "GBps_per_source_link_gather_only": arrived / t_xfer / 1e9 / max(world - 1, 1),
            "GBps_per_source_alone": per_source,
            "root_consumed": "sgx_checksum_add over every piece (its own included), nothing kept",
            "checksum_all_columns": int(acc[0]) & (2**64 - 1),
            "sharded_equals_single_gpu_on_first_chunk_of_every_rank": bool(ok),
        }
        if n_probe:
            import numpy as np
            path = getattr(args, "config5_probe_file", None) or os.path.join(
                ROOT, "gpurun_out" if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "", "config5_probes.npz")
            np.savez(path, frames=np.array([t for t, _ in probed], np.int64),
                     rgba=np.stack([c for _, c in probed]) if probed else np.zeros((0, R, 4), np.uint8))
            out["probes"], out["probes_file"] = len(probed), os.path.relpath(path, ROOT)
    return out


def cpu_baseline_leg(args, eng, pcm):
    import numpy as np

    import oracle

    info = host_info()
    # the threads actually started: one per CPU this process may run on (its affinity mask), all of them -- the oracle's
    # frame loop is embarrassingly parallel and every thread owns its scratch
    cores = info["affinity"] or info["nproc"] or 1
    if info["cgroup_cpu_quota"]:                       # more threads than the quota's CPUs only take turns
        cores = max(1, min(cores, int(info["cgroup_cpu_quota"] + 0.5)))
    Fc = args.cpu_frames
    piece = 65_536                       # frames per oracle call: bounds host memory to ~1 GB of output
    host = oracle.white_noise((min(piece, Fc) - 1) * H + W)
    oracle.stream_process(host[:W + 64 * H], 1, W, H, threads=cores)  # plan + page-in
    cdt, done, ref = 0.0, 0, None
    while done < Fc:
        m = min(piece, Fc - done)
        host = oracle.white_noise((m - 1) * H + W, first=done * H)
        c0 = time.perf_counter()
        out = oracle.stream_process(host, 1, W, H, threads=cores)
        cdt += time.perf_counter() - c0
        if ref is None:
            ref = out[:64].copy()
        done += m
        del out
    got = eng.stft_batch(pcm, max_frames=64).cpu().numpy()[:, 0]
    peak = np.abs(ref[:, 0]).max(axis=(1, 2), keepdims=True)
    ok = bool((np.abs(got - ref[:, 0]) <= 2e-5 * np.maximum(np.abs(ref[:, 0]), 0.02 * peak)).all())
    cpu = {
        "value": Fc / cdt, "unit": "frames/s", "cores": cores, "kind": "port", "impl": "plain C (oracle/spectro_oracle.c)",
        "nproc": info["nproc"], "affinity": info["affinity"], "cgroup_cpu_quota": info["cgroup_cpu_quota"], "cpu_model": info["cpu_model"],
        "sample": f"first {Fc} frames of the same white-noise stream, float32 oracle (oracle/spectro_oracle.c), {cores} threads, "
                  f"{cdt * cores:.1f} thread-seconds",
        "parity_on_sample": ok,
    }
    # the same restatement on ONE thread (a short prefix of the sample: ~2 s)
    n1 = min(Fc, 8192)
    host = oracle.white_noise((n1 - 1) * H + W)
    c0 = time.perf_counter()
    oracle.stream_process(host, 1, W, H, threads=1)
    cpu["single_thread"] = {"value": n1 / (time.perf_counter() - c0), "unit": "frames/s", "frames": n1}
    # the reference's own FFT library, if this host has it (dlopen libfftw3f.so.3): "as written" (fft.rs:61,68,76:
    # per-frame allocations and cosf) and hoisted, all threads
    try:
        n_f = min(Fc, 65_536)
        host = oracle.white_noise((n_f - 1) * H + W)
        if oracle.fftw_available():
            fig = {}
            for key, aw in (("as_written", True), ("hoisted", False)):
                oracle.fftw_stream_process(host[:W + 64 * H], 1, W, H, threads=cores, as_written=aw)
                c0 = time.perf_counter()
                o = oracle.fftw_stream_process(host, 1, W, H, threads=cores, as_written=aw)
                fig[key] = n_f / (time.perf_counter() - c0)
                pk = np.abs(ref[:, 0]).max(axis=(1, 2), keepdims=True)
                fig[key + "_matches_oracle"] = bool((np.abs(o[:64, 0] - ref[:, 0]) <= 2e-5 * np.maximum(np.abs(ref[:, 0]), 0.02 * pk)).all())
            cpu["fftw"] = dict(fig, unit="frames/s", cores=cores, frames=n_f, plan="fftwf_plan_dft_1d(4096, FORWARD, MEASURE)")
        else:
            cpu["fftw"] = "not found (dlopen libfftw3f.so.3 failed on this host; the image ships no FFTW)"
    except Exception as e:  # noqa: BLE001
        cpu["fftw"] = {"error": f"{type(e).__name__}: {e}"}
    # the FFT call alone through an optimised library FFT (scipy's pocketfft, complex64, all host threads) ...
    try:
        cpu["library_fft"] = library_fft_rate(np, oracle, W, H, min(Fc, 32768), cores, ref)
    except Exception as e:  # noqa: BLE001
        cpu["library_fft"] = {"error": f"{type(e).__name__}: {e}"}
    # ... and the WHOLE frame loop around it: the fairer CPU baseline.  `value` is the faster of the two complete loops (the oracle's
    # plain-C port, a checker with a textbook recursive FFT, and this one); both are reported
    cpu["port"] = {"value": cpu["value"], "unit": "frames/s", "cores": cores, "sample": cpu["sample"], "parity_on_sample": ok, "impl": cpu["impl"]}
    try:
        lib = library_loop_rate(np, oracle, W, H, min(Fc, 131072), cores, ref)
        cpu["library"] = lib
        if lib["matches_oracle"] and lib["value"] > cpu["value"]:
            # still a PORT of the reference's frame loop (kind): a stand-in for its FFTW path, which this image cannot build
            cpu.update(value=lib["value"], kind="port", impl=lib["impl"],
                       sample=f"first {lib['frames']} frames of the same white-noise stream, fft.rs:47-99 in numpy + pocketfft c64, {cores} threads")
    except Exception as e:  # noqa: BLE001
        cpu["library"] = {"error": f"{type(e).__name__}: {e}"}
    return cpu


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return launch_children(args, argv)
    return main_rank(args)


if __name__ == "__main__":
    sys.exit(main())
