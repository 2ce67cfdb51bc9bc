#!/usr/bin/env python3
"""bench.py -- STFT frames/s (4096-point, hop 256) on N MI355X, with roofline and CPU baseline.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic PCM that is already resident in
HBM: BASELINE config 2 -- 1e6 frames of mono white noise (256 001 792 samples), W 2048 / P 4096 /
H 256, output [1e6][2047][2] float32 magnitudes.  With N GPUs every rank transforms its own
1e6-frame shard of one long stream (contiguous frame ranges, sample offset rank * 1e6 * H; no
data-path collective: the path shards by frame) => weak scaling; value = N * 1e6 * K / t.

Extra objects on the same line:
  roofline     -- dominant kernel (the STFT kernel): algorithmic bytes per launch / measured
                  launch duration (HIP events on the launch stream) against the 8 TB/s HBM peak
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference's algorithm; the reference itself
                  is Rust + FFTW and cannot be built in this image) timed on this host's cores, on a
                  bounded sample of the same stream (rank 0, N = 1 only)
  pixel_path   -- BASELINE config 3 (N = 1) / config 5 shape (N > 1): PCM -> RGBA columns, and for
                  N > 1 the RCCL gather of pixel columns to rank 0 (reported, not the headline value)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
W, H, M, R = 2048, 256, 2047, 1024
ALGO_BYTES_STFT = H * 1 * 4 + M * 2 * 4  # 17 400 B / frame: each input sample once, each output byte once
ALGO_BYTES_PIXEL = H * 1 * 4 + R * 4     # 5 120 B / frame
KERNEL_NAMES = {
    0: ("generic power-of-two (workgroup per frame, LDS radix-4)", "sgx::stft_generic_kernel"),
    1: ("stft4096 wave-per-transform", "sgx::stft4096_kernel<6, true>"),
    2: ("stft4096 workgroup-per-transform (256 threads x 16 points, radix-16 x3, mono frame pairs), scalar codelets",
        "sgx::wg::stft4096_wg_kernel<true, 0, false, false>"),
    3: ("stft4096 workgroup-per-transform (256 threads x 16 points, radix-16 x3, mono frame pairs), packed (re, im) codelets",
        "sgx::wgp::stft4096_wgp_kernel<true, 0, false, false>"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=1_000_000, help="frames per GPU per step (config 2: 1e6)")
    ap.add_argument("--pixel-frames", type=int, default=262_144, help="frames per GPU for the pixel-path leg (0 = skip)")
    ap.add_argument("--pixel-timeout", type=float, default=300.0, help="seconds after which a stalled pixel-path leg is given up")
    ap.add_argument("--cpu-frames", type=int, default=262_144, help="frames of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--generic", action="store_true", help="force the generic power-of-two kernel")
    return ap.parse_args()


def load_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/hbm_traffic.json), or None."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")  # the latest round's copy (profiles/rNN_hbm_traffic.json)
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def library_fft_rate(np, oracle, W, H, frames, cores, ref):
    """frames/s of fft.rs:43-99 done with scipy.fft on the host (bounded sample; checked against the oracle's first
    frames): the whole frame (numpy window / pack / split around the FFT) and the FFT call alone"""
    import scipy.fft

    win = oracle.hann_window(W)
    batch, done, dt, dt_fft, first = 4096, 0, 0.0, 0.0, None
    z = np.zeros((batch, 2 * W), np.complex64)                    # the padding half stays zero (out-of-place FFT)
    while done < frames:
        m = min(batch, frames - done)
        host = oracle.white_noise((m - 1) * H + W, first=done * H)
        t0 = time.perf_counter()
        fr = np.lib.stride_tricks.as_strided(host, shape=(m, W), strides=(host.strides[0] * H, host.strides[0]))
        sw = fr * win
        z.real[:m, :W] = sw                                       # mono -> (s, s): l + i r
        z.imag[:m, :W] = sw
        t1 = time.perf_counter()
        F = scipy.fft.fft(z[:m], axis=1, workers=cores)
        dt_fft += time.perf_counter() - t1
        a, b = F[:, 1:W], F[:, 2 * W - 1:W:-1]                    # F[k], F[P - k], k = 1 .. W-1
        out = np.stack([np.abs(a + np.conj(b)), np.abs(a - np.conj(b))], axis=2) * np.float32(1.0 / W)
        dt += time.perf_counter() - t0
        if first is None:
            first = out[:8].copy()
        done += m
    peak = np.abs(ref[:8, 0]).max(axis=(1, 2), keepdims=True)
    ok = bool((np.abs(first - ref[:8, 0]) <= 2e-5 * np.maximum(np.abs(ref[:8, 0]), 0.05 * peak)).all())
    return {"value": frames / dt, "fft_call_only": frames / dt_fft, "unit": "frames/s", "cores": cores, "frames": frames,
            "what": "scipy.fft (pocketfft) complex64 on all cores; numpy (one thread) does window / pack / split", "matches_oracle": ok}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback"
    # rehearsal knobs (one-GPU box): BENCH_BACKEND=gloo BENCH_SINGLE_DEVICE=1 run every rank on cuda:0 so
    # that the N > 1 control flow can be exercised without a second GPU; the driver never sets them
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if os.environ.get("BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from spectrogram_rs_amd import SpectrogramEngine
    from spectrogram_rs_amd.sharding import chunks, frame_range, gather_columns, sample_range

    def barrier():
        if world > 1:
            dist.barrier()

    F = args.frames
    eng = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=1, device=local_rank,
                            force_generic=args.generic, interp=1, gradient="viridis")
    # weak scaling: one stream of world*F frames; rank g owns the contiguous range frame_range(g)
    # and generates exactly its samples (with the W-H halo) -- no input exchange
    first_frame, n_own = frame_range(rank, world, world * F)
    assert n_own == F
    first_sample, n_samples = sample_range(first_frame, F, W, H)
    pcm = eng.white_noise(n_samples, first=first_sample)
    mags = torch.empty((F, 1, M, 2), dtype=torch.float32, device=eng.device)
    mags.zero_()  # first touch of the 16 GB output buffer belongs to the allocation, not to a step

    for _ in range(args.warmup):
        eng.stft_batch(pcm, out=mags)
    torch.cuda.synchronize()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        eng.stft_batch(pcm, out=mags)
        b.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / max(len(evs), 1)
    red_dev = eng.device if backend == "nccl" else "cpu"
    t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms = float(t[0]), float(t[1])
    checksum = eng.checksum(mags[:4096])

    def build_line(pixel, cpu):
        total_frames = world * F * args.steps
        value = total_frames / elapsed
        achieved = F * ALGO_BYTES_STFT / (kernel_ms * 1e-3) / 1e9
        traffic = load_traffic()
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
                "kernel": KERNEL_NAMES[eng.info.stft_kernel][0],
                "sharding": "contiguous frame ranges per rank, no data-path collective" if world > 1 else "single GPU",
            },
            "achieved_GBps_algorithmic": value * ALGO_BYTES_STFT / 1e9,
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": ((traffic or {}).get("stft_bytes_per_frame") or 0) * F or None,
                "kernel": KERNEL_NAMES[eng.info.stft_kernel][1],
                "launch_ms": kernel_ms, "bytes_per_frame": ALGO_BYTES_STFT, "frames_per_launch": F,
            },
            "cpu_baseline": cpu,
            "pixel_path": pixel,
            "checksum_first_4096_frames": checksum,
        }
        return line

    def pixel_leg():
        if args.pixel_frames <= 0:
            return None
        Fp = min(args.pixel_frames, F)
        chunk = min(65_536, Fp)
        rgba = torch.empty((Fp, 1, R, 4), dtype=torch.uint8, device=eng.device)
        counts = [Fp] * world
        root_seen = [0]

        def consume(first_col, piece):
            # rank 0 consumes every gathered piece (here: touches it) instead of materialising the
            # whole image -- 1e8 columns would be 410 GB, more than one GPU's HBM
            root_seen[0] += int(piece.shape[0])

        def render(c0, cn):
            eng.render_batch(pcm, first_frame=c0, max_frames=cn, out=rgba[c0:c0 + cn])

        def pixel_pass():
            if world == 1:
                for c0, cn in chunks(Fp, chunk):
                    render(c0, cn)
            else:
                # the one exchange step of the path: finished pixel columns to rank 0 (RCCL over xGMI), world-1
                # concurrent point-to-point flows per chunk; chunk i's transfer overlaps chunk i+1's kernel
                gather_columns(rgba[:, 0], counts, dst=0, chunk=chunk, consume=consume, produce=render)

        pixel_pass()
        torch.cuda.synchronize()
        barrier()
        tp0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            pixel_pass()
        torch.cuda.synchronize()
        barrier()
        tp = torch.tensor([time.perf_counter() - tp0], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        fps = world * Fp * reps / float(tp[0])
        pixel = {
            "workload": ("config 3: " if world == 1 else "config 5 shape: ") + f"{Fp} frames/GPU -> 1024 log rows (cosine), Viridis RGBA"
                        + ("" if world == 1 else f", RCCL gather of {chunk}-column chunks to rank 0"),
            "frames_per_s": fps,
            "algorithmic_GBps": fps * ALGO_BYTES_PIXEL / 1e9,
            "gathered": world > 1,
        }
        return pixel

    # ---- pixel path leg (config 3 / config 5 shape), not the headline value -----------------------
    # The headline number above is already final.  The leg below contains the one collective exchange of the
    # path; if it fails or stalls on some rank (a collective cannot be interrupted from Python), a watchdog
    # still prints the JSON line -- with the failure recorded in "pixel_path" -- and ends the process.
    pixel = None
    import threading

    def on_stall():
        if rank == 0:
            print(json.dumps(build_line({"error": f"pixel-path leg did not finish within {args.pixel_timeout} s"}, None)), flush=True)
        os._exit(0)

    watchdog = threading.Timer(args.pixel_timeout, on_stall)
    watchdog.daemon = True
    if args.pixel_frames > 0:
        watchdog.start()
    try:
        pixel = pixel_leg()
    except Exception as e:  # noqa: BLE001 -- reported in the line, the headline stands
        pixel = {"error": f"{type(e).__name__}: {e}"}
        if world > 1:      # the other ranks may be waiting in the exchange: nothing collective from here on
            if rank == 0:
                print(json.dumps(build_line(pixel, None)), flush=True)
            os._exit(0)
    watchdog.cancel()
    # ---- CPU baseline: the oracle on this host's cores (rank 0, N = 1 only) ------------------------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        import numpy as np

        import oracle

        cores = min(os.cpu_count() or 1, 16)
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
        got = mags[:64, 0].cpu().numpy()
        peak = np.abs(ref[:, 0]).max(axis=(1, 2), keepdims=True)
        ok = bool((np.abs(got - ref[:, 0]) <= 2e-5 * np.maximum(np.abs(ref[:, 0]), 0.05 * peak)).all())
        cpu = {
            "value": Fc / cdt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"first {Fc} frames of the same white-noise stream, float32 oracle (oracle/spectro_oracle.c), {cores} threads, "
                      f"{cdt * cores:.1f} thread-seconds",
            "parity_on_sample": ok,
        }
        # a second CPU figure for orientation: the same frames through an optimised library FFT (scipy's
        # pocketfft, complex64, all cores) with numpy doing window, pack and split -- the nearest thing to the
        # reference's FFTW-backed path that this image holds (no libfftw3f here or on the GPU box)
        try:
            cpu["library_fft"] = library_fft_rate(np, oracle, W, H, min(Fc, 32768), cores, ref)
        except Exception as e:  # noqa: BLE001
            cpu["library_fft"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        print(json.dumps(build_line(pixel, cpu)), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
