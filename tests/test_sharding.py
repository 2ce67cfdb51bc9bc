"""N > 1 path on CPU: gloo, world sizes 2, 3 and 8.  Each rank computes the pixel columns of its own frame
range (with the CPU oracle standing in for the GPU) from its own halo-padded slice of the stream;
the gathered image must equal the single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest

from spectrogram_rs_amd.sharding import chunks, combine_checksums, frame_range, sample_range

W, H, SR, R = 256, 32, 48000, 64


def test_frame_and_sample_ranges_cover_exactly():
    for total in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            got = [frame_range(r, world, total) for r in range(world)]
            assert got[0][0] == 0 and sum(c for _, c in got) == total
            for (f0, c0), (f1, _) in zip(got, got[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in got) - min(c for _, c in got) <= 3
            assert all(f % 2 == 0 for f, c in got if c)  # even starts: mono frame pairs never straddle ranks
    assert sample_range(10, 5, 2048, 256) == (2560, 4 * 256 + 2048)
    assert sample_range(3, 0, 2048, 256)[1] == 0
    # neighbours share exactly W - H samples
    a, b = sample_range(0, 100, 2048, 256), sample_range(100, 100, 2048, 256)
    assert a[0] + a[1] - b[0] == 2048 - 256
    assert list(chunks(10, 4)) == [(0, 4), (4, 4), (8, 2)]
    assert combine_checksums([2**64 - 1, 2]) == 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_frames, out_path):
    import torch
    import torch.distributed as dist

    import oracle
    from spectrogram_rs_amd.sharding import chunks, frame_range, gather_columns, sample_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grad = np.load(os.path.join(os.path.dirname(__file__), "golden", "gradients.npz"))["viridis"]
    first, count = frame_range(rank, world, total_frames)
    s0, n = sample_range(first, count, W, H)
    pcm = oracle.white_noise(n, first=s0) * np.float32(0.05)   # the rank's own slice (+ halo) of ONE stream
    mags = oracle.stream_process(pcm, 1, W, H)[:, 0]
    assert mags.shape[0] == count
    cols = torch.from_numpy(oracle.render_columns(mags, SR, grad, R=R, f_min=200.0, f_max=20000.0)
                            if count else np.zeros((0, R, 4), np.uint8))
    counts = [frame_range(r, world, total_frames)[1] for r in range(world)]
    # (a) concatenating gather, deliberately tiny chunks so several rounds happen
    img = gather_columns(cols, counts, dst=0, chunk=5)
    # (b) streaming gather: the root only ever sees one piece at a time
    seen = {}
    gather_columns(cols, counts, dst=0, chunk=4, consume=lambda g0, t: seen.__setitem__(g0, t.clone()))
    # (c) pipelined gather: the columns of a round are produced (rendered) only when that round is posted, one
    #     round ahead of the round being waited for
    lazy, calls = torch.zeros_like(cols), []

    def produce(c0, n_cols):
        calls.append((c0, n_cols))
        lazy[c0:c0 + n_cols] = cols[c0:c0 + n_cols]

    img2 = gather_columns(lazy, counts, dst=0, chunk=3, produce=produce)
    assert calls == list(chunks(count, 3))
    # (d) the config-5 shape: columns exist one chunk at a time on every rank (a ring of two buffers), the root consumes
    #     every piece and keeps nothing; then the transfer-only and compute-only legs the bench times beside it
    from spectrogram_rs_amd.sharding import stream_columns

    got_stream, made = {}, []

    def fill(c0, n_cols, buf):
        made.append((c0, n_cols))
        buf[:n_cols] = cols[c0:c0 + n_cols]

    like = torch.empty((0, R, 4), dtype=torch.uint8)
    arrived = stream_columns(counts, 3, fill, lambda g0, t: got_stream.__setitem__(g0, t.clone()), like=like, dst=0)
    assert made == list(chunks(count, 3))
    assert stream_columns(counts, 3, None, lambda g0, t: None, like=like, dst=0) == arrived      # transfer only
    assert stream_columns(counts, 3, fill, None, like=like, dst=0, send=False) == 0                # compute only
    if rank == 0:
        assert arrived == (total_frames - count) * R * 4
        assert torch.equal(torch.cat([got_stream[k] for k in sorted(got_stream)]), img)
    if rank == 0:
        assert img.shape[0] == total_frames
        pieces = torch.cat([seen[k] for k in sorted(seen)])
        assert torch.equal(pieces, img) and torch.equal(img2, img)
        np.save(out_path, img.numpy())
    dist.barrier()
    dist.destroy_process_group()


# (8, 45): 23 frame pairs over 8 ranks -> seven ranks of 6 frames (two rounds of the 5-column gather) and a last one of 3 (one
# round): uneven counts, a total not divisible by 16, one rank with fewer rounds than the others.  (8, 5): five ranks empty.
@pytest.mark.parametrize("world,total", [(2, 23), (2, 1), (3, 10), (8, 45), (8, 5)])
def test_gloo_sharded_columns_equal_single_process(tmp_path, world, total):
    import torch.multiprocessing as mp

    import oracle

    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), total, out), nprocs=world, join=True)
    got = np.load(out)
    grad = np.load(os.path.join(os.path.dirname(__file__), "golden", "gradients.npz"))["viridis"]
    n = (total - 1) * H + W
    mags = oracle.stream_process(oracle.white_noise(n) * np.float32(0.05), 1, W, H)[:, 0]
    ref = oracle.render_columns(mags, SR, grad, R=R, f_min=200.0, f_max=20000.0)
    assert got.shape == ref.shape and np.array_equal(got, ref)
