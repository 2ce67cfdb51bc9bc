"""Frame sharding across the GPUs of a node, and the one exchange step of the path.

The reference is a single-process desktop app (no collectives).  The hop loop
(src/fourier/audio_transform.rs:34-42) makes frame t depend only on samples [t*H, t*H + W), so a
long stream shards embarrassingly by contiguous frame ranges; neighbouring ranks share a halo of
W - H samples of *input*, never any intermediate.  The only exchange is the gather of finished
pixel columns (4 KB each) to the rank that owns the image.

One process per GPU; `torch.distributed` with backend "nccl" (= RCCL over xGMI) on MI355X, "gloo"
in the CPU tests.  Nothing here computes: it only says who owns what and moves finished bytes.
"""
from __future__ import annotations

from typing import Callable, Iterator, List, Optional, Tuple


def frame_range(rank: int, world: int, total_frames: int, granule: int = 2) -> Tuple[int, int]:
    """Contiguous frame range [first, first + count) owned by `rank`.  Boundaries fall on multiples of
    `granule` frames (default 2: a mono transform carries the frame pair (2j, 2j+1), so even
    boundaries make every rank compute exactly the transforms a single GPU would -- same bytes)."""
    units = (total_frames + granule - 1) // granule
    base, extra = divmod(units, world)
    u0 = rank * base + min(rank, extra)
    u1 = u0 + base + (1 if rank < extra else 0)
    first = min(u0 * granule, total_frames)
    return first, min(u1 * granule, total_frames) - first


def sample_range(first_frame: int, n_frames: int, W: int, H: int) -> Tuple[int, int]:
    """Samples (per channel) a rank must hold for its frames: [first*H, (first+n-1)*H + W) --
    the W - H sample halo is shared with the next rank."""
    if n_frames <= 0:
        return first_frame * H, 0
    return first_frame * H, (n_frames - 1) * H + W


def chunks(n: int, chunk: int) -> Iterator[Tuple[int, int]]:
    for c0 in range(0, n, chunk):
        yield c0, min(chunk, n - c0)


def _post_round(mine, n_per_rank: List[int], dst: int, staged: bool, like, group):
    """Post one round of the gather without waiting: the root posts one receive per contributing rank, every other
    rank one send, as ONE batch (RCCL: one grouped launch, world - 1 concurrent point-to-point flows, one per
    xGMI link into the root).  Returns (pieces, requests); pieces[r] is None where rank r sends nothing."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    pieces, ops = [None] * world, []
    if rank == dst:
        for r in range(world):
            if n_per_rank[r] == 0:
                continue
            if r == dst:
                pieces[r] = mine
                continue
            pieces[r] = torch.empty((n_per_rank[r],) + tuple(like.shape[1:]), dtype=like.dtype,
                                    device="cpu" if staged else like.device)
            ops.append(dist.P2POp(dist.irecv, pieces[r], r, group))
    elif n_per_rank[rank] > 0:
        ops.append(dist.P2POp(dist.isend, mine.cpu() if staged else mine.contiguous(), dst, group))
    return pieces, (dist.batch_isend_irecv(ops) if ops else [])


def _finish_round(pieces, reqs, staged: bool, device):
    for q in reqs:
        q.wait()  # RCCL: orders the current stream after the transfer; gloo: blocks the host
    if staged:
        pieces = [None if p is None else (p if p.device == device else p.to(device)) for p in pieces]
    return pieces


def gather_columns(local, counts: List[int], dst: int = 0, chunk: int = 65536,
                   consume: Optional[Callable] = None, group=None, produce: Optional[Callable] = None):
    """Gather pixel columns [n_local][...] from every rank to `dst`, in chunks of at most `chunk`
    columns per rank so that the root never holds more than world * chunk columns at once (1e8
    columns are 410 GB: more than one GPU's HBM).  `counts[r]` is rank r's column count (known from
    frame_range).  On the root, `consume(global_first_column, tensor)` is called for every piece in
    stream order per rank; without `consume` the pieces are concatenated and returned (small runs).

    `produce(c0, n)`, if given, is called before round i's columns local[c0:c0+n] are sent: the caller
    renders them there.  Round i's transfer is posted and round i+1 is produced before round i is
    waited for, so on MI355X the xGMI transfer of one chunk overlaps the kernel of the next.

    Traffic shape on MI355X: world - 1 independent point-to-point flows into the root, one per
    xGMI link, posted as one batch per round -- per-link bound, no ring, no reduction."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    assert len(counts) == world and counts[rank] == local.shape[0]
    # gloo moves host memory only: stage device tensors through the host there (rehearsals, CPU tests);
    # with nccl (RCCL) the pieces go GPU to GPU over xGMI
    staged = local.is_cuda and dist.get_backend(group) != "nccl"
    starts = [sum(counts[:r]) for r in range(world)]
    rounds = max((c + chunk - 1) // chunk for c in counts) if counts else 0
    kept = [[] for _ in range(world)]

    def post(i):
        c0 = i * chunk
        n_per_rank = [max(min(chunk, counts[r] - c0), 0) for r in range(world)]
        if produce is not None and n_per_rank[rank] > 0:
            produce(c0, n_per_rank[rank])
        return c0, _post_round(local[c0:c0 + chunk], n_per_rank, dst, staged, local, group)

    pending = post(0) if rounds else None
    for i in range(rounds):
        c0, (pieces, reqs) = pending
        pending = post(i + 1) if i + 1 < rounds else None
        pieces = _finish_round(pieces, reqs, staged, local.device)
        if rank != dst:
            continue
        for r in range(world):
            if pieces[r] is None:
                continue
            if consume is not None:
                consume(starts[r] + c0, pieces[r])
            else:
                kept[r].append(pieces[r].clone() if r == dst else pieces[r])
    if rank == dst and consume is None:
        flat = [p for r in range(world) for p in kept[r]]
        return torch.cat(flat) if flat else local[:0]
    return None


def stream_columns(counts: List[int], chunk: int, produce: Optional[Callable], consume: Optional[Callable], *,
                   like, dst: int = 0, group=None, slots: int = 2, send: bool = True):
    """The gather of `gather_columns` for runs whose columns are never all resident: rank r renders its counts[r]
    columns chunk by chunk into a ring of `slots` buffers (`produce(c0, n, buf)` fills buf[:n] with its columns
    [c0, c0 + n)) and sends each chunk to `dst`, which hands every piece -- its own included -- to
    `consume(global_first_column, tensor)` and keeps nothing.  BASELINE config 5: 1e8 columns = 410 GB, more than one
    GPU's HBM, so neither the producers nor the root ever hold more than `slots` chunks per peer.

    Round i's transfers are posted, round i + 1 is produced, then round i is waited for: with RCCL the xGMI
    transfer of one chunk overlaps the kernel of the next (world - 1 concurrent point-to-point flows into the
    root, one per link; no ring, no reduction).  `produce=None` re-sends whatever the ring holds (transfer only),
    `send=False` only produces (compute only): the two legs `bench.py` times beside the overlapped run.

    `like`: a tensor giving dtype / device / trailing shape of one column block, e.g. torch.empty((0, R, 4), uint8).
    Returns the number of bytes that arrived at `dst` from other ranks (on `dst`; 0 elsewhere)."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    assert len(counts) == world and slots >= 2
    staged = like.is_cuda and dist.get_backend(group) != "nccl"
    starts = [sum(counts[:r]) for r in range(world)]
    rounds = max((c + chunk - 1) // chunk for c in counts) if counts else 0
    tail = tuple(like.shape[1:])
    ring = [torch.empty((chunk,) + tail, dtype=like.dtype, device=like.device) for _ in range(slots)]
    # the root's receive buffers: one ring per peer, allocated once (a fresh allocation per round would put the
    # caching allocator on the critical path of every chunk)
    recv = None
    if rank == dst and send:
        recv = [[None if r == dst or counts[r] == 0 else
                 torch.empty((chunk,) + tail, dtype=like.dtype, device="cpu" if staged else like.device)
                 for r in range(world)] for _ in range(slots)]
    arrived = 0

    def post(i):
        c0 = i * chunk
        n_per_rank = [max(min(chunk, counts[r] - c0), 0) for r in range(world)]
        mine = ring[i % slots]
        if produce is not None and n_per_rank[rank] > 0:
            produce(c0, n_per_rank[rank], mine)
        pieces, ops = [None] * world, []
        if not send:
            pieces[rank] = mine[:n_per_rank[rank]] if n_per_rank[rank] else None
            return c0, pieces, []
        if rank == dst:
            for r in range(world):
                if n_per_rank[r] == 0:
                    continue
                if r == dst:
                    pieces[r] = mine[:n_per_rank[r]]
                    continue
                pieces[r] = recv[i % slots][r][:n_per_rank[r]]
                ops.append(dist.P2POp(dist.irecv, pieces[r], r, group))
        elif n_per_rank[rank] > 0:
            ops.append(dist.P2POp(dist.isend, mine[:n_per_rank[rank]].cpu() if staged else mine[:n_per_rank[rank]], dst, group))
        return c0, pieces, (dist.batch_isend_irecv(ops) if ops else [])

    pending = post(0) if rounds else None
    for i in range(rounds):
        c0, pieces, reqs = pending
        pending = post(i + 1) if i + 1 < rounds else None
        pieces = _finish_round(pieces, reqs, staged, like.device)
        if rank != dst or consume is None:
            continue
        for r in range(world):
            if pieces[r] is None:
                continue
            if r != dst:
                arrived += pieces[r].numel() * pieces[r].element_size()
            consume(starts[r] + c0, pieces[r])
    return arrived


def combine_checksums(parts: List[int]) -> int:
    """sgx_checksum is a sum of per-word mixes with global word indices: shards add (mod 2^64)."""
    return sum(parts) & ((1 << 64) - 1)
