"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

The extraction path shards embarrassingly: frames are independent, each rank runs its own batch and
no collective touches the data path.  The only exchange is AFTER compute: the gather of SiftData to
rank 0 (BASELINE config 4) — an all-gather of the per-frame counts (B ints per rank) followed by one
point-to-point message per sender carrying exactly the valid 576-byte records, packed contiguously.
xGMI is a full mesh, so the 7 senders use 7 distinct links into rank 0 concurrently.

The matcher shards by row blocks of set 1; set 2 starts sharded the same way and is replicated with one
all-gather before the sweep (BASELINE config 5) — see bench.py.

Backend-agnostic (tensors may live on the CPU with gloo), so the logic is covered by world_size-2
gloo tests in tests/test_dist_cpu.py.
"""
import numpy as np

RECORD_BYTES = 576


def shard_range(total, rank, world):
    """Contiguous block partition of `total` units (frames / rows): first (total % world) ranks get one more."""
    q, r = divmod(total, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def pack_valid_records(torch, pts2d, counts):
    """pts2d: [B, max_pts*576] uint8 (frame f's records at row f); counts[f] valid records each.
    Returns one contiguous uint8 tensor holding only the valid records, frame after frame."""
    parts = [pts2d[f, : int(counts[f]) * RECORD_BYTES] for f in range(pts2d.shape[0]) if int(counts[f]) > 0]
    if not parts:
        return pts2d.new_empty((0,))
    return torch.cat(parts)


def gather_sift_records(dist, torch, pts2d, counts, rank, world, device, dst=0):
    """Gather the valid SiftPoint records of every rank's batch on rank `dst`.

    Returns (all_counts [world, B] int32 numpy, list of per-rank packed uint8 tensors) on `dst`,
    (all_counts, None) elsewhere."""
    B = pts2d.shape[0]
    cnt = torch.as_tensor(np.ascontiguousarray(counts, dtype=np.int32)).to(device)
    gathered = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(gathered, cnt)
    all_counts = torch.stack(gathered).cpu().numpy()
    packed = pack_valid_records(torch, pts2d, counts)
    if rank == dst:
        bufs = []
        ops = []
        for r in range(world):
            nbytes = int(all_counts[r].sum()) * RECORD_BYTES
            if r == dst:
                bufs.append(packed)
                continue
            buf = torch.empty((nbytes,), dtype=torch.uint8, device=device)
            bufs.append(buf)
            if nbytes:
                ops.append(dist.P2POp(dist.irecv, buf, r))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return all_counts, bufs
    if packed.numel():
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, packed, dst)]):
            w.wait()
    return all_counts, None


def unpack_records(packed, counts_row):
    """Split one rank's packed byte tensor back into per-frame [n_f, 576] views."""
    out, off = [], 0
    for n in counts_row:
        nb = int(n) * RECORD_BYTES
        out.append(packed[off:off + nb].view(int(n), RECORD_BYTES) if nb else packed[:0].view(0, RECORD_BYTES))
        off += nb
    return out
