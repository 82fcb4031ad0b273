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


class RecordGather:
    """Pipelined gather of packed SiftPoint records to rank `dst` (BASELINE config 4).

    Per batch a rank hands over what misift_extract_batch_packed_async produced — the per-frame counts and the
    packed valid records, both still being computed on the GPU — with `post(slot, ...)`; one step later
    `complete(slot)` all-gathers the counts (B ints per rank), and every sender ships exactly its valid bytes
    to `dst` with one point-to-point message (xGMI is a full mesh: 7 senders use 7 distinct links into rank 0).
    On a GPU everything here runs on a separate communication stream that waits for the batch's `ready` event,
    so the transfer of batch k overlaps the extraction of batch k+1 (a 64-frame batch is ~77 MB per sender:
    ~1.5 ms on one xGMI link — as long as the extraction itself); `free_event(slot)` tells the compute stream
    when the slot's buffers may be overwritten.  Backend-agnostic: with gloo / CPU tensors there are no streams
    and the same logic runs synchronously (tests/test_dist_cpu.py)."""

    def __init__(self, dist, torch, rank, world, device, dst=0, nslots=2, force_collectives=False):
        self.dist, self.torch = dist, torch
        self.force = force_collectives          # run the count all-gather even with one rank (single-GPU self-test)
        self.rank, self.world, self.device, self.dst = rank, world, device, dst
        self.cuda = device.type == "cuda"
        # high priority: the small collective / copy kernels must not queue behind a saturating extraction launch
        self.comm = torch.cuda.Stream(device=device, priority=-1) if self.cuda else None
        self.slots = [dict(counts=None, packed=None, ready=None, free=None) for _ in range(nslots)]
        self.recv = {}

    def post(self, slot, counts, packed, ready_event=None):
        """counts: int32 tensor [B] (-1 = overflowed frame, counted as 0 records); packed: uint8 tensor whose
        first sum(counts)*576 bytes are the records; ready_event: recorded on the compute stream after packing."""
        s = self.slots[slot]
        s["counts"], s["packed"], s["ready"] = counts, packed, ready_event

    def free_event(self, slot):
        return self.slots[slot]["free"]

    def _recv_buf(self, slot, r, nbytes):
        key = (slot, r)
        buf = self.recv.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = self.torch.empty((max(nbytes, 1) * 5 // 4,), dtype=self.torch.uint8, device=self.device)
            self.recv[key] = buf
        return buf[:nbytes]

    def _run(self, slot):
        torch, dist = self.torch, self.dist
        s = self.slots[slot]
        if self.cuda and s["ready"] is not None:
            self.comm.wait_event(s["ready"])
        cnt = s["counts"].to(torch.int32)
        if self.world > 1 or self.force:
            gathered = [torch.empty_like(cnt) for _ in range(self.world)]
            dist.all_gather(gathered, cnt)
            all_counts = torch.stack(gathered).cpu().numpy()
        else:
            all_counts = cnt.cpu().numpy()[None, :]
        nbytes = np.clip(all_counts, 0, None).sum(axis=1).astype(np.int64) * RECORD_BYTES
        out = None
        if self.rank == self.dst:
            out, ops = [], []
            for r in range(self.world):
                if r == self.dst:
                    out.append(s["packed"][: int(nbytes[r])])
                    continue
                buf = self._recv_buf(slot, r, int(nbytes[r]))
                out.append(buf)
                if nbytes[r]:
                    ops.append(dist.P2POp(dist.irecv, buf, r))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
        elif nbytes[self.rank]:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, s["packed"][: int(nbytes[self.rank])], self.dst)]):
                w.wait()
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(self.comm)
            s["free"] = ev
        return all_counts, out

    def complete(self, slot):
        """Returns (all_counts [world, B] numpy, per-rank packed byte tensors on `dst` / None elsewhere)."""
        if self.cuda:
            with self.torch.cuda.stream(self.comm):
                return self._run(slot)
        return self._run(slot)
