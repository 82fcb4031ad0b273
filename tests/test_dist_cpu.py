"""Multi-process CPU tests (gloo, world_size 2 / 3) of the multi-GPU exchange.  Since r04 the gather of SiftData is
libmisift.so's OWN code: a HOST communicator (misift_comm_create_host, include/misift.h) runs misift_gather_post /
misift_gather_complete — count staging, per-rank record counts and offsets, root placement, the -1 frames, the collective
ENOMEM decision — on host memory, and the five transport primitives are torch.distributed (gloo) calls made from the
callbacks below.  (r01-r03 tested a Python mirror of that logic, cudasift_amd/dist.py, which could drift from the C++;
it is gone.)  The row-block matcher split of BASELINE config 5 is exercised with gloo collectives and the CPU oracle as
the matcher, through bench.py's own matcher leg."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

RECORD_BYTES = 576


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def shard_range(total, rank, world):
    """Contiguous block partition of `total` units (frames / rows): the first (total % world) ranks get one more."""
    q, r = divmod(total, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def gloo_host_comm(capi, dist, torch, rank, world):
    """capi.HostComm whose primitives are gloo collectives / point-to-point messages on the caller's host buffers."""
    pending = []

    def view(ptr, nbytes):
        return torch.frombuffer((C.c_char * nbytes).from_address(ptr), dtype=torch.uint8)

    def allgather(send, recv, nbytes):
        if nbytes == 0:
            return
        outs = [view(recv + r * nbytes, nbytes) for r in range(world)]
        dist.all_gather(outs, view(send, nbytes).clone())          # (in-place contributions: the source is copied first)

    def send(ptr, nbytes, peer):
        pending.append(dist.P2POp(dist.isend, view(ptr, nbytes), peer))

    def recv(ptr, nbytes, peer):
        pending.append(dist.P2POp(dist.irecv, view(ptr, nbytes), peer))

    def group_end():
        if pending:
            for w in dist.batch_isend_irecv(list(pending)):
                w.wait()
        pending.clear()

    return capi.HostComm(world, rank, allgather, send, recv, group_end)


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from cudasift_amd import capi
    from oracle import pyoracle as orc
    from synth import descriptors_to_points, synth_descriptors
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = gloo_host_comm(capi, dist, torch, rank, world)
    ok = comm.rank == rank and comm.size == world
    comm.barrier()
    # ---- frames shard: every rank owns B frames with different valid counts; gather on root 1 (not 0)
    B, max_pts, root = 3, 16, world - 1

    def batch(r):
        rr = np.random.default_rng(100 + r)
        c = rr.integers(0, max_pts + 1, size=B).astype(np.int32)
        c[r % B] = 0                                        # an empty frame
        body = rr.integers(0, 255, size=int(c.sum()) * RECORD_BYTES, dtype=np.uint8)
        return c, body

    counts, body = batch(rank)
    packed = np.zeros(B * max_pts * RECORD_BYTES, np.uint8)
    packed[: body.size] = body
    cap = world * B * max_pts
    recv = np.zeros(cap * RECORD_BYTES, np.uint8) if rank == root else None
    comm.gather_post(0, counts, B, packed)
    all_counts, offs = comm.gather_complete(0, B, root, recv, cap)
    for r in range(world):
        c, b = batch(r)
        ok &= bool(np.array_equal(all_counts[r], c))
        ok &= int(offs[r + 1] - offs[r]) == int(c.sum())
        if rank == root:
            ok &= bool(np.array_equal(recv[int(offs[r]) * RECORD_BYTES: int(offs[r + 1]) * RECORD_BYTES], b))
    # too little room on the root: every rank returns ENOMEM, nothing is exchanged, nobody waits
    comm.gather_post(1, counts, B, packed)
    try:
        comm.gather_complete(1, B, root, recv, 1)
        ok = False
    except RuntimeError as e:
        ok &= "room for 1" in str(e)
    comm.barrier()
    rx, tx = comm.wire_bytes()
    mine = int(counts.sum()) * RECORD_BYTES
    total = sum(int(batch(r)[0].sum()) for r in range(world)) * RECORD_BYTES
    ok &= tx >= (mine if rank != root else 0) and (rx >= total - mine if rank == root else True)
    comm.close()
    # ---- matcher: row blocks of set 1, set 2 sharded then all-gathered (BASELINE config 5)
    n1, n2 = 96, 64
    p1 = descriptors_to_points(synth_descriptors(n1, 1), orc.POINT_DTYPE)
    p2 = descriptors_to_points(synth_descriptors(n2, 2), orc.POINT_DTYPE)
    b2, e2 = shard_range(n2, rank, world)
    shard = torch.from_numpy(p2[b2:e2].view(np.uint8).reshape(e2 - b2, RECORD_BYTES).copy())
    parts = [torch.empty((shard_range(n2, r, world)[1] - shard_range(n2, r, world)[0], RECORD_BYTES), dtype=torch.uint8)
             for r in range(world)]
    dist.all_gather(parts, shard)
    full2 = torch.cat(parts).numpy().view(orc.POINT_DTYPE).reshape(-1)
    ok &= bool(np.array_equal(full2["data"], p2["data"]))
    b1, e1 = shard_range(n1, rank, world)
    mine = p1.copy()
    orc.match_rows(mine, b1, e1 - b1, full2, n2)           # the CPU oracle stands in for misift_match_rows
    res = torch.from_numpy(np.stack([mine["score"][b1:e1], mine["ambiguity"][b1:e1],
                                     mine["match"][b1:e1].astype(np.float32)], axis=1).copy())
    outs = [torch.empty((shard_range(n1, r, world)[1] - shard_range(n1, r, world)[0], 3)) for r in range(world)]
    dist.all_gather(outs, res)
    ref = p1.copy()
    orc.match(ref, n1, p2, n2)
    got = torch.cat(outs).numpy()
    ok &= bool(np.array_equal(got[:, 0], ref["score"]) and np.array_equal(got[:, 2], ref["match"].astype(np.float32)))
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("1" if ok else "0")
    dist.destroy_process_group()


def test_shard_range_partitions():
    for total in (0, 1, 7, 512, 100000):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1


def test_gather_and_row_block_match_world2(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "ok0").read() == "1" and open(tmp_path / "ok1").read() == "1"


def _pipe_worker(rank, world, port, out_dir):
    """Five pipelined batches through misift_gather_post / misift_gather_complete of a host communicator, rotating root
    (root = k % world, bench.py --gather-root rotate), a rank with nothing to send, an overflowed (-1) frame."""
    import torch
    import torch.distributed as dist
    from cudasift_amd import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = gloo_host_comm(capi, dist, torch, rank, world)
    B, cap, steps = 4, 12, 5

    def batch(r, k):
        rng = np.random.default_rng(1000 * r + k)
        c = rng.integers(0, cap + 1, size=B).astype(np.int32)
        if r == 2 and k % 2 == 1:
            c[:] = 0                                       # a rank with nothing to send this step (no message at all)
        if k == 2:
            c[r] = -1                                      # an overflowed frame: counted as 0 records
        n = int(np.clip(c, 0, None).sum())
        body = rng.integers(0, 255, size=n * RECORD_BYTES, dtype=np.uint8)
        packed = np.zeros(B * cap * RECORD_BYTES, np.uint8)
        packed[: body.size] = body
        return c, body, packed

    ok = True
    live = {}
    room = world * B * cap
    recv = np.zeros(room * RECORD_BYTES, np.uint8)

    def complete(k):
        nonlocal ok
        root = k % world
        all_counts, offs = comm.gather_complete(k % 2, B, root, recv if rank == root else None, room)
        for r in range(world):
            c, body, _ = batch(r, k)
            ok &= bool(np.array_equal(all_counts[r], c))
            if rank == root:
                ok &= bool(np.array_equal(recv[int(offs[r]) * RECORD_BYTES: int(offs[r + 1]) * RECORD_BYTES], body))
        live.pop(k % 2)

    for k in range(steps):                                  # software pipeline: complete(k-1) after post(k)
        c, _, packed = batch(rank, k)
        live[k % 2] = (c, packed)                           # the C side holds pointers into these until complete()
        comm.gather_post(k % 2, c, B, packed)
        if k > 0:
            complete(k - 1)
    complete(steps - 1)
    comm.close()
    open(os.path.join(out_dir, "pok%d" % rank), "w").write("1" if ok else "0")
    dist.destroy_process_group()


def test_pipelined_record_gather_world2(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_pipe_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "pok0").read() == "1" and open(tmp_path / "pok1").read() == "1"


def test_pipelined_record_gather_world3_with_silent_rank(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_pipe_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert all(open(tmp_path / ("pok%d" % r)).read() == "1" for r in range(3))


# ------------------------------------------------------------------ bench.py's N > 1 matcher branch under gloo
class _GlooOracleOps:
    """Stands in for bench.CabiMatchOps on a machine without GPUs: memory = numpy arrays, the per-step call does what
    misift_match_sharded does (all-gather of the set-2 shards, row-block sweep, all-gather of the 12 B/row results)
    with gloo collectives and the CPU oracle as the matcher."""

    def __init__(self, dist, torch, orc, rank, world):
        self.dist, self.torch, self.orc, self.rank, self.world = dist, torch, orc, rank, world

    def to_device(self, recs):
        return recs.copy()

    def empty(self, nbytes):
        return np.zeros(nbytes, np.uint8)

    def match_step(self, rows1, nrows, shard2, nshard, set2_all, results_all):
        from bench import RESULT_DTYPE_NP as RD
        t, dist = self.torch, self.dist
        if self.world == 1:
            self.orc.match_rows(rows1, 0, nrows, shard2, nshard)
            return
        parts = [t.empty((nshard * 576,), dtype=t.uint8) for _ in range(self.world)]
        dist.all_gather(parts, t.from_numpy(shard2.view(np.uint8).reshape(-1)))
        set2_all[:] = t.cat(parts).numpy()
        full = set2_all.view(self.orc.POINT_DTYPE)
        self.orc.match_rows(rows1, 0, nrows, full, nshard * self.world)
        mine = np.zeros(nrows, RD)
        for f in ("score", "ambiguity", "match"):
            mine[f] = rows1[f]
        outs = [t.empty((nrows * 12,), dtype=t.uint8) for _ in range(self.world)]
        dist.all_gather(outs, t.from_numpy(mine.view(np.uint8).reshape(-1)))
        results_all[:] = t.cat(outs).numpy()

    def to_host(self, a, count, dtype):
        return a.view(np.uint8)[: count * np.dtype(dtype).itemsize].view(dtype).copy()

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        v = self.torch.tensor([x], dtype=self.torch.float64)
        self.dist.all_reduce(v, op=self.dist.ReduceOp.MAX)
        return float(v.item())

    def kernel_ms(self, fn):
        import time
        t0 = time.perf_counter()
        fn()
        return 1e3 * (time.perf_counter() - t0)


def _bench_match_worker(rank, world, port, out_dir):
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from oracle import pyoracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ops = _GlooOracleOps(dist, torch, orc, rank, world)
    res = bench.matcher_leg(ops, rank, world, 640, msteps=1, validate_rows=25, l2=(rank < 0), point_dtype=orc.POINT_DTYPE,
                            result_dtype=bench.RESULT_DTYPE_NP, oracle=orc)
    ok = res["n1"] == 640 and res["validated_rows"] >= 20 and res["split"] == "row-block x%d" % world and res["value"] > 0
    open(os.path.join(out_dir, "bok%d" % rank), "w").write("1" if ok else "0")
    dist.destroy_process_group()


def test_bench_matcher_branch_world2_gloo_oracle(tmp_path):
    """bench.py's N > 1 matcher leg (row blocks, set-2 shards, gathered 12-byte results, per-rank spot check against
    the sequential-FMA oracle, max-over-ranks timing) executed with world_size 2 — gloo + the oracle standing in for
    the RCCL / MFMA calls of the C-ABI."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_bench_match_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "bok0").read() == "1" and open(tmp_path / "bok1").read() == "1"
