"""GPU tests of the multi-GPU entry points of the C-ABI (include/misift.h: misift_comm_*, misift_gather_*,
misift_match_sharded) on RCCL.  With one visible device the communicator has one rank (every collective still
goes through RCCL); with two or more, two host threads drive one context + communicator each, the way a C++
caller of the boundary would (one thread per GPU)."""
import threading

import numpy as np
import pytest

from conftest import record
from synth import descriptors_to_points, synth_descriptors, synth_frame

pytestmark = pytest.mark.gpu


def _canon(recs):
    k = [recs[f].view(np.uint32) for f in ("orientation", "scale", "ypos", "xpos")]
    return recs[np.lexsort(k)].tobytes()


def _extract_packed(capi, c, frames, mp, slot_bufs):
    """misift_extract_batch_packed_async of `frames` [B,h,w] on context c; returns the device buffers."""
    B, h, w = frames.shape
    d = c.upload(frames)
    sc = capi.DevBuf(4 * capi.scratch_floats(w, h, 5, False) * B)
    cnt = c.zeros(4 * (2 * B + 1))
    packed = c.zeros(576 * mp * B)
    capi.check(capi.lib().misift_extract_batch_packed_async(c.h, d.ptr, B, h * w, w, h, w, 5, 1.0, 2.0, 0.0, sc.ptr, None,
                                                            mp, cnt.ptr, cnt.ptr + 4 * B, packed.ptr),
               "misift_extract_batch_packed_async")
    slot_bufs.extend([d, sc])                    # keep alive until the stream is done
    return cnt, packed


def _rank_body(capi, rank, world, uid, frames_of, mp, out, errs):
    try:
        c = capi.Context(rank)
        comm = capi.Comm(c, world, rank, uid)
        assert comm.rank == rank and comm.size == world
        keep = []
        B = frames_of(0).shape[0]
        nslot = 2
        results = []
        # two batches in flight: post(k) ... complete(k-1), like bench.py's software pipeline
        bufs = {}
        for k in range(3):
            cnt, packed = _extract_packed(capi, c, frames_of(rank * 10 + k), mp, keep)
            comm.gather_post(k % nslot, cnt.ptr, B, packed.ptr)
            bufs[k] = (cnt, packed)
            if k >= 1:
                recv = c.zeros(576 * mp * B * world) if rank == 0 else None
                counts, offs = comm.gather_complete((k - 1) % nslot, B, 0, recv.ptr if recv else None, mp * B * world)
                recs = c.download(recv, (int(offs[-1]),), capi.POINT_DTYPE) if rank == 0 else None
                results.append((k - 1, counts, offs, recs))
        recv = c.zeros(576 * mp * B * world) if rank == 0 else None
        counts, offs = comm.gather_complete(2 % nslot, B, 0, recv.ptr if recv else None, mp * B * world)
        results.append((2, counts, offs, c.download(recv, (int(offs[-1]),), capi.POINT_DTYPE) if rank == 0 else None))
        comm.barrier()
        # ---- too little room on the root: every rank says so (same decision everywhere, nothing is exchanged, nobody hangs)
        cnt, packed = _extract_packed(capi, c, frames_of(rank * 10), mp, keep)
        comm.gather_post(0, cnt.ptr, B, packed.ptr)
        try:
            comm.gather_complete(0, B, 0, recv.ptr if recv else None, 7)
            raise AssertionError("capacity overflow not reported")
        except RuntimeError as e:
            assert "room for 7" in str(e), e
        comm.barrier()
        # ---- matcher: row blocks of set 1, shards of set 2
        n1, n2 = 512 * world, 384 * world
        p1 = descriptors_to_points(synth_descriptors(n1, 31), capi.POINT_DTYPE)
        p2 = descriptors_to_points(synth_descriptors(n2, 32), capi.POINT_DTYPE)
        rows, shard = n1 // world, n2 // world
        d1 = c.upload(p1[rank * rows:(rank + 1) * rows])
        d2 = c.upload(p2[rank * shard:(rank + 1) * shard])
        all2 = c.zeros(576 * n2)
        res = c.zeros(12 * n1)
        comm.match_sharded(d1.ptr, rows, d2.ptr, shard, all2.ptr, res.ptr)
        out[rank] = dict(results=results, rows=c.download(d1, (rows,), capi.POINT_DTYPE),
                         res=c.download(res, (n1,), capi.RESULT_DTYPE), set2=c.download(all2, (n2,), capi.COLUMN_DTYPE),
                         p1=p1, p2=p2)
        comm.close()
        c.close()
    except Exception as e:                         # noqa: BLE001 — reported by the main thread
        import traceback
        errs.append("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))


def _check(capi, ctx, world, out, frames_of, mp):
    # gather: the root holds, per batch, the records of rank 0's frames then rank 1's ... exactly as a single-GPU
    # batch call of the same frames returns them
    for k, counts, offs, recs in out[0]["results"]:
        assert counts.shape[0] == world
        for r in range(world):
            rp, rn = ctx.extract_batch(frames_of(r * 10 + k), thresh=2.0, max_pts=mp)
            assert np.array_equal(counts[r], rn), (k, r, counts[r], rn)
            off = int(offs[r])
            for f in range(len(rn)):
                assert _canon(recs[off:off + rn[f]]) == _canon(rp[f, :rn[f]]), (k, r, f)
                off += rn[f]
            assert off == offs[r + 1]
        for r in range(1, world):                  # every rank sees the same counts
            assert np.array_equal(out[r]["results"][k][1], counts)
    # matcher: every rank's row block equals the single-GPU misift_match of those rows; the all-gathered 12-byte
    # results are the whole answer on every rank
    p1, p2 = out[0]["p1"], out[0]["p2"]
    ref = ctx.match(p1, len(p1), p2, len(p2))
    rows = len(p1) // world
    for r in range(world):
        assert np.array_equal(out[r]["set2"]["data"], p2["data"])          # the gathered set 2: 528-byte match columns
        assert np.array_equal(out[r]["set2"]["xpos"], p2["xpos"]) and np.array_equal(out[r]["set2"]["ypos"], p2["ypos"])
        blk = out[r]["rows"]
        for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos"):
            assert np.array_equal(blk[f], ref[f][r * rows:(r + 1) * rows]), (r, f)
        for f in ("score", "ambiguity", "match"):
            assert np.array_equal(out[r]["res"][f], ref[f]), (r, f)


def _frames_of(seed):
    return np.stack([synth_frame(9000 + 7 * seed + i, width=480, height=272) for i in range(3)]).astype(np.float32)


def test_comm_single_rank_gather_and_match_sharded(ctx):
    """One rank: counts all-gather, the root's own records (device-to-device) and both all-gathers of the matcher
    run through RCCL on this GPU."""
    from cudasift_amd import capi
    out, errs = {}, []
    uid = capi.comm_unique_id()
    _rank_body(capi, 0, 1, uid, _frames_of, 4096, out, errs)
    assert not errs, errs
    _check(capi, ctx, 1, out, _frames_of, 4096)
    record("comm_single_rank", batches=3, ok=True)


def test_match_sharded_refuses_aliased_buffers(ctx):
    """The shard is re-packed into d_set2_all: an in-place call (legal in r03, when records were all-gathered as they
    were) is refused instead of racing (advisor r04) — with a single-rank communicator too."""
    from cudasift_amd import capi
    from synth import descriptors_to_points, synth_descriptors
    comm = capi.Comm(ctx, 1, 0, capi.comm_unique_id())
    try:
        n = 256
        a = descriptors_to_points(synth_descriptors(n, 1, l2=True), capi.POINT_DTYPE)
        d1, d2 = ctx.upload(a), ctx.upload(a)
        with pytest.raises(capi.MisiftError, match="overlaps"):
            comm.match_sharded(d1.ptr, n, d2.ptr, n, d2.ptr, None)
        tail = capi.DevBuf(576 * 2 * n)                  # shard in the upper half of the gather buffer: still overlapping
        capi.check(capi.lib().misift_copy_h2d(ctx.h, tail.ptr + 200 * n, a.ctypes.data, 576 * n))
        with pytest.raises(capi.MisiftError, match="overlaps"):
            comm.match_sharded(d1.ptr, n, tail.ptr + 200 * n, n, tail.ptr, None)
        all2 = capi.DevBuf(576 * n)
        comm.match_sharded(d1.ptr, n, d2.ptr, n, all2.ptr, None)             # separate buffers: fine
    finally:
        comm.close()


def test_comm_two_devices_threads(ctx):
    """Two GPUs, one host thread each (what a C++ caller of the boundary does): pipelined gather to rank 0 and the
    row-block matcher.  Skipped on a one-GPU box."""
    from cudasift_amd import capi
    if capi.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    world = 2
    out, errs = {}, []
    uid = capi.comm_unique_id()
    ts = [threading.Thread(target=_rank_body, args=(capi, r, world, uid, _frames_of, 4096, out, errs)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not errs, errs
    assert len(out) == world
    _check(capi, ctx, world, out, _frames_of, 4096)
    record("comm_two_devices", ok=True)


# ------------------------------------------------------------------------------------------- loopback world
# the N > 1 branches of multigpu.hip (count all-gather, per-rank record counts and offsets, root
# placement, grouped send/recv, -1 frames, the collective ENOMEM decision, set-2 placement, result all-gather) executed on
# ONE GPU: N host threads, N contexts of device 0, N communicators of a misift_loopback_world.  Same entry points and
# the same code above the transport table as with RCCL.
def _loop_rank(capi, rank, world, lw, plan, out, errs):
    """plan: dict(root, mp, frames(rank, k) -> [B,h,w], thresh(rank, k), nbatches, capacity)"""
    try:
        c = capi.Context(0)
        comm = capi.Comm(c, world, rank, lw)
        assert comm.rank == rank and comm.size == world
        root, mp, nb = plan["root"], plan["mp"], plan["nbatches"]
        keep, results, nslot = [], [], 2
        B = plan["frames"](0, 0).shape[0]
        cap = plan["capacity"]

        def extract(k):
            fr = plan["frames"](rank, k)
            h, w = fr.shape[1:]
            d = c.upload(fr)
            sc = capi.DevBuf(4 * capi.scratch_floats(w, h, 5, False) * B)
            cnt = c.zeros(4 * (2 * B + 1))
            packed = c.zeros(576 * mp * B)
            capi.check(capi.lib().misift_extract_batch_packed_async(c.h, d.ptr, B, h * w, w, h, w, 5, 1.0, plan["thresh"](rank, k),
                                                                    0.0, sc.ptr, None, mp, cnt.ptr, cnt.ptr + 4 * B, packed.ptr),
                       "misift_extract_batch_packed_async")
            keep.extend([d, sc, cnt, packed])
            return cnt, packed

        def complete(k):
            rt = (k % world) if plan.get("rotate") else root        # rotating root: batch k is gathered on rank k % world
            recv = c.zeros(576 * cap) if rank == rt else None
            polls = 0
            while not comm.gather_test(k % nslot):          # the non-blocking companion: poll instead of parking
                polls += 1
            counts, offs = comm.gather_complete(k % nslot, B, rt, recv.ptr if recv else None, cap)
            recs = c.download(recv, (int(offs[-1]),), capi.POINT_DTYPE) if rank == rt else None
            results.append((k, counts, offs, recs))

        for k in range(nb):                                  # post(k) ... complete(k-1): two batches in flight
            cnt, packed = extract(k)
            comm.gather_post(k % nslot, cnt.ptr, B, packed.ptr)
            if k >= 1:
                complete(k - 1)
        complete(nb - 1)
        comm.barrier()
        wire_after_gathers = comm.wire_bytes()
        # too little room on the root: every rank takes the same decision, nothing is exchanged, nobody hangs
        cnt, packed = extract(0)
        comm.gather_post(0, cnt.ptr, B, packed.ptr)
        try:
            comm.gather_complete(0, B, root, c.zeros(576 * 8).ptr if rank == root else None, 7)
            raise AssertionError("capacity overflow not reported")
        except RuntimeError as e:
            assert "room for 7" in str(e), e
        comm.barrier()
        # matcher: row blocks of set 1, shards of set 2 (shard size deliberately not a multiple of 32)
        rows, shard = plan["rows"], plan["shard"]
        n1, n2 = rows * world, shard * world
        p1 = descriptors_to_points(synth_descriptors(n1, 31), capi.POINT_DTYPE)
        p2 = descriptors_to_points(synth_descriptors(n2, 32), capi.POINT_DTYPE)
        d1 = c.upload(p1[rank * rows:(rank + 1) * rows])
        d2 = c.upload(p2[rank * shard:(rank + 1) * shard])
        all2 = c.zeros(576 * n2)
        res = c.zeros(12 * n1)
        comm.match_sharded(d1.ptr, rows, d2.ptr, shard, all2.ptr, res.ptr)
        out[rank] = dict(results=results, rows=c.download(d1, (rows,), capi.POINT_DTYPE),
                         res=c.download(res, (n1,), capi.RESULT_DTYPE), set2=c.download(all2, (n2,), capi.COLUMN_DTYPE),
                         p1=p1, p2=p2, wire=wire_after_gathers)
        comm.close()
        c.close()
    except Exception as e:                         # noqa: BLE001 — reported by the main thread
        import traceback
        errs.append("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))


def _run_loopback(capi, world, plan):
    lw = capi.LoopbackWorld(world)
    out, errs = {}, []
    ts = [threading.Thread(target=_loop_rank, args=(capi, r, world, lw, plan, out, errs)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=900)
    assert not any(t.is_alive() for t in ts), "a loopback rank is stuck"
    assert not errs, "\n".join(errs)
    assert len(out) == world
    lw.close()
    return out


def _check_loopback(capi, ctx, world, plan, out):
    root, mp = plan["root"], plan["mp"]
    n_overflowed = 0
    for k in range(plan["nbatches"]):
        rt = (k % world) if plan.get("rotate") else root
        k_, counts, offs, recs = out[rt]["results"][k]
        assert k_ == k and recs is not None
        assert all(out[r]["results"][k][3] is None for r in range(world) if r != rt)      # only that batch's root holds records
        assert counts.shape[0] == world and offs[0] == 0
        for r in range(world):
            th = plan["thresh"](r, k)
            fr = plan["frames"](r, k)
            assert np.array_equal(out[r]["results"][k][1], counts)          # every rank holds every count
            assert np.array_equal(out[r]["results"][k][2], offs)
            off = int(offs[r])
            if th < 0.01:                                                   # candidate lists overflow: -1, no records
                assert (counts[r] == -1).all(), counts[r]
                n_overflowed += len(counts[r])
            else:
                rp, rn = ctx.extract_batch(fr, thresh=th, max_pts=mp)
                assert np.array_equal(counts[r], rn), (k, r, counts[r], rn)
                for f in range(len(rn)):
                    assert _canon(recs[off:off + rn[f]]) == _canon(rp[f, :rn[f]]), (k, r, f)
                    off += rn[f]
            assert off == offs[r + 1], (k, r)
        assert len(recs) == offs[world]
    p1, p2 = out[0]["p1"], out[0]["p2"]
    ref = ctx.match(p1, len(p1), p2, len(p2))
    rows = plan["rows"]
    for r in range(world):
        # set 2 replicated in rank order as 528-byte match columns (descriptor + position: what the sweep reads)
        assert np.array_equal(out[r]["set2"]["data"], p2["data"])
        assert np.array_equal(out[r]["set2"]["xpos"], p2["xpos"]) and np.array_equal(out[r]["set2"]["ypos"], p2["ypos"])
        blk = out[r]["rows"]
        for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos"):
            assert np.array_equal(blk[f], ref[f][r * rows:(r + 1) * rows]), (r, f)
        for f in ("score", "ambiguity", "match"):
            assert np.array_equal(out[r]["res"][f], ref[f]), (r, f)
    return n_overflowed


def _plan(world, root, empty_rank=None, overflow_rank=None, B=2, w=320, h=208, mp=2048):
    flat = np.full((B, h, w), 77.0, np.float32)                              # no keypoints at all: an empty rank

    def frames(rank, k):
        if rank == empty_rank:
            return flat
        if rank == overflow_rank and k == 1:         # white noise at a near-zero threshold: more in-row extrema than the
            rng = np.random.default_rng(123 + rank)  # per-octave candidate list max(16384, w*h/4) holds
            return rng.uniform(0.0, 255.0, (B, h, w)).astype(np.float32)
        # ragged: different frames (and counts) on every rank and batch
        return np.stack([synth_frame(7000 + 31 * rank + 5 * k + i, width=w, height=h) for i in range(B)]).astype(np.float32)

    def thresh(rank, k):
        return 0.0005 if (rank == overflow_rank and k == 1) else 2.0 + 0.25 * (rank % 3)

    return dict(root=root, mp=mp, frames=frames, thresh=thresh, nbatches=3, capacity=mp * B * world, rows=96, shard=100)


@pytest.mark.parametrize("world,root,empty,overflow", [(2, 0, None, None), (2, 1, 0, None), (4, 2, 1, 3), (8, 5, 6, 0)])
def test_loopback_world_gather_and_match_sharded(ctx, world, root, empty, overflow):
    """world ranks on ONE GPU through the loopback transport: ragged counts, an empty rank, a rank whose frames
    overflow their candidate lists (-1 counts), root != 0, capacity too small, shard size % 32 != 0."""
    from cudasift_amd import capi
    plan = _plan(world, root, empty, overflow)
    out = _run_loopback(capi, world, plan)
    nov = _check_loopback(capi, ctx, world, plan, out)
    if overflow is not None:
        assert nov > 0
    record("loopback_world_%d" % world, root=root, empty_rank=empty, overflow_rank=overflow, overflowed_frames=nov, ok=True)


def test_loopback_world_rotating_gather_root(ctx):
    """The gather root moves with the batch (root = k % world; bench.py --gather-root rotate): one fixed root would take
    7 senders' records of EVERY batch over its own xGMI links.  Same records as a single-GPU batch."""
    from cudasift_amd import capi
    world = 4
    plan = _plan(world, 0, 2, None)
    plan["rotate"] = True
    plan["nbatches"] = 5                                   # batches 0..4: roots 0, 1, 2, 3, 0
    out = _run_loopback(capi, world, plan)
    _check_loopback(capi, ctx, world, plan, out)
    # bytes: a rank receives records only for the batches it is the root of
    recv = [out[r]["wire"][0] for r in range(world)]
    assert recv[0] > recv[1] and min(recv) > 0, recv       # rank 0 was the root twice
    record("loopback_world_rotating_root", received_bytes=recv, ok=True)


def test_loopback_world_reports_a_missing_rank(ctx):
    """A rank that never makes the matching call: its peer gets an error after the timeout instead of hanging."""
    import os
    from cudasift_amd import capi
    os.environ["MISIFT_LOOPBACK_TIMEOUT_S"] = "2"
    try:
        lw = capi.LoopbackWorld(2)
    finally:
        del os.environ["MISIFT_LOOPBACK_TIMEOUT_S"]
    c = capi.Context(0)
    comm = capi.Comm(c, 2, 0, lw)
    with pytest.raises(RuntimeError, match="timed out"):
        comm.barrier()
    comm.close()
    c.close()
    lw.close()


# the set-2 all-gather used to precede the whole sweep.  With more than one rank the sweep of the
# super-tiles inside the rank's OWN shard now runs while the exchange is on the communication stream, the other columns
# follow behind its event, one merge covers both launches.
def _match_rank(capi, rank, world, lw, rows, shard, out, errs):
    try:
        c = capi.Context(0)
        comm = capi.Comm(c, world, rank, lw)
        n1, n2 = rows * world, shard * world
        p1 = descriptors_to_points(synth_descriptors(n1, 41), capi.POINT_DTYPE)
        p2 = descriptors_to_points(synth_descriptors(n2, 42), capi.POINT_DTYPE)
        p2["data"][3 * shard // 2] = p2["data"][shard // 3]            # an exact tie across two shards: the smaller column wins
        d1 = c.upload(p1[rank * rows:(rank + 1) * rows])
        d2 = c.upload(p2[rank * shard:(rank + 1) * shard])
        all2 = c.zeros(576 * n2)
        res = c.zeros(12 * n1)
        c.profile_enable(True)
        for _ in range(2):                                            # twice: the buffers and events are reused
            comm.match_sharded(d1.ptr, rows, d2.ptr, shard, all2.ptr, res.ptr)
        prof = c.profile_read()
        c.profile_enable(False)
        out[rank] = dict(rows=c.download(d1, (rows,), capi.POINT_DTYPE), res=c.download(res, (n1,), capi.RESULT_DTYPE),
                         p1=p1, p2=p2, sweeps=prof["match_mfma"]["calls"], wire=comm.wire_bytes())
        comm.close()
        c.close()
    except Exception as e:                         # noqa: BLE001
        import traceback
        errs.append("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))


@pytest.mark.parametrize("world,rows,shard,overlap", [(3, 70, 333, True), (3, 70, 333, False), (8, 40, 130, True), (2, 33, 40, True)])
def test_loopback_match_sharded_overlaps_the_exchange(ctx, world, rows, shard, overlap):
    """Own-shard sweep beside the all-gather + the rest behind it == one sweep over all of set 2, bit for bit (shard sizes
    that are not multiples of 64, a partial last tile, an exact tie across shards, a shard too small to hold a super-tile)."""
    import os
    from cudasift_amd import capi
    if not overlap:
        os.environ["MISIFT_MATCH_NO_OVERLAP"] = "1"
    try:
        lw = capi.LoopbackWorld(world)
        out, errs = {}, []
        ts = [threading.Thread(target=_match_rank, args=(capi, r, world, lw, rows, shard, out, errs)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=600)
    finally:
        os.environ.pop("MISIFT_MATCH_NO_OVERLAP", None)
    assert not any(t.is_alive() for t in ts), "a loopback rank is stuck"
    assert not errs, "\n".join(errs)
    lw.close()
    p1, p2 = out[0]["p1"], out[0]["p2"]
    ref = ctx.match(p1, len(p1), p2, len(p2))
    from oracle import pyoracle
    exp = p1.copy()
    pyoracle.match(exp, len(exp), p2, len(p2))
    for f in ("score", "ambiguity", "match"):
        assert np.array_equal(ref[f], exp[f]), f
    for r in range(world):
        for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos"):
            assert np.array_equal(out[r]["rows"][f], ref[f][r * rows:(r + 1) * rows]), (r, f)
        for f in ("score", "ambiguity", "match"):
            assert np.array_equal(out[r]["res"][f], ref[f]), (r, f)
        lo, hi = (r * shard + 63) // 64, ((r + 1) * shard) // 64
        two = overlap and hi > lo
        assert out[r]["sweeps"] == (4 if two else 2), (r, out[r]["sweeps"])      # two calls: 2 launches each when split
        # bytes on the wire (r04): 528-byte match columns, not 576-byte records, + the 12-byte results; two calls
        per_call = (world - 1) * (shard * capi.COLUMN_DTYPE.itemsize + rows * 12)
        assert out[r]["wire"] == (2 * per_call, 2 * per_call), (r, out[r]["wire"], per_call)
    record("loopback_match_overlap_%d_%d_%d" % (world, shard, overlap), ok=True)
