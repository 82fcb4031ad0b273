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
                         res=c.download(res, (n1,), capi.RESULT_DTYPE), set2=c.download(all2, (n2,), capi.POINT_DTYPE),
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
        assert np.array_equal(out[r]["set2"]["data"], p2["data"])
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
