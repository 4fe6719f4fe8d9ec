"""CPU, world_size 2 over gloo: the multi-GPU path of bench.py / af_mi355x.parallel - contiguous clip
sharding with no data-path collective, then one all-gather of per-clip logits - must reproduce the
single-process result exactly, for even and ragged global batches."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _per_clip_logit(clips):
    """stand-in forward with the real contract: (b,T,H,W,3) uint8 -> (b,1) fp32, each clip independent"""
    x = clips.float().flatten(1)
    w = torch.linspace(-1, 1, x.shape[1])
    return (x * w).sum(1, keepdim=True) / x.shape[1]


def _worker(rank, world, port, global_batch, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import af_mi355x  # noqa: F401
    from af_mi355x import parallel, synth
    r, lr, w = parallel.init(backend="gloo")
    assert (r, w) == (rank, world)
    clips = synth.synthetic_clips_u8(global_batch, seed=11, num_frames=2, size=8)
    mine = parallel.shard_clips(clips, rank, world)
    lo, hi = parallel.shard_bounds(global_batch, rank, world)
    assert mine.shape[0] == hi - lo
    out = parallel.gather_logits(_per_clip_logit(mine), global_batch)
    q.put((rank, out.clone()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [8, 7])
def test_shard_and_gather_world2(global_batch):
    from af_mi355x import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, global_batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _per_clip_logit(synth.synthetic_clips_u8(global_batch, seed=11, num_frames=2, size=8))
    for r in range(2):
        assert got[r].shape == (global_batch, 1)
        assert torch.equal(got[r], want)


def test_shard_bounds_cover_everything():
    from af_mi355x import parallel
    for gb in (1, 7, 16, 128):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
