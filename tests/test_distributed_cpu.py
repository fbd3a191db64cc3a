"""CPU, world_size 2 over gloo: the N>1 plumbing (batch sharding without a collective, per-rank
streams, max-over-ranks timing, bucketed mean all-reduce of a flat gradient buffer)."""

import os
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dmme_amd import distributed as D


def test_shard_and_bucket_helpers():
    assert [D.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert D.shard_range(0, 0, 2) == (0, 0)
    sl = D.bucket_slices(10, 4)
    assert sl == [(6, 10), (2, 6), (0, 2)] and sum(e - b for b, e in sl) == 10
    assert D.rank_seed(1337, 0) != D.rank_seed(1337, 1)


def _worker(rank, world, path):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "RANK": str(rank), "WORLD_SIZE": str(world)})
    dist.init_process_group("gloo", init_method=f"file://{path}", rank=rank, world_size=world)
    try:
        # sampling shards: disjoint, covering, no collective involved
        b, e = D.shard_range(7, rank, world)
        owned = torch.zeros(7)
        owned[b:e] = 1
        dist.all_reduce(owned)
        assert torch.equal(owned, torch.ones(7))
        assert D.max_over_ranks(1.0 + rank) == float(world)
        # training exchange: mean of per-rank flat grads, bucketed, equals the big-batch gradient
        torch.manual_seed(0)
        full = torch.randn(world, 1000)
        flat = full[rank].clone()
        hs = D.allreduce_mean_flat(flat, bucket_elems=300, async_op=True)
        assert len(hs) == 4
        for h in hs:
            h.wait()
        assert torch.allclose(flat, full.mean(0), atol=1e-6)
        # the overlapped form: buckets reported mid-backward (tail of the buffer first), reduced asynchronously, joined by finish()
        class FakeModel:
            def __init__(self, g):
                self.g = g

            def flat_grad(self):
                return self.g

        m = FakeModel(full[rank].clone())
        red = D.OverlappedGradReducer(m, bucket_elems=256)
        assert m._bucket_hook == red.bucket_ready
        red.bucket_ready(600, 400)   # what dmme_unet_backward_buckets reports first
        red.bucket_ready(0, 600)
        assert red.finish() is True
        assert torch.allclose(m.g, full.mean(0), atol=1e-6)
        m.g.copy_(full[rank])        # a backward that reported nothing is reduced in one piece
        assert red.finish() is True and torch.allclose(m.g, full.mean(0), atol=1e-6)
        m.g.copy_(full[rank])
        red.bucket_ready(600, 400)   # a partial report is an error, not a silent half-reduction
        try:
            red.finish()
            raise AssertionError("expected RuntimeError")
        except RuntimeError:
            red.handles.clear(); red.reported.clear()
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, os.path.join(d, "rdv")), nprocs=2, join=True)
