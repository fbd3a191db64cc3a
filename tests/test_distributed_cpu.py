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
        assert m._bucket_hook == red.bucket_ready and red.exchange == "fp32-allreduce"
        red.bucket_ready(600, 400)   # what dmme_unet_backward_buckets reports first
        red.bucket_ready(0, 600)
        assert red.finish() is True
        # the buffer holds rank SUMS: the mean's divide is folded into the fused clip + Adam pass (FusedAdam.grad_scale)
        assert red.grad_scale() == 1.0 / world and torch.allclose(m.g * red.grad_scale(), full.mean(0), atol=1e-6)
        m.g.copy_(full[rank])        # a backward that reported nothing is reduced in one piece
        assert red.finish() is True and torch.allclose(m.g * red.grad_scale(), full.mean(0), atol=1e-6)
        m.g.copy_(full[rank])        # a step WITHOUT exchange on a model that has a reducer: the hook is gone, nothing is pending
        red.detach()
        assert m._bucket_hook is None and not red.handles and not red.reported
        red.attach()
        assert m._bucket_hook == red.bucket_ready
        m2 = FakeModel(full[rank].clone())  # the unfolded form divides before the all-reduce, as round 2 did
        red2 = D.OverlappedGradReducer(m2, bucket_elems=256, fold_mean=False)
        red2.bucket_ready(0, 1000)
        assert red2.finish() is True and red2.grad_scale() == 1.0 and torch.allclose(m2.g, full.mean(0), atol=1e-6)
        # bf16 on the wire, fp32 accumulation at the shard owner, identical bits on every rank (Bf16ShardExchange)
        m3 = FakeModel(full[rank].clone())
        red3 = D.make_reducer(m3, "bf16-rs-ag")
        red3.bucket_elems = 301      # sub-buckets that do not divide by the world size: padding path; >= 4 exchanges per step
        assert red3.exchange == "bf16-rs-ag" and len(D.bucket_slices(1000, red3.bucket_elems)) >= 4
        red3.bucket_ready(600, 400)
        red3.bucket_ready(0, 600)
        assert red3.finish() is True and red3.grad_scale() == 1.0
        want = (full.to(torch.bfloat16).to(torch.float32).sum(0) / world).to(torch.bfloat16).to(torch.float32)
        assert torch.equal(m3.g, want), float((m3.g - want).abs().max())          # exactly: rounded contributions, fp32 sum, one rounding
        assert float((m3.g - full.mean(0)).abs().max()) <= 2.0 ** -7 * float(full.abs().max())  # and within bf16 rounding of the fp32 mean
        allg = [torch.empty(1000) for _ in range(world)]
        dist.all_gather(allg, m3.g)
        assert all(torch.equal(a, allg[0]) for a in allg)                      # rank-identical: clip / Adam / EMA need no further exchange
        m.g.copy_(full[rank])
        red.bucket_ready(600, 400)   # a partial report is an error, not a silent half-reduction
        try:
            red.finish()
            raise AssertionError("expected RuntimeError")
        except RuntimeError:
            red.handles.clear(); red.reported.clear()
        # data-parallel start: every rank takes rank 0's parameters and EMA copy (DDP's broadcast at wrap time); replicas that began
        # from different random weights end up identical, and the module is told its buffer changed under it
        class FakeUNet:
            def __init__(self, w):
                self.w, self.epoch = w, 0

            def flat_parameters(self):
                return self.w

            def mark_params_updated(self):
                self.epoch += 1

        class FakeOpt:
            def __init__(self, e):
                self.e = e

            def ema_parameters(self, model):
                return self.e

        torch.manual_seed(100 + rank)
        net, opt = FakeUNet(torch.randn(500)), FakeOpt(torch.randn(500))
        assert D.sync_parameters(net, opt) is True and net.epoch == 1
        gathered = [torch.empty(500) for _ in range(world)]
        dist.all_gather(gathered, net.w)
        assert all(torch.equal(g, gathered[0]) for g in gathered)
        torch.manual_seed(100)
        assert torch.equal(net.w, torch.randn(500)) and torch.equal(opt.e, torch.randn(500))  # rank 0's draws, in its order
        # one train_step over the process group with the real loop (a stand-in module: no GPU here): gradients averaged, step taken
        from dmme_amd.train_loop import train_step

        class Lit:
            class _DM:
                pass

            def __init__(self, model):
                self.diffusion_model = Lit._DM()
                self.diffusion_model.model = model

            def training_step(self, batch, idx):
                m = self.diffusion_model.model
                loss = ((m.p * batch[0]).sum()) ** 2
                return loss

        class TinyModel(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.p = torch.nn.Parameter(torch.ones(4) * (1 + rank))  # differs per rank until synchronised
                self._g = torch.zeros(4)

            def flat_parameters(self):
                return self.p.data

            def flat_grad(self):
                if self.p.grad is None:
                    self.p.grad = self._g
                return self.p.grad

            def mark_params_updated(self):
                pass

        tm = TinyModel()
        sgd = torch.optim.SGD(tm.parameters(), lr=0.1)
        x0 = torch.arange(4.0) + rank
        train_step(Lit(tm), sgd, None, x0)
        ref = torch.ones(4, requires_grad=True)  # what a single process computes on the mean of the per-rank losses
        (sum(((ref * (torch.arange(4.0) + r)).sum()) ** 2 for r in range(world)) / world).backward()
        assert torch.allclose(tm.p.data, torch.ones(4) - 0.1 * ref.grad, atol=1e-6), (tm.p.data, ref.grad)
        # reduce=False on a model that already has a reducer: its backward must not reach the reducer at all
        assert tm._grad_reducer is not None and tm._bucket_hook is not None
        before = tm.p.data.clone()
        train_step(Lit(tm), sgd, None, x0, reduce=False)
        assert tm._bucket_hook is None and not tm._grad_reducer.handles and not tm._grad_reducer.reported
        loc = before.clone().requires_grad_(True)
        (((loc * x0).sum()) ** 2).backward()
        assert torch.allclose(tm.p.data, before - 0.1 * loc.grad, atol=1e-5)   # a purely local step
        # ... and the next exchanged step reinstalls the hook; here with the bf16 wire format
        tm.p.data.copy_(torch.ones(4))
        train_step(Lit(tm), sgd, None, x0, exchange="bf16-rs-ag")
        assert tm._bucket_hook is not None and tm._grad_reducer.exchange == "bf16-rs-ag"
        assert torch.allclose(tm.p.data, torch.ones(4) - 0.1 * ref.grad, rtol=2.0 ** -7, atol=1e-3), (tm.p.data, ref.grad)
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, os.path.join(d, "rdv")), nprocs=2, join=True)


def test_four_rank_gloo():
    """the same exchanges at world size 4 (shard owners 0..3, sub-buckets that do not divide by 4)"""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(4, os.path.join(d, "rdv")), nprocs=4, join=True)


def test_default_exchange_follows_the_model_precision_and_is_decided_once():
    """bf16 on the wire only for models that compute in 16 bits anyway - fp32 / bf16x3 keep the reference's fp32
    all-reduce at every batch size (their 1e-5 parity does not survive bf16-rounded gradients) - and the choice of a run is made once
    (a short last batch must not swap the reducer)."""
    import dmme_amd
    from dmme_amd import distributed as D

    os.environ.pop("DMME_EXCHANGE", None)
    from oracle import unet as O

    cfg = O.TINY
    nets = {p: dmme_amd.UNet(cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups, cfg.dropout, cfg.channels_per_depth, cfg.num_blocks,
                             cfg.attention_depths, precision=p) for p in ("fp32", "bf16x3", "bf16", "fp16")}
    assert D.default_exchange(16, nets["fp32"]) == "fp32-allreduce" and D.default_exchange(16, nets["bf16x3"]) == "fp32-allreduce"
    assert D.default_exchange(16, nets["bf16"]) == "bf16-rs-ag" and D.default_exchange(16, nets["fp16"]) == "bf16-rs-ag"
    assert D.default_exchange(128, nets["bf16"]) == "fp32-allreduce"
    assert D.run_exchange(nets["bf16"], 16) == "bf16-rs-ag"
    assert D.run_exchange(nets["bf16"], 128) == "bf16-rs-ag"  # decided at the first step, kept for the run
    assert D.run_exchange(nets["bf16"], 128, "fp32-allreduce") == "fp32-allreduce"  # an explicit request still wins
    os.environ["DMME_EXCHANGE"] = "bf16-rs-ag"
    try:
        assert D.default_exchange(16, nets["fp32"]) == "bf16-rs-ag"  # the override is the operator's own decision
    finally:
        os.environ.pop("DMME_EXCHANGE", None)
