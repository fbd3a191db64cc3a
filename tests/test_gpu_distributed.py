"""-m gpu: two ranks sharing the one GPU of the test box (gloo transports the CUDA tensors; RCCL refuses two ranks on one
device): the overlapped gradient exchange end to end - HIP backward in two buckets, an event per bucket, the collectives on a side
stream, the join before the optimiser - against the mean of the two ranks' local gradients."""

import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _local_grads(net, rank):
    from oracle import synth

    x = synth.normal(10 + rank, (3, 3, 32, 32)).cuda()
    t = torch.tensor([5, 60, 99]).cuda() + rank
    w = synth.normal(20 + rank, (3, 3, 32, 32)).cuda()
    net.zero_grad(set_to_none=True)
    (net(x, t) * w).sum().backward()
    return net.flat_grad()


def _worker(rank, world, path):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "RANK": str(rank), "WORLD_SIZE": str(world)})
    dist.init_process_group("gloo", init_method=f"file://{path}", rank=rank, world_size=world)
    try:
        import dmme_amd
        from dmme_amd import distributed as D
        from oracle import unet as O

        torch.cuda.set_device(0)
        cfg = O.TINY
        net = dmme_amd.UNet(cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups, 0.0, cfg.channels_per_depth, cfg.num_blocks, cfg.attention_depths)
        sd = O.make_state_dict(O.UNetConfig(pos_dim=cfg.pos_dim, emb_dim=cfg.emb_dim, num_groups=cfg.num_groups, dropout=0.0, channels_per_depth=cfg.channels_per_depth,
                                            num_blocks=cfg.num_blocks, attention_depths=cfg.attention_depths), 11)
        net.load_state_dict(sd, strict=True)
        net.cuda().train()
        want = sum(_local_grads(net, r).clone() for r in range(world)) / world  # what the exchange must produce (same weights on every rank)
        red = D.OverlappedGradReducer(net, bucket_elems=4096)
        for _ in range(3):  # repeated steps: events, side stream and handles are reused
            got = _local_grads(net, rank)
            assert len(red.reported) == 2
            assert red.finish() is True
            torch.cuda.synchronize()
            assert torch.allclose(got, want, rtol=1e-5, atol=1e-6 * float(want.abs().max())), float((got - want).abs().max())
        # every rank ends with the same bits
        mine = got.clone()
        other = got.clone()
        dist.broadcast(other, src=0)
        assert torch.equal(mine, other)
    finally:
        dist.destroy_process_group()


def test_two_ranks_overlapped_gradient_exchange():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, os.path.join(d, "rdv")), nprocs=2, join=True)
