import os, sys, tempfile, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
backend = sys.argv[1]
os.environ.update({"MASTER_ADDR": "127.0.0.1", "RANK": "0", "WORLD_SIZE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
torch.cuda.set_device(0)
d = tempfile.mkdtemp()
dist.init_process_group(backend, init_method=f"file://{d}/rdv", rank=0, world_size=1)
import dmme_amd
from dmme_amd import distributed as D
from oracle import unet as O
from tests.test_gpu_distributed import _local_grads
cfg = O.TINY
net = dmme_amd.UNet(cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups, 0.0, cfg.channels_per_depth, cfg.num_blocks, cfg.attention_depths)
sd = O.make_state_dict(O.UNetConfig(pos_dim=cfg.pos_dim, emb_dim=cfg.emb_dim, num_groups=cfg.num_groups, dropout=0.0, channels_per_depth=cfg.channels_per_depth,
                                    num_blocks=cfg.num_blocks, attention_depths=cfg.attention_depths), 11)
net.load_state_dict(sd, strict=True); net.cuda().train()
want = _local_grads(net, 0).clone()
w2 = _local_grads(net, 0).clone()
print("repeatable local grads:", torch.equal(want, w2), "numel", want.numel())
# direct exchange of a plain tensor, no backward involved
red = D.Bf16ShardExchange(net, bucket_elems=4096); red.active = lambda: True
v = want.clone()
red._reduce(v, 1); torch.cuda.synchronize()
print("direct _reduce on current stream: max diff", float((v - want.to(torch.bfloat16).float()).abs().max()))
for it in range(3):
    got = _local_grads(net, 0)
    rep = list(red.reported)
    ok = red.finish(); torch.cuda.synchronize()
    df = (got - want.to(torch.bfloat16).float()).abs()
    bad = (df > 0).nonzero().flatten()
    print(f"iter {it}: reported {rep[:8]} ok={ok} max diff {float(df.max()):.4g} nbad {bad.numel()} first bad {bad[:3].tolist()} last bad {bad[-3:].tolist()}")
    raw = (got - want).abs()
    print("   vs unrounded want: max", float(raw.max()), " vs 2*want:", float((got - 2 * want).abs().max()))
dist.destroy_process_group()
