"""-m gpu: the HBM-resident input pipeline (dmme_image_batch) against the CPU oracle (oracle/data.py): bit-exact."""

import numpy as np
import pytest
import torch

from oracle import data as OD

pytestmark = pytest.mark.gpu


def _set(n, c=3, h=32, w=32, seed=0):
    return torch.from_numpy(np.random.RandomState(seed).randint(0, 256, size=(n, c, h, w)).astype(np.uint8))


@pytest.mark.parametrize("shape", [(3, 32, 32), (3, 64, 64), (1, 8, 12)])
def test_image_batch_bit_exact(golden, shape):
    from dmme_amd.data_modules import GpuBatchLoader

    data = _set(57, *shape)
    loader = GpuBatchLoader(data.cuda(), None, 16, shuffle=False, flip_p=0.0)
    rs = np.random.RandomState(1)
    idx = torch.from_numpy(rs.randint(0, 57, size=23))
    flip = torch.from_numpy(rs.randint(0, 2, size=23).astype(np.uint8))
    for fl in (None, flip):
        got = loader.batch(idx.cuda(), None if fl is None else fl.cuda()).cpu()
        want = OD.image_batch(data, idx, fl)
        assert torch.equal(got, want)
    # every byte value lands on the reference's norm(ToTensor(.)) value
    ramp = torch.arange(256, dtype=torch.uint8).reshape(1, 1, 8, 32)
    got = GpuBatchLoader(ramp.cuda(), None, 1, False, 0.0).batch(torch.tensor([0]).cuda(), None).cpu().reshape(-1)
    assert np.array_equal(got.numpy(), golden("data")["norm_u8_table"])


def test_loader_epoch_covers_the_set_once_and_flips_about_half():
    from dmme_amd.data_modules import GpuBatchLoader

    n = 1000
    data = torch.zeros((n, 1, 4, 4), dtype=torch.uint8)
    data[:, 0, 0, 0] = torch.arange(n) % 251   # id marker (left column)
    data[:, 0, 1, 0] = torch.arange(n) // 251
    data[:, 0, 2, 3] = 255                      # orientation marker (right column)
    loader = GpuBatchLoader(data.cuda(), torch.arange(n).cuda(), 128, shuffle=True, flip_p=0.5, seed=3)
    seen, flipped, nb = [], 0, 0
    for x, y in loader:
        nb += 1
        u = torch.round((x / 2 + 0.5) * 255).to(torch.int64).cpu()
        fl = u[:, 0, 2, 0] == 255
        ids = torch.where(fl, u[:, 0, 0, 3] + 251 * u[:, 0, 1, 3], u[:, 0, 0, 0] + 251 * u[:, 0, 1, 0])
        assert torch.equal(ids, y.cpu())
        seen += ids.tolist()
        flipped += int(fl.sum())
    assert nb == len(loader) == 8 and sorted(seen) == list(range(n))
    assert 400 < flipped < 600
    first_epoch = seen
    second = [int(v) for _, y in loader for v in y.cpu()]
    assert second != first_epoch and sorted(second) == list(range(n))


def test_trainer_fit_with_the_yaml_data_module(tmp_path, capsys):
    """`dmme.trainer fit --config ... --data config` with a CIFAR10 directory in the reference's on-disk format."""
    import json
    import pickle

    import yaml

    from dmme_amd import trainer

    d = tmp_path / "cifar-10-batches-py"
    d.mkdir()
    rs = np.random.RandomState(0)
    for i in range(1, 6):
        with open(d / f"data_batch_{i}", "wb") as f:
            pickle.dump({"data": rs.randint(0, 256, size=(40, 3072)).astype(np.uint8), "labels": [0] * 40}, f)
    cfg = {
        "seed_everything": 1337,
        "trainer": {"max_steps": 4, "gradient_clip_val": 1.0, "precision": 32, "log_every_n_steps": 2},
        "model": {"class_path": "dmme.LitDDPM", "init_args": {"lr": 2e-4, "warmup": 10, "decay": 0.99,
                  "model": {"class_path": "dmme.UNet", "init_args": {"pos_dim": 8, "emb_dim": 16, "num_groups": 2, "channels_per_depth": [8, 16], "num_blocks": 1, "attention_depths": [2]}}}},
        "data": {"class_path": "dmme.CIFAR10", "init_args": {"data_dir": str(tmp_path), "batch_size": 64,
                 "augs": [{"class_path": "torchvision.transforms.RandomHorizontalFlip"}]}},
    }
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    ck = str(tmp_path / "last.ckpt")
    assert trainer.main(["fit", "--config", str(path), "--data", "config", "--save-checkpoint", ck]) == 0
    lines = [json.loads(l) for l in capsys.readouterr().out.strip().splitlines()]
    assert lines[-1]["step"] == 4 and np.isfinite(lines[-1]["train/loss"])
    # resume from the Lightning-layout checkpoint for two more steps, then sample from it
    assert trainer.main(["fit", "--config", str(path), "--data", "config", "--ckpt-path", ck, "--max-steps", "6"]) == 0
    lines = [json.loads(l) for l in capsys.readouterr().out.strip().splitlines()]
    assert [l["step"] for l in lines] == [6]
    assert trainer.main(["sample", "--config", str(path), "--ckpt-path", ck, "--num-images", "2", "--steps", "3"]) == 0
    assert json.loads(capsys.readouterr().out.strip().splitlines()[-1])["finite"]
