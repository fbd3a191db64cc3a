"""CPU: the C-ABI library loads and exports every symbol include/dmme_hip.h declares; the
ctypes prototypes list exactly that set; host-only entry points behave (no GPU compute)."""

import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dmme_hip.h")).read()
    return sorted(set(re.findall(r"DMME_API[^;(]*?\b(dmme_\w+)\s*\(", text)))


def test_header_declares_entry_points():
    names = declared_symbols()
    assert len(names) >= 30
    for must in ("dmme_unet_plan_create", "dmme_unet_forward", "dmme_q_sample", "dmme_ddpm_step", "dmme_ddim_step", "dmme_conv2d"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from dmme_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    h = C.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(h, name), f"{name} declared in include/dmme_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES) == declared_symbols()
    assert _lib.lib().dmme_version() >= 100


def test_plan_is_host_only_and_reports_errors():
    from dmme_amd import _lib

    lib = _lib.lib()
    cfg = _lib.UNetCfg()
    cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups, cfg.dropout = 3, 128, 512, 32, 0.1
    cfg.num_depths, cfg.num_blocks, cfg.num_attention_depths = 4, 2, 1
    for i, c in enumerate((128, 256, 256, 256)):
        cfg.channels_per_depth[i] = c
    cfg.attention_depths[0] = 2
    h = C.c_void_p()
    assert lib.dmme_unet_plan_create(C.byref(cfg), 128, 32, 32, _lib.BF16, -1, C.byref(h)) == 0
    assert lib.dmme_unet_plan_num_params(h) == 305
    assert lib.dmme_unet_plan_ref_numel(h) == 32_416_643 + 64  # parameters + the sinusoid buffer
    assert lib.dmme_unet_plan_workspace_bytes(h) > 0 and lib.dmme_unet_plan_packed_bytes(h) > 2 * 32_416_643
    assert lib.dmme_unet_plan_dropmask_numel(h) == 128 * 4736
    n_ops = lib.dmme_unet_plan_num_ops(h)
    # launches = ops minus the GroupNorms finished by their producing convs' epilogues (8x8 / 4x4 / 16x16 whole-image tiles) or by
    # their consuming conv's parameter fill (the norms in front of the persistent 3x3 / the activation-stationary 1x1 kernel)
    n_launch = lib.dmme_unet_plan_num_launches(h)
    # ... and, since round 3, minus the layers of the 8x8 / 4x4 maps: three level-engine launches stand for 57 convs, norms and the
    # middle attention (csrc/lvl_engine.hip)
    buf = C.create_string_buffer(1024)
    assert lib.dmme_unet_plan_level_info(h, buf, 1024) == 0
    info = buf.value.decode()
    assert info.startswith("runs=3") and info.count("map=8x8") == 2 and info.count("map=4x4") == 1 and info.count("workgroups=256") == 3, info
    # ... and, since round 4, minus the seven 1x1 residual convs of the 32x32 / 16x16 levels that run inside their block's conv2 (51 now)
    # ... and, since round 5, minus the proj convs of the five 16x16 attention blocks, which run inside the attention launch (45 now)
    assert n_ops >= n_launch >= 43 and n_launch <= 50, (n_ops, n_launch)
    label = C.create_string_buffer(128)
    fl, by = C.c_double(), C.c_double()
    total = 0.0
    for i in range(n_ops):
        assert lib.dmme_unet_plan_op_info(h, i, label, 128, C.byref(fl), C.byref(by)) == 0
        total += fl.value
    # algorithmic FLOPs of one batch-128 forward: 128 x 9.809 GFLOP, time-MLP counted once
    assert abs(total / 128 / 9.809e9 - 1) < 0.01
    lib.dmme_unet_plan_destroy(h)
    # invalid arguments come back as a status + message, never an exception / crash
    cfg.num_groups = 7
    assert lib.dmme_unet_plan_create(C.byref(cfg), 1, 32, 32, _lib.F32, -1, C.byref(h)) == -1
    assert b"num_groups" in lib.dmme_last_error()
    with pytest.raises(ValueError):
        _lib.check(-1, "x")
    with pytest.raises(NotImplementedError):
        _lib.check(-2, "x")


def _default_cfg():
    from dmme_amd import _lib

    cfg = _lib.UNetCfg()
    cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups, cfg.dropout = 3, 128, 512, 32, 0.1
    cfg.num_depths, cfg.num_blocks, cfg.num_attention_depths = 4, 2, 1
    for i, c in enumerate((128, 256, 256, 256)):
        cfg.channels_per_depth[i] = c
    cfg.attention_depths[0] = 2
    return cfg


def _plan_labels(batch, dtype, route=None):
    from dmme_amd import _lib

    lib = _lib.lib()
    old = os.environ.get("DMME_DEBUG_ROUTE")
    if route is not None:
        os.environ["DMME_DEBUG_ROUTE"] = route
    try:
        h = C.c_void_p()
        cfg = _default_cfg()
        assert lib.dmme_unet_plan_create(C.byref(cfg), batch, 32, 32, dtype, -1, C.byref(h)) == 0
    finally:
        if route is not None:
            if old is None:
                os.environ.pop("DMME_DEBUG_ROUTE", None)
            else:
                os.environ["DMME_DEBUG_ROUTE"] = old
    label = C.create_string_buffer(128)
    fl, by = C.c_double(), C.c_double()
    out, total = [], 0.0
    for i in range(lib.dmme_unet_plan_num_ops(h)):
        assert lib.dmme_unet_plan_op_info(h, i, label, 128, C.byref(fl), C.byref(by)) == 0
        out.append(label.value.decode())
        total += fl.value
    n = lib.dmme_unet_plan_num_launches(h)
    lib.dmme_unet_plan_destroy(h)
    return out, n, total


def test_plan_fuses_residual_convs_into_conv2_where_the_persistent_kernel_runs_it():
    """host logic of assign_rseg (csrc/plan.hip), no GPU: at the benchmark batch the seven channel-changing ResBlocks of the 32x32 /
    16x16 levels lose their 1x1 launch (five on the 256-pixel tiles, two on the 128-pixel ones), at batch 32 the three whose conv2 the
    persistent kernel takes, at batch 1 none; never in fp32 / fp16r32 plans; the switch restores the launches; FLOPs are conserved."""
    from dmme_amd import _lib

    fused = lambda labels: sum(1 for l in labels if l.startswith("(residual 1x1 conv"))
    l128, n128, f128 = _plan_labels(128, _lib.BF16)
    l128_off, n128_off, f128_off = _plan_labels(128, _lib.BF16, "no_rseg")
    assert fused(l128) == 7 and fused(l128_off) == 0 and n128_off == n128 + 7
    assert sum(1 for l in l128 if l == "conv3x3_ws2_kernel<11,res>") == 5 and sum(1 for l in l128 if l == "conv3x3_ws2_kernel<7,128,res>") == 2
    assert abs(f128 / f128_off - 1) < 1e-12
    assert fused(_plan_labels(128, _lib.F16)[0]) == 7
    assert fused(_plan_labels(32, _lib.BF16)[0]) == 3
    assert fused(_plan_labels(1, _lib.BF16)[0]) == 0
    for dt in (_lib.F32, _lib.F16R32):
        assert fused(_plan_labels(128, dt)[0]) == 0
