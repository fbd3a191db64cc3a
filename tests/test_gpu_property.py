"""-m gpu: property test of the single-op convolution entry point (dmme_conv2d) over randomly drawn geometries - channel counts on and
off the MFMA kernels' 64 / 32-channel grid, square maps from 2x2 to 32x32, batch sizes that leave multi-image tiles partly empty,
stride 2, fused nearest upsample, channel concat, fused GroupNorm-affine / SiLU / Dropout2d prologue, time-embedding rows and residual
epilogue - against a plain fp64 torch convolution of the same operands, in all three precisions.  Whatever kernel the dispatcher picks
for a geometry (generic, first-generation MFMA, pipelined, persistent), the contract is the same."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

pytestmark = pytest.mark.gpu

CH = st.sampled_from([3, 4, 12, 32, 48, 64, 96, 128, 192, 256])


@st.composite
def conv_case(draw):
    k = draw(st.sampled_from([1, 3]))
    stride = draw(st.sampled_from([1, 1, 1, 2])) if k == 3 else 1
    up = draw(st.booleans()) if (k == 3 and stride == 1) else False
    H = draw(st.sampled_from([2, 4, 8, 16, 32]))
    if stride == 2 and H < 4:
        H = 4
    c1 = draw(CH)
    c2 = draw(st.sampled_from([0, 0, c1])) if c1 >= 32 else 0
    cout = draw(CH)
    n = draw(st.integers(1, 9))
    pro = draw(st.booleans()) and c1 > 4
    ntp = draw(st.sampled_from([0, 1, n])) if k == 3 else 0
    res = draw(st.booleans()) and cout >= 4
    return dict(k=k, stride=stride, up=up, H=H, c1=c1, c2=c2, cout=cout, n=n, pro=pro, ntp=ntp, res=res, seed=draw(st.integers(0, 10_000)))


def _run(case, dtname):
    from dmme_amd import _lib
    from tests import gpu_util as G

    g = torch.Generator().manual_seed(case["seed"])
    rn = lambda *s: torch.randn(*s, generator=g)
    N, C1, C2, H, Cout, k = case["n"], case["c1"], case["c2"], case["H"], case["cout"], case["k"]
    Cin = C1 + C2
    x1 = rn(N, C1, H, H)
    x2 = rn(N, C2, H, H) if C2 else None
    w = rn(Cout, Cin, k, k) / np.sqrt(Cin * k * k)
    b = 0.1 * rn(Cout)
    scale = 1 + 0.3 * rn(N, Cin) if case["pro"] else None
    shift = 0.2 * rn(N, Cin) if case["pro"] else None
    dmask = ((torch.rand(N, Cin, generator=g) < 0.9).float() / 0.9) if case["pro"] else None
    tproj = 0.3 * rn(case["ntp"], Cout) if case["ntp"] else None
    Ho = (2 * H if case["up"] else H) // case["stride"]
    res = rn(N, Cout, Ho, Ho) if case["res"] else None
    bf = (lambda t: t.to(torch.bfloat16).to(torch.float32)) if dtname == "bf16" else (lambda t: t)
    x = x1 if x2 is None else torch.cat([x1, x2], 1)
    xr = bf(x)  # (the helper hands every input over as an NHWC tensor in the compute dtype)
    if scale is not None:
        xr = F.silu(xr * scale[:, :, None, None] + shift[:, :, None, None]) * dmask[:, :, None, None]
        xr = bf(xr)
    if case["up"]:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    want = F.conv2d(xr.double(), bf(w).double(), b.double(), stride=case["stride"], padding=k // 2)
    if tproj is not None:
        want = want + (tproj if tproj.shape[0] > 1 else tproj.expand(N, -1)).double()[:, :, None, None]
    if res is not None:
        want = want + bf(res).double()
    cu = lambda t: None if t is None else t.cuda()
    got = G.conv2d(_lib.dtype_code(dtname), cu(x1), cu(w), cu(b), cu(x2), cu(scale), cu(shift), cu(dmask), cu(tproj), cu(res), case["stride"], case["up"],
                   case["pro"], False, 0)
    err = float((got.double().cpu() - want).abs().max())
    ref = max(1.0, float(want.abs().max()))
    tol = {"fp32": 1e-5, "bf16x3": 1e-4, "bf16": 2.0**-8}[dtname] * ref
    assert err <= tol, f"{dtname} {case}: max err {err:.3e} > {tol:.3e}"


@pytest.mark.parametrize("dtname", ["fp32", "bf16", "bf16x3"])
@settings(max_examples=100, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(case=conv_case())
def test_conv2d_random_geometries(dtname, case):
    _run(case, dtname)
