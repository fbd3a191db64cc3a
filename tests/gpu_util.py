"""Helpers for the -m gpu parity tests: call the single-op C-ABI entry points with torch
tensors (NCHW fp32 at the Python level, NHWC compute dtype inside)."""

import ctypes as C

import torch

from dmme_amd import _lib

TD = {_lib.F32: torch.float32, _lib.BF16: torch.bfloat16, _lib.BF16X3: torch.float32, _lib.F16: torch.float16, _lib.F16R32: torch.float32}


def _st(dt):
    """storage dtype code of a compute dtype (the accurate mode keeps fp32 buffers)"""
    return _lib.F32 if dt in (_lib.BF16X3, _lib.F16R32) else dt  # (fp16r32 single ops: the fp32 level's tensors)


def dev():
    return torch.device("cuda:0")


def to_nhwc(x_nchw, dt):
    """fp32 NCHW (cuda) -> NHWC tensor in the compute dtype through dmme_nchw_to_nhwc"""
    N, Cc, H, W = x_nchw.shape
    out = torch.empty((N, H, W, Cc), dtype=TD[dt], device=x_nchw.device)
    _lib.check(_lib.lib().dmme_nchw_to_nhwc(_st(dt), _lib.ptr(x_nchw.contiguous()), N, Cc, H * W, _lib.ptr(out), _lib.stream_ptr()))
    return out


def to_nchw(x_nhwc, dt):
    N, H, W, Cc = x_nhwc.shape
    out = torch.empty((N, Cc, H, W), dtype=torch.float32, device=x_nhwc.device)
    _lib.check(_lib.lib().dmme_nhwc_to_nchw(_st(dt), _lib.ptr(x_nhwc), N, Cc, H * W, _lib.ptr(out), _lib.stream_ptr()))
    return out


def pack_w(w_oihw, dt):
    co, ci, k, _ = w_oihw.shape
    if dt == _lib.F16R32:  # hi / lo halves: [co][taps][ci / 32][hi 32 | lo 32]
        out = torch.empty((co, k * k, 2 * ci), dtype=torch.float16, device=w_oihw.device)
        _lib.check(_lib.lib().dmme_pack_weight(dt, _lib.ptr(w_oihw.contiguous()), co, ci, k * k, _lib.ptr(out), _lib.stream_ptr()))
        return out
    out = torch.empty((co, k * k, ci), dtype=TD[dt], device=w_oihw.device)
    _lib.check(_lib.lib().dmme_pack_weight(_st(dt), _lib.ptr(w_oihw.contiguous()), co, ci, k * k, _lib.ptr(out), _lib.stream_ptr()))
    return out


def conv2d(dt, x1, w, b, x2=None, scale=None, shift=None, dmask=None, tproj=None, res=None, stride=1, upsample=False,
           pro_silu=False, out_silu=False, force_generic=False, out_nchw=False):
    """x1/x2/res: fp32 NCHW cuda tensors; w: (Cout, Cin, k, k); returns fp32 NCHW (out_nchw: written that way by the kernel itself,
    like the network's output conv)."""
    N, C1, H, W = x1.shape
    k = w.shape[-1]
    d = _lib.ConvDesc()
    d.dtype, d.N, d.Hin, d.Win, d.C1 = dt, N, H, W, C1
    d.C2 = 0 if x2 is None else x2.shape[1]
    d.upsample, d.stride, d.taps, d.Cout = int(upsample), stride, k * k, w.shape[0]
    d.pro_silu, d.out_silu = int(pro_silu), int(out_silu)
    d.nt = 0 if tproj is None else tproj.shape[0]
    d.tproj_ld = 0 if tproj is None else tproj.shape[1]
    d.in_nchw, d.out_nchw = 0, int(out_nchw)
    d.force_generic = int(force_generic)
    Hv, Wv = (2 * H, 2 * W) if upsample else (H, W)
    Ho, Wo = Hv // stride, Wv // stride
    a1 = to_nhwc(x1, dt)
    a2 = None if x2 is None else to_nhwc(x2, dt)
    r1 = None if res is None else to_nhwc(res, dt)
    wp = pack_w(w, dt)
    out = (torch.empty((N, w.shape[0], Ho, Wo), dtype=torch.float32, device=x1.device) if out_nchw
           else torch.empty((N, Ho, Wo, w.shape[0]), dtype=TD[dt], device=x1.device))
    f = lambda t: None if t is None else t.to(torch.float32).contiguous()
    sc, sh, dm, tp, bb = f(scale), f(shift), f(dmask), f(tproj), f(b)
    _lib.check(
        _lib.lib().dmme_conv2d(C.byref(d), _lib.ptr(a1), _lib.ptr(a2), _lib.ptr(wp), _lib.ptr(bb), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(dm),
                               _lib.ptr(tp), _lib.ptr(r1), _lib.ptr(None), w.shape[0], _lib.ptr(out), _lib.stream_ptr()),
        "dmme_conv2d",
    )
    return out if out_nchw else to_nchw(out, dt)


def conv2d_res(dt, x1, w, b, r1, wr, br, x2=None, r2=None, scale=None, shift=None, dmask=None, pro_silu=False):
    """dmme_conv2d_res: conv3x3(prologue(x1 ++ x2)) + conv1x1(r1 ++ r2) in one launch.  fp32 NCHW cuda tensors in, fp32 NCHW out."""
    N, C1, H, W = x1.shape
    d = _lib.ConvDesc()
    d.dtype, d.N, d.Hin, d.Win, d.C1 = dt, N, H, W, C1
    d.C2 = 0 if x2 is None else x2.shape[1]
    d.upsample, d.stride, d.taps, d.Cout = 0, 1, 9, w.shape[0]
    d.pro_silu, d.out_silu = int(pro_silu), 0
    d.nt = d.tproj_ld = d.in_nchw = d.out_nchw = d.force_generic = 0
    a1 = to_nhwc(x1, dt)
    a2 = None if x2 is None else to_nhwc(x2, dt)
    q1 = to_nhwc(r1, dt)
    q2 = None if r2 is None else to_nhwc(r2, dt)
    wp, wrp = pack_w(w, dt), pack_w(wr, dt)
    out = torch.empty((N, H, W, w.shape[0]), dtype=TD[dt], device=x1.device)
    f = lambda t: None if t is None else t.to(torch.float32).contiguous()
    sc, sh, dm, bb, bbr = f(scale), f(shift), f(dmask), f(b), f(br)
    _lib.check(
        _lib.lib().dmme_conv2d_res(C.byref(d), _lib.ptr(a1), _lib.ptr(a2), _lib.ptr(wp), _lib.ptr(bb), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(dm),
                                   _lib.ptr(q1), _lib.ptr(q2), r1.shape[1], 0 if r2 is None else r2.shape[1], _lib.ptr(wrp), _lib.ptr(bbr),
                                   _lib.ptr(out), _lib.stream_ptr()),
        "dmme_conv2d_res",
    )
    return to_nchw(out, dt)


def gn_scale_shift(dt, x1, gamma, beta, groups, x2=None, force_generic=False, eps=1e-5):
    N, C1, H, W = x1.shape
    C2 = 0 if x2 is None else x2.shape[1]
    a1 = to_nhwc(x1, dt)
    a2 = None if x2 is None else to_nhwc(x2, dt)
    scale = torch.empty((N, C1 + C2), dtype=torch.float32, device=x1.device)
    shift = torch.empty_like(scale)
    scratch = torch.empty(N * H * W * groups * 2 + 1024, dtype=torch.float32, device=x1.device)
    _lib.check(
        _lib.lib().dmme_groupnorm_scale_shift(dt, _lib.ptr(a1), _lib.ptr(a2), N, H * W, C1, C2, groups, _lib.ptr(gamma.contiguous()),
                                              _lib.ptr(beta.contiguous()), eps, _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(scratch),
                                              int(force_generic), _lib.stream_ptr()),
        "dmme_groupnorm_scale_shift",
    )
    return scale, shift


def attention(dt, qkv_nsc, force_generic=False):
    """qkv: (N, S, 3C) fp32 cuda -> (N, S, C) fp32"""
    N, S, C3 = qkv_nsc.shape
    q = qkv_nsc.to(TD[dt]).contiguous()
    out = torch.empty((N, S, C3 // 3), dtype=TD[dt], device=q.device)
    _lib.check(_lib.lib().dmme_attention(dt, _lib.ptr(q), N, S, C3 // 3, _lib.ptr(out), int(force_generic), _lib.stream_ptr()), "dmme_attention")
    return out.to(torch.float32)


# ---- A/B route switches ------------------------------------------------------------------------------------------------------------
# The library reads a dozen documented product switches as environment variables of their own; every other
# experiment / comparison route lives behind ONE variable, DMME_DEBUG_ROUTE="key[=int],...".  Tests name all of them the historical
# way (DMME_NO_WS128, DMME_LVL_MASK=4, ...); this context manager sets each the way the library reads it.
PRODUCT_SWITCHES = {"DMME_NO_LVL", "DMME_NO_WS", "DMME_NO_KW", "DMME_NO_CONV1X1_AS", "DMME_NO_FUSED_GN", "DMME_NO_GN_IN", "DMME_NO_GN_DIRECT", "DMME_NO_PREACT",
                    "DMME_NO_CONV_THIN", "DMME_NO_ATTN_FULL", "DMME_NO_WGRAD_GROUP", "DMME_NO_GN_BWD_REGS", "DMME_NO_GRAD_BUCKETS"}


class route_env:
    def __init__(self, kv):
        self.kv = dict(kv) if not isinstance(kv, str) else {kv: "1"}

    def __enter__(self):
        import os

        self._saved_route = os.environ.get("DMME_DEBUG_ROUTE")
        keys = [] if not self._saved_route else [self._saved_route]
        self._set = []
        for k, v in self.kv.items():
            if k in PRODUCT_SWITCHES or not k.startswith("DMME_"):
                os.environ[k] = str(v)
                self._set.append(k)
            else:
                keys.append(k[5:].lower() + ("" if str(v) == "1" else f"={v}"))
        if keys:
            os.environ["DMME_DEBUG_ROUTE"] = ",".join(keys)
        return self

    def __exit__(self, *a):
        import os

        for k in self._set:
            os.environ.pop(k, None)
        if self._saved_route is None:
            os.environ.pop("DMME_DEBUG_ROUTE", None)
        else:
            os.environ["DMME_DEBUG_ROUTE"] = self._saved_route
