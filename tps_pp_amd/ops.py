"""Tensor-level wrappers of the C ABI (include/tpspp.h).

Each function mirrors one PyTorch call site of the reference (cited in the docstring) and hands raw
device pointers + the current HIP stream to libtpspp_hip.so.  Inputs must be fp32 CUDA(HIP) tensors;
anything else raises -- there is no eager / CPU fallback on purpose.
"""
import torch

from . import _lib


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _chk(name, t, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise _lib.TpsppError(f"{name}: tensor is on {t.device}; the HIP path needs a GPU tensor "
                              "(no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got {tuple(t.shape)}")
    return t.contiguous()


def require_gpu(t, who, training=False):
    """Module forwards call this first: the product has no CPU path and no library-kernel path."""
    if not t.is_cuda:
        raise _lib.TpsppError(f"{who}: tensor is on {t.device}; the HIP path needs a GPU tensor "
                              "(no CPU fallback)")
    if training:
        raise NotImplementedError(f"{who} (HIP path) is forward-only: call .eval() and run under "
                                  "torch.no_grad() (SURVEY.md section 8f, row F2)")


def warn_detached_once(module, who):
    """Eval mode with autograd enabled and trainable parameters (the usual frozen-BatchNorm fine-tuning set-up): the HIP
    inference path records no graph, so a loss computed from its output has no grad_fn.  Said once per module instead of
    failing silently; `.train()` (or an input that requires grad) selects the PyTorch training graph."""
    if not torch.is_grad_enabled() or getattr(module, "_warned_detached", False):
        return
    if any(p.requires_grad for p in module.parameters()):
        import logging
        logging.getLogger("tps_pp_amd").warning(
            "%s in eval mode with autograd enabled: the HIP inference path records no graph, so no gradient will reach "
            "this module's parameters (call .train() -- forward_train needs it --, or wrap inference in torch.no_grad())", who)
    module._warned_detached = True


def solve_T(inv_delta_C, ctrl):
    """torch.bmm(inv_delta_C.repeat(N,1,1), cat(ctrl, zeros(N,3,2)))  -> (N, F+3, 2)
    (tps_preprocessor.py:273-280, tps_pp.py:484-494)."""
    inv_delta_C, ctrl = _chk("inv_delta_C", inv_delta_C, 2), _chk("ctrl", ctrl, 3)
    N, F, two = ctrl.shape
    if two != 2 or tuple(inv_delta_C.shape) != (F + 3, F + 3):
        raise ValueError("solve_T: shape mismatch")
    T = torch.empty((N, F + 3, 2), device=ctrl.device, dtype=torch.float32)
    with torch.cuda.device(ctrl.device):
        rc = _lib.lib().tpspp_solve_T(_ptr(inv_delta_C), _ptr(ctrl), N, F, _ptr(T), _stream(ctrl))
    _lib.check(rc, "tpspp_solve_T")
    return T


def build_grid(P_hat, T, P_xy=None, score=None):
    """torch.bmm(batch_P_hat, T) -> (N, n, 2); with P_xy/score the TPS_PP form
    `cat[1, P, P_hat*(score*0.5+1)] @ T` (tps_preprocessor.py:281, tps_pp.py:467-479,495)."""
    P_hat, T = _chk("P_hat", P_hat, 2), _chk("T", T, 3)
    N, K, _ = T.shape
    F, n = K - 3, P_hat.shape[0]
    if P_xy is not None:
        P_xy = _chk("P_xy", P_xy, 2)
    if score is not None:
        score = _chk("score", score, 3)
        if tuple(score.shape) != (N, n, F):
            raise ValueError("build_grid: score must be (N, n, F)")
    if P_hat.shape[1] != (F if P_xy is not None else F + 3):
        raise ValueError("build_grid: P_hat has the wrong number of columns")
    grid = torch.empty((N, n, 2), device=T.device, dtype=torch.float32)
    with torch.cuda.device(T.device):
        rc = _lib.lib().tpspp_build_grid(_ptr(P_hat), P_hat.shape[1], _ptr(P_xy), _ptr(score),
                                         _ptr(T), N, n, F, _ptr(grid), _stream(T))
    _lib.check(rc, "tpspp_build_grid")
    return grid


def grid_sample(inp, grid, return_idx=False):
    """F.grid_sample(inp, grid, mode='bilinear', padding_mode='border', align_corners=True)
    (tps_preprocessor.py:79-83, tps_pp.py:606-615).  grid: (N, Ho, Wo, 2)."""
    inp, grid = _chk("input", inp, 4), _chk("grid", grid, 4)
    N, C, H, W = inp.shape
    if grid.shape[0] != N or grid.shape[3] != 2:
        raise ValueError("grid_sample: grid must be (N, Ho, Wo, 2)")
    Ho, Wo = int(grid.shape[1]), int(grid.shape[2])
    out = torch.empty((N, C, Ho, Wo), device=inp.device, dtype=torch.float32)
    idx = torch.empty((N, Ho * Wo, 2), device=inp.device, dtype=torch.int32) if return_idx else None
    with torch.cuda.device(inp.device):
        rc = _lib.lib().tpspp_grid_sample(_ptr(inp), _ptr(grid), N, C, H, W, Ho, Wo, _ptr(out),
                                          _ptr(idx), _stream(inp))
    _lib.check(rc, "tpspp_grid_sample")
    return (out, idx) if return_idx else out


def transpose_p_hat(P_hat):
    """(n, cols) -> (cols, n) device copy for the coalesced kernels (one-off, per module buffer)."""
    P_hat = _chk("P_hat", P_hat, 2)
    n, cols = P_hat.shape
    out = torch.empty((cols, n), device=P_hat.device, dtype=torch.float32)
    with torch.cuda.device(P_hat.device):
        rc = _lib.lib().tpspp_transpose_p_hat(_ptr(P_hat), cols, n, cols, _ptr(out), _stream(P_hat))
    _lib.check(rc, "tpspp_transpose_p_hat")
    return out


TABLE_MIRROR4 = 1
SCORE_TRANSPOSED = 2
IO_BF16 = 4              # TPSPP_IO_BF16: in0 / in1 / out0 / out1 are bfloat16
BWD_FIXED_POINT = 16     # TPSPP_BWD_FIXED_POINT: tpspp_warp_bwd accumulates dL/d input in 64-bit fixed point (this call only)
BWD_TWO_KERNELS = 64     # TPSPP_BWD_TWO_KERNELS: tpspp_warp_bwd never takes the classic one-launch form (this call only)
TABLE_PACKED = 8         # TPSPP_TABLE_PACKED: P_hat_t is the head of a prepare_mirror_table() buffer
TABLE_SPAN = 32          # TPSPP_TABLE_SPAN: ... of the current three-section layout (the span-staging kernel reads the third)


def prepare_mirror_table(P_hat, out_hw):
    """One-off preparation of a classic-layout table for the image-pair kernel (`tpspp_prepare_mirror_table`):
    returns `(P_hat_t, flags)` where `P_hat_t` is the usual (F+3, n) transposed table -- a view of the head of a
    larger buffer whose tail is the packed copy -- and `flags` is `TABLE_PACKED`; for a geometry without a
    prepared form: `(transpose_p_hat(P_hat), 0)`.  OR `TABLE_MIRROR4` in once the symmetry has been verified.
    `flags` is `TABLE_PACKED | TABLE_SPAN`: the buffer has the current three-section layout."""
    P_hat = _chk("P_hat", P_hat, 2)
    n, cols = P_hat.shape
    Ho, Wo = int(out_hw[0]), int(out_hw[1])
    if n != Ho * Wo:
        raise ValueError("prepare_mirror_table: P_hat must have Ho * Wo rows")
    total = int(_lib.lib().tpspp_prepared_table_floats(Ho, Wo, cols - 3)) if cols > 3 else 0
    if total == 0:
        return transpose_p_hat(P_hat), 0
    buf = torch.empty((total,), device=P_hat.device, dtype=torch.float32)
    with torch.cuda.device(P_hat.device):
        rc = _lib.lib().tpspp_prepare_mirror_table(_ptr(P_hat), cols, Ho, Wo, cols - 3, _ptr(buf), _stream(P_hat))
    _lib.check(rc, "tpspp_prepare_mirror_table")
    return buf[:cols * n].view(cols, n), TABLE_PACKED | TABLE_SPAN


def _chk_table(who, P_hat, P_hat_t, n, out_hw, table_flags):
    """P_hat_t must be P_hat transposed; with TABLE_PACKED it must be the head of a prepare_mirror_table buffer
    (the kernel reads the packed copy behind it: anything else would be an out-of-bounds read)."""
    P_hat_t = _chk("P_hat_t", P_hat_t, 2)
    if tuple(P_hat_t.shape) != (P_hat.shape[1], n) or P_hat_t.device != P_hat.device:
        raise ValueError(f"{who}: P_hat_t must be P_hat transposed, on the same device")
    if int(table_flags) & TABLE_PACKED:
        need = int(_lib.lib().tpspp_prepared_table_floats(int(out_hw[0]), int(out_hw[1]), P_hat.shape[1] - 3))
        base = P_hat_t._base
        if need == 0 or base is None or base.numel() < need or base.data_ptr() != P_hat_t.data_ptr():
            raise ValueError(f"{who}: TABLE_PACKED needs the P_hat_t that prepare_mirror_table() returned")
    return P_hat_t


def _chk_out(who, name, o, shape, dtype, device):
    """Caller-supplied output buffers reach the kernels as raw pointers: everything is checked here."""
    if not isinstance(o, torch.Tensor) or tuple(o.shape) != tuple(shape) or o.dtype != dtype \
            or o.device != device or not o.is_contiguous():
        got = (tuple(o.shape), o.dtype, str(o.device)) if isinstance(o, torch.Tensor) else type(o)
        raise ValueError(f"{who}: {name} must be a contiguous {dtype} tensor of shape {tuple(shape)} on {device}, got {got}")
    return o


def table_mirror_symmetry(P_hat_host, out_hw, F):
    """1 if the HOST tensor/array `P_hat_host` (n, F+3) has the exact 4-fold mirror symmetry of the
    reference's GridGenerator table (then `table_flags=TABLE_MIRROR4` may be passed to warp)."""
    import numpy as np
    a = np.ascontiguousarray(P_hat_host.detach().cpu().numpy() if isinstance(P_hat_host, torch.Tensor)
                             else P_hat_host, dtype=np.float32)
    if a.ndim != 2 or a.shape[0] != out_hw[0] * out_hw[1]:
        return 0
    return int(_lib.lib().tpspp_table_mirror_symmetry(a.ctypes.data, a.shape[1], int(out_hw[0]),
                                                      int(out_hw[1]), int(F)))


def warp(in0, ctrl, inv_delta_C, P_hat, out_hw, P_xy=None, score=None, in1=None,
         want_grid=False, want_idx=False, out0=None, out1=None, P_hat_t=None, table_flags=0):
    """Fused build_P_prime + grid_sample(s): one kernel, T in LDS, grid in registers.

    Returns (out0, out1 | None, grid | None, idx | None).
    classic:  in0 = image, P_hat = GridGenerator.P_hat (n, F+3)     (tps_preprocessor.py:71-83)
    TPS_PP :  in0 = feat_grid, in1 = x, P_hat (n, F) + P_xy (n, 2) + score (N, n, F)
                                                                     (tps_pp.py:597-615)
    """
    io16 = isinstance(in0, torch.Tensor) and in0.dtype == torch.bfloat16
    if io16:          # bf16 images in and out (TPSPP_IO_BF16): T, grid and interpolation stay fp32
        if in1 is not None and in1.dtype != torch.bfloat16:
            raise TypeError("warp: in0 and in1 must share their dtype")
        in0 = _chk16("in0", in0, 4)
        table_flags = int(table_flags) | IO_BF16
    else:
        in0 = _chk("in0", in0, 4)
    io_dtype = torch.bfloat16 if io16 else torch.float32
    ctrl = _chk("ctrl", ctrl, 3)
    inv_delta_C, P_hat = _chk("inv_delta_C", inv_delta_C, 2), _chk("P_hat", P_hat, 2)
    N, C0, H0, W0 = in0.shape
    F = int(ctrl.shape[1])
    Ho, Wo = int(out_hw[0]), int(out_hw[1])
    n = Ho * Wo
    if ctrl.shape[0] != N or ctrl.shape[2] != 2:
        raise ValueError("warp: ctrl must be (N, F, 2)")
    if tuple(inv_delta_C.shape) != (F + 3, F + 3):
        raise ValueError("warp: inv_delta_C must be (F+3, F+3)")
    if P_xy is not None:
        P_xy = _chk("P_xy", P_xy, 2)
        if tuple(P_xy.shape) != (n, 2):
            raise ValueError("warp: P_xy must be (n, 2)")
    if tuple(P_hat.shape) != (n, F if P_xy is not None else F + 3):
        raise ValueError(f"warp: P_hat has shape {tuple(P_hat.shape)}")
    if score is not None:
        # (N, n, F) as the reference produces it, or a transposed VIEW of an (N, F, n) buffer: the
        # latter is what lets lanes that own consecutive pixels read the score coalesced
        if not isinstance(score, torch.Tensor) or score.dim() != 3 or tuple(score.shape) != (N, n, F):
            raise ValueError("warp: score must be (N, n, F)")
        if score.stride() == (F * n, 1, n) and n > 1 and F > 1:
            table_flags = int(table_flags) | SCORE_TRANSPOSED
            score = _chk("score", score.transpose(1, 2), 3)      # the underlying (N, F, n) buffer
        else:
            score = _chk("score", score, 3)
    if P_hat_t is not None:
        P_hat_t = _chk_table("warp", P_hat, P_hat_t, n, (Ho, Wo), table_flags)
    elif int(table_flags) & TABLE_PACKED:
        raise ValueError("warp: TABLE_PACKED without P_hat_t")
    C1 = H1 = W1 = 0
    if in1 is not None:
        in1 = _chk16("in1", in1, 4) if io16 else _chk("in1", in1, 4)
        if in1.shape[0] != N:
            raise ValueError("warp: in1 batch mismatch")
        _, C1, H1, W1 = in1.shape
    dev = in0.device
    if out0 is None:
        out0 = torch.empty((N, C0, Ho, Wo), device=dev, dtype=io_dtype)
    if in1 is not None and out1 is None:
        out1 = torch.empty((N, C1, Ho, Wo), device=dev, dtype=io_dtype)
    _chk_out("warp", "out0", out0, (N, C0, Ho, Wo), io_dtype, dev)
    if in1 is not None:
        if in1.device != dev:
            raise ValueError("warp: in1 must be on in0's device")
        _chk_out("warp", "out1", out1, (N, C1, Ho, Wo), io_dtype, dev)
    elif out1 is not None:
        raise ValueError("warp: out1 given without in1")
    for nm, t in (("ctrl", ctrl), ("inv_delta_C", inv_delta_C), ("P_hat", P_hat), ("P_xy", P_xy), ("score", score)):
        if t is not None and t.device != dev:
            raise ValueError(f"warp: {nm} must be on in0's device")
    grid = torch.empty((N, n, 2), device=dev, dtype=torch.float32) if want_grid else None
    idx = torch.empty((N, n, 2), device=dev, dtype=torch.int32) if want_idx else None
    with torch.cuda.device(dev):
        rc = _lib.lib().tpspp_warp_fwd(_ptr(in0), C0, H0, W0, _ptr(in1), C1, H1, W1, _ptr(ctrl),
                                       _ptr(score), _ptr(inv_delta_C), _ptr(P_hat), P_hat.shape[1],
                                       _ptr(P_xy), _ptr(P_hat_t), int(table_flags), N, F, Ho, Wo,
                                       _ptr(out0), _ptr(out1),
                                       _ptr(grid), _ptr(idx), _stream(in0))
    _lib.check(rc, "tpspp_warp_fwd")
    return out0, out1, grid, idx


class WarpPlan:
    """A validated, pre-marshalled `tpspp_warp_fwd` call on fixed buffers: `ops.warp` checks its tensors and
    marshals 25 arguments on every call (~12 us of Python, about one launch period of the classic geometry);
    a caller that rectifies batch after batch into the same buffers (a serving loop, bench.py) builds the
    plan once and pays one foreign call per batch.  The tensors are kept alive by the plan; the launch goes to
    the stream that was current on `in0.device` when the plan was built, and that device must be the
    thread's current device when `run()` is called (one process per GPU: always true)."""

    def __init__(self, in0, ctrl, inv_delta_C, P_hat, out_hw, out0, P_xy=None, score=None, in1=None, out1=None,
                 P_hat_t=None, table_flags=0):
        import ctypes
        in0, ctrl = _chk("in0", in0, 4), _chk("ctrl", ctrl, 3)
        inv_delta_C, P_hat = _chk("inv_delta_C", inv_delta_C, 2), _chk("P_hat", P_hat, 2)
        N, C0, H0, W0 = in0.shape
        F = int(ctrl.shape[1])
        Ho, Wo = int(out_hw[0]), int(out_hw[1])
        n = Ho * Wo
        if ctrl.shape[0] != N or ctrl.shape[2] != 2 or tuple(inv_delta_C.shape) != (F + 3, F + 3):
            raise ValueError("WarpPlan: ctrl must be (N, F, 2), inv_delta_C (F+3, F+3)")
        if P_xy is not None:
            P_xy = _chk("P_xy", P_xy, 2)
            if tuple(P_xy.shape) != (n, 2):
                raise ValueError("WarpPlan: P_xy must be (n, 2)")
        if tuple(P_hat.shape) != (n, F if P_xy is not None else F + 3):
            raise ValueError(f"WarpPlan: P_hat has shape {tuple(P_hat.shape)}")
        if score is not None:
            if tuple(score.shape) != (N, n, F):
                raise ValueError("WarpPlan: score must be (N, n, F)")
            if score.stride() == (F * n, 1, n) and n > 1 and F > 1:
                table_flags = int(table_flags) | SCORE_TRANSPOSED
                score = _chk("score", score.transpose(1, 2), 3)
            else:
                score = _chk("score", score, 3)
        if P_hat_t is not None:
            P_hat_t = _chk_table("WarpPlan", P_hat, P_hat_t, n, (Ho, Wo), table_flags)
        elif int(table_flags) & TABLE_PACKED:
            raise ValueError("WarpPlan: TABLE_PACKED without P_hat_t")
        dev = in0.device
        C1 = H1 = W1 = 0
        if in1 is not None:
            in1 = _chk("in1", in1, 4)
            _, C1, H1, W1 = in1.shape
            if in1.shape[0] != N or in1.device != dev:
                raise ValueError("WarpPlan: in1 must have in0's batch size and device")
            _chk_out("WarpPlan", "out1", out1, (N, C1, Ho, Wo), torch.float32, dev)
        elif out1 is not None:
            raise ValueError("WarpPlan: out1 given without in1")
        _chk_out("WarpPlan", "out0", out0, (N, C0, Ho, Wo), torch.float32, dev)
        for nm, t in (("ctrl", ctrl), ("inv_delta_C", inv_delta_C), ("P_hat", P_hat), ("P_xy", P_xy), ("score", score)):
            if t is not None and t.device != dev:
                raise ValueError(f"WarpPlan: {nm} must be on in0's device")
        self._keep = (in0, in1, ctrl, score, inv_delta_C, P_hat, P_xy, P_hat_t, out0, out1)
        vp, ci = ctypes.c_void_p, ctypes.c_int
        self._args = (vp(_ptr(in0)), ci(C0), ci(H0), ci(W0), vp(_ptr(in1)), ci(C1), ci(H1), ci(W1), vp(_ptr(ctrl)),
                      vp(_ptr(score)), vp(_ptr(inv_delta_C)), vp(_ptr(P_hat)), ci(P_hat.shape[1]), vp(_ptr(P_xy)),
                      vp(_ptr(P_hat_t)), ci(int(table_flags)), ci(N), ci(F), ci(Ho), ci(Wo), vp(_ptr(out0)),
                      vp(_ptr(out1)), vp(0), vp(0), vp(_stream(in0)))
        self._fn = _lib.lib().tpspp_warp_fwd
        self.out0, self.out1 = out0, out1
        self._dev = in0.device
        # round 6: the arguments live on the library's side (tpspp_warp_plan_create); run() hands over one pointer instead of
        # marshalling 25 arguments per launch (1.6 of ~4.3 us of host time per call)
        L = _lib.lib()
        handle = ctypes.c_void_p()
        _lib.check(L.tpspp_warp_plan_create(*self._args, ctypes.byref(handle)), "tpspp_warp_plan_create")
        self._handle, self._run, self._run_on, self._destroy = handle, L.tpspp_warp_plan_run, L.tpspp_warp_plan_run_on, L.tpspp_warp_plan_destroy

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h is not None and getattr(self, "_destroy", None) is not None:
            try:
                self._destroy(h)
            except Exception:                                  # noqa: BLE001  (interpreter shutdown)
                pass

    def run(self, stream=None):
        """One launch on the stream that was current when the plan was built; pass `stream` (a torch.cuda.Stream)
        to launch on another one (the caller orders the buffers' producers / consumers on it)."""
        if stream is not None:
            import ctypes
            rc = self._run_on(self._handle, ctypes.c_void_p(stream.cuda_stream))
        else:
            rc = self._run(self._handle)
        if rc != 0:
            _lib.check(rc, "tpspp_warp_fwd")
        return self.out0, self.out1


class ConvWeight:
    """Device-side weights of one fused convolution, prepared once: `wt` (K, Cout) for the generic
    kernel, `tiled` [chunk][tap][channel-in-chunk][Cout] for the tiled kernel, `bias` (Cout) | None."""

    def __init__(self, wt, tiled, bias, kernel, post_scale=None, post_shift=None):
        self.wt, self.tiled, self.bias, self.kernel = wt, tiled, bias, kernel
        self.post_scale, self.post_shift = post_scale, post_shift


def prep_conv_weight(weight, bn=None, conv_bias=None, eps=1e-5, src_channels=None, post_bn=None):
    """PyTorch conv weight (Cout, Cin, KH, KW) [+ eval-mode BatchNorm (gamma, beta, mean, var), folded:
    y = gamma * (conv(x) + b - mean) / sqrt(var + eps) + beta] -> ConvWeight.
    `src_channels`: channel counts of the concatenated sources (chunks never straddle a source).
    `post_bn`: a BatchNorm that FOLLOWS the activation (applied as a per-channel affine in the epilogue)."""
    w = weight.detach().float()
    b = None if conv_bias is None else conv_bias.detach().float()
    if bn is not None:
        gamma, beta, mean, var = (t.detach().float() for t in bn)
        scale = gamma / torch.sqrt(var + eps)
        w = w * scale.view(-1, 1, 1, 1)
        b = beta - mean * scale if b is None else beta + (b - mean) * scale
    cout, cin, kh, kw = w.shape
    wt = w.reshape(cout, -1).t().contiguous()
    kc = int(_lib.lib().tpspp_conv_chunk_channels(int(kh)))
    tiled = None
    if src_channels is None or len(src_channels) == 1 or all(c % kc == 0 for c in src_channels):
        nch = (cin + kc - 1) // kc
        wp = torch.zeros((cout, nch * kc, kh, kw), device=w.device, dtype=torch.float32)
        wp[:, :cin] = w
        # (cout, chunk, ci, ky, kx) -> (chunk, ky, kx, ci, cout)
        tiled = wp.view(cout, nch, kc, kh, kw).permute(1, 3, 4, 2, 0).contiguous()
    ps = pb = None
    if post_bn is not None:
        gamma, beta, mean, var = (t.detach().float() for t in post_bn)
        ps = (gamma / torch.sqrt(var + eps)).contiguous()
        pb = (beta - mean * ps).contiguous()
    return ConvWeight(wt, tiled, None if b is None else b.contiguous(), kh, ps, pb)


def _chk_conv_sources(who, ts, dims, stride_of_dims):
    """Every source reaches the kernel as a raw pointer with only its (C, H, W, uh, uw): batch size, device and the
    shared logical size (H*uh, W*uw) are checked here."""
    N, dev = ts[0].shape[0], ts[0].device
    Hi, Wi = ts[0].shape[2] * dims[3], ts[0].shape[3] * dims[4]
    for i, t in enumerate(ts):
        d = dims[stride_of_dims * i: stride_of_dims * (i + 1)]
        if t.shape[0] != N or t.device != dev:
            raise ValueError(f"{who}: source {i} must have the first source's batch size and device")
        if d[3] < 1 or d[4] < 1 or (t.shape[2] * d[3], t.shape[3] * d[4]) != (Hi, Wi):
            raise ValueError(f"{who}: source {i} does not have the shared logical size {(Hi, Wi)}")
    if not 1 <= len(ts) <= 3:
        raise ValueError(f"{who}: 1..3 sources")
    return N, dev, Hi, Wi


def conv2d(srcs, cw, stride=(1, 1), relu=True, residual=None, res_mode=0, out=None):
    """Fused conv on the fp32 matrix cores (`tpspp_conv2d_fwd`).

    srcs: list of 1..3 entries `tensor` or `(tensor, uh, uw)`: channel-concatenated, each nearest-
    upsampled by (uh, uw) on the fly.  cw: ConvWeight from `prep_conv_weight` ("same" padding).
    residual/res_mode: 1 = act(conv)+res, 2 = act(conv+res).  relu: False/0 none, True/1 ReLU, 2 GELU (erf)."""
    import ctypes
    weight_t, bias, kernel = cw.wt, cw.bias, cw.kernel
    ts, dims = [], []
    for e in srcs:
        t, uh, uw = (e, 1, 1) if isinstance(e, torch.Tensor) else e
        t = _chk("conv source", t, 4)
        ts.append(t)
        dims += [t.shape[1], t.shape[2], t.shape[3], int(uh), int(uw)]
    N, dev, Hi, Wi = _chk_conv_sources("conv2d", ts, dims, 5)
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    pad = (kernel - 1) // 2
    Ho = (Hi + 2 * pad - kernel) // sh + 1
    Wo = (Wi + 2 * pad - kernel) // sw + 1
    weight_t = _chk("weight_t", weight_t, 2)
    Cout = weight_t.shape[1]
    if weight_t.shape[0] != sum(t.shape[1] for t in ts) * kernel * kernel:
        raise ValueError("conv2d: weight_t rows != Cin*KH*KW")
    if bias is not None:
        bias = _chk("bias", bias, 1)
    if residual is not None:
        residual = _chk("residual", residual, 4)
        if tuple(residual.shape) != (N, Cout, Ho, Wo) or res_mode not in (1, 2) or residual.device != dev:
            raise ValueError("conv2d: residual shape / device / res_mode")
    elif res_mode != 0:
        raise ValueError("conv2d: res_mode without residual")
    if out is None:
        out = torch.empty((N, Cout, Ho, Wo), device=dev, dtype=torch.float32)
    else:
        _chk_out("conv2d", "out", out, (N, Cout, Ho, Wo), torch.float32, dev)
    if N == 0:          # an empty batch has no device pointer to hand over
        return out
    ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    dim_arr = (ctypes.c_int * len(dims))(*dims)
    with torch.cuda.device(ts[0].device):
        rc = _lib.lib().tpspp_conv2d_fwd(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(dim_arr, ctypes.c_void_p),
                                         len(ts), _ptr(weight_t), _ptr(cw.tiled), _ptr(bias), _ptr(residual),
                                         _ptr(cw.post_scale), _ptr(cw.post_shift), int(res_mode),
                                         int(relu), N, Cout, kernel, kernel, sh, sw, _ptr(out), Ho, Wo,
                                         _stream(ts[0]))
    _lib.check(rc, "tpspp_conv2d_fwd")
    return out


def _chk16(name, t, ndim=None):
    """bf16 path: a GPU tensor that is float32 or bfloat16."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise _lib.TpsppError(f"{name}: tensor is on {t.device}; the HIP path needs a GPU tensor "
                              "(no CPU fallback)")
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"{name}: expected float32 or bfloat16, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got {tuple(t.shape)}")
    return t.contiguous()


class ConvWeightBf16:
    """Device-side weights of one fused bf16 convolution: `arranged` bf16
    [Cout/64][Cin/KC][taps][KC/8][64][8] (include/tpspp.h, tpspp_conv2d_bf16_fwd), fp32 `bias` | None."""

    def __init__(self, arranged, bias, kernel, cin, cout, post_scale=None, post_shift=None, x3=False):
        self.arranged, self.bias, self.kernel, self.cin, self.cout = arranged, bias, kernel, cin, cout
        self.post_scale, self.post_shift, self.x3 = post_scale, post_shift, x3


def prep_conv_weight_bf16(weight, bn=None, conv_bias=None, eps=1e-5, post_bn=None, x3=False):
    """PyTorch conv weight (Cout, Cin, KH, KW) [+ eval-mode BatchNorm folded in fp32, then rounded] ->
    ConvWeightBf16.  Same folding rules as `prep_conv_weight`.  x3: the "bf16x3" split (hi and lo slabs per chunk)."""
    w = weight.detach().float()
    b = None if conv_bias is None else conv_bias.detach().float()
    if bn is not None:
        gamma, beta, mean, var = (t.detach().float() for t in bn)
        scale = gamma / torch.sqrt(var + eps)
        w = w * scale.view(-1, 1, 1, 1)
        b = beta - mean * scale if b is None else beta + (b - mean) * scale
    cout, cin, kh, kw = w.shape
    kc = int(_lib.lib().tpspp_conv_bf16_chunk_channels(int(kh)))
    ct, nch = (cout + 63) // 64, (cin + kc - 1) // kc
    wp = torch.zeros((ct * 64, nch * kc, kh, kw), device=w.device, dtype=torch.float32)
    wp[:cout, :cin] = w
    # (ctile, co, chunk, kgroup, k8, ky, kx) -> (ctile, chunk, ky, kx, kgroup, co, k8)
    arranged = wp.view(ct, 64, nch, kc // 8, 8, kh, kw).permute(0, 2, 5, 6, 3, 1, 4).contiguous()
    if x3:          # [ctile][chunk][hi|lo][tap...]: hi = bf16(w), lo = bf16(w - hi)
        hi = arranged.to(torch.bfloat16)
        lo = (arranged - hi.float()).to(torch.bfloat16)
        arranged = torch.stack([hi, lo], dim=2).contiguous()
    else:
        arranged = arranged.to(torch.bfloat16)
    ps = pb = None
    if post_bn is not None:
        gamma, beta, mean, var = (t.detach().float() for t in post_bn)
        ps = (gamma / torch.sqrt(var + eps)).contiguous()
        pb = (beta - mean * ps).contiguous()
    return ConvWeightBf16(arranged, None if b is None else b.contiguous(), kh, cin, cout, ps, pb, x3)


class Blocked:
    """A bf16 feature map in the blocked layout (N, C/8, H, W, 8) -- the eight channels of a group next to each other
    per pixel -- that the bf16 convolutions exchange among themselves (`tpspp_conv2d_bf16_fwd`, layout code 2: a
    16-byte unit of it is a unit of the kernel's LDS patch).  `shape` is the logical (N, C, H, W)."""

    def __init__(self, t):
        if t.dtype != torch.bfloat16 or t.dim() != 5 or t.shape[4] != 8 or not t.is_contiguous():
            raise ValueError("Blocked: needs a contiguous bfloat16 (N, C/8, H, W, 8) tensor")
        self.t = t
        self.shape = (t.shape[0], t.shape[1] * 8, t.shape[2], t.shape[3])
        self.dtype, self.device = t.dtype, t.device
        self.requires_grad, self.is_cuda = False, t.is_cuda      # (a product of the HIP inference kernels: no autograd graph)

    @staticmethod
    def from_nchw(x):
        n, c, h, w = x.shape
        return Blocked(x.to(torch.bfloat16).reshape(n, c // 8, 8, h, w).permute(0, 1, 3, 4, 2).contiguous())

    def nchw(self):
        n, c, h, w = self.shape
        return self.t.permute(0, 1, 4, 2, 3).reshape(n, c, h, w).contiguous()

    def nchw_hip(self):
        """`tpspp_blocked_to_nchw_bf16`: the same tensor as `nchw()` by a HIP kernel (the product path; `nchw()` is a
        PyTorch composition for tests and edges); the result remembers its blocked twin (`_tpspp_blocked`) so that a
        convolution further on can take the cheaper source."""
        n, c, h, w = self.shape
        if not self.t.is_cuda or (h * w) % 64:
            raise _lib.TpsppError("Blocked.nchw_hip: needs a GPU tensor with H * W a multiple of 64 (no CPU fallback)")
        out = torch.empty((n, c, h, w), device=self.t.device, dtype=torch.bfloat16)
        with torch.cuda.device(self.t.device):
            rc = _lib.lib().tpspp_blocked_to_nchw_bf16(self.t.data_ptr(), n, c, h * w, out.data_ptr(), _stream(self.t))
        _lib.check(rc, "tpspp_blocked_to_nchw_bf16")
        out._tpspp_blocked = self
        return out

    def data_ptr(self):
        return self.t.data_ptr()


class Blocked32(Blocked):
    """The blocked layout in float32, (N, C/8, H, W, 8) fp32 -- what the maps of the three-term-split ("bf16x3") configuration
    use between convolutions (layout code 3: a patch position's 8 channels are two 16-byte loads instead of eight 4-byte
    loads from eight planes)."""

    def __init__(self, t):
        if t.dtype != torch.float32 or t.dim() != 5 or t.shape[4] != 8 or not t.is_contiguous():
            raise ValueError("Blocked32: needs a contiguous float32 (N, C/8, H, W, 8) tensor")
        self.t = t
        self.shape = (t.shape[0], t.shape[1] * 8, t.shape[2], t.shape[3])
        self.dtype, self.device = t.dtype, t.device

    @staticmethod
    def from_nchw(x):
        n, c, h, w = x.shape
        return Blocked32(x.float().reshape(n, c // 8, 8, h, w).permute(0, 1, 3, 4, 2).contiguous())


def _layout_code(t):
    if isinstance(t, Blocked32):
        return 3
    return 2 if isinstance(t, Blocked) else int(t.dtype == torch.float32)


def conv2d_bf16(srcs, cw, stride=(1, 1), relu=True, residual=None, res_mode=0, out_dtype=torch.bfloat16, out_blocked=False):
    """Fused conv on the bf16 matrix cores (`tpspp_conv2d_bf16_fwd`): same call shape as `conv2d`; every
    source / the residual may be float32 or bfloat16 NCHW or a `Blocked` bf16 map, the output is `out_dtype`
    (`out_blocked`: a `Blocked` bf16 map, for layers that only feed other convolutions)."""
    import ctypes
    ts, dims = [], []
    for e in srcs:
        t, uh, uw = e if isinstance(e, tuple) else (e, 1, 1)
        code = _layout_code(t)
        t = _BlkView(t) if isinstance(t, Blocked) else _chk16("conv source", t, 4)
        ts.append(t)
        dims += [t.shape[1], t.shape[2], t.shape[3], int(uh), int(uw), code]
    N, dev, Hi, Wi = _chk_conv_sources("conv2d_bf16", ts, dims, 6)
    if sum(t.shape[1] for t in ts) != cw.cin:
        raise ValueError("conv2d_bf16: source channels != Cin of the weight")
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    kernel, Cout = cw.kernel, cw.cout
    pad = (kernel - 1) // 2
    Ho = (Hi + 2 * pad - kernel) // sh + 1
    Wo = (Wi + 2 * pad - kernel) // sw + 1
    res_code = 0
    if residual is not None:
        if not isinstance(residual, Blocked):
            residual = _chk16("residual", residual, 4)
        if tuple(residual.shape) != (N, Cout, Ho, Wo) or res_mode not in (1, 2) or residual.device != dev:
            raise ValueError("conv2d_bf16: residual shape / device / res_mode")
        res_code = _layout_code(residual)
    elif res_mode != 0:
        raise ValueError("conv2d_bf16: res_mode without residual")
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("conv2d_bf16: out_dtype must be float32 or bfloat16")
    if out_blocked:
        if Cout % 8:
            raise ValueError("conv2d_bf16: a blocked output needs a multiple of 8 channels")
        if cw.x3:                                   # the three-term split: fp32 maps, fp32 blocked
            if out_dtype != torch.float32:
                raise ValueError("conv2d_bf16: with x3 weights a blocked output is float32 (Blocked32)")
            out = Blocked32(torch.empty((N, Cout // 8, Ho, Wo, 8), device=dev, dtype=torch.float32))
        else:
            if out_dtype != torch.bfloat16:
                raise ValueError("conv2d_bf16: a blocked output is bfloat16 (float32 with x3 weights)")
            out = Blocked(torch.empty((N, Cout // 8, Ho, Wo, 8), device=dev, dtype=torch.bfloat16))
    else:
        out = torch.empty((N, Cout, Ho, Wo), device=dev, dtype=out_dtype)
    codes = list(dims[5::6]) + [res_code]
    if (cw.x3 and 2 in codes) or (not cw.x3 and 3 in codes):
        raise ValueError("conv2d_bf16: Blocked (bfloat16) maps go with plain bf16 weights, Blocked32 (float32) maps with x3 weights")
    if N == 0:          # an empty batch has no device pointer to hand over
        return out
    ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    dim_arr = (ctypes.c_int * len(dims))(*dims)
    first = ts[0].keep if isinstance(ts[0], _BlkView) else ts[0]
    with torch.cuda.device(dev):
        rc = _lib.lib().tpspp_conv2d_bf16_fwd(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(dim_arr, ctypes.c_void_p),
                                              len(ts), _ptr(cw.arranged), _ptr(cw.bias),
                                              residual.data_ptr() if residual is not None else None, res_code,
                                              _ptr(cw.post_scale), _ptr(cw.post_shift), int(res_mode), int(relu),
                                              N, Cout, kernel, kernel, sh, sw, out.data_ptr(),
                                              (3 if cw.x3 else 2) if out_blocked else int(out_dtype == torch.float32), Ho, Wo, int(cw.x3),
                                              _stream(first))
    _lib.check(rc, "tpspp_conv2d_bf16_fwd")
    return out


class _BlkView:
    """What the shared source checks look at (logical shape, device, is_cuda, dtype, data_ptr) for a `Blocked` map."""

    def __init__(self, b):
        self.shape, self.device, self.dtype, self.is_cuda = b.shape, b.device, b.dtype, b.t.is_cuda
        self.keep = b.t

    def data_ptr(self):
        return self.keep.data_ptr()


def _mfma_feature_perm(device):
    """slot 2*ks + half -> input feature held by (accumulator register ks, half-wavefront `half`) in the
    C/D layout of v_mfma_f32_32x32x2_f32 (see tpspp_dgab.hip)."""
    perm = []
    for ks in range(32):
        for half in range(2):
            perm.append(32 * (ks >> 4) + (ks & 3) + 8 * ((ks & 15) >> 2) + 4 * half)
    return torch.tensor(perm, device=device, dtype=torch.long)


class DgabWeights:
    def __init__(self, blk):
        """blk: the DGAB module (norm1, attn.{mlp_h,mlp_w,proj}, norm2, mlp.{fc1,fc2})."""
        f = lambda t: t.detach().float().contiguous()
        dev = blk.norm1.weight.device
        perm = _mfma_feature_perm(dev)
        self.ln1_w, self.ln1_b = f(blk.norm1.weight), f(blk.norm1.bias)
        self.ln2_w, self.ln2_b = f(blk.norm2.weight), f(blk.norm2.bias)
        self.mw_t = f(blk.attn.mlp_w[0].weight.t())              # (96, 65)
        self.mh_t = f(blk.attn.mlp_h[0].weight.t())              # (48, 17)
        self.proj_slab = f(blk.attn.proj.weight[:, perm].t())    # [slot][out]
        self.proj_b = f(blk.attn.proj.bias)
        w1, w2 = blk.mlp.fc1.weight.detach().float(), blk.mlp.fc2.weight.detach().float()
        self.fc1_slab = torch.stack([w1[hb * 64:(hb + 1) * 64][:, perm].t() for hb in range(4)]).contiguous()
        self.fc2_slab = torch.stack([w2[:, hb * 64 + perm].t() for hb in range(4)]).contiguous()
        self.fc1_b, self.fc2_b = f(blk.mlp.fc1.bias), f(blk.mlp.fc2.bias)


def dgab(x, y, dw):
    """DGAB.forward (DGAB.py:74-77) fused: x (N, C, 16, 64), y (N, C, 32) -> (N, C, 16, 64)."""
    x, y = _chk("x", x, 4), _chk("y", y, 3)
    N, C, H, W = x.shape
    if (H, W) != (16, 64) or tuple(y.shape) != (N, C, 32):
        raise ValueError("dgab: needs x (N, C, 16, 64) and y (N, C, 32)")
    out = torch.empty_like(x)
    scratch = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = _lib.lib().tpspp_dgab_fwd(_ptr(x), _ptr(y), _ptr(dw.ln1_w), _ptr(dw.ln1_b), _ptr(dw.mw_t),
                                       _ptr(dw.mh_t), _ptr(dw.proj_slab), _ptr(dw.proj_b), _ptr(dw.ln2_w),
                                       _ptr(dw.ln2_b), _ptr(dw.fc1_slab), _ptr(dw.fc1_b), _ptr(dw.fc2_slab),
                                       _ptr(dw.fc2_b), _ptr(scratch), _ptr(out), N, C, _stream(x))
    _lib.check(rc, "tpspp_dgab_fwd")
    return out


_PERM16 = [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15]


def _bf16_slab(w, chain, x3=False):
    """(out 64, K) fp32 -> [K/16][2][64][8] bf16 A-operand slab of v_mfma_f32_32x32x16_bf16; `chain`: k-slots in the
    order in which the previous layer's result registers come back as operands (include/tpspp.h).  x3: the hi and lo
    halves of the three-term split, stacked [hi|lo][K/16][2][64][8]."""
    cout, cin = w.shape
    idx = torch.arange(cin, device=w.device).view(cin // 16, 16)
    if chain:
        idx = idx[:, torch.tensor(_PERM16, device=w.device)]
    a = w[:, idx.reshape(-1)].view(cout, cin // 16, 2, 8).permute(1, 2, 0, 3).contiguous()
    if not x3:
        return a.to(torch.bfloat16)
    hi = a.to(torch.bfloat16)
    return torch.stack([hi, (a - hi.float()).to(torch.bfloat16)]).contiguous()


class DgabWeightsBf16:
    def __init__(self, blk, x3=False):
        """blk: the DGAB module.  Slabs for tpspp_dgab_bf16_fwd (x3: hi + lo halves, split3)."""
        self.x3 = x3
        f = lambda t: t.detach().float().contiguous()          # noqa: E731
        self.ln1_w, self.ln1_b = f(blk.norm1.weight), f(blk.norm1.bias)
        self.ln2_w, self.ln2_b = f(blk.norm2.weight), f(blk.norm2.bias)
        self.mw_t = f(blk.attn.mlp_w[0].weight.t())
        self.mh_t = f(blk.attn.mlp_h[0].weight.t())
        self.proj_slab = _bf16_slab(f(blk.attn.proj.weight), False, x3)
        self.proj_b = f(blk.attn.proj.bias)
        w1, w2 = f(blk.mlp.fc1.weight), f(blk.mlp.fc2.weight)
        self.fc1_slab = torch.stack([_bf16_slab(w1[hb * 64:(hb + 1) * 64], True, x3) for hb in range(4)]).contiguous()
        self.fc2_slab = torch.stack([_bf16_slab(w2[:, hb * 64:(hb + 1) * 64], True, x3) for hb in range(4)]).contiguous()
        self.fc1_b, self.fc2_b = f(blk.mlp.fc1.bias), f(blk.mlp.fc2.bias)


def dgab_bf16(x, y, dw):
    """`dgab` with proj / fc1 / fc2 on the bf16 matrix cores (fp32 in, fp32 out)."""
    x, y = _chk("x", x, 4), _chk("y", y, 3)
    N, C, H, W = x.shape
    if (H, W) != (16, 64) or tuple(y.shape) != (N, C, 32):
        raise ValueError("dgab_bf16: needs x (N, C, 16, 64) and y (N, C, 32)")
    out = torch.empty_like(x)
    scratch = torch.empty(x.shape, device=x.device, dtype=torch.float32 if dw.x3 else torch.bfloat16)
    with torch.cuda.device(x.device):
        rc = _lib.lib().tpspp_dgab_bf16_fwd(_ptr(x), _ptr(y), _ptr(dw.ln1_w), _ptr(dw.ln1_b), _ptr(dw.mw_t),
                                            _ptr(dw.mh_t), _ptr(dw.proj_slab), _ptr(dw.proj_b), _ptr(dw.ln2_w),
                                            _ptr(dw.ln2_b), _ptr(dw.fc1_slab), _ptr(dw.fc1_b), _ptr(dw.fc2_slab),
                                            _ptr(dw.fc2_b), _ptr(scratch), _ptr(out), N, C, int(dw.x3), _stream(x))
    _lib.check(rc, "tpspp_dgab_bf16_fwd")
    return out


class ScoreWeights:
    def __init__(self, feat_linear):
        """feat_linear: nn.Sequential(Linear(64, 32), Linear(32, 128)) (tps_pp.py:257-260)."""
        l1, l2 = feat_linear[0], feat_linear[1]
        dev = l1.weight.device
        perm16 = torch.tensor([(ks & 3) + 8 * (ks >> 2) + 4 * half for ks in range(16) for half in range(2)],
                              device=dev, dtype=torch.long)
        self.w1_slab = l1.weight.detach().float().t().contiguous()              # [64][32]
        self.w2_slab = l2.weight.detach().float()[:, perm16].t().contiguous()   # [32 slots][128]
        self.b1 = l1.bias.detach().float().contiguous()
        self.b2 = l2.bias.detach().float().contiguous()
        # hi / lo slabs of the three-term split (tpspp_score_x3_fwd)
        self.w1_x3 = _bf16_slab(l1.weight.detach().float(), False, True)
        self.w2_x3 = _bf16_slab(l2.weight.detach().float(), True, True)


def score(de_feat, p, sw, scale, x3=False):
    """get_score (tps_pp.py:303-312) fused: de_feat (N, 64, H, W), p (N, 32, 128) -> (N, H*W, 32) as
    the transposed VIEW of an (N, 32, H*W) buffer.  x3: the three-term bf16 split in the three products."""
    de_feat, p = _chk("de_feat", de_feat, 4), _chk("p", p, 3)
    N, C, H, W = de_feat.shape
    if C != 64 or tuple(p.shape) != (N, 32, 128):
        raise ValueError("score: needs de_feat (N, 64, H, W) and p (N, 32, 128)")
    n = H * W
    out = torch.empty((N, 32, n), device=de_feat.device, dtype=torch.float32)
    with torch.cuda.device(de_feat.device):
        if x3:
            rc = _lib.lib().tpspp_score_x3_fwd(_ptr(de_feat), _ptr(sw.w1_x3), _ptr(sw.b1), _ptr(sw.w2_x3),
                                               _ptr(sw.b2), _ptr(p), float(scale), _ptr(out), N, n, _stream(de_feat))
        else:
            rc = _lib.lib().tpspp_score_fwd(_ptr(de_feat), _ptr(sw.w1_slab), _ptr(sw.b1), _ptr(sw.w2_slab),
                                            _ptr(sw.b2), _ptr(p), float(scale), _ptr(out), N, n, _stream(de_feat))
    _lib.check(rc, "tpspp_score_fwd")
    return out.transpose(1, 2)


class FrontWeights:
    def __init__(self, m):
        """m: TPS_PP (ResNet45v2 wiring): down0, down1, down2, down_feat ConvModules."""
        f = lambda t: t.detach().float().contiguous()
        perm = _mfma_feature_perm(m.down0.conv.weight.device)
        self.w0 = f(m.down0.conv.weight.view(64, 32).t())
        self.w1 = f(m.down1.conv.weight.view(64, 32).t())
        self.w2 = f(m.down2.conv.weight.view(64, 64).t())
        wg = m.down_feat.conv.weight.detach().float().view(64, 192)
        self.wg = torch.stack([wg[:, blk * 64 + perm].t() for blk in range(3)]).contiguous()
        self.b0, self.b1, self.b2, self.bg = (f(c.conv.bias) for c in (m.down0, m.down1, m.down2, m.down_feat))


def front(o0, o1, x, fw, store01=True):
    """down0/down1/down2 + grid() of TPS_PP.forward (tps_pp.py:560-562,581-585) in one kernel.
    Returns (feat0, feat1, feat2, feat_grid); `store01=False`: feat0 / feat1 stay operands of feat_grid and are returned as
    None (`down_fused_f32` recomputes them where they are consumed)."""
    o0, o1, x = _chk("outs[0]", o0, 4), _chk("outs[1]", o1, 4), _chk("x", x, 4)
    N, c0, H, W = o0.shape
    if c0 != 32 or tuple(o1.shape) != (N, 32, H, W) or tuple(x.shape) != (N, 64, H // 2, W // 2):
        raise ValueError("front: needs outs (N,32,H,W) x2 and x (N,64,H/2,W/2)")
    dev = o0.device
    feat_grid = torch.empty((N, 64, H, W), device=dev, dtype=torch.float32)
    feat0 = torch.empty_like(feat_grid) if store01 else None
    feat1 = torch.empty_like(feat_grid) if store01 else None
    feat2 = torch.empty((N, 64, H // 2, W // 2), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        rc = _lib.lib().tpspp_front_fwd(_ptr(o0), _ptr(o1), _ptr(x), _ptr(fw.w0), _ptr(fw.b0), _ptr(fw.w1),
                                        _ptr(fw.b1), _ptr(fw.w2), _ptr(fw.b2), _ptr(fw.wg), _ptr(fw.bg),
                                        _ptr(feat0), _ptr(feat1), _ptr(feat2), _ptr(feat_grid), N, H, W,
                                        _stream(o0))
    _lib.check(rc, "tpspp_front_fwd")
    return feat0, feat1, feat2, feat_grid


def down_fused_f32_applicable(o, cw):
    """`down_fused_f32` takes a float32 (N, 32, H, 128) map with an even H and a 64 -> 64 3x3 fp32 weight with a bias."""
    return (o.dtype == torch.float32 and o.dim() == 4 and o.shape[1] == 32 and o.shape[2] % 2 == 0 and o.shape[3] == 128
            and cw.kernel == 3 and cw.tiled is not None and tuple(cw.tiled.shape) == (16, 3, 3, 4, 64) and cw.bias is not None
            and cw.post_scale is None)


def down_fused_f32(o, w0_slab, b0, cw, relu=True):
    """`down0_1(down0(outs[0]))` / `down1_1(down1(outs[1]))` of TPS_PP.forward (tps_pp.py:560-563) in one exact-fp32 kernel
    (`tpspp_down_fused_f32_fwd`); `w0_slab`, `b0`: the 1x1 layer as `FrontWeights` holds it; `cw`: the 3x3 stride-2 layer
    (`prep_conv_weight`).  Returns (N, 64, H/2, 64) float32."""
    o = _chk("outs", o, 4)
    if not down_fused_f32_applicable(o, cw):
        raise ValueError("down_fused_f32: needs a float32 (N, 32, H, 128) map with an even H and a 64 -> 64 3x3 weight with a bias")
    N, _, H, W = o.shape
    out = torch.empty((N, 64, H // 2, W // 2), device=o.device, dtype=torch.float32)
    if N == 0:
        return out
    with torch.cuda.device(o.device):
        rc = _lib.lib().tpspp_down_fused_f32_fwd(_ptr(o), _ptr(w0_slab), _ptr(b0), _ptr(cw.tiled), _ptr(cw.bias), _ptr(out),
                                                 N, H, W, int(relu), _stream(o))
    _lib.check(rc, "tpspp_down_fused_f32_fwd")
    return out


class FrontWeightsBf16:
    def __init__(self, m, x3=False):
        """m: TPS_PP (ResNet45v2 wiring).  Slabs for tpspp_front_bf16_fwd (include/tpspp.h); x3: hi + lo halves."""
        f = lambda t: t.detach().float().contiguous()          # noqa: E731
        self.x3 = x3
        self.w0 = _bf16_slab(f(m.down0.conv.weight).view(64, 32), False, x3)
        self.w1 = _bf16_slab(f(m.down1.conv.weight).view(64, 32), False, x3)
        self.w2 = _bf16_slab(f(m.down2.conv.weight).view(64, 64), False, x3)
        self.wg = _bf16_slab(f(m.down_feat.conv.weight).view(64, 192), True, x3)
        self.b0, self.b1, self.b2, self.bg = (f(c.conv.bias) for c in (m.down0, m.down1, m.down2, m.down_feat))


def front_bf16_applicable(o0, o1, x, x3=False):
    dt = torch.float32 if x3 else torch.bfloat16
    return all(t.dtype == dt for t in (o0, o1, x)) and o0.shape[2] % 2 == 0 and o0.shape[3] % 32 == 0


def front_bf16(o0, o1, x, fw, feat_grid_dtype=torch.bfloat16, blocked=False, store01=True):
    """`front` on the bf16 matrix cores: bf16 in, bf16 feat0 / feat1 / feat2, feat_grid bf16 or fp32; with x3 weights
    (`FrontWeightsBf16(m, x3=True)`) fp32 in and out, three-term split.  `store01=False` (blocked bf16 only): feat0 / feat1
    stay operands of feat_grid and are returned as None -- `down_fused_bf16` recomputes them where they are consumed."""
    o0, o1, x = _chk16("outs[0]", o0, 4), _chk16("outs[1]", o1, 4), _chk16("x", x, 4)
    N, c0, H, W = o0.shape
    if c0 != 32 or tuple(o1.shape) != (N, 32, H, W) or tuple(x.shape) != (N, 64, H // 2, W // 2):
        raise ValueError("front_bf16: needs outs (N,32,H,W) x2 and x (N,64,H/2,W/2)")
    if not front_bf16_applicable(o0, o1, x, fw.x3):
        raise ValueError("front_bf16: needs bfloat16 inputs (float32 with x3 weights), an even height and a width that "
                         "is a multiple of 32")
    dev = o0.device
    bf = torch.float32 if fw.x3 else torch.bfloat16
    if fw.x3:
        feat_grid_dtype = torch.float32
    blocked = bool(blocked)                     # feat0 / feat1 / feat2 as `Blocked` / `Blocked32` maps (they only feed convolutions)
    if blocked:
        feat0 = torch.empty((N, 8, H, W, 8), device=dev, dtype=bf)
        feat2 = torch.empty((N, 8, H // 2, W // 2, 8), device=dev, dtype=bf)
    else:
        feat0 = torch.empty((N, 64, H, W), device=dev, dtype=bf)
        feat2 = torch.empty((N, 64, H // 2, W // 2), device=dev, dtype=bf)
    feat1 = torch.empty_like(feat0)
    if not store01:
        if not blocked:
            raise ValueError("front_bf16: store01=False goes with the blocked form")
        feat0 = feat1 = None
    feat_grid = torch.empty((N, 64, H, W), device=dev, dtype=feat_grid_dtype)
    with torch.cuda.device(dev):
        rc = _lib.lib().tpspp_front_bf16_fwd(_ptr(o0), _ptr(o1), _ptr(x), _ptr(fw.w0), _ptr(fw.b0), _ptr(fw.w1),
                                             _ptr(fw.b1), _ptr(fw.w2), _ptr(fw.b2), _ptr(fw.wg), _ptr(fw.bg),
                                             _ptr(feat0) if store01 else None, _ptr(feat1) if store01 else None,
                                             _ptr(feat2), _ptr(feat_grid),
                                             int(feat_grid_dtype == torch.float32) | (2 if blocked else 0), N, H, W,
                                             int(fw.x3), _stream(o0))
    _lib.check(rc, "tpspp_front_bf16_fwd")
    if blocked:
        B = Blocked32 if fw.x3 else Blocked
        return (B(feat0), B(feat1), B(feat2), feat_grid) if store01 else (None, None, B(feat2), feat_grid)
    return feat0, feat1, feat2, feat_grid


def token_gemm_bf16(x_cm, cw, act=None, residual=None, out_dtype=torch.float32):
    """`out (Co, M) = act(W^T x + bias) [+ residual]` for channel-major tokens `x_cm` (K, M) float32 on the bf16 matrix cores
    (`tpspp_token_gemm_bf16_fwd`); `cw`: a 1x1 `prep_conv_weight_bf16` weight (x3: the three-term split); act None | "gelu"."""
    x_cm = _chk("x_cm", x_cm, 2)
    K, M = x_cm.shape
    if cw.kernel != 1 or cw.cin != K or cw.post_scale is not None:
        raise ValueError("token_gemm_bf16: needs a 1x1 weight with Cin = x_cm.shape[0] and no post-affine")
    if residual is not None:
        residual = _chk("residual", residual, 2)
        if tuple(residual.shape) != (cw.cout, M):
            raise ValueError("token_gemm_bf16: residual must be (Co, M)")
    if out_dtype not in (torch.float32, torch.bfloat16) or act not in (None, "gelu"):
        raise ValueError("token_gemm_bf16: out_dtype float32 | bfloat16, act None | 'gelu'")
    out = torch.empty((cw.cout, M), device=x_cm.device, dtype=out_dtype)
    with torch.cuda.device(x_cm.device):
        rc = _lib.lib().tpspp_token_gemm_bf16_fwd(_ptr(x_cm), _ptr(cw.arranged), _ptr(cw.bias), _ptr(residual), out.data_ptr(),
                                                  int(out_dtype == torch.float32), K, cw.cout, M, 2 if act == "gelu" else 0,
                                                  int(cw.x3), _stream(x_cm))
    _lib.check(rc, "tpspp_token_gemm_bf16_fwd")
    return out


def down_fused_bf16_applicable(o, cw):
    """`down_fused_bf16` takes a (N, 32, H, 128) map with an even H -- bfloat16 with a plain-bf16 64 -> 64 3x3 weight, float32
    with an x3 (three-term split) weight."""
    return (o.dtype == (torch.float32 if cw.x3 else torch.bfloat16) and o.dim() == 4 and o.shape[1] == 32
            and o.shape[2] % 2 == 0 and o.shape[3] == 128
            and cw.kernel == 3 and cw.cin == 64 and cw.cout == 64 and cw.bias is not None and cw.post_scale is None)


def down_fused_bf16(o, w0_slab, b0, cw, relu=True):
    """`down0_1(down0(outs[0]))` / `down1_1(down1(outs[1]))` of TPS_PP.forward (tps_pp.py:560-563) in one kernel
    (`tpspp_down_fused_bf16_fwd`): the 1x1 result never exists in HBM.  `w0_slab`, `b0`: the 1x1 layer as
    `FrontWeightsBf16` holds it; `cw`: the 3x3 stride-2 layer (`prep_conv_weight_bf16`).  Returns a `Blocked` map."""
    o = _chk16("outs", o, 4)
    if not down_fused_bf16_applicable(o, cw):
        raise ValueError("down_fused_bf16: needs a (N, 32, H, 128) map with an even H -- bfloat16 with a 64 -> 64 3x3 bf16 weight, "
                         "float32 with an x3 weight -- and a bias")
    N, _, H, W = o.shape
    if cw.x3:                                           # three-term split: fp32 maps, `w0_slab` with its hi and lo halves
        out = Blocked32(torch.empty((N, 8, H // 2, W // 2, 8), device=o.device, dtype=torch.float32))
        fn, name = _lib.lib().tpspp_down_fused_x3_fwd, "tpspp_down_fused_x3_fwd"
    else:
        out = Blocked(torch.empty((N, 8, H // 2, W // 2, 8), device=o.device, dtype=torch.bfloat16))
        fn, name = _lib.lib().tpspp_down_fused_bf16_fwd, "tpspp_down_fused_bf16_fwd"
    if N == 0:
        return out
    with torch.cuda.device(o.device):
        rc = fn(_ptr(o), _ptr(w0_slab), _ptr(b0), _ptr(cw.arranged), _ptr(cw.bias), out.t.data_ptr(), N, H, W, int(relu),
                _stream(o))
    _lib.check(rc, name)
    return out


def cbam(x, atten):
    """CBAM.forward (tps_pp.py:77-82) on the (N, 64, 2, 16) bottleneck map, one fused kernel."""
    x = _chk("x", x, 4)
    if tuple(x.shape[1:]) != (64, 2, 16):
        raise ValueError("cbam: needs (N, 64, 2, 16)")
    f = lambda t: t.detach().float().contiguous()
    ca, sa = atten.channel_attention, atten.spatial_attention
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = _lib.lib().tpspp_cbam_fwd(_ptr(x), _ptr(f(ca.shared_MLP[0].weight.view(4, 64))),
                                       _ptr(f(ca.shared_MLP[2].weight.view(64, 4))), _ptr(f(sa.conv2d.weight)),
                                       _ptr(f(sa.conv2d.bias)), _ptr(out), x.shape[0], _stream(x))
    _lib.check(rc, "tpspp_cbam_fwd")
    return out


def tpe_points(en_feat, tpe):
    """localization_fc1/fc2 and p_linear of Transformation_Parameter_Estimation (tps_pp.py:305,321-323)
    on en_feat (N, 64, 2, 16): returns (ctrl (N, 32, 2), p (N, 32, 128))."""
    en_feat = _chk("en_feat", en_feat, 4)
    if tuple(en_feat.shape[1:]) != (64, 2, 16):
        raise ValueError("tpe_points: needs (N, 64, 2, 16)")
    N = en_feat.shape[0]
    f = lambda t: t.detach().float().contiguous()
    l1a, l1b, l2 = tpe.localization_fc1[0], tpe.localization_fc1[2], tpe.localization_fc2
    p0, p1 = tpe.p_linear[0], tpe.p_linear[1]
    ctrl = torch.empty((N, 32, 2), device=en_feat.device, dtype=torch.float32)
    p = torch.empty((N, 32, 128), device=en_feat.device, dtype=torch.float32)
    ws = [f(t) for t in (l1a.weight, l1a.bias, l1b.weight, l1b.bias, l2.weight, l2.bias, p0.weight, p0.bias,
                         p1.weight, p1.bias)]
    with torch.cuda.device(en_feat.device):
        rc = _lib.lib().tpspp_tpe_points_fwd(_ptr(en_feat), *[_ptr(w) for w in ws], _ptr(ctrl), _ptr(p), N,
                                             _stream(en_feat))
    _lib.check(rc, "tpspp_tpe_points_fwd")
    return ctrl, p


def maxpool2x2(x):
    """nn.MaxPool2d(2, 2) (tps_preprocessor.py:110,114,118)."""
    x = _chk("input", x, 4)
    N, C, H, W = x.shape
    out = torch.empty((N, C, H // 2, W // 2), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        rc = _lib.lib().tpspp_maxpool2x2_fwd(_ptr(x), N, C, H, W, _ptr(out), _stream(x))
    _lib.check(rc, "tpspp_maxpool2x2_fwd")
    return out


def global_avgpool(x):
    """nn.AdaptiveAvgPool2d(1) -> (N, C) (tps_preprocessor.py:126)."""
    x = _chk("input", x, 4)
    N, C, H, W = x.shape
    out = torch.empty((N, C), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        rc = _lib.lib().tpspp_global_avgpool_fwd(_ptr(x), N, C, H, W, _ptr(out), _stream(x))
    _lib.check(rc, "tpspp_global_avgpool_fwd")
    return out


def linear(x, cw, relu=False):
    """y = act(x @ W^T + b) for x (N, Cin) on the conv kernel: the batch plays the role of the pixels
    of a 1x1 convolution over a (1, Cin, 1, N) image.  cw = prep_conv_weight(W.view(Cout, Cin, 1, 1), conv_bias=b).
    Returns (N, Cout)."""
    x = _chk("input", x, 2)
    n, cin = x.shape
    xt = x.t().contiguous().view(1, cin, 1, n)
    y = conv2d([xt], cw, 1, relu)                      # (1, Cout, 1, N)
    return y.view(-1, n).t().contiguous()


def set_warp_bwd_accumulator(fixed_point: bool = False):
    """`tpspp_warp_bwd_set_accumulator`: the process-wide default of how dL/d input is accumulated in LDS (fp64 atomics;
    True: round 3's 64-bit fixed point, bitwise reproducible but with a per-pass scale).  Measurement scripts only: prefer
    `warp_backward(..., fixed_point=True)`, which is per call and per stream."""
    _lib.check(_lib.lib().tpspp_warp_bwd_set_accumulator(1 if fixed_point else 0), "tpspp_warp_bwd_set_accumulator")


def set_warp_tuning(images_per_group=0, threads_per_group=0, kernel_choice=0, bands=0):
    """`tpspp_warp_set_tuning` (process-wide lab knobs; 0 everywhere = the automatic choice).
    kernel_choice: 0 automatic, 1 gather kernel, 2 LDS-staged kernel, 3 the same without the mirror trick, 4 plane-streaming
    kernel, 5 image-pair kernel, 6 instantiated in-place kernel (tpspp_warp_img.h), 7 a run-time-geometry kernel (the
    in-place kernel of tpspp_warp_geo.h where one workgroup covers the image, else the span-staging kernel), 8 span-staging
    kernel (tpspp_warp_span.h), 9 the in-place kernel of tpspp_warp_geo.h in its banded form as well (2..9: error if not
    applicable to the call's shapes);
    bands: kernel_choice 0 / 2 / 3: workgroups per image (pair) in the LDS-staged kernels, 0..8 (0 = heuristic);
    kernel_choice 7 / 9: bits 0-2 = workgroups per image (0 = heuristic), bit 3 (value 8) = never an image pair per
    workgroup, so 0..15; kernel_choice 8: bits 0-5 = workgroups per image, bit 6 (64) = every workgroup on the
    global-memory path, bit 7 (128) = measured row spans instead of the windows requested at launch, bits 8-15 = LDS budget
    in KB; values above 8 are rejected for every other choice."""
    _lib.check(_lib.lib().tpspp_warp_set_tuning(int(images_per_group), int(threads_per_group),
                                                int(kernel_choice), int(bands)),
               "tpspp_warp_set_tuning")


# ---- recogniser head (NRTR encoder / decoder): channel-major matrices, K-major weights -------------------
def transpose2d(x):
    """(rows, cols) -> (cols, rows) on the device (`tpspp_transpose2d`)."""
    x = _chk("input", x, 2)
    r, c = x.shape
    out = torch.empty((c, r), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        rc = _lib.lib().tpspp_transpose2d(_ptr(x), r, c, _ptr(out), _stream(x))
    _lib.check(rc, "tpspp_transpose2d")
    return out


def layernorm_cm(x, gamma, beta, eps=1e-5):
    """LayerNorm over the rows of a channel-major (C, M) matrix (`tpspp_layernorm_cm_fwd`)."""
    x = _chk("input", x, 2)
    C, M = x.shape
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = _lib.lib().tpspp_layernorm_cm_fwd(_ptr(x), _ptr(_chk("gamma", gamma, 1)), _ptr(_chk("beta", beta, 1)),
                                               C, M, float(eps), _ptr(y), _stream(x))
    _lib.check(rc, "tpspp_layernorm_cm_fwd")
    return y


def fold_layernorm(gamma, beta, w_kmajor, bias=None):
    """LayerNorm affine folded into the following projection (done once per weight):
    -> (w_gamma = diag(gamma) W, w_colsum = column sums of w_gamma, bias_eff = beta^T W (+ bias))."""
    w = w_kmajor.detach().float()
    wg = (gamma.detach().float()[:, None] * w).contiguous()
    be = beta.detach().float() @ w
    if bias is not None:
        be = be + bias.detach().float()
    return wg, wg.sum(0).contiguous(), be.contiguous()


def linear_ln(x, folded, eps=1e-5, act=0, residual=None, token_major=False):
    """Fused LayerNorm -> Linear on a channel-major (K, M) activation (`tpspp_linear_ln_fwd`);
    `folded` = fold_layernorm(gamma, beta, w_kmajor, bias)."""
    wg, colsum, bias_eff = folded
    x, wg = _chk("x", x, 2), _chk("w_gamma", wg, 2)
    K, M = x.shape
    Cout = wg.shape[1]
    out = torch.empty((M, Cout) if token_major else (Cout, M), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        rc = _lib.lib().tpspp_linear_ln_fwd(_ptr(x), K, M, float(eps), _ptr(wg), _ptr(_chk("w_colsum", colsum, 1)), Cout,
                                            _ptr(bias_eff), int(act), _ptr(residual), int(bool(token_major)),
                                            _ptr(out), _stream(x))
    _lib.check(rc, "tpspp_linear_ln_fwd")
    return out


def attn_enc(qkv, N, T, valid_len=None):
    """Encoder multi-head self-attention on projected (3C, N*T) q/k/v (`tpspp_attn_enc_fwd`) -> (C, N*T)."""
    qkv = _chk("qkv", qkv, 2)
    C = qkv.shape[0] // 3
    out = torch.empty((C, N * T), device=qkv.device, dtype=torch.float32)
    with torch.cuda.device(qkv.device):
        rc = _lib.lib().tpspp_attn_enc_fwd(_ptr(qkv), N, C, T, _ptr(valid_len), _ptr(out), _stream(qkv))
    _lib.check(rc, "tpspp_attn_enc_fwd")
    return out


def kmajor(weight):
    """PyTorch Linear weight (out, in) -> K-major (in, out) fp32 contiguous device copy."""
    return weight.detach().float().t().contiguous()


def arrange_x3(w_kmajor):
    """K-major (K, Co) fp32 weight -> the step GEMM's operand (`dec_gemm_x3_kernel`, include/tpspp.h): hi = bf16(w),
    lo = bf16(w - hi), Co zero-padded to a multiple of 32, [Co/32][K/16][hi|lo][2 k halves][32 outputs][8 k] bf16."""
    w = w_kmajor.detach().float()
    K, Co = w.shape
    if K % 16:
        raise ValueError("arrange_x3: K must be a multiple of 16")
    Cop = (Co + 31) // 32 * 32
    if Cop != Co:
        w = torch.cat([w, w.new_zeros((K, Cop - Co))], dim=1)
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    st = torch.stack([hi, lo])                                         # (s, K, Cop)
    st = st.reshape(2, K // 16, 2, 8, Cop // 32, 32)                   # (s, ks, h, e, ct, r)
    return st.permute(4, 1, 0, 2, 5, 3).contiguous()                   # (ct, ks, s, h, r, e)


def arrange_f32(w_kmajor):
    """K-major (K, Co) fp32 weight -> the exact-fp32 step GEMM's operand (`dec_gemm_f32_kernel`): Co zero-padded to a
    multiple of 32, [Co/32][K/8][2 k halves][32 outputs][4 k] fp32 with k = 8 u + 4 half + e."""
    w = w_kmajor.detach().float()
    K, Co = w.shape
    if K % 8:
        raise ValueError("arrange_f32: K must be a multiple of 8")
    Cop = (Co + 31) // 32 * 32
    if Cop != Co:
        w = torch.cat([w, w.new_zeros((K, Cop - Co))], dim=1)
    return w.reshape(K // 8, 2, 4, Cop // 32, 32).permute(3, 0, 1, 4, 2).contiguous()     # (u, h, e, ct, r) -> (ct, u, h, r, e)


class PtrTable:
    """Host array of device pointers (`const float* const*`) + the tensors it points at (kept alive)."""

    def __init__(self, tensors):
        import ctypes
        self.keep = list(tensors)
        self.arr = (ctypes.c_void_p * len(self.keep))(*[None if t is None else t.data_ptr() for t in self.keep])
        self.ptr = ctypes.cast(self.arr, ctypes.c_void_p)

    def __len__(self):
        return len(self.keep)


def _workspace(holder, nbytes, device):
    """Scratch of the encoder / decoder, cached on the module per (device, stream): two calls on different HIP streams
    must not share a buffer (they could overlap on the GPU)."""
    key = (str(device), _stream(torch.empty(0, device=device)))
    cache = getattr(holder, "_tpspp_ws", None)
    if not isinstance(cache, dict):
        cache = {}
        holder._tpspp_ws = cache
    ws = cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), device=device, dtype=torch.uint8)
        cache[key] = ws
    return ws


HEAD_BF16 = 1            # TPSPP_HEAD_BF16
HEAD_BF16X3 = 2          # TPSPP_HEAD_BF16X3


def nrtr_encoder(feat, table, n_layers, d_inner, ln_g, ln_b, valid_len=None, holder=None, want_ntc=True, flags=0):
    """`tpspp_nrtr_encoder_fwd`: feat (N, C, H, W) -> (out (N, T, C) | None, out_cm (C, N*T))."""
    feat = _chk("feat", feat, 4)
    N, C, H, W = feat.shape
    T = H * W
    L = _lib.lib()
    nbytes = L.tpspp_nrtr_encoder_workspace(N, C, T, d_inner)
    ws = _workspace(holder if holder is not None else table, nbytes, feat.device)
    out_cm = torch.empty((C, N * T), device=feat.device, dtype=torch.float32)
    out = torch.empty((N, T, C), device=feat.device, dtype=torch.float32) if want_ntc else None
    with torch.cuda.device(feat.device):
        rc = L.tpspp_nrtr_encoder_fwd(_ptr(feat), N, C, T, d_inner, n_layers, table.ptr, _ptr(ln_g), _ptr(ln_b),
                                      _ptr(valid_len), ws.data_ptr(), ws.numel(), _ptr(out_cm), _ptr(out),
                                      int(flags), _stream(feat))
    _lib.check(rc, "tpspp_nrtr_encoder_fwd")
    return out, out_cm


def nrtr_decoder(enc_cm, N, T, table, n_layers, d_inner, emb, pos_table, cls_folded, max_seq_len,
                 start_idx, padding_idx, valid_len=None, forced_tokens=None, holder=None, flags=0):
    """`tpspp_nrtr_decoder_fwd` -> (out (N, L, num_out), tokens (N, L+1) int32, status (1,) int32).
    `status` is the device word the decode leaves behind itself: 0 = every step completed, 1 = a cluster barrier of the
    persistent step kernel timed out (the affected images' scores are NaN from that step on; include/tpspp.h).  Nothing
    here synchronises; whoever copies the scores / tokens to the host reads it in the same copy (`check_decoder_status`,
    `attn_tensor2idx`)."""
    enc_cm = _chk("enc_cm", enc_cm, 2)
    C = enc_cm.shape[0]
    if enc_cm.shape[1] != N * T:
        raise ValueError("nrtr_decoder: enc_cm must be (C, N*T)")
    w_cls, cls_colsum, b_cls = cls_folded              # fold_layernorm(final layer_norm, classifier)
    num_out = w_cls.shape[1]
    L = _lib.lib()
    nbytes = L.tpspp_nrtr_decoder_workspace(N, C, T, d_inner, n_layers, max_seq_len, num_out)
    ws = _workspace(holder if holder is not None else table, nbytes, enc_cm.device)
    out = torch.empty((N, max_seq_len, num_out), device=enc_cm.device, dtype=torch.float32)
    # tokens and the status word in one buffer: one device->host copy fetches both
    tok_status = torch.empty((N * (max_seq_len + 1) + 1,), device=enc_cm.device, dtype=torch.int32)
    tokens = tok_status[:-1].view(N, max_seq_len + 1)
    status = tok_status[-1:]
    if forced_tokens is not None:
        if forced_tokens.dtype != torch.int32 or tuple(forced_tokens.shape) != (N, max_seq_len) or \
                not forced_tokens.is_contiguous() or forced_tokens.device != enc_cm.device:
            raise ValueError("nrtr_decoder: forced_tokens must be a contiguous (N, max_seq_len) int32 device tensor")
    with torch.cuda.device(enc_cm.device):
        rc = L.tpspp_nrtr_decoder_fwd(_ptr(enc_cm), N, C, T, d_inner, n_layers, table.ptr, len(table),
                                      _ptr(emb), _ptr(pos_table), pos_table.shape[0], _ptr(w_cls), _ptr(cls_colsum),
                                      _ptr(b_cls), num_out, max_seq_len, int(start_idx), int(padding_idx), _ptr(valid_len),
                                      _ptr(forced_tokens), ws.data_ptr(), ws.numel(), _ptr(out), tokens.data_ptr(),
                                      status.data_ptr(), int(flags), _stream(enc_cm))
    _lib.check(rc, "tpspp_nrtr_decoder_fwd")
    return out, tokens, status


DECODER_TIMEOUT_MESSAGE = (
    "tpspp_nrtr_decoder_fwd: a cluster barrier of the persistent decoder kernel timed out (status 1): the scores of the "
    "affected images are NaN.  The kernel needs its workgroups co-resident; this happens only when another PROCESS runs a "
    "persistent decode on the same GPU or the device stayed occupied for longer than TPSPP_HEAD_TIMEOUT_MS "
    "(include/tpspp.h).  TPSPP_HEAD_NO_PERSIST=1 selects the launch-per-phase pipeline.")


def check_decoder_status(status):
    """Synchronising check of a decode's status word (one 4-byte device->host copy): raises `TpsppError` on a timeout."""
    if status is not None and int(status.cpu()[0]) != 0:
        raise _lib.TpsppError(DECODER_TIMEOUT_MESSAGE)


def attn_tensor2idx(scores, end_idx, padding_idx, status=None):
    """`tpspp_attn_tensor2idx_fwd` + ONE device->host copy: (idx (N, L) int32 numpy with -1 where the reference's scan drops
    the position, val (N, L) float32 numpy).  `status`: a decode's status word, fetched in the same copy; non-zero raises."""
    scores = _chk("scores", scores, 3)
    N, L, C = scores.shape
    # [idx (N*L) | val (N*L) as raw bits | status]: one int32 buffer, one copy
    buf = torch.empty((2 * N * L + 1,), device=scores.device, dtype=torch.int32)
    with torch.cuda.device(scores.device):
        rc = _lib.lib().tpspp_attn_tensor2idx_fwd(_ptr(scores), N, L, C, int(end_idx), int(padding_idx), buf.data_ptr(),
                                                  buf.data_ptr() + 4 * N * L, _stream(scores))
    _lib.check(rc, "tpspp_attn_tensor2idx_fwd")
    if status is not None and status.device == scores.device:
        buf[-1:].copy_(status)
    else:
        buf[-1:].zero_()
    host = buf.cpu().numpy()
    if host[-1] != 0:
        raise _lib.TpsppError(DECODER_TIMEOUT_MESSAGE)
    return host[:N * L].reshape(N, L), host[N * L:2 * N * L].view("float32").reshape(N, L)


RESIZE_CV2, RESIZE_PILLOW = 0, 1


def resize_normalize(packed, offsets, src_h, src_w, resize_w, lut, pad_value, N, C, H, W, interpolation=RESIZE_CV2):
    """`tpspp_resize_normalize_fwd`: packed uint8 HWC images -> (N, C, H, W) fp32 (ocr_transforms.py:67-156).
    `interpolation`: RESIZE_CV2 (OpenCV's INTER_LINEAR arithmetic, unpinned) or RESIZE_PILLOW (Pillow's BILINEAR, pinned)."""
    if interpolation not in (RESIZE_CV2, RESIZE_PILLOW):
        raise ValueError("resize_normalize: interpolation must be RESIZE_CV2 or RESIZE_PILLOW")
    for name, t, dt in (("packed", packed, torch.uint8), ("offsets", offsets, torch.int64), ("src_h", src_h, torch.int32),
                        ("src_w", src_w, torch.int32), ("resize_w", resize_w, torch.int32), ("lut", lut, torch.float32)):
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise _lib.TpsppError(f"resize_normalize: {name} must be a GPU tensor (no CPU fallback)")
        if t.dtype != dt or not t.is_contiguous():
            raise TypeError(f"resize_normalize: {name} must be a contiguous {dt} tensor")
    if offsets.numel() != N or src_h.numel() != N or src_w.numel() != N or resize_w.numel() != N or \
            tuple(lut.shape) != (C, 256):
        raise ValueError("resize_normalize: per-image arrays need N entries, lut must be (C, 256)")
    out = torch.empty((N, C, H, W), device=packed.device, dtype=torch.float32)
    with torch.cuda.device(packed.device):
        rc = _lib.lib().tpspp_resize_normalize_fwd(_ptr(packed), _ptr(offsets), _ptr(src_h), _ptr(src_w), _ptr(resize_w),
                                                   _ptr(lut), int(pad_value), int(N), int(C), int(H), int(W), _ptr(out),
                                                   int(interpolation), _stream(packed))
    _lib.check(rc, "tpspp_resize_normalize_fwd")
    return out


# ---- backward of the fused warp ------------------------------------------------
def warp_backward(g_out0, in0, grid, ctrl, inv_delta_C, P_hat, out_hw, P_xy=None, score=None, in1=None,
                  g_out1=None, P_hat_t=None, need_in0=True, need_in1=True, need_score=True, fixed_point=False,
                  two_kernels=False):
    """`tpspp_warp_bwd`: (g_in0 | None, g_in1 | None, g_ctrl, g_score | None) for the call
    `warp(in0, ctrl, inv_delta_C, P_hat, out_hw, P_xy, score, in1)`; `grid` is that call's grid output.
    `fixed_point=True` (TPSPP_BWD_FIXED_POINT, this call only): dL/d input accumulated in 64-bit fixed point -- bitwise
    reproducible from run to run; the default fp64 LDS atomics keep every term's bits but their sum depends on arrival
    order, so two runs may differ by one fp32 ulp at rounding ties (include/tpspp.h).  `two_kernels=True`
    (TPSPP_BWD_TWO_KERNELS, this call only): the sampling + parameter kernels even where the classic rectifier's one-launch
    form would run."""
    g_out0, in0 = _chk("g_out0", g_out0, 4), _chk("in0", in0, 4)
    grid, ctrl = _chk("grid", grid, 3), _chk("ctrl", ctrl, 3)
    inv_delta_C, P_hat = _chk("inv_delta_C", inv_delta_C, 2), _chk("P_hat", P_hat, 2)
    N, C0, H0, W0 = in0.shape
    F = int(ctrl.shape[1])
    Ho, Wo = int(out_hw[0]), int(out_hw[1])
    n = Ho * Wo
    if tuple(g_out0.shape) != (N, C0, Ho, Wo) or tuple(grid.shape) != (N, n, 2):
        raise ValueError("warp_backward: g_out0 / grid shape")
    flags = (BWD_FIXED_POINT if fixed_point else 0) | (BWD_TWO_KERNELS if two_kernels else 0)
    g_score = None
    if score is not None:
        if tuple(score.shape) != (N, n, F):
            raise ValueError("warp_backward: score must be (N, n, F)")
        if score.stride() == (F * n, 1, n) and n > 1 and F > 1:
            flags |= SCORE_TRANSPOSED
            score = _chk("score", score.transpose(1, 2), 3)
        else:
            score = _chk("score", score, 3)
        if need_score:
            g_score = torch.empty_like(score)
    if P_xy is not None:
        P_xy = _chk("P_xy", P_xy, 2)
    if P_hat_t is not None:
        P_hat_t = _chk("P_hat_t", P_hat_t, 2)
    C1 = H1 = W1 = 0
    if in1 is not None:
        in1, g_out1 = _chk("in1", in1, 4), _chk("g_out1", g_out1, 4)
        _, C1, H1, W1 = in1.shape
        if tuple(g_out1.shape) != (N, C1, Ho, Wo):
            raise ValueError("warp_backward: g_out1 shape")
    T = solve_T(inv_delta_C, ctrl) if score is not None else None      # (only dL/d score reads T)
    g_in0 = torch.empty_like(in0) if need_in0 else None
    g_in1 = torch.empty_like(in1) if (in1 is not None and need_in1) else None
    g_ctrl = torch.empty((N, F, 2), device=in0.device, dtype=torch.float32)
    g_grid = torch.empty((int(_lib.lib().tpspp_warp_bwd_workspace_floats(N, Ho, Wo)),), device=in0.device, dtype=torch.float32)
    with torch.cuda.device(in0.device):
        rc = _lib.lib().tpspp_warp_bwd(_ptr(g_out0), _ptr(in0), C0, H0, W0, _ptr(g_out1), _ptr(in1), C1, H1, W1,
                                       _ptr(grid), _ptr(T), _ptr(inv_delta_C), _ptr(P_hat), P_hat.shape[1],
                                       _ptr(P_xy), _ptr(score), _ptr(P_hat_t), flags, N, F, Ho, Wo,
                                       _ptr(g_in0), _ptr(g_in1), _ptr(g_ctrl), _ptr(g_score), _ptr(g_grid),
                                       g_grid.numel(), _stream(in0))
    _lib.check(rc, "tpspp_warp_bwd")
    if g_score is not None and (flags & SCORE_TRANSPOSED):
        g_score = g_score.transpose(1, 2)                  # back to the logical (N, n, F) view
    return g_in0, g_in1, g_ctrl, g_score


class _WarpFunction(torch.autograd.Function):
    """Differentiable `warp`: HIP forward (`tpspp_warp_fwd`) and HIP backward (`tpspp_warp_bwd`).  The tables
    (inv_delta_C, P_hat, P_xy) are constants of the module and get no gradient."""

    @staticmethod
    def forward(ctx, in0, ctrl, score, in1, inv_delta_C, P_hat, P_xy, P_hat_t, out_hw, table_flags):
        out0, out1, grid, _ = warp(in0, ctrl, inv_delta_C, P_hat, out_hw, P_xy=P_xy, score=score, in1=in1,
                                   want_grid=True, P_hat_t=P_hat_t, table_flags=table_flags)
        ctx.out_hw = out_hw
        ctx.has = (score is not None, in1 is not None)
        ctx.save_for_backward(in0, ctrl, score, in1, inv_delta_C, P_hat, P_xy, P_hat_t, grid)
        if out1 is None:
            return out0
        return out0, out1

    @staticmethod
    def backward(ctx, *g):
        in0, ctrl, score, in1, inv_delta_C, P_hat, P_xy, P_hat_t, grid = ctx.saved_tensors
        g0 = g[0]
        g1 = g[1] if len(g) > 1 else None
        if g0 is None:
            g0 = torch.zeros((in0.shape[0], in0.shape[1]) + tuple(ctx.out_hw), device=in0.device)
        if in1 is not None and g1 is None:
            g1 = torch.zeros((in1.shape[0], in1.shape[1]) + tuple(ctx.out_hw), device=in0.device)
        need = ctx.needs_input_grad
        g_in0, g_in1, g_ctrl, g_score = warp_backward(
            g0.float(), in0, grid, ctrl, inv_delta_C, P_hat, ctx.out_hw, P_xy=P_xy, score=score, in1=in1,
            g_out1=None if g1 is None else g1.float(), P_hat_t=P_hat_t, need_in0=need[0],
            need_in1=need[3], need_score=need[2])
        return (g_in0, g_ctrl if need[1] else None, g_score, g_in1, None, None, None, None, None, None)


def warp_autograd(in0, ctrl, inv_delta_C, P_hat, out_hw, P_xy=None, score=None, in1=None, P_hat_t=None,
                  table_flags=0):
    """`warp` inside an autograd graph: returns out0 or (out0, out1); gradients flow to in0, ctrl, score, in1."""
    return _WarpFunction.apply(in0, ctrl, score, in1, inv_delta_C, P_hat, P_xy, P_hat_t,
                               (int(out_hw[0]), int(out_hw[1])), int(table_flags))
