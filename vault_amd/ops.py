"""Thin host wrappers: torch tensors (device memory + stream only) -> C-ABI calls of libvault_hip.so.

No arithmetic happens here; every function enqueues HIP kernels on torch's current stream.
"""
from __future__ import annotations

import contextvars
import ctypes as C
from typing import Optional

import torch

from . import lib as L

EPI_BF16, EPI_BF16_GELU, EPI_BF16_DGELU, EPI_F32_RES, EPI_F32_PATCH, EPI_F32_ATOMIC = range(6)


class LnFwdArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("post_add", C.c_void_p),
                ("y_bf16", C.c_void_p), ("y_f32", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("rows", C.c_int), ("H", C.c_int), ("eps", C.c_float),
                ("x_rpg", C.c_int), ("x_gstride", C.c_int), ("x_goff", C.c_int),
                ("y_rpg", C.c_int), ("y_gstride", C.c_int), ("y_goff", C.c_int),
                ("drop_thresh", C.c_uint32), ("drop_seed", C.c_uint32), ("drop_stream", C.c_uint32),
                ("drop_scale", C.c_float), ("y_split3", C.c_void_p), ("y_q", C.c_void_p), ("y_scale", C.c_void_p)]


class LnBwdArgs(C.Structure):
    _fields_ = [("dy_bf16", C.c_void_p), ("dy_f32", C.c_void_p), ("x", C.c_void_p), ("mean", C.c_void_p),
                ("rstd", C.c_void_p), ("gamma", C.c_void_p), ("dres", C.c_void_p),
                ("dx_f32", C.c_void_p), ("dx_bf16", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("rows", C.c_int), ("H", C.c_int),
                ("dy_rpg", C.c_int), ("dy_gstride", C.c_int), ("dy_goff", C.c_int),
                ("x_rpg", C.c_int), ("x_gstride", C.c_int), ("x_goff", C.c_int),
                ("dx_rpg", C.c_int), ("dx_gstride", C.c_int), ("dx_goff", C.c_int),
                ("drop_thresh", C.c_uint32), ("drop_seed", C.c_uint32), ("drop_stream", C.c_uint32),
                ("drop_scale", C.c_float), ("drop_on_dy", C.c_int), ("dbias", C.c_void_p), ("dres_bf16", C.c_void_p)]


class AttnArgs(C.Structure):
    _fields_ = [("qkv", C.c_void_p), ("keymask", C.c_void_p), ("ctx", C.c_void_p), ("lse", C.c_void_p),
                ("dctx", C.c_void_p), ("dqkv", C.c_void_p),
                ("B", C.c_int), ("S", C.c_int), ("H", C.c_int), ("heads", C.c_int),
                ("drop_thresh", C.c_uint32), ("drop_seed", C.c_uint32), ("drop_stream", C.c_uint32),
                ("drop_scale", C.c_float), ("ctx_split3", C.c_void_p), ("qkv_hm", C.c_int),
                ("bias_partials", C.c_void_p), ("bias_thirds", C.c_int)]


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# ---- 16-bit operand format -------------------------------------------------------------------------------------
# Every wrapper below launches through the library of the CURRENT format ("bf16": libvault_hip.so, "fp16":
# libvault_hip_f16.so - the same kernels compiled for the other operand type, csrc/common.h).  An engine makes its own
# format current around its entry points (``with ops.operand_format(fmt):``); a recorded tape holds the function
# pointers of the library it was recorded on.  The current format is a context variable: another host thread (a prefetch /
# preprocessing thread, a second engine) starts from the default and is never re-routed by an engine that is inside its
# own ``with`` on this thread.
_FMT: contextvars.ContextVar = contextvars.ContextVar("vault_operand_format", default="bf16")
HALF_DTYPE = {"bf16": torch.bfloat16, "fp16": torch.float16}


class operand_format:
    """Context manager: the launches inside run on the library of ``fmt``."""

    def __init__(self, fmt: str):
        if fmt not in HALF_DTYPE:
            raise ValueError("operand format must be 'bf16' or 'fp16'")
        self.fmt = fmt

    def __enter__(self):
        self._token = _FMT.set(self.fmt)
        return self

    def __exit__(self, *exc):
        _FMT.reset(self._token)
        return False


def current_format() -> str:
    return _FMT.get()


def _h(t: Optional[torch.Tensor]):
    """Pointer of a 16-bit operand tensor; its dtype must be the current library's (a bf16 tensor handed to the fp16
    kernels, or the reverse, would be read as garbage without any error from the device)."""
    if t is None:
        return None
    fmt = _FMT.get()
    if t.dtype != HALF_DTYPE[fmt] and t.dtype in (torch.bfloat16, torch.float16):
        raise TypeError(f"{t.dtype} operand in a launch on the {fmt} library")
    return t.data_ptr()


class Tape:
    """A recorded sequence of C-ABI calls (function pointer + prepared ctypes arguments) and host
    callbacks.  Every buffer of the engine is persistent, so a train step is the SAME call list with the
    same pointers every time: replaying it costs a few microseconds per launch instead of re-marshalling
    ~700 argument structs in Python.  Only the dropout seed changes between replays (``seeded``)."""

    def __init__(self):
        self.calls = []
        self.seeded = []
        # one operand that changes address between replays (the caller's `pixel_patches` tensor, adopted as the patch
        # GEMM's operand): [lo, hi) while recording, and the (argument struct, member, offset) places that pointed into it
        self.rebind_range = None
        self.rebinds = []

    def rebind(self, base: int):
        """Point every recorded use of the rebindable operand at its new address."""
        for s, f, off in self.rebinds:
            setattr(s, f, base + off)

    def replay(self, seed: Optional[int] = None):
        if seed is not None:
            sd = seed & 0xFFFFFFFF
            for a in self.seeded:
                a.drop_seed = sd
        for fn, args in self.calls:
            rc = fn(*args)
            if isinstance(rc, int) and rc != 0:
                raise RuntimeError(f"{getattr(fn, '__name__', fn)} failed with code {rc} during replay")


_TAPE: Optional[Tape] = None


def start_tape() -> Tape:
    global _TAPE
    _TAPE = Tape()
    return _TAPE


def stop_tape() -> Optional[Tape]:
    global _TAPE
    t, _TAPE = _TAPE, None
    return t


def taping() -> bool:
    return _TAPE is not None


def pycall(fn):
    """Run a host-side action now and, when recording, put it on the tape (must return None / 0).  Kernel launches made
    INSIDE the action are live every time it runs (they are not recorded a second time): the data-parallel gradient
    exchange launches its pack / unpack kernels from such callbacks."""
    global _TAPE
    t = _TAPE
    if t is not None:
        t.calls.append((fn, ()))
    _TAPE = None
    try:
        fn()
    finally:
        _TAPE = t


def _invoke(name: str, *args, struct=None, drop=None):
    fn = getattr(L.load(_FMT.get()), name)
    if _TAPE is not None:
        _TAPE.calls.append((fn, args))
        if struct is not None and drop is not None and drop.thresh:
            _TAPE.seeded.append(struct)
    L.check(fn(*args), name)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Drop:
    """Dropout descriptor: keep iff hash(seed, stream, element) >= p * 2^32; kept values scale by 1/(1-p)."""

    __slots__ = ("thresh", "seed", "stream", "scale")

    def __init__(self, p: float = 0.0, seed: int = 0, stream: int = 0):
        self.thresh = 0 if p <= 0.0 else min(int(p * 4294967296.0), 4294967295)
        self.seed = seed & 0xFFFFFFFF
        self.stream = stream & 0xFFFFFFFF
        self.scale = 1.0 if p <= 0.0 else 1.0 / (1.0 - p)


NO_DROP = Drop()


# GEMM scheduling mode (vault_gemm_args.persist): 3 while the GEMMs share the GPU with RCCL collectives on another
# stream (train.TrainStep sets it for world size > 1): tiles are handed out dynamically / one block per tile, so a CU
# held by the collective costs its share of throughput instead of a second pass over a static tile list
# (tools/contention_probe.py: 196-230 us instead of 271-281 us with 8-64 CUs held, 189 us alone)
GEMM_SCHED = 0       # (tests set it directly to exercise the mode on one GPU)


def gemm(A, B, out, M, N, K, lda, ldb, ldo, a_mode, b_mode, epi, *, cfg=-1, m_valid=0, splits=1, accumulate=0,
         bias=None, res=None, aux=None, out2=None, addtab=None, rpg=0, gstride=0, goff=0, drop: Drop = NO_DROP,
         colsum=None, split3=False, batch=0, batch_a=0, batch_b=0, batch_o=0, aux_u8=False, plan_only=False, out_hm=0, a_hm=0,
         splitk_ws=None):
    a = L.GemmArgs()
    if splitk_ws is not None:       # caller-owned workspace of the in-launch split-K reduction (vault_gemm_args.splitk_ws)
        a.splitk_ws, a.splitk_bytes = _p(splitk_ws), splitk_ws.numel() * splitk_ws.element_size()
    a.aux_u8 = 1 if aux_u8 else 0
    a.out_hm, a.a_hm = int(out_hm), int(a_hm)     # head-major output / A operand (vault_gemm_args.out_hm / a_hm)
    a.A, a.B, a.out, a.out2 = _h(A), _h(B), _h(out), _h(out2)
    a.bias, a.res, a.aux, a.addtab = _p(bias), _p(res), _h(aux), _p(addtab)
    a.colsum = _p(colsum)
    a.split3 = 1 if split3 else 0
    a.M, a.N, a.K, a.lda, a.ldb, a.ldo, a.m_valid = M, N, K, lda, ldb, ldo, m_valid
    a.a_mode, a.b_mode, a.epi, a.cfg, a.splits, a.accumulate = a_mode, b_mode, epi, cfg, splits, accumulate
    a.rpg, a.gstride, a.goff = rpg, gstride, goff
    a.persist = GEMM_SCHED
    a.batch, a.batch_a, a.batch_b, a.batch_o = batch, batch_a, batch_b, batch_o   # batched weight gradients (ABI 3)
    a.drop_thresh, a.drop_seed, a.drop_stream, a.drop_scale = drop.thresh, drop.seed, drop.stream, drop.scale
    if plan_only:      # the kernel configuration these arguments would run on (vault_gemm_plan): >= 0, or -EINVAL
        return int(L.load(_FMT.get()).vault_gemm_plan(C.byref(a)))
    if _TAPE is not None and _TAPE.rebind_range is not None:
        lo, hi = _TAPE.rebind_range
        for f in ("A", "B"):
            p = getattr(a, f) or 0
            if lo <= p < hi:
                _TAPE.rebinds.append((a, f, p - lo))
    _invoke("vault_gemm", C.byref(a), _stream(), struct=a, drop=drop)


class WgradSeg(C.Structure):
    """vault_wgrad_seg (include/vault_hip.h)"""
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p),
                ("n_out", C.c_int), ("n_in", C.c_int), ("ld_dy", C.c_int), ("ld_x", C.c_int), ("ld_dw", C.c_int),
                ("batch", C.c_int), ("first", C.c_int), ("count", C.c_int),
                ("batch_dy", C.c_longlong), ("batch_x", C.c_longlong), ("batch_dw", C.c_longlong), ("dy_hm", C.c_int)]


class WgradGroupedArgs(C.Structure):
    """vault_wgrad_grouped_args"""
    _fields_ = [("nseg", C.c_int), ("seg", WgradSeg * 3), ("tokens", C.c_int), ("splits", C.c_int), ("accumulate", C.c_int),
                ("persist", C.c_int)]


def wgrad_grouped(segs, tokens, splits=1, accumulate=1):
    """One launch over up to three segments of weight-gradient tiles (vault_wgrad_grouped).  ``segs``: dicts with dy, x
    (16-bit [tokens, n] tensors of the kind's FIRST layer), dw (f32 view of its dW), n_out, n_in, batch, first, count,
    batch_dy, batch_x, batch_dw."""
    a = WgradGroupedArgs()
    a.nseg, a.tokens, a.splits, a.accumulate, a.persist = len(segs), tokens, splits, accumulate, GEMM_SCHED
    for k, g in enumerate(segs):
        t = a.seg[k]
        t.dy, t.x, t.dw = _h(g["dy"]), _h(g["x"]), _p(g["dw"])
        t.n_out, t.n_in, t.ld_dy, t.ld_x, t.ld_dw = g["n_out"], g["n_in"], g["n_out"], g["n_in"], g["n_in"]
        t.batch, t.first, t.count = g["batch"], g["first"], g["count"]
        t.batch_dy, t.batch_x, t.batch_dw = g["batch_dy"], g["batch_x"], g["batch_dw"]
        t.dy_hm = int(g.get("dy_hm", 0))
    _invoke("vault_wgrad_grouped", C.byref(a), _stream(), struct=a)


def quant_mxfp8(src_bf16, rows, K, ld, dst_q, dst_scale):
    """bf16 [rows][ld >= K] -> MXFP8: e4m3 bytes [rows][K] + E8M0 block scales [rows][K/32] (include/vault_hip.h)."""
    _invoke("vault_quant_mxfp8", C.c_void_p(_p(src_bf16)), C.c_longlong(rows), C.c_int(K), C.c_int(ld),
            C.c_void_p(_p(dst_q)), C.c_void_p(_p(dst_scale)), _stream())


def gemm_mxfp8(Aq, As, Bq, Bs, out, M, N, K, ldo, epi, *, m_valid=0, bias=None, res=None, out2=None,
               drop: Drop = NO_DROP, cfg=-1, aux_u8=False, out_hm=0, out_q=None, out_scale=None, plan_only=False):
    """out = epilogue(A . B^T) on MXFP8 operands (forward Linear layers of the fp8-forward configuration).  ``cfg``: -1 = the
    8-wave kernel's MXFP8 form where it takes the call (else the simple kernel), 0 = the simple kernel, 5 / 6 = the 8-wave form
    with 256- / 192-wide tiles; ``aux_u8`` (8-bit gelu' in tile order) and ``out_hm`` (head-major output) as ``gemm``: 8-wave
    form only; ``out_q`` / ``out_scale``: also write the MXFP8 image of the 16-bit output (kernel 5, vault_gemm_args.out_q)."""
    a = L.GemmArgs()
    a.out_q, a.out_scale = _p(out_q), _p(out_scale)
    a.A, a.B, a.out, a.out2 = _p(Aq), _p(Bq), _p(out), _p(out2)
    a.bias, a.res = _p(bias), _p(res)
    a.M, a.N, a.K, a.lda, a.ldb, a.ldo, a.m_valid = M, N, K, K, K, ldo, m_valid
    a.a_mode, a.b_mode, a.epi, a.cfg, a.splits, a.accumulate = 0, 0, epi, cfg, 1, 0
    a.aux_u8, a.out_hm = int(bool(aux_u8)), int(out_hm)
    a.persist = GEMM_SCHED
    a.drop_thresh, a.drop_seed, a.drop_stream, a.drop_scale = drop.thresh, drop.seed, drop.stream, drop.scale
    if plan_only:      # the kernel these arguments would run on (vault_gemm_mxfp8_plan): 0, 5, 6, or -EINVAL
        return int(L.load(_FMT.get()).vault_gemm_mxfp8_plan(C.byref(a)))
    _invoke("vault_gemm_mxfp8", C.byref(a), C.c_void_p(_p(As)), C.c_void_p(_p(Bs)), _stream(), struct=a, drop=drop)


def layernorm_fwd(x, gamma, beta, eps, rows, H, *, y_bf16=None, y_f32=None, mean=None, rstd=None, post_add=None,
                  xmap=(0, 0, 0), ymap=(0, 0, 0), drop: Drop = NO_DROP, y_split3=None, y_q=None, y_scale=None):
    a = LnFwdArgs()
    a.x, a.gamma, a.beta, a.post_add = _p(x), _p(gamma), _p(beta), _p(post_add)
    a.y_bf16, a.y_f32, a.mean, a.rstd = _h(y_bf16), _p(y_f32), _p(mean), _p(rstd)
    a.rows, a.H, a.eps = rows, H, eps
    a.x_rpg, a.x_gstride, a.x_goff = xmap
    a.y_rpg, a.y_gstride, a.y_goff = ymap
    a.drop_thresh, a.drop_seed, a.drop_stream, a.drop_scale = drop.thresh, drop.seed, drop.stream, drop.scale
    a.y_split3 = _h(y_split3)
    a.y_q, a.y_scale = _p(y_q), _p(y_scale)
    _invoke("vault_layernorm_fwd", C.byref(a), _stream(), struct=a, drop=drop)


def layernorm_bwd(x, mean, rstd, gamma, rows, H, *, dy_bf16=None, dy_f32=None, dres=None, dx_f32=None, dx_bf16=None,
                  dgamma=None, dbeta=None, dymap=(0, 0, 0), xmap=(0, 0, 0), dxmap=(0, 0, 0), drop: Drop = NO_DROP,
                  drop_on_dy: bool = False, dbias=None, dres_bf16=None):
    a = LnBwdArgs()
    a.dres_bf16 = _h(dres_bf16)
    a.dy_bf16, a.dy_f32, a.x, a.mean, a.rstd, a.gamma, a.dres = (_h(dy_bf16), _p(dy_f32), _p(x), _p(mean),
                                                                  _p(rstd), _p(gamma), _p(dres))
    a.dx_f32, a.dx_bf16, a.dgamma, a.dbeta = _p(dx_f32), _h(dx_bf16), _p(dgamma), _p(dbeta)
    a.rows, a.H = rows, H
    a.dy_rpg, a.dy_gstride, a.dy_goff = dymap
    a.x_rpg, a.x_gstride, a.x_goff = xmap
    a.dx_rpg, a.dx_gstride, a.dx_goff = dxmap
    a.drop_thresh, a.drop_seed, a.drop_stream, a.drop_scale = drop.thresh, drop.seed, drop.stream, drop.scale
    a.drop_on_dy = 1 if drop_on_dy else 0
    a.dbias = _p(dbias)
    _invoke("vault_layernorm_bwd", C.byref(a), _stream(), struct=a, drop=drop)


def colsum_batched(x_bf16, ld, rows, N, out, batch, batch_in, batch_out):
    _invoke("vault_colsum_batched", C.c_void_p(_p(x_bf16)), C.c_int(ld), C.c_int(rows), C.c_int(N), C.c_void_p(_p(out)),
            C.c_int(batch), C.c_longlong(batch_in), C.c_longlong(batch_out), _stream())


def colsum_hm(x_bf16, rows, hm_rows, planes, out, batch=1, batch_in=0, batch_out=0):
    _invoke("vault_colsum_hm", C.c_void_p(_h(x_bf16)), C.c_int(rows), C.c_int(hm_rows), C.c_int(planes), C.c_void_p(_p(out)),
            C.c_int(batch), C.c_longlong(batch_in), C.c_longlong(batch_out), _stream())


def colsum(x_bf16, ld, rows, N, out):
    _invoke("vault_colsum", C.c_void_p(_p(x_bf16)), C.c_int(ld), C.c_int(rows), C.c_int(N), C.c_void_p(_p(out)), _stream())


def _attn_args(qkv, keymask, ctx, lse, B, S, H, heads, dctx=None, dqkv=None, drop: Drop = NO_DROP, ctx_split3=None, qkv_hm=0):
    a = AttnArgs()
    a.qkv_hm = int(qkv_hm)      # > 0: qkv / dqkv in the head-major layout [3][heads][qkv_hm rows][64] (vault_attn_args.qkv_hm)
    a.ctx_split3 = _h(ctx_split3)
    a.qkv, a.keymask, a.ctx, a.lse, a.dctx, a.dqkv = _h(qkv), _p(keymask), _h(ctx), _p(lse), _h(dctx), _h(dqkv)
    a.B, a.S, a.H, a.heads = B, S, H, heads
    a.drop_thresh, a.drop_seed, a.drop_stream, a.drop_scale = drop.thresh, drop.seed, drop.stream, drop.scale
    return a


def attention_fwd(qkv, keymask, ctx, lse, B, S, H, heads, drop: Drop = NO_DROP, ctx_split3=None, qkv_hm=0):
    a = _attn_args(qkv, keymask, ctx, lse, B, S, H, heads, drop=drop, ctx_split3=ctx_split3, qkv_hm=qkv_hm)
    _invoke("vault_attention_fwd", C.byref(a), _stream(), struct=a, drop=drop)


def attention_bwd(qkv, keymask, ctx, lse, dctx, dqkv, B, S, H, heads, drop: Drop = NO_DROP, qkv_hm=0, bias_partials=None,
                  bias_thirds=1):
    """``bias_partials``: f32 [attention_bwd_partials(...), bias_thirds * H] - the launch's workgroups leave the column sums of
    dq (| dk | dv) over their items there (the QKV bias gradient without a pass over dqkv; colsum_partials adds the rows)."""
    a = _attn_args(qkv, keymask, ctx, lse, B, S, H, heads, dctx, dqkv, drop, qkv_hm=qkv_hm)
    a.bias_partials, a.bias_thirds = _p(bias_partials), (int(bias_thirds) if bias_partials is not None else 0)
    _invoke("vault_attention_bwd", C.byref(a), _stream(), struct=a, drop=drop)


def attention_bwd_partials(B, S, H, heads, thirds) -> int:
    """Rows of ``bias_partials`` a backward launch of this shape writes; 0 = the shape has no such form."""
    a = AttnArgs()
    a.B, a.S, a.H, a.heads, a.bias_thirds = B, S, H, heads, thirds
    return int(L.load(_FMT.get()).vault_attention_bwd_partials(C.byref(a)))


def colsum_partials(part_f32, nparts, n, out, batch=1, batch_in=0, batch_out=0):
    _invoke("vault_colsum_partials", C.c_void_p(_p(part_f32)), C.c_int(nparts), C.c_int(n), C.c_void_p(_p(out)), C.c_int(batch),
            C.c_longlong(batch_in), C.c_longlong(batch_out), _stream())


class GatherArgs(C.Structure):
    _fields_ = [("src", C.c_void_p), ("out", C.c_void_p),
                ("tab", C.c_void_p * 3), ("idx", C.c_void_p * 3), ("is64", C.c_int * 3), ("fixed", C.c_int * 3),
                ("period", C.c_int), ("rows", C.c_int), ("H", C.c_int), ("rowmask", C.c_void_p)]


class HeadArgs(C.Structure):
    _fields_ = [("pre", C.c_void_p), ("Wc", C.c_void_p), ("bc", C.c_void_p), ("labels", C.c_void_p),
                ("dlogits", C.c_void_p),
                ("pooled", C.c_void_p), ("logits", C.c_void_p), ("loss_sum", C.c_void_p), ("dWc", C.c_void_p),
                ("dbc", C.c_void_p), ("dpre_bf16", C.c_void_p),
                ("B", C.c_int), ("H", C.c_int), ("C", C.c_int), ("loss_scale", C.c_float), ("grad_scale", C.c_float),
                ("drop_thresh", C.c_uint32), ("drop_seed", C.c_uint32), ("drop_stream", C.c_uint32),
                ("drop_scale", C.c_float), ("targets", C.c_void_p), ("loss_kind", C.c_int)]


def position_ids(ids_i64, pos_i32, B, T, mode, pad):
    _invoke("vault_position_ids", C.c_void_p(_p(ids_i64)), C.c_void_p(_p(pos_i32)), C.c_int(B), C.c_int(T),
            C.c_int(mode), C.c_int(pad), _stream())


def _gather_args(src, out, tables, rows, H, period=1):
    """tables: list of up to 3 (table, index) with index = tensor (int64/int32) | int | "mod" (row % period)."""
    a = GatherArgs()
    a.src, a.out, a.rows, a.H, a.period = _p(src), _p(out), rows, H, period
    for k in range(3):
        if k < len(tables) and tables[k] is not None:
            tab, idx = tables[k]
            a.tab[k] = _p(tab)
            if isinstance(idx, torch.Tensor):
                a.idx[k] = _p(idx)
                a.is64[k] = 1 if idx.dtype == torch.int64 else 0
                a.fixed[k] = 0
            elif idx == "mod":
                a.idx[k], a.is64[k], a.fixed[k] = None, 0, -2
            else:
                a.idx[k], a.is64[k], a.fixed[k] = None, 0, int(idx)
        else:
            a.tab[k], a.idx[k], a.is64[k], a.fixed[k] = None, None, 0, -1
    return a


def gather_sum(src, out, tables, rows, H, period=1):
    a = _gather_args(src, out, tables, rows, H, period)
    _invoke("vault_gather_sum", C.byref(a), _stream())


def scatter_add(src, grad_tables, rows, H, period=1, rowmask=None):
    a = _gather_args(src, None, grad_tables, rows, H, period)
    a.rowmask = _p(rowmask)
    _invoke("vault_scatter_add", C.byref(a), _stream())


def im2col(pix, out_bf16, B, Cn, IMG, ps, split3=False):
    _invoke("vault_im2col", C.c_void_p(_p(pix)), C.c_void_p(_p(out_bf16)), C.c_int(B), C.c_int(Cn), C.c_int(IMG),
            C.c_int(ps), C.c_int(1 if split3 else 0), _stream())


def image_consts(bias, pos, mtype1, cls, addtab, x, P, H, B, S, T):
    _invoke("vault_image_consts", C.c_void_p(_p(bias)), C.c_void_p(_p(pos)), C.c_void_p(_p(mtype1)),
            C.c_void_p(_p(cls)), C.c_void_p(_p(addtab)), C.c_void_p(_p(x)), C.c_int(P), C.c_int(H), C.c_int(B),
            C.c_int(S), C.c_int(T), _stream())


def image_rows_bwd(dx, dpos, dmtype1, dcls, dbias, dyp_bf16, P, H, B, S, T):
    _invoke("vault_image_rows_bwd", C.c_void_p(_p(dx)), C.c_void_p(_p(dpos)), C.c_void_p(_p(dmtype1)),
            C.c_void_p(_p(dcls)), C.c_void_p(_p(dbias)), C.c_void_p(_p(dyp_bf16)), C.c_int(P), C.c_int(H), C.c_int(B),
            C.c_int(S), C.c_int(T), _stream())


def im2col_sel(pix, out_bf16, sel, B, L, Cn, HP, WP, ps, split3=False):
    _invoke("vault_im2col_sel", C.c_void_p(_p(pix)), C.c_void_p(_p(out_bf16)), C.c_void_p(_p(sel)), C.c_int(B), C.c_int(L),
            C.c_int(Cn), C.c_int(HP), C.c_int(WP), C.c_int(ps), C.c_int(1 if split3 else 0), _stream())


def image_sel_consts(bias, pos, mtype1, cls, addtab, x, L, H, B, S, T):
    _invoke("vault_image_sel_consts", C.c_void_p(_p(bias)), C.c_void_p(_p(pos)), C.c_void_p(_p(mtype1)),
            C.c_void_p(_p(cls)), C.c_void_p(_p(addtab)), C.c_void_p(_p(x)), C.c_int(L), C.c_int(H), C.c_int(B),
            C.c_int(S), C.c_int(T), _stream())


def image_pos_sel_fwd(x, pos, sel, hw, B, L, S, T, H, gw, G):
    _invoke("vault_image_pos_sel_fwd", C.c_void_p(_p(x)), C.c_void_p(_p(pos)), C.c_void_p(_p(sel)), C.c_void_p(_p(hw)),
            C.c_int(B), C.c_int(L), C.c_int(S), C.c_int(T), C.c_int(H), C.c_int(gw), C.c_int(G), _stream())


def image_sel_bwd(dx, dpos, dmtype1, dcls, dbias, dyp_bf16, sel, hw, B, L, S, T, H, gw, G):
    _invoke("vault_image_sel_bwd", C.c_void_p(_p(dx)), C.c_void_p(_p(dpos)), C.c_void_p(_p(dmtype1)), C.c_void_p(_p(dcls)),
            C.c_void_p(_p(dbias)), C.c_void_p(_p(dyp_bf16)), C.c_void_p(_p(sel)), C.c_void_p(_p(hw)), C.c_int(B),
            C.c_int(L), C.c_int(S), C.c_int(T), C.c_int(H), C.c_int(gw), C.c_int(G), _stream())


def axpy(dst, src, a, n):
    _invoke("vault_axpy_f32", C.c_void_p(_p(dst)), C.c_void_p(_p(src)), C.c_float(a), C.c_longlong(n), _stream())


def scale(x, a, n):
    """x *= a over n floats."""
    _invoke("vault_scale_f32", C.c_void_p(_p(x)), C.c_float(a), C.c_longlong(n), _stream())


def _head_args(pre, Wc, bc, labels, pooled, logits, loss_sum, B, H, Cc, loss_scale=1.0, grad_scale=1.0, dlogits=None,
               dWc=None, dbc=None, dpre=None, drop: Drop = NO_DROP):
    a = HeadArgs()
    a.pre, a.Wc, a.bc, a.labels, a.dlogits = _p(pre), _p(Wc), _p(bc), _p(labels), _p(dlogits)
    if labels is not None and labels.dtype.is_floating_point:   # float targets: BCE-with-logits (n_classes = 1)
        if labels.dtype != torch.float32:
            raise TypeError("BCE targets must be float32")
        a.labels, a.targets, a.loss_kind = None, _p(labels), 1
    a.pooled, a.logits, a.loss_sum, a.dWc, a.dbc, a.dpre_bf16 = (_p(pooled), _p(logits), _p(loss_sum), _p(dWc),
                                                                 _p(dbc), _p(dpre))
    a.B, a.H, a.C, a.loss_scale, a.grad_scale = B, H, Cc, loss_scale, grad_scale
    a.drop_thresh, a.drop_seed, a.drop_stream, a.drop_scale = drop.thresh, drop.seed, drop.stream, drop.scale
    return a


def head_fwd(pre, Wc, bc, labels, pooled, logits, loss_sum, B, H, Cc, loss_scale, drop: Drop = NO_DROP):
    a = _head_args(pre, Wc, bc, labels, pooled, logits, loss_sum, B, H, Cc, loss_scale=loss_scale, drop=drop)
    _invoke("vault_head_fwd", C.byref(a), _stream(), struct=a, drop=drop)


def head_bwd(pooled, logits, labels, Wc, dWc, dbc, dpre, B, H, Cc, grad_scale, dlogits=None, drop: Drop = NO_DROP):
    a = _head_args(None, Wc, None, labels, pooled, logits, None, B, H, Cc, grad_scale=grad_scale, dlogits=dlogits,
                   dWc=dWc, dbc=dbc, dpre=dpre, drop=drop)
    _invoke("vault_head_bwd", C.byref(a), _stream(), struct=a, drop=drop)


def tanh_bwd(pooled, dpooled, dpre_bf16, n):
    _invoke("vault_tanh_bwd", C.c_void_p(_p(pooled)), C.c_void_p(_p(dpooled)), C.c_void_p(_p(dpre_bf16)),
            C.c_longlong(n), _stream())


def gelu_fwd(x, y_bf16, n):
    _invoke("vault_gelu_fwd", C.c_void_p(_p(x)), C.c_void_p(_p(y_bf16)), C.c_longlong(n), _stream())


def gelu_fwd_f32(x, y, n):
    _invoke("vault_gelu_fwd_f32", C.c_void_p(_p(x)), C.c_void_p(_p(y)), C.c_longlong(n), _stream())


def gelu_bwd(x, dy, dx, n):
    _invoke("vault_gelu_bwd", C.c_void_p(_p(x)), C.c_void_p(_p(dy)), C.c_void_p(_p(dx)), C.c_longlong(n), _stream())


def adamw_step(p, g, m, v, p_bf16, n, lr, beta1, beta2, eps, weight_decay, bias_corr_factor=1.0, grad_scale=1.0,
               zero_grad=True, zero_mask=None):
    """``zero_mask``: uint8, one byte per 64 elements of the range (0 = the next backward stores there: not zeroed)."""
    _invoke("vault_adamw_step", C.c_void_p(_p(p)), C.c_void_p(_p(g)), C.c_void_p(_p(m)), C.c_void_p(_p(v)),
            C.c_void_p(_h(p_bf16)), C.c_longlong(n), C.c_float(lr), C.c_float(beta1), C.c_float(beta2), C.c_float(eps),
            C.c_float(weight_decay), C.c_float(bias_corr_factor), C.c_float(grad_scale),
            C.c_int(1 if zero_grad else 0), C.c_void_p(_p(zero_mask)), _stream())


def cast_bf16(x, y_bf16, n):
    _invoke("vault_cast_bf16", C.c_void_p(_p(x)), C.c_void_p(_h(y_bf16)), C.c_longlong(n), _stream())


def h16_census(x_h16, out4_u64):
    """Debug (VaultEngine VAULT_H16_CENSUS=1): out4 += [largest-finite-magnitude, non-finite, subnormal, zero] element counts
    of a 16-bit tensor in the current operand format (vault_h16_census)."""
    _invoke("vault_h16_census", C.c_void_p(_h(x_h16)), C.c_longlong(x_h16.numel()), C.c_void_p(_p(out4_u64)), _stream())


# ---- data-parallel gradient exchange (csrc/exchange.hip) ----
def rows_union(keys_i64, n_keys, V, flags_i32, uniq_i64, count_i32):
    _invoke("vault_rows_union", C.c_void_p(_p(keys_i64)), C.c_longlong(n_keys), C.c_int(V), C.c_void_p(_p(flags_i32)),
            C.c_void_p(_p(uniq_i64)), C.c_void_p(_p(count_i32)), _stream())


def rows_gather(table_f32, idx_i64, n_rows, H, out_f32):
    _invoke("vault_rows_gather_f32", C.c_void_p(_p(table_f32)), C.c_void_p(_p(idx_i64)), C.c_int(n_rows), C.c_int(H),
            C.c_void_p(_p(out_f32)), _stream())


def rows_scatter(src_f32, idx_i64, n_rows, H, table_f32):
    _invoke("vault_rows_scatter_f32", C.c_void_p(_p(src_f32)), C.c_void_p(_p(idx_i64)), C.c_int(n_rows), C.c_int(H),
            C.c_void_p(_p(table_f32)), _stream())


def sum_chunks_bf16(src_bf16, n_src, chunk, out_bf16):
    _invoke("vault_sum_chunks_bf16", C.c_void_p(_p(src_bf16)), C.c_int(n_src), C.c_longlong(chunk), C.c_void_p(_p(out_bf16)),
            _stream())


def widen_bf16(x_bf16, y_f32, n):
    _invoke("vault_widen_bf16", C.c_void_p(_p(x_bf16)), C.c_void_p(_p(y_f32)), C.c_longlong(n), _stream())


def split3_bf16(x_f32, out_bf16, rows, K, layout):
    _invoke("vault_split3_bf16", C.c_void_p(_p(x_f32)), C.c_void_p(_p(out_bf16)), C.c_longlong(rows), C.c_int(K),
            C.c_int(layout), _stream())


def transpose_bf16(src, dst, rows, cols, batch=1, stride_src=0, stride_dst=0):
    """dst[b][c][r] = src[b][r][c] (bf16; rows, cols multiples of 64): transposed weight shadow for the data gradients."""
    _invoke("vault_transpose_bf16", C.c_void_p(_h(src)), C.c_void_p(_h(dst)), C.c_int(rows), C.c_int(cols), C.c_int(batch),
            C.c_longlong(stride_src), C.c_longlong(stride_dst), _stream())


def rows_add(src, vec, out, rows, H, rpg, gstride, goff):
    _invoke("vault_rows_add_f32", C.c_void_p(_p(src)), C.c_void_p(_p(vec)), C.c_void_p(_p(out)), C.c_int(rows), C.c_int(H),
            C.c_int(rpg), C.c_int(gstride), C.c_int(goff), _stream())


def rows_gather_bwd(dx, dsrc, dvec, rows, H, rpg, gstride, goff):
    _invoke("vault_rows_gather_bwd_f32", C.c_void_p(_p(dx)), C.c_void_p(_p(dsrc)), C.c_void_p(_p(dvec)), C.c_int(rows),
            C.c_int(H), C.c_int(rpg), C.c_int(gstride), C.c_int(goff), _stream())


class LayerArgs(C.Structure):
    """vault_layer_args (include/vault_hip.h): one encoder layer's weights, activations and dropout description."""
    _fields_ = ([(n, C.c_int) for n in ("B", "S", "H", "FF", "heads", "rows", "rows_pad")] + [("eps", C.c_float)] +
                [(n, C.c_void_p) for n in ("wqkv", "wo", "wi", "wf", "wo_t", "wf_t", "bqkv", "bo", "bi", "bf", "ln1w", "ln1b",
                                           "ln2w", "ln2b", "x_in", "x_in_bf16", "x_out", "x_out_bf16", "keymask", "n1", "qkv",
                                           "ctx", "lse", "xm", "y1", "n2", "act", "u", "h2", "m1", "r1", "m2", "r2")] +
                [(n, C.c_uint32) for n in ("attn_drop_thresh", "hid_drop_thresh", "drop_seed", "drop_stream_base")] +
                [("attn_drop_scale", C.c_float), ("hid_drop_scale", C.c_float), ("persist", C.c_int),
                 ("splitk_ws", C.c_void_p), ("splitk_bytes", C.c_longlong)])


class LayerBwdArgs(C.Structure):
    _fields_ = ([("fwd", C.POINTER(LayerArgs))] +
                [(n, C.c_void_p) for n in ("dy_bf16", "dy_f32", "dx_f32", "dx_bf16", "dU", "dN", "dctx", "dqkv", "dmid_bf16",
                                           "dh1_bf16", "dmid_f32", "g_wqkv", "g_bqkv", "g_wo", "g_bo", "g_wi", "g_bi", "g_wf",
                                           "g_bf", "g_ln1w", "g_ln1b", "g_ln2w", "g_ln2b", "g_bf_below")] +
                [("do_wgrad", C.c_int)])


def _struct(name, doc, spec):
    """ctypes.Structure from a compact field list: ``"i:a b c"`` ints, ``"f:..."`` floats, ``"p:..."`` pointers, ``"u:..."`` uint32."""
    kinds = {"i": C.c_int, "f": C.c_float, "p": C.c_void_p, "u": C.c_uint32}
    fields = []
    for part in spec:
        k, names = part.split(":")
        fields += [(n, kinds[k]) for n in names.split()]
    return type(name, (C.Structure,), {"_fields_": fields, "__doc__": doc})


# ABI 5 stage structs (include/vault_hip.h), member for member
LmEmbedArgs = _struct("LmEmbedArgs", "vault_lm_embed_args", [
    "i:B T H rows_pad pos_mode pad_id", "f:eps", "p:ids token_type_ids inputs_embeds word pos type lnw lnb pos_ids esum mean rstd y y_bf16",
    "u:drop_thresh drop_seed drop_stream", "f:drop_scale", "p:dy_bf16 dy_f32 desum rowmask g_word g_pos g_type g_lnw g_lnb"])
TextEmbedArgs = _struct("TextEmbedArgs", "vault_text_embed_args", [
    "i:B T S H rows_pad", "f:eps", "p:text_src ids token_type_ids word pos type lnw lnb mtype0 vsum mean rstd x",
    "p:dx dvsum dbeta_scratch g_word g_pos g_type g_lnw g_lnb g_mtype0"])
PatchEmbedArgs = _struct("PatchEmbedArgs", "vault_patch_embed_args", [
    "i:B C IMG ps T S H", "p:pixel_values w_bf16 conv_bias pos_emb mtype1 cls apatch addtab x", "i:persist",
    "p:dx dyp g_w g_conv_bias g_pos g_mtype1 g_cls"])
HeadLossArgs = _struct("HeadLossArgs", "vault_head_loss_args", [
    "i:B S H C seq_rows_pad", "f:eps", "p:x lnw lnb wp_bf16 bp Wc bc labels targets", "i:loss_kind", "f:loss_scale grad_scale",
    "p:h0_bf16 mean rstd pre pooled logits loss", "u:drop_thresh drop_seed drop_stream", "f:drop_scale", "i:persist",
    "p:dpre dh0 dx_f32 dx_bf16 g_Wc g_bc g_wp g_bp g_lnw g_lnb g_bf_last"])
ModelDims = _struct("ModelDims", "vault_model_dims", ["i:H FF heads lm_layers vilt_layers IMG ps C n_classes"])


def stage_args(struct_type, /, **kw):
    """Fill one of the stage structs from tensors / numbers (``persist`` = the process-wide GEMM scheduling mode)."""
    a = struct_type()
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            v = _p(v)
        setattr(a, k, v)
    if hasattr(a, "persist"):
        a.persist = GEMM_SCHED
    a._keep = keep
    return a


def workspace_bytes(H, FF, heads, lm_layers, vilt_layers, IMG, ps, Cn, n_classes, B, T, train) -> int:
    fn = L.load(_FMT.get()).vault_workspace_bytes
    fn.restype = C.c_longlong
    d = ModelDims(H, FF, heads, lm_layers, vilt_layers, IMG, ps, Cn, n_classes)
    return int(fn(C.byref(d), C.c_int(B), C.c_int(T), C.c_int(1 if train else 0)))


def layer_args(**kw) -> LayerArgs:
    a = LayerArgs()
    for k, v in kw.items():
        setattr(a, k, _p(v) if isinstance(v, torch.Tensor) else v)
    a.persist = GEMM_SCHED
    return a


def layer_bwd_args(fwd: LayerArgs, **kw) -> LayerBwdArgs:
    g = LayerBwdArgs()
    g.fwd = C.pointer(fwd)
    for k, v in kw.items():
        setattr(g, k, _p(v) if isinstance(v, torch.Tensor) else v)
    g._fwd_keep = fwd     # (the struct holds a raw pointer: keep the Python object alive with it)
    return g


def layer_call(name: str, args, seeded: bool = False):
    """vault_{vilt,lm}_layer_{fwd,bwd}: one layer per C call.  ``seeded``: the struct carries a dropout seed the tape
    re-keys between replays (for the backward form: its forward struct)."""
    fn = getattr(L.load(_FMT.get()), name)
    call = (C.byref(args), _stream())
    if _TAPE is not None:
        _TAPE.calls.append((fn, call))
        if seeded:
            _TAPE.seeded.append(args if isinstance(args, LayerArgs) else args._fwd_keep)
    L.check(fn(*call), name)
