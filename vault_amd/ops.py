"""Thin host wrappers: torch tensors (device memory + stream only) -> C-ABI calls of libvault_hip.so.

No arithmetic happens here; every function enqueues HIP kernels on torch's current stream.
"""
from __future__ import annotations

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
                ("drop_scale", C.c_float)]


class LnBwdArgs(C.Structure):
    _fields_ = [("dy_bf16", C.c_void_p), ("dy_f32", C.c_void_p), ("x", C.c_void_p), ("mean", C.c_void_p),
                ("rstd", C.c_void_p), ("gamma", C.c_void_p), ("dres", C.c_void_p),
                ("dx_f32", C.c_void_p), ("dx_bf16", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("rows", C.c_int), ("H", C.c_int),
                ("dy_rpg", C.c_int), ("dy_gstride", C.c_int), ("dy_goff", C.c_int),
                ("x_rpg", C.c_int), ("x_gstride", C.c_int), ("x_goff", C.c_int),
                ("dx_rpg", C.c_int), ("dx_gstride", C.c_int), ("dx_goff", C.c_int),
                ("drop_thresh", C.c_uint32), ("drop_seed", C.c_uint32), ("drop_stream", C.c_uint32),
                ("drop_scale", C.c_float)]


class AttnArgs(C.Structure):
    _fields_ = [("qkv", C.c_void_p), ("keymask", C.c_void_p), ("ctx", C.c_void_p), ("lse", C.c_void_p),
                ("dctx", C.c_void_p), ("dqkv", C.c_void_p),
                ("B", C.c_int), ("S", C.c_int), ("H", C.c_int), ("heads", C.c_int),
                ("drop_thresh", C.c_uint32), ("drop_seed", C.c_uint32), ("drop_stream", C.c_uint32),
                ("drop_scale", C.c_float)]


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


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


def gemm(A, B, out, M, N, K, lda, ldb, ldo, a_mode, b_mode, epi, *, cfg=-1, m_valid=0, splits=1, accumulate=0,
         bias=None, res=None, aux=None, out2=None, addtab=None, rpg=0, gstride=0, goff=0, drop: Drop = NO_DROP):
    a = L.GemmArgs()
    a.A, a.B, a.out, a.out2 = _p(A), _p(B), _p(out), _p(out2)
    a.bias, a.res, a.aux, a.addtab = _p(bias), _p(res), _p(aux), _p(addtab)
    a.M, a.N, a.K, a.lda, a.ldb, a.ldo, a.m_valid = M, N, K, lda, ldb, ldo, m_valid
    a.a_mode, a.b_mode, a.epi, a.cfg, a.splits, a.accumulate = a_mode, b_mode, epi, cfg, splits, accumulate
    a.rpg, a.gstride, a.goff = rpg, gstride, goff
    a.drop_thresh, a.drop_seed, a.drop_stream, a.drop_scale = drop.thresh, drop.seed, drop.stream, drop.scale
    L.check(L.load().vault_gemm(C.byref(a), _stream()), "vault_gemm")


def layernorm_fwd(x, gamma, beta, eps, rows, H, *, y_bf16=None, y_f32=None, mean=None, rstd=None, post_add=None,
                  xmap=(0, 0, 0), ymap=(0, 0, 0), drop: Drop = NO_DROP):
    a = LnFwdArgs()
    a.x, a.gamma, a.beta, a.post_add = _p(x), _p(gamma), _p(beta), _p(post_add)
    a.y_bf16, a.y_f32, a.mean, a.rstd = _p(y_bf16), _p(y_f32), _p(mean), _p(rstd)
    a.rows, a.H, a.eps = rows, H, eps
    a.x_rpg, a.x_gstride, a.x_goff = xmap
    a.y_rpg, a.y_gstride, a.y_goff = ymap
    a.drop_thresh, a.drop_seed, a.drop_stream, a.drop_scale = drop.thresh, drop.seed, drop.stream, drop.scale
    L.check(L.load().vault_layernorm_fwd(C.byref(a), _stream()), "vault_layernorm_fwd")


def layernorm_bwd(x, mean, rstd, gamma, rows, H, *, dy_bf16=None, dy_f32=None, dres=None, dx_f32=None, dx_bf16=None,
                  dgamma=None, dbeta=None, dymap=(0, 0, 0), xmap=(0, 0, 0), dxmap=(0, 0, 0), drop: Drop = NO_DROP):
    a = LnBwdArgs()
    a.dy_bf16, a.dy_f32, a.x, a.mean, a.rstd, a.gamma, a.dres = (_p(dy_bf16), _p(dy_f32), _p(x), _p(mean),
                                                                  _p(rstd), _p(gamma), _p(dres))
    a.dx_f32, a.dx_bf16, a.dgamma, a.dbeta = _p(dx_f32), _p(dx_bf16), _p(dgamma), _p(dbeta)
    a.rows, a.H = rows, H
    a.dy_rpg, a.dy_gstride, a.dy_goff = dymap
    a.x_rpg, a.x_gstride, a.x_goff = xmap
    a.dx_rpg, a.dx_gstride, a.dx_goff = dxmap
    a.drop_thresh, a.drop_seed, a.drop_stream, a.drop_scale = drop.thresh, drop.seed, drop.stream, drop.scale
    L.check(L.load().vault_layernorm_bwd(C.byref(a), _stream()), "vault_layernorm_bwd")


def colsum(x_bf16, ld, rows, N, out):
    lib = L.load()
    L.check(lib.vault_colsum(C.c_void_p(_p(x_bf16)), C.c_int(ld), C.c_int(rows), C.c_int(N), C.c_void_p(_p(out)),
                             _stream()), "vault_colsum")


def _attn_args(qkv, keymask, ctx, lse, B, S, H, heads, dctx=None, dqkv=None, drop: Drop = NO_DROP):
    a = AttnArgs()
    a.qkv, a.keymask, a.ctx, a.lse, a.dctx, a.dqkv = _p(qkv), _p(keymask), _p(ctx), _p(lse), _p(dctx), _p(dqkv)
    a.B, a.S, a.H, a.heads = B, S, H, heads
    a.drop_thresh, a.drop_seed, a.drop_stream, a.drop_scale = drop.thresh, drop.seed, drop.stream, drop.scale
    return a


def attention_fwd(qkv, keymask, ctx, lse, B, S, H, heads, drop: Drop = NO_DROP):
    a = _attn_args(qkv, keymask, ctx, lse, B, S, H, heads, drop=drop)
    L.check(L.load().vault_attention_fwd(C.byref(a), _stream()), "vault_attention_fwd")


def attention_bwd(qkv, keymask, ctx, lse, dctx, dqkv, B, S, H, heads, drop: Drop = NO_DROP):
    a = _attn_args(qkv, keymask, ctx, lse, B, S, H, heads, dctx, dqkv, drop)
    L.check(L.load().vault_attention_bwd(C.byref(a), _stream()), "vault_attention_bwd")
