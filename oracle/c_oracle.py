"""ctypes loader for oracle/_build/libgdkvm_oracle.so (scalar C restatement).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by gdkvm_amd/."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgdkvm_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_u8 = ctypes.POINTER(ctypes.c_uint8)
_i32 = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, n) for n in ("gdkvm_oracle.c", "gdkvm_oracle_impl.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        for suf in ("_f64", "_f32"):
            fn = getattr(_lib, "gdkvm_oracle_scan" + suf)
            fn.restype = ctypes.c_int
            fn.argtypes = [_f] * 6 + [_f, _f] + [ctypes.c_int] * 8
            fn = getattr(_lib, "gdkvm_oracle_kpff" + suf)
            fn.restype = ctypes.c_int
            fn.argtypes = [_f] * 7 + [_f] + [ctypes.c_int] * 6
            fn = getattr(_lib, "gdkvm_oracle_scan_normalizer" + suf)
            fn.restype = ctypes.c_int
            fn.argtypes = [_f] * 7 + [_f] * 3 + [ctypes.c_int] * 8 + [ctypes.c_double]
        _lib.gdkvm_oracle_argmax_dice.restype = ctypes.c_int
        _lib.gdkvm_oracle_argmax_dice.argtypes = [_f, _u8, _u8, _i32] + [ctypes.c_int] * 4
        _lib.gdkvm_oracle_upsample_argmax_dice.restype = ctypes.c_int
        _lib.gdkvm_oracle_upsample_argmax_dice.argtypes = [_f, _u8, _u8, _i32] + [ctypes.c_int] * 6
        _lib.gdkvm_oracle_num_threads.restype = ctypes.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f) if a is not None else None


def _c32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def scan(q, k, v, alpha, beta, s0=None, rule=2, flags=0, math="f64"):
    """q,k [B,T,N,Hh,Dk] v [B,T,N,Hh,Dv] alpha [B,T,Hh] beta [B,T,N,Hh] -> (R fp32, S_T fp32)."""
    q, k, v, alpha, beta, s0 = map(_c32, (q, k, v, alpha, beta, s0))
    B, T, N, Hh, Dk = q.shape
    Dv = v.shape[-1]
    r = np.empty((B, T, N, Hh, Dv), dtype=np.float32)
    s = np.empty((B, Hh, Dk, Dv), dtype=np.float32)
    rc = getattr(lib(), "gdkvm_oracle_scan_" + math)(_fp(q), _fp(k), _fp(v), _fp(alpha), _fp(beta), _fp(s0),
                                                     _fp(r), _fp(s), B, T, N, Hh, Dk, Dv, rule, flags)
    if rc != 0:
        raise RuntimeError(f"gdkvm_oracle_scan failed: {rc}")
    return r, s


def scan_normalizer(q, k, v, alpha, beta, s0=None, z0=None, rule=2, flags=0, eps=1e-6, math="f64"):
    """The scan with the `normalizer` flag: z [B,Hh,Dk] carried beside S, read-out divided by |q . z| + eps.
    -> (R fp32, S_T fp32, z_T fp32)."""
    q, k, v, alpha, beta, s0, z0 = map(_c32, (q, k, v, alpha, beta, s0, z0))
    B, T, N, Hh, Dk = q.shape
    Dv = v.shape[-1]
    r = np.empty((B, T, N, Hh, Dv), dtype=np.float32)
    s = np.empty((B, Hh, Dk, Dv), dtype=np.float32)
    z = np.empty((B, Hh, Dk), dtype=np.float32)
    rc = getattr(lib(), "gdkvm_oracle_scan_normalizer_" + math)(_fp(q), _fp(k), _fp(v), _fp(alpha), _fp(beta), _fp(s0), _fp(z0),
                                                                _fp(r), _fp(s), _fp(z), B, T, N, Hh, Dk, Dv, rule, flags, float(eps))
    if rc != 0:
        raise RuntimeError(f"gdkvm_oracle_scan_normalizer failed: {rc}")
    return r, s, z


def kpff(L, G, P, Wa, ba, Wl, Wg, h, w, math="f64"):
    L, G, P, Wa, ba, Wl, Wg = map(_c32, (L, G, P, Wa, ba, Wl, Wg))
    BT, N, Ck = L.shape
    Cv, Cp = G.shape[-1], P.shape[-1]
    assert N == h * w
    F = np.empty((BT, N, Cp), dtype=np.float32)
    rc = getattr(lib(), "gdkvm_oracle_kpff_" + math)(_fp(L), _fp(G), _fp(P), _fp(Wa), _fp(ba), _fp(Wl), _fp(Wg),
                                                     _fp(F), BT, Ck, Cv, Cp, h, w)
    if rc != 0:
        raise RuntimeError(f"gdkvm_oracle_kpff failed: {rc}")
    return F


def argmax_dice(logits, target=None):
    """logits [BT,ncls,H,W] -> (mask uint8 [BT,H,W], counts int32 [BT,ncls,3] or None)."""
    logits = _c32(logits)
    BT, ncls, H, W = logits.shape
    mask = np.empty((BT, H, W), dtype=np.uint8)
    counts = None
    tp = None
    if target is not None:
        target = np.ascontiguousarray(target, dtype=np.uint8)
        counts = np.zeros((BT, ncls, 3), dtype=np.int32)
        tp = target.ctypes.data_as(_u8)
    rc = lib().gdkvm_oracle_argmax_dice(_fp(logits), tp, mask.ctypes.data_as(_u8),
                                        counts.ctypes.data_as(_i32) if counts is not None else None,
                                        BT, ncls, H, W)
    if rc != 0:
        raise RuntimeError(f"gdkvm_oracle_argmax_dice failed: {rc}")
    return mask, counts


def upsample_argmax_dice(logits, H, W, target=None):
    """low-res logits [BT,ncls,hl,wl] -> bilinear (align_corners=False) to HxW -> (mask u8 [BT,H,W], counts | None)."""
    logits = _c32(logits)
    BT, ncls, hl, wl = logits.shape
    mask = np.empty((BT, H, W), dtype=np.uint8)
    counts, tp = None, None
    if target is not None:
        target = np.ascontiguousarray(target, dtype=np.uint8)
        counts = np.zeros((BT, ncls, 3), dtype=np.int32)
        tp = target.ctypes.data_as(_u8)
    rc = lib().gdkvm_oracle_upsample_argmax_dice(_fp(logits), tp, mask.ctypes.data_as(_u8),
                                                 counts.ctypes.data_as(_i32) if counts is not None else None,
                                                 BT, ncls, hl, wl, H, W)
    if rc != 0:
        raise RuntimeError(f"gdkvm_oracle_upsample_argmax_dice failed: {rc}")
    return mask, counts


def num_threads() -> int:
    return lib().gdkvm_oracle_num_threads()
