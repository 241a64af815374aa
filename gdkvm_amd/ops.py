"""Host-side bindings of the C ABI (include/gdkvm.h) for torch tensors.

Every function checks shapes on the host, hands raw device pointers + the current HIP stream to
libgdkvm_hip.so and returns torch tensors.  No fallback path exists: a missing library or a CPU tensor
raises (the product must never silently run anything but the HIP kernels).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch

RULE_GATED_LINEAR, RULE_DELTA_PARALLEL, RULE_DELTA_SEQUENTIAL = 0, 1, 2
FLAG_NORMALIZE_QK, FLAG_GATE_LOGITS, FLAG_TRAIN, FLAG_WIDE_RANGE = 1, 2, 4, 8


def recurrence_flags(rule: int, flags: int) -> int:
    """Flags for the split prep / apply / transition entry points: rule delta_parallel is not contractive, so its state
    recurrence runs on the full-range three-term operands (gdkvm_scan_fwd adds the flag itself)."""
    return flags | FLAG_WIDE_RANGE if rule == RULE_DELTA_PARALLEL else flags
F32, BF16 = 0, 1

_PKG = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_PKG, "libgdkvm_hip.so")
_lib = None
_vp, _sz, _i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int

# symbol -> (restype, argtypes): must list every entry point include/gdkvm.h declares
SIGNATURES = {
    "gdkvm_abi_version": (_i, []),
    "gdkvm_last_error": (ctypes.c_char_p, []),
    "gdkvm_scan_workspace_bytes": (_sz, [_i] * 6),
    "gdkvm_scan_fwd": (_i, [_vp] * 10 + [_sz] + [_i] * 9 + [_vp]),
    "gdkvm_scan_prep": (_i, [_vp] * 5 + [_sz] + [_i] * 9 + [_vp]),
    "gdkvm_scan_apply": (_i, [_vp] * 7 + [_sz] + [_i] * 8 + [_vp]),
    "gdkvm_scan_prep_normed": (_i, [_vp] * 6 + [_sz] + [_i] * 9 + [_vp]),
    "gdkvm_scan_fwd_normed": (_i, [_vp] * 10 + [_sz] + [_i] * 9 + [_vp]),
    "gdkvm_scan_transition": (_i, [_vp] * 4 + [_sz] + [_i] * 8 + [_vp]),
    "gdkvm_scan_stitch": (_i, [_vp] * 5 + [_i] * 5 + [_vp]),
    "gdkvm_scan_status": (_i, [_vp, _sz] + [_i] * 7 + [_vp]),
    "gdkvm_scan_bwd_workspace_bytes": (_sz, [_i] * 6),
    "gdkvm_scan_state_bwd": (_i, [_vp] * 6 + [_sz] + [_vp] * 8 + [_sz] + [_i] * 9 + [_vp]),
    "gdkvm_scan_bwd": (_i, [_vp] * 7 + [_sz] + [_vp] * 9 + [_sz] + [_i] * 9 + [_vp]),
    "gdkvm_scan_segments": (_i, [_i] * 5),
    "gdkvm_scan_segmented_workspace_bytes": (_sz, [_i] * 7),
    "gdkvm_scan_fwd_segmented": (_i, [_vp] * 9 + [_sz] + [_i] * 10 + [_vp]),
    "gdkvm_scan_normalizer_workspace_bytes": (_sz, [_i] * 7),
    "gdkvm_scan_fwd_normalizer": (_i, [_vp] * 11 + [_sz] + [_i] * 9 + [ctypes.c_float, _vp]),
    "gdkvm_lkva_read": (_i, [_vp] * 4 + [_i] * 7 + [_vp]),
    "gdkvm_mask_embed_add": (_i, [_vp] * 3 + [_i] * 7 + [_vp]),
    "gdkvm_scan_train_workspace_bytes": (_sz, [_i] * 7),
    "gdkvm_scan_train_fwd": (_i, [_vp] * 9 + [_sz] + [_i] * 9 + [_vp]),
    "gdkvm_scan_train_bwd": (_i, [_vp] * 14 + [_sz] + [_i] * 9 + [_vp]),
    "gdkvm_readout_fwd": (_i, [_vp] * 3 + [_i] * 9 + [_vp]),
    "gdkvm_readout_bwd": (_i, [_vp] * 5 + [_i] * 9 + [_vp]),
    "gdkvm_gemm_nt": (_i, [_vp] * 4 + [_i] * 4 + [_vp]),
    "gdkvm_gemm_tn_workspace_bytes": (_sz, [_i] * 3),
    "gdkvm_gemm_tn": (_i, [_vp] * 4 + [_sz] + [_i] * 4 + [_vp]),
    "gdkvm_gemm_tn_colsum_workspace_bytes": (_sz, [_i] * 3),
    "gdkvm_gemm_tn_colsum": (_i, [_vp] * 5 + [_sz] + [_i] * 4 + [_vp]),
    "gdkvm_kpff_workspace_bytes": (_sz, [_i] * 4),
    "gdkvm_kpff_fwd": (_i, [_vp] * 9 + [_sz] + [_i] * 7 + [_vp]),
    "gdkvm_kpff_fwd_packed": (_i, [_vp] * 9 + [_sz] + [_i] * 7 + [_vp]),
    "gdkvm_kpff_fwd_train": (_i, [_vp] * 13 + [_sz] + [_i] * 7 + [_vp]),
    "gdkvm_kpff_bwd_pre": (_i, [_vp] * 7 + [_i] * 4 + [_vp]),
    "gdkvm_kpff_bwd_post": (_i, [_vp] * 7 + [_i] * 7 + [_vp]),
    "gdkvm_argmax_dice": (_i, [_vp] * 4 + [_i] * 5 + [_vp]),
    "gdkvm_bias_act": (_i, [_vp] * 4 + [_sz] + [_i] * 3 + [_vp]),
    "gdkvm_head_logits": (_i, [_vp] * 4 + [_i] * 6 + [_vp]),
    "gdkvm_head_bwd_workspace_bytes": (_sz, [_i] * 2),
    "gdkvm_head_bwd": (_i, [_vp] * 7 + [_sz] + [_i] * 6 + [_vp]),
    "gdkvm_proj_rows": (_i, [_vp] * 6 + [ctypes.c_longlong] + [_i] * 5 + [_vp]),
    "gdkvm_stem_conv_pool": (_i, [_vp] * 4 + [_i] * 4 + [_vp]),
    "gdkvm_stem_conv_pool_nchw": (_i, [_vp] * 4 + [_i] * 5 + [_vp]),
    "gdkvm_stem_conv_nchw": (_i, [_vp] * 3 + [_i] * 5 + [_vp]),
    "gdkvm_stem_pack_s2d": (_i, [_vp, _vp, _i] + [ctypes.c_longlong] * 4 + [_vp]),
    "gdkvm_stem_wgrad_workspace_bytes": (_sz, [_i] * 3),
    "gdkvm_stem_wgrad_nchw": (_i, [_vp] * 3 + [ctypes.c_longlong] * 4 + [_vp, _sz] + [_i] * 5 + [_vp]),
    "gdkvm_gate_logits": (_i, [_vp] * 7 + [_i] * 5 + [_vp]),
    "gdkvm_proj_gates": (_i, [_vp] * 13 + [_i] * 7 + [_vp]),
    "gdkvm_conv_bias_act": (_i, [_vp] * 5 + [_i] * 12 + [_vp]),
    "gdkvm_conv3x3_pack_weights": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "gdkvm_conv_igemm_pack_weights": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "gdkvm_conv_down_bias_act": (_i, [_vp] * 4 + [_i] + [_vp] * 3 + [_i] * 10 + [_vp]),
    "gdkvm_conv_cat_bias_act": (_i, [_vp] * 6 + [_i] * 9 + [_vp]),
    "gdkvm_conv_s2_dgrad_pack_bytes": (_sz, [_i] * 3),
    "gdkvm_conv_s2_pack_train": (_i, [_vp] * 7 + [_i, _i, _vp]),
    "gdkvm_conv_s2_dgrad": (_i, [_vp] * 4 + [_i] * 7 + [_vp]),
    "gdkvm_conv_wgrad_strided_workspace_bytes": (_sz, [_i] * 9),
    "gdkvm_conv_wgrad_strided": (_i, [_vp] * 3 + [ctypes.c_longlong] * 4 + [_vp, _sz] + [_i] * 10 + [_vp]),
    "gdkvm_conv3x3_pack_weights_dgrad": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "gdkvm_conv3x3_pack_weights_train": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gdkvm_conv3x3_wgrad_workspace_bytes": (_sz, [_i] * 5),
    "gdkvm_conv3x3_wgrad": (_i, [_vp, _vp, _vp, _vp, _sz] + [_i] * 6 + [_vp]),
    "gdkvm_conv3x3_wgrad_krsc": (_i, [_vp, _vp, _vp, _vp, _sz] + [_i] * 6 + [_vp]),
    "gdkvm_upsample_cat": (_i, [_vp] * 3 + [_i] * 8 + [_vp]),
    "gdkvm_upsample_cat_bwd": (_i, [_vp] * 3 + [_i] * 8 + [_vp]),
    "gdkvm_bias_relu_maxpool": (_i, [_vp] * 3 + [_i] * 7 + [_vp]),
    "gdkvm_stem_s2d": (_i, [_vp] * 2 + [_i] * 6 + [_vp]),
    "gdkvm_upsample_argmax_dice": (_i, [_vp] * 4 + [_i] * 7 + [_vp]),
    "gdkvm_head_upsample_argmax_dice": (_i, [_vp] * 6 + [_i] * 8 + [_vp]),
    "gdkvm_maxpool_fwd": (_i, [_vp] * 3 + [_i] * 5 + [_vp]),
    "gdkvm_maxpool_bwd": (_i, [_vp] * 3 + [_i] * 5 + [_vp]),
    "gdkvm_seg_loss_workspace_bytes": (_sz, [_i]),
    "gdkvm_seg_loss_fwd": (_i, [_vp] * 4 + [_sz] + [_i] * 6 + [ctypes.c_float] * 2 + [_i, _i, _vp]),
    "gdkvm_seg_loss_bwd": (_i, [_vp] * 3 + [_sz] + [_vp] * 2 + [_i] * 8 + [_vp]),
    "gdkvm_bn_workspace_bytes": (_sz, [_i]),
    "gdkvm_bn_fwd_train": (_i, [_vp] * 9 + [_sz, ctypes.c_longlong, _i, ctypes.c_float, ctypes.c_float, _i, _i, _vp]),
    "gdkvm_bn_bwd": (_i, [_vp] * 10 + [_sz, ctypes.c_longlong, _i, _i, _i, _vp]),
    "gdkvm_bn_pool_fwd_train": (_i, [_vp] * 9 + [_sz] + [_i] * 4 + [ctypes.c_float] * 2 + [_i, _vp]),
    "gdkvm_bn_pool_bwd": (_i, [_vp] * 9 + [_sz] + [_i] * 5 + [_vp]),
}


class GdkvmError(RuntimeError):
    pass


def library_path() -> str:
    return _SO


def load():
    """dlopen libgdkvm_hip.so and bind every ABI symbol.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise GdkvmError(f"{_SO} not found: build it with `python -m gdkvm_amd.build` "
                             "(there is no fallback path)")
        lib = ctypes.CDLL(_SO)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.gdkvm_abi_version() != 1:
            raise GdkvmError(f"ABI version mismatch: {lib.gdkvm_abi_version()}")
        _lib = lib
    return _lib


def require_native():
    """Fail loudly unless the HIP library is loaded and a GPU is present."""
    load()
    if not torch.cuda.is_available():
        raise GdkvmError("no HIP device visible: the GDKVM ops have no CPU path")


def _check(rc: int, what: str):
    if rc != 0:
        raise GdkvmError(f"{what} failed ({rc}): {load().gdkvm_last_error().decode()}")


def _io_dtype(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise GdkvmError(f"unsupported io dtype {t.dtype} (float32 or bfloat16)")


def _dev(*ts):
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise GdkvmError("GDKVM ops need device tensors (no CPU path)")
        if not t.is_contiguous():
            raise GdkvmError("GDKVM ops need contiguous tensors")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise GdkvmError("tensors on different devices")
    return dev


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None or t.numel() == 0 else t.data_ptr()


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


KERNEL_DK = 64                # per-head key dim the fast kernels are built for (include/gdkvm.h; narrower keys run on them through zero
                              # channels, wider ones -- up to 256 -- on the general kernel of csrc/gdr_general.hip, inference only)


def _pad_keys(q, k, state):
    """Key dims below 64 run on the Dk = 64 kernels with zero channels appended: norms, Gram matrices and read-outs are
    unchanged, P stays the identity on the extra rows and a zero extra block of the state stays zero -- the result on the real
    rows is exactly that of the narrower problem (tests/test_scan_gpu.py).  Layout plumbing only; nothing is computed here."""
    pad = KERNEL_DK - q.shape[-1]
    Fn = torch.nn.functional
    return Fn.pad(q, (0, pad)), Fn.pad(k, (0, pad)), (None if state is None else Fn.pad(state, (0, 0, 0, pad)))


def scan_workspace_bytes(B, T, Hh, N, Dk, Dv) -> int:
    return int(load().gdkvm_scan_workspace_bytes(B, T, Hh, N, Dk, Dv))


def scan_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, alpha: torch.Tensor, beta: torch.Tensor,
             state: Optional[torch.Tensor] = None, rule: int = RULE_DELTA_SEQUENTIAL, flags: int = 0,
             workspace: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
             state_out: Optional[torch.Tensor] = None, state_hist: Optional[torch.Tensor] = None,
             readout: bool = True, norms: Optional[torch.Tensor] = None, check: bool = False) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
    """Fused LKVA read + GDR write over T frames (gdkvm_scan_fwd).

    q,k [B,T,N,Hh,Dk]  v [B,T,N,Hh,Dv]  (f32|bf16)   alpha [B,T,Hh]  beta [B,T,N,Hh]  state [B,Hh,Dk,Dv] (f32)
    returns (R [B,T,N,Hh,Dv] in the io dtype, S_T [B,Hh,Dk,Dv] f32).
    Range: the default recurrence carries the state as fp16 pairs at 2^-e with e sized from the call's own bound on the state
    (include/gdkvm.h, GDKVM_FLAG_WIDE_RANGE), so values and carried states of any magnitude are served for rules 0 and 2 (with unit-norm keys and gates in [0, 1]: flags=3, or inputs the caller normalised); the
    one refusal -- frames of more than 64 tokens with values ~1e5x the usual -- returns NaNs, and FLAG_WIDE_RANGE serves it.
    check=True: wait for the call and raise GdkvmError if the data left that range (gdkvm_scan_status; synchronises the stream).
    norms [B*T*N, Hh, 2] fp32 (ops.proj_gates): the inverse key / query norms came with the projections (gdkvm_scan_fwd_normed:
    the frame-parallel kernel neither reads q nor reduces anything in its first phase); needs FLAG_NORMALIZE_QK, Dk = 64."""
    lib = load()
    if q.dim() != 5 or k.shape != q.shape or v.dim() != 5 or v.shape[:4] != q.shape[:4]:
        raise GdkvmError(f"bad shapes q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)}")
    if q.shape[-1] < KERNEL_DK and state_hist is None and q.shape[-1] % 8:      # key widths that are no multiple of 8: padded here
        qp, kp, sp = _pad_keys(q, k, state)                # (multiples of 8 below 64: gdkvm_scan_fwd zero-extends them itself)
        r, s = scan_fwd(qp, kp, v, alpha, beta, sp, rule, flags, workspace, out, None, None, readout, None, check)
        s = s[:, :, :q.shape[-1]].contiguous()
        if state_out is not None:
            state_out.copy_(s)
            s = state_out
        return r, s
    B, T, N, Hh, Dk = q.shape
    Dv = v.shape[-1]
    if tuple(alpha.shape) != (B, T, Hh) or tuple(beta.shape) != (B, T, N, Hh):
        raise GdkvmError(f"bad gate shapes alpha{tuple(alpha.shape)} beta{tuple(beta.shape)}")
    if k.dtype != q.dtype or v.dtype != q.dtype:
        raise GdkvmError("q, k, v must share one dtype")
    if alpha.dtype != torch.float32 or beta.dtype != torch.float32:
        raise GdkvmError("alpha / beta must be float32")
    if state is not None and (tuple(state.shape) != (B, Hh, Dk, Dv) or state.dtype != torch.float32):
        raise GdkvmError("state must be float32 [B,Hh,Dk,Dv]")
    dev = _dev(q, k, v, alpha, beta, state, workspace, out, state_out, state_hist)
    io = _io_dtype(q)
    need = scan_workspace_bytes(B, T, Hh, N, Dk, Dv)
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    r = None if not readout else (out if out is not None else torch.empty((B, T, N, Hh, Dv), dtype=q.dtype, device=dev))
    s = state_out if state_out is not None else torch.empty((B, Hh, Dk, Dv), dtype=torch.float32, device=dev)
    if norms is not None:
        if state_hist is not None or norms.dtype != torch.float32 or norms.numel() != B * T * N * Hh * 2 or not norms.is_contiguous() \
                or norms.device != dev:
            raise GdkvmError("scan_fwd: norms must be contiguous float32 [B*T*N, Hh, 2] on the inputs' device (inference: no state_hist)")
        with torch.cuda.device(dev):
            rc = lib.gdkvm_scan_fwd_normed(_ptr(q), _ptr(k), _ptr(v), _ptr(alpha), _ptr(beta), norms.data_ptr(), _ptr(state), _ptr(r), _ptr(s),
                                           workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                           B, T, Hh, N, Dk, Dv, io, rule, flags, _stream(dev))
        _check(rc, "gdkvm_scan_fwd_normed")
        if check:
            scan_status(workspace, B, T, Hh, N, Dk, Dv, flags | (FLAG_WIDE_RANGE if rule == RULE_DELTA_PARALLEL else 0))
        return r, s
    with torch.cuda.device(dev):
        rc = lib.gdkvm_scan_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(alpha), _ptr(beta), _ptr(state), _ptr(r), _ptr(s),
                                _ptr(state_hist), workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                B, T, Hh, N, Dk, Dv, io, rule, flags, _stream(dev))
    _check(rc, "gdkvm_scan_fwd")
    if check:
        scan_status(workspace, B, T, Hh, N, Dk, Dv, flags | (FLAG_WIDE_RANGE if rule == RULE_DELTA_PARALLEL else 0))
    return r, s


def scan_fwd_normalizer(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, alpha: torch.Tensor, beta: torch.Tensor,
                        state: Optional[torch.Tensor] = None, z: Optional[torch.Tensor] = None, rule: int = RULE_DELTA_SEQUENTIAL,
                        flags: int = 0, eps: float = 1e-6):
    """gdkvm_scan_fwd_normalizer (SURVEY.md A.1's `normalizer` flag): the scan with z [B,Hh,Dk] carried beside S and the read-out divided
    by |q . z| + eps.  Returns (R [B,T,N,Hh,Dv], S_T [B,Hh,Dk,Dv] fp32, z_T [B,Hh,Dk] fp32).  Inference only; Dk = 64."""
    lib = load()
    if q.dim() != 5 or k.shape != q.shape or v.dim() != 5 or v.shape[:4] != q.shape[:4]:
        raise GdkvmError(f"bad shapes q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)}")
    B, T, N, Hh, Dk = q.shape
    Dv = v.shape[-1]
    if tuple(alpha.shape) != (B, T, Hh) or tuple(beta.shape) != (B, T, N, Hh) or alpha.dtype != torch.float32 or beta.dtype != torch.float32:
        raise GdkvmError(f"bad gates alpha{tuple(alpha.shape)} beta{tuple(beta.shape)} (float32)")
    if k.dtype != q.dtype or v.dtype != q.dtype:
        raise GdkvmError("q, k, v must share one dtype")
    if state is not None and (tuple(state.shape) != (B, Hh, Dk, Dv) or state.dtype != torch.float32):
        raise GdkvmError("state must be float32 [B,Hh,Dk,Dv]")
    if z is not None and (tuple(z.shape) != (B, Hh, Dk) or z.dtype != torch.float32):
        raise GdkvmError("z must be float32 [B,Hh,Dk]")
    dev = _dev(q, k, v, alpha, beta, state, z)
    io = _io_dtype(q)
    ws = torch.empty(int(lib.gdkvm_scan_normalizer_workspace_bytes(B, T, Hh, N, Dk, Dv, io)), dtype=torch.uint8, device=dev)
    r = torch.empty((B, T, N, Hh, Dv), dtype=q.dtype, device=dev)
    s = torch.empty((B, Hh, Dk, Dv), dtype=torch.float32, device=dev)
    zo = torch.empty((B, Hh, Dk), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.gdkvm_scan_fwd_normalizer(_ptr(q), _ptr(k), _ptr(v), _ptr(alpha), _ptr(beta), _ptr(state), _ptr(z), _ptr(r), _ptr(s), _ptr(zo),
                                           ws.data_ptr(), ws.numel(), B, T, Hh, N, Dk, Dv, io, rule, flags, float(eps), _stream(dev))
    _check(rc, "gdkvm_scan_fwd_normalizer")
    return r, s, zo


def lkva_read(q: torch.Tensor, state: torch.Tensor, flags: int = 0, norms: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
    """gdkvm_lkva_read: R [B,N,Hh,Dv] = Qn S for ONE frame per clip (the read half of a per-frame step).  q [B,N,Hh,Dk] f32|bf16,
    state [B,Hh,Dk,Dv] fp32, norms [B*N,Hh,2] fp32 from ops.proj_gates or None."""
    lib = load()
    if q.dim() != 4 or state.dim() != 4 or state.shape[0] != q.shape[0] or state.shape[1] != q.shape[2] or state.shape[2] != q.shape[3]:
        raise GdkvmError(f"lkva_read: bad shapes q{tuple(q.shape)} state{tuple(state.shape)}")
    if state.dtype != torch.float32:
        raise GdkvmError("lkva_read: state must be float32")
    B, N, Hh, Dk = q.shape
    Dv = state.shape[-1]
    dev = _dev(q, state, norms, out)
    if norms is not None and (norms.dtype != torch.float32 or norms.numel() != B * N * Hh * 2):
        raise GdkvmError("lkva_read: norms must be float32 [B*N, Hh, 2]")
    r = _out_like(out, (B, N, Hh, Dv), q.dtype, dev, "lkva_read") if out is not None else torch.empty((B, N, Hh, Dv), dtype=q.dtype, device=dev)
    with torch.cuda.device(dev):
        rc = lib.gdkvm_lkva_read(_ptr(q), _ptr(norms), _ptr(state), _ptr(r), B, N, Hh, Dk, Dv, _io_dtype(q), flags, _stream(dev))
    _check(rc, "gdkvm_lkva_read")
    return r


def mask_embed_add_(v: torch.Tensor, mask: torch.Tensor, w_embed: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """gdkvm_mask_embed_add, in place: v [BT, h*w, C] += w_embed[c] * adaptive_avg_pool(mask != 0) per token.  mask uint8 [BT,H,W],
    w_embed fp32 [C]."""
    lib = load()
    if v.dim() != 3 or mask.dim() != 3 or mask.dtype != torch.uint8 or mask.shape[0] != v.shape[0] or v.shape[1] != h * w:
        raise GdkvmError(f"mask_embed_add_: bad shapes v{tuple(v.shape)} mask{tuple(mask.shape)} {mask.dtype} tokens {h}x{w}")
    if w_embed.dtype != torch.float32 or w_embed.numel() != v.shape[2]:
        raise GdkvmError("mask_embed_add_: w_embed must be float32 [C]")
    dev = _dev(v, mask, w_embed)
    with torch.cuda.device(dev):
        rc = lib.gdkvm_mask_embed_add(_ptr(mask), _ptr(w_embed), _ptr(v), v.shape[0], mask.shape[1], mask.shape[2], h, w, v.shape[2],
                                      _io_dtype(v), _stream(dev))
    _check(rc, "gdkvm_mask_embed_add")
    return v


def scan_status(workspace: torch.Tensor, B: int, T: int, Hh: int, N: int, Dk: int, Dv: int, flags: int = 0) -> None:
    """Raises GdkvmError (GDKVM_ERR_RANGE) if the last scan_fwd / scan_apply on ``workspace`` left the range of its fp16-pair operands
    (its results are NaNs then); returns None otherwise.  Waits for the current stream (gdkvm_scan_status)."""
    dev = _dev(workspace)
    with torch.cuda.device(dev):
        rc = load().gdkvm_scan_status(workspace.data_ptr(), workspace.numel() * workspace.element_size(), B, T, Hh, N, Dk, Dv, flags, _stream(dev))
    _check(rc, "gdkvm_scan_status")


def scan_bwd(q, k, v, alpha, beta, state_hist, workspace, d_r, d_state_out=None, rule=RULE_DELTA_SEQUENTIAL, flags=0,
             need_d_state_in=True):
    """Backward of scan_fwd (gdkvm_scan_bwd).  ``state_hist`` and ``workspace`` are the ones the forward call filled.
    Returns (d_q, d_k, d_v, d_alpha, d_beta, d_state_in | None)."""
    lib = load()
    B, T, N, Hh, Dk = q.shape
    Dv = v.shape[-1]
    dev = _dev(q, k, v, alpha, beta, state_hist, workspace, d_r, d_state_out)
    if d_r.dtype != q.dtype or tuple(d_r.shape) != (B, T, N, Hh, Dv):
        raise GdkvmError("d_r must match the read-out's shape and dtype")
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    da = torch.empty((B, T, Hh), dtype=torch.float32, device=dev)
    db = torch.empty((B, T, N, Hh), dtype=torch.float32, device=dev)
    ds = torch.empty((B, Hh, Dk, Dv), dtype=torch.float32, device=dev) if need_d_state_in else None
    bws = torch.empty(int(lib.gdkvm_scan_bwd_workspace_bytes(B, T, Hh, N, Dk, Dv)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.gdkvm_scan_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(alpha), _ptr(beta), _ptr(state_hist), workspace.data_ptr(),
                                workspace.numel(), _ptr(d_r), _ptr(d_state_out), _ptr(dq), _ptr(dk), _ptr(dv), _ptr(da),
                                _ptr(db), _ptr(ds), bws.data_ptr(), bws.numel(), B, T, Hh, N, Dk, Dv, _io_dtype(q), rule,
                                flags, _stream(dev))
    _check(rc, "gdkvm_scan_bwd")
    return dq, dk, dv, da, db, ds


class _ScanFunction(torch.autograd.Function):
    """Differentiable gdkvm_scan_fwd: forward saves the state history and the WY workspace, backward is gdkvm_scan_bwd."""

    @staticmethod
    def forward(ctx, q, k, v, alpha, beta, state, rule, flags):
        B, T, N, Hh, Dk = q.shape
        Dv = v.shape[-1]
        ws = torch.empty(scan_workspace_bytes(B, T, Hh, N, Dk, Dv), dtype=torch.uint8, device=q.device)
        hist = torch.empty((B, T, Hh, Dk, Dv), dtype=torch.float32, device=q.device)
        r, s = scan_fwd(q, k, v, alpha, beta, state, rule=rule, flags=flags, workspace=ws, state_hist=hist)
        ctx.save_for_backward(q, k, v, alpha, beta, hist, ws)
        ctx.rule, ctx.flags, ctx.has_state = rule, flags, state is not None
        return r, s

    @staticmethod
    def backward(ctx, d_r, d_s):
        q, k, v, alpha, beta, hist, ws = ctx.saved_tensors
        d_r = d_r.contiguous()
        d_s = None if d_s is None else d_s.contiguous().float()
        dq, dk, dv, da, db, ds = scan_bwd(q, k, v, alpha, beta, hist, ws, d_r, d_s, ctx.rule, ctx.flags,
                                          need_d_state_in=ctx.has_state)
        return dq, dk, dv, da, db, ds, None, None


def scan_state_bwd(k, v, alpha, beta, state_hist, workspace, d_hist=None, d_state_out=None, rule=RULE_DELTA_SEQUENTIAL, flags=0,
                   need_d_state_in=True):
    """Backward of the state recurrence alone (gdkvm_scan_state_bwd): ``d_hist`` [B,T,Hh,Dk,Dv] is the gradient with respect to
    the state before every frame.  Returns (d_k, d_v, d_alpha, d_beta, d_state_in | None)."""
    lib = load()
    B, T, N, Hh, Dk = k.shape
    Dv = v.shape[-1]
    dev = _dev(k, v, alpha, beta, state_hist, workspace, d_hist, d_state_out)
    dk, dv = torch.empty_like(k), torch.empty_like(v)
    da = torch.empty((B, T, Hh), dtype=torch.float32, device=dev)
    db = torch.empty((B, T, N, Hh), dtype=torch.float32, device=dev)
    ds = torch.empty((B, Hh, Dk, Dv), dtype=torch.float32, device=dev) if need_d_state_in else None
    bws = torch.empty(int(lib.gdkvm_scan_bwd_workspace_bytes(B, T, Hh, N, Dk, Dv)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.gdkvm_scan_state_bwd(_ptr(k), _ptr(v), _ptr(alpha), _ptr(beta), _ptr(state_hist), workspace.data_ptr(),
                                      workspace.numel(), _ptr(d_hist), _ptr(d_state_out), _ptr(dk), _ptr(dv), _ptr(da), _ptr(db),
                                      _ptr(ds), bws.data_ptr(), bws.numel(), B, T, Hh, N, Dk, Dv, _io_dtype(k), rule, flags,
                                      _stream(dev))
    _check(rc, "gdkvm_scan_state_bwd")
    return dk, dv, da, db, ds


class _ReadoutFunction(torch.autograd.Function):
    """The LKVA read-out of frames of more than 64 tokens from the saved state history (gdkvm_readout_fwd), differentiable:
    the backward (gdkvm_readout_bwd) returns d_q and the gradient with respect to the states the frames read, in the history's
    own layout (what gdkvm_scan_state_bwd takes as d_hist).  gdkvm_scan_train_fwd / _bwd compose the same calls in C; this
    autograd node exposes the read-out on its own."""

    @staticmethod
    def forward(ctx, q, hist, chunks, flags):
        B, T, N, Hh, Dk = q.shape
        Dv = hist.shape[-1]
        dev = _dev(q, hist)
        if hist.dtype != torch.float32 or tuple(hist.shape) != (B, T * chunks, Hh, Dk, Dv):
            raise GdkvmError(f"state history must be float32 [B, T*chunks, Hh, Dk, Dv], got {tuple(hist.shape)}")
        r = torch.empty((B, T, N, Hh, Dv), dtype=q.dtype, device=dev)
        with torch.cuda.device(dev):
            rc = load().gdkvm_readout_fwd(_ptr(q), _ptr(hist), _ptr(r), B, T, Hh, N, Dk, Dv, chunks, _io_dtype(q), flags, _stream(dev))
        _check(rc, "gdkvm_readout_fwd")
        ctx.save_for_backward(q, hist)
        ctx.chunks, ctx.flags = chunks, flags
        return r

    @staticmethod
    def backward(ctx, d_r):
        q, hist = ctx.saved_tensors
        B, T, N, Hh, Dk = q.shape
        Dv = hist.shape[-1]
        d_r = d_r.contiguous()
        if d_r.dtype != q.dtype:
            d_r = d_r.to(q.dtype)
        dq = torch.empty_like(q)
        d_hist = torch.zeros_like(hist)                     # only the states before the frames receive a gradient here
        with torch.cuda.device(q.device):
            rc = load().gdkvm_readout_bwd(_ptr(q), _ptr(hist), _ptr(d_r), _ptr(dq), _ptr(d_hist), B, T, Hh, N, Dk, Dv, ctx.chunks,
                                          _io_dtype(q), ctx.flags, _stream(q.device))
        _check(rc, "gdkvm_readout_bwd")
        return dq, d_hist, None, None


class _ScanTrainFunction(torch.autograd.Function):
    """Differentiable scan for frames of any token count: gdkvm_scan_train_fwd / gdkvm_scan_train_bwd, with ONE workspace carrying
    the state history, the WY factors and (N > 64) the frame re-cut into 64-token pseudo-frames from the forward to the backward
    call.  No framework op between the HIP calls."""

    @staticmethod
    def forward(ctx, q, k, v, alpha, beta, state, rule, flags):
        lib = load()
        B, T, N, Hh, Dk = q.shape
        Dv = v.shape[-1]
        q, k, v, alpha, beta = (t.contiguous() for t in (q, k, v, alpha, beta))
        dev = _dev(q, k, v, alpha, beta, state)
        io = _io_dtype(q)
        ws = torch.empty(int(lib.gdkvm_scan_train_workspace_bytes(B, T, Hh, N, Dk, Dv, io)), dtype=torch.uint8, device=dev)
        r = torch.empty((B, T, N, Hh, Dv), dtype=q.dtype, device=dev)
        s = torch.empty((B, Hh, Dk, Dv), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.gdkvm_scan_train_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(alpha), _ptr(beta), _ptr(state), _ptr(r), _ptr(s),
                                          ws.data_ptr(), ws.numel(), B, T, Hh, N, Dk, Dv, io, rule, flags, _stream(dev))
        _check(rc, "gdkvm_scan_train_fwd")
        ctx.save_for_backward(q, k, v, alpha, beta, ws)
        ctx.rule, ctx.flags, ctx.has_state = rule, flags, state is not None
        return r, s

    @staticmethod
    def backward(ctx, d_r, d_s):
        lib = load()
        q, k, v, alpha, beta, ws = ctx.saved_tensors
        B, T, N, Hh, Dk = q.shape
        Dv = v.shape[-1]
        dev = q.device
        d_r = d_r.contiguous()
        if d_r.dtype != q.dtype:
            d_r = d_r.to(q.dtype)
        d_s = None if d_s is None else d_s.contiguous().float()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        da = torch.empty((B, T, Hh), dtype=torch.float32, device=dev)
        db = torch.empty((B, T, N, Hh), dtype=torch.float32, device=dev)
        ds = torch.empty((B, Hh, Dk, Dv), dtype=torch.float32, device=dev) if ctx.has_state else None
        with torch.cuda.device(dev):
            rc = lib.gdkvm_scan_train_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(alpha), _ptr(beta), _ptr(d_r), _ptr(d_s), _ptr(dq), _ptr(dk),
                                          _ptr(dv), _ptr(da), _ptr(db), _ptr(ds), ws.data_ptr(), ws.numel(), B, T, Hh, N, Dk, Dv,
                                          _io_dtype(q), ctx.rule, ctx.flags, _stream(dev))
        _check(rc, "gdkvm_scan_train_bwd")
        return dq, dk, dv, da, db, ds, None, None


def scan(q, k, v, alpha, beta, state=None, rule: int = RULE_DELTA_SEQUENTIAL, flags: int = 0):
    """scan_fwd with autograd support (training).  Inference callers should use scan_fwd directly (no history)."""
    if q.shape[-1] < KERNEL_DK:                            # narrower keys (differentiable: padding and slicing are autograd ops)
        qp, kp, sp = _pad_keys(q, k, state)
        r, s = scan(qp, kp, v, alpha, beta, sp, rule, flags)
        return r, s[:, :, :q.shape[-1]]
    if q.shape[2] > 64:
        return _ScanTrainFunction.apply(q, k, v, alpha, beta, state, rule, flags)
    return _ScanFunction.apply(q, k, v, alpha, beta, state, rule, flags)


def scan_prep(q, k, v, beta, workspace, rule=RULE_DELTA_SEQUENTIAL, flags=0):
    """Stage 1 of scan_fwd alone (gdkvm_scan_prep): folds every frame into its affine map S' = a P S + G in ``workspace``
    (with FLAG_TRAIN also the WY factors the backward consumes)."""
    B, T, N, Hh, Dk = k.shape
    Dv = v.shape[-1]
    dev = _dev(q, k, v, beta, workspace)
    with torch.cuda.device(dev):
        rc = load().gdkvm_scan_prep(_ptr(q), _ptr(k), _ptr(v), _ptr(beta), workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                    B, T, Hh, N, Dk, Dv, _io_dtype(k), rule, flags, _stream(dev))
    _check(rc, "gdkvm_scan_prep")


def new_workspace(B, T, Hh, N, Dk, Dv, device) -> torch.Tensor:
    return torch.empty(scan_workspace_bytes(B, T, Hh, N, Dk, Dv), dtype=torch.uint8, device=device)


def scan_transition(q, alpha, workspace, Dv, flags=0):
    """State-transition matrix Phi [B,Hh,Dk,Dk] of the frames prepared in ``workspace`` (gdkvm_scan_transition):
    S_out = Phi @ S_in + S_loc for these frames."""
    B, T, N, Hh, Dk = q.shape
    dev = _dev(q, alpha, workspace)
    phi = torch.empty((B, Hh, Dk, Dk), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = load().gdkvm_scan_transition(_ptr(q), _ptr(alpha), _ptr(phi), workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                          B, T, Hh, N, Dk, Dv, _io_dtype(q), flags, _stream(dev))
    _check(rc, "gdkvm_scan_transition")
    return phi


def scan_stitch(phi: torch.Tensor, s_loc: torch.Tensor, state: Optional[torch.Tensor] = None):
    """starts [B,S,Hh,Dk,Dv] and the end state of S consecutive segments from their transition matrices phi [B,S,Hh,Dk,Dk] and
    zero-start end states s_loc [B,S,Hh,Dk,Dv] (gdkvm_scan_stitch): starts[:,0] = state, starts[:,c+1] = phi[:,c] starts[:,c] + s_loc[:,c]."""
    B, S, Hh, Dk, Dv = s_loc.shape
    if tuple(phi.shape) != (B, S, Hh, Dk, Dk) or phi.dtype != torch.float32 or s_loc.dtype != torch.float32:
        raise GdkvmError("scan_stitch: phi [B,S,Hh,Dk,Dk] and s_loc [B,S,Hh,Dk,Dv] in float32")
    dev = _dev(phi, s_loc, state)
    phi, s_loc = phi.contiguous(), s_loc.contiguous()
    starts = torch.empty_like(s_loc)
    end = torch.empty((B, Hh, Dk, Dv), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = load().gdkvm_scan_stitch(phi.data_ptr(), s_loc.data_ptr(), _ptr(state), starts.data_ptr(), end.data_ptr(), B, S, Hh, Dk, Dv, _stream(dev))
    _check(rc, "gdkvm_scan_stitch")
    return starts, end


def scan_fwd_segmented(q, k, v, alpha, beta, state=None, segments: int = 0, rule: int = RULE_DELTA_SEQUENTIAL, flags: int = 0,
                       workspace: Optional[torch.Tensor] = None):
    """Long-clip scan with the time axis cut into ``segments`` equal pieces that run concurrently (gdkvm_scan_fwd_segmented; SURVEY
    §8f n3 inside one GPU): per segment the transition matrix Phi_c and the zero-start end state S_loc_c, a tiny sequential
    stitch S_start_{c+1} = Phi_c S_start_c + S_loc_c, then every segment is scanned from its true start state.  2.25x the
    recurrence work on `segments`x the workgroups.  ``segments`` = 0: chosen by shape (gdkvm_scan_segments; 1 = the serial scan).
    Equal to scan_fwd up to fp32 re-association (not bit-identical)."""
    lib = load()
    B, T, N, Hh, Dk = q.shape
    Dv = v.shape[-1]
    if segments < 0 or (segments and T % segments):
        raise GdkvmError(f"segments={segments} must divide T={T}")
    if k.shape != q.shape or tuple(v.shape[:4]) != (B, T, N, Hh) or tuple(alpha.shape) != (B, T, Hh) or tuple(beta.shape) != (B, T, N, Hh):
        raise GdkvmError("bad scan shapes")
    if k.dtype != q.dtype or v.dtype != q.dtype or alpha.dtype != torch.float32 or beta.dtype != torch.float32:
        raise GdkvmError("q, k, v share one dtype; alpha / beta are float32")
    if state is not None and (tuple(state.shape) != (B, Hh, Dk, Dv) or state.dtype != torch.float32):
        raise GdkvmError("state must be float32 [B,Hh,Dk,Dv]")
    dev = _dev(q, k, v, alpha, beta, state, workspace)
    q, k, v, alpha, beta = (t.contiguous() for t in (q, k, v, alpha, beta))
    with torch.cuda.device(dev):                           # (the choice by shape asks the current device for its CU count)
        if workspace is None:
            workspace = torch.empty(int(lib.gdkvm_scan_segmented_workspace_bytes(B, T, Hh, N, Dk, Dv, segments)), dtype=torch.uint8, device=dev)
        r = torch.empty((B, T, N, Hh, Dv), dtype=q.dtype, device=dev)
        s = torch.empty((B, Hh, Dk, Dv), dtype=torch.float32, device=dev)
        rc = lib.gdkvm_scan_fwd_segmented(_ptr(q), _ptr(k), _ptr(v), _ptr(alpha), _ptr(beta), _ptr(state), _ptr(r), _ptr(s),
                                          workspace.data_ptr(), workspace.numel(), B, T, Hh, N, Dk, Dv, segments, _io_dtype(q), rule,
                                          flags, _stream(dev))
    _check(rc, "gdkvm_scan_fwd_segmented")
    return r, s


def scan_apply(q, alpha, workspace, Dv, state=None, flags=0, out=None, state_out=None, want_readout=True):
    """Stage 2 of scan_fwd alone (gdkvm_scan_apply): the serial read/write recurrence over a prepared workspace."""
    B, T, N, Hh, Dk = q.shape
    dev = _dev(q, alpha, workspace, state, out, state_out)
    if not want_readout:
        s = state_out if state_out is not None else torch.empty((B, Hh, Dk, Dv), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = load().gdkvm_scan_apply(_ptr(q), _ptr(alpha), _ptr(state), None, _ptr(s), None, workspace.data_ptr(),
                                         workspace.numel() * workspace.element_size(), B, T, Hh, N, Dk, Dv, _io_dtype(q), flags,
                                         _stream(dev))
        _check(rc, "gdkvm_scan_apply")
        return None, s
    r = out if out is not None else torch.empty((B, T, N, Hh, Dv), dtype=q.dtype, device=dev)
    s = state_out if state_out is not None else torch.empty((B, Hh, Dk, Dv), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = load().gdkvm_scan_apply(_ptr(q), _ptr(alpha), _ptr(state), _ptr(r), _ptr(s), None, workspace.data_ptr(),
                                     workspace.numel() * workspace.element_size(), B, T, Hh, N, Dk, Dv, _io_dtype(q), flags,
                                     _stream(dev))
    _check(rc, "gdkvm_scan_apply")
    return r, s


def kpff_fwd(local: torch.Tensor, glob: torch.Tensor, pixel: torch.Tensor, wa: torch.Tensor, ba: torch.Tensor,
             wl: torch.Tensor, wg: torch.Tensor, h: int, w: int, out: Optional[torch.Tensor] = None,
             workspace: Optional[torch.Tensor] = None, packed: bool = False, exact: bool = False) -> torch.Tensor:
    """Key-Pixel Feature Fusion (gdkvm_kpff_fwd).  local [BT,N,Ck] glob [BT,N,Cv] pixel [BT,N,Cp], N=h*w;
    wa [2Cp,Cp+Ck+Cv] ba [2Cp] wl [Cp,Ck] wg [Cp,Cv] float32.  Returns F [BT,N,Cp] in the io dtype.
    exact (float32 features only): no workspace is handed over, which selects the exact fp32-MFMA arm instead of the arm on
    bf16 splits (include/gdkvm.h)."""
    lib = load()
    BT, N, Ck = local.shape
    Cv, Cp = glob.shape[-1], pixel.shape[-1]
    if N != h * w or glob.shape[:2] != (BT, N) or pixel.shape[:2] != (BT, N):
        raise GdkvmError("bad KPFF feature shapes")
    if tuple(wa.shape) != (2 * Cp, Cp + Ck + Cv) or tuple(ba.shape) != (2 * Cp,) or \
            tuple(wl.shape) != (Cp, Ck) or tuple(wg.shape) != (Cp, Cv):
        raise GdkvmError("bad KPFF weight shapes")
    for t in (wa, ba, wl, wg):
        if t.dtype != torch.float32:
            raise GdkvmError("KPFF weights must be float32")
    if glob.dtype != local.dtype or pixel.dtype != local.dtype:
        raise GdkvmError("KPFF features must share one dtype")
    dev = _dev(local, glob, pixel, wa, ba, wl, wg, out, workspace)
    io = _io_dtype(local)
    f = out if out is not None else torch.empty((BT, N, Cp), dtype=local.dtype, device=dev)
    if exact:
        if local.dtype != torch.float32 or workspace is not None or packed:
            raise GdkvmError("kpff_fwd(exact=True) is the float32 arm without a workspace")
    elif workspace is None:
        workspace = torch.empty(int(lib.gdkvm_kpff_workspace_bytes(Ck, Cv, Cp, io)), dtype=torch.uint8, device=dev)
    fn = lib.gdkvm_kpff_fwd_packed if packed else lib.gdkvm_kpff_fwd     # packed: `workspace` already holds these weights
    with torch.cuda.device(dev):
        rc = fn(_ptr(local), _ptr(glob), _ptr(pixel), _ptr(wa), _ptr(ba), _ptr(wl), _ptr(wg), _ptr(f),
                workspace.data_ptr() if workspace is not None else None, workspace.numel() if workspace is not None else 0,
                BT, Ck, Cv, Cp, h, w, io, _stream(dev))
    _check(rc, "gdkvm_kpff_fwd")
    return f


class _KpffFunction(torch.autograd.Function):
    """Differentiable KPFF: HIP forward (saving gates / mixes / pooled feature), HIP elementwise + pooling backward, and the
    six plain products of the backward on the hand-written MFMA kernels of csrc/gemm.hip."""

    @staticmethod
    def forward(ctx, local, glob, pixel, wa, ba, wl, wg, h, w):
        lib = load()
        BT, N, Ck = local.shape
        Cv, Cp = glob.shape[-1], pixel.shape[-1]
        dev = _dev(local, glob, pixel, wa, ba, wl, wg)
        io = _io_dtype(local)
        f = torch.empty((BT, N, Cp), dtype=local.dtype, device=dev)
        gates = torch.empty((BT * N, 2 * Cp), dtype=local.dtype, device=dev)
        lp, gp = (torch.empty((BT * N, Cp), dtype=local.dtype, device=dev) for _ in range(2))
        gms = torch.empty((BT * N, Cv), dtype=local.dtype, device=dev)
        ws = torch.empty(int(lib.gdkvm_kpff_workspace_bytes(Ck, Cv, Cp, io)), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = lib.gdkvm_kpff_fwd_train(_ptr(local), _ptr(glob), _ptr(pixel), _ptr(wa), _ptr(ba), _ptr(wl), _ptr(wg), _ptr(f),
                                          _ptr(gates), _ptr(lp), _ptr(gp), _ptr(gms), ws.data_ptr(), ws.numel(),
                                          BT, Ck, Cv, Cp, h, w, io, _stream(dev))
        _check(rc, "gdkvm_kpff_fwd_train")
        ctx.save_for_backward(local, pixel, wa, wl, wg, gates, lp, gp, gms)
        ctx.hw = (h, w)
        return f

    @staticmethod
    def backward(ctx, d_f):
        lib = load()
        local, pixel, wa, wl, wg, gates, lp, gp, gms = ctx.saved_tensors
        h, w = ctx.hw
        BT, N, Ck = local.shape
        Cp, Cv = pixel.shape[-1], gms.shape[-1]
        M, dt, dev = BT * N, local.dtype, local.device
        io = _io_dtype(local)
        d_f = d_f.contiguous()
        dz = torch.empty((M, 2 * Cp), dtype=dt, device=dev)
        dlp, dgp = (torch.empty((M, Cp), dtype=dt, device=dev) for _ in range(2))
        with torch.cuda.device(dev):
            _check(lib.gdkvm_kpff_bwd_pre(_ptr(d_f), _ptr(gates), _ptr(lp), _ptr(gp), _ptr(dz), _ptr(dlp), _ptr(dgp),
                                          BT, N, Cp, io, _stream(dev)), "gdkvm_kpff_bwd_pre")
        L2, P2 = local.reshape(M, Ck), pixel.reshape(M, Cp)
        # the six products of the backward on the hand-written kernels (csrc/gemm.hip): dX-type ones as gdkvm_gemm_nt against
        # the weight with its input index leading (a one-off transpose of a small matrix), dW-type ones as gdkvm_gemm_tn
        def t_of(wt):                                       # cast and transpose in ONE copy kernel
            return torch.empty((wt.shape[1], wt.shape[0]), dtype=dt, device=dev).copy_(wt.t())
        dx = gemm_nt(dz, t_of(wa))                          # [M, Cin]
        dl_add = gemm_nt(dlp, t_of(wl))                     # [M, Ck]
        dg_add = gemm_nt(dgp, t_of(wg))                     # [M, Cv]
        # K = B*T*N tokens: split over workgroups, fp32 partials; the bias gradient (column sums of dz) rides on the first product's launch
        d_wa_p, d_ba = wgrad(dz, P2, colsum=True)
        d_wa = torch.cat([d_wa_p, wgrad(dz, L2), wgrad(dz, gms)], 1)
        d_wl = wgrad(dlp, L2)
        d_wg = wgrad(dgp, gms)
        d_p, d_l, d_g = torch.empty_like(pixel), torch.empty_like(local), torch.empty((BT, N, Cv), dtype=dt, device=dev)
        with torch.cuda.device(dev):
            _check(lib.gdkvm_kpff_bwd_post(_ptr(d_f), _ptr(dx), _ptr(dl_add), _ptr(dg_add), _ptr(d_p), _ptr(d_l), _ptr(d_g),
                                           BT, Ck, Cv, Cp, h, w, io, _stream(dev)), "gdkvm_kpff_bwd_post")
        return d_l, d_g, d_p, d_wa, d_ba, d_wl, d_wg, None, None


def kpff(local, glob, pixel, wa, ba, wl, wg, h: int, w: int):
    """kpff_fwd with autograd support (training)."""
    return _KpffFunction.apply(local, glob, pixel, wa, ba, wl, wg, h, w)


def argmax_dice(logits: torch.Tensor, target: Optional[torch.Tensor] = None):
    """mask = argmax over classes (ties -> lowest index) and, with a target, integer Dice counts
    (gdkvm_argmax_dice).  logits [BT,ncls,H,W]; target [BT,H,W] uint8.  Returns (mask u8, counts i32|None)."""
    lib = load()
    BT, ncls, H, W = logits.shape
    dev = _dev(logits, target)
    if target is not None and (target.dtype != torch.uint8 or tuple(target.shape) != (BT, H, W)):
        raise GdkvmError("target must be uint8 [BT,H,W]")
    mask = torch.empty((BT, H, W), dtype=torch.uint8, device=dev)
    counts = torch.empty((BT, ncls, 3), dtype=torch.int32, device=dev) if target is not None else None
    with torch.cuda.device(dev):
        rc = lib.gdkvm_argmax_dice(_ptr(logits), _ptr(target), _ptr(mask), _ptr(counts), BT, ncls, H, W,
                                   _io_dtype(logits), _stream(dev))
    _check(rc, "gdkvm_argmax_dice")
    return mask, counts


def upsample_argmax_dice(logits: torch.Tensor, H: int, W: int, target: Optional[torch.Tensor] = None,
                         mask_out: Optional[torch.Tensor] = None, counts_out: Optional[torch.Tensor] = None):
    """Fused bilinear upsample (align_corners=False) + argmax + Dice counts (gdkvm_upsample_argmax_dice).
    logits [BT,ncls,hl,wl] low-resolution; returns (mask u8 [BT,H,W], counts i32 [BT,ncls,3] | None); mask_out / counts_out: write there."""
    lib = load()
    BT, ncls, hl, wl = logits.shape
    dev = _dev(logits, target)
    if target is not None and (target.dtype != torch.uint8 or tuple(target.shape) != (BT, H, W)):
        raise GdkvmError("target must be uint8 [BT,H,W]")
    mask = _out_like(mask_out, (BT, H, W), torch.uint8, dev, "mask_out")
    counts = _out_like(counts_out, (BT, ncls, 3), torch.int32, dev, "counts_out") if target is not None else None
    with torch.cuda.device(dev):
        rc = lib.gdkvm_upsample_argmax_dice(_ptr(logits), _ptr(target), _ptr(mask), _ptr(counts), BT, ncls, hl, wl, H, W,
                                            _io_dtype(logits), _stream(dev))
    _check(rc, "gdkvm_upsample_argmax_dice")
    return mask, counts


def _out_like(out, shape, dtype, dev, what):
    """A caller-owned output (e.g. a slice of a larger result, so that parts of a batch computed on different streams land in ONE tensor
    without a concatenation pass) or a fresh tensor."""
    if out is None:
        return torch.empty(shape, dtype=dtype, device=dev)
    if tuple(out.shape) != tuple(shape) or out.dtype != dtype or out.device != dev or not out.is_contiguous():
        raise GdkvmError(f"{what} must be a contiguous {dtype} tensor of shape {tuple(shape)} on the inputs' device")
    return out


def head_upsample_argmax_dice(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, H: int, W: int,
                              target: Optional[torch.Tensor] = None, mask_out: Optional[torch.Tensor] = None,
                              counts_out: Optional[torch.Tensor] = None):
    """The decoder's 1x1 head + bilinear upsample + argmax + Dice counts in one kernel (gdkvm_head_upsample_argmax_dice): x is the
    channels_last stride-4 feature [BT,C,hl,wl], weight fp32 [classes, C], bias fp32 [classes]; the class planes never reach memory.
    Bit-identical to head_logits followed by upsample_argmax_dice.  Returns (mask u8 [BT,H,W], counts i32 [BT,classes,3] | None)."""
    lib = load()
    if x.dim() != 4 or not x.is_cuda or not x.is_contiguous(memory_format=torch.channels_last):
        raise GdkvmError("head_upsample_argmax_dice needs a channels_last [BT,C,hl,wl] device tensor (no CPU path)")
    BT, C, hl, wl = x.shape
    ncls = weight.shape[0]
    if weight.dtype != torch.float32 or bias.dtype != torch.float32 or tuple(weight.shape) != (ncls, C) or bias.numel() != ncls \
            or not weight.is_contiguous():
        raise GdkvmError("head_upsample_argmax_dice: weight fp32 [classes, C] (contiguous), bias fp32 [classes]")
    dev = _dev(weight, bias, target)                       # (x is channels_last: checked above)
    if dev != x.device:
        raise GdkvmError("all tensors must live on one device")
    if target is not None and (target.dtype != torch.uint8 or tuple(target.shape) != (BT, H, W)):
        raise GdkvmError("target must be uint8 [BT,H,W]")
    mask = _out_like(mask_out, (BT, H, W), torch.uint8, dev, "mask_out")
    counts = _out_like(counts_out, (BT, ncls, 3), torch.int32, dev, "counts_out") if target is not None else None
    with torch.cuda.device(dev):
        rc = lib.gdkvm_head_upsample_argmax_dice(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), _ptr(target), _ptr(mask), _ptr(counts),
                                                 BT, C, ncls, hl, wl, H, W, _io_dtype(x), _stream(dev))
    _check(rc, "gdkvm_head_upsample_argmax_dice")
    return mask, counts


def bias_act_(x: torch.Tensor, bias: torch.Tensor, residual: Optional[torch.Tensor] = None, relu: bool = True) -> torch.Tensor:
    """In-place fused epilogue on an NHWC (channels_last) conv output: x <- act(x + bias[c] (+ residual))  (gdkvm_bias_act)."""
    lib = load()
    if x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last):
        raise GdkvmError("bias_act_ needs a channels_last [N,C,H,W] tensor")
    if residual is not None and (residual.shape != x.shape or residual.dtype != x.dtype or
                                 not residual.is_contiguous(memory_format=torch.channels_last)):
        raise GdkvmError("residual must match x (shape, dtype, channels_last)")
    if bias.dtype != torch.float32 or bias.numel() != x.shape[1]:
        raise GdkvmError("bias must be float32 [C]")
    if not x.is_cuda:
        raise GdkvmError("GDKVM ops need device tensors (no CPU path)")
    n, c, hh, ww = x.shape
    with torch.cuda.device(x.device):
        rc = lib.gdkvm_bias_act(x.data_ptr(), bias.data_ptr(), None if residual is None else residual.data_ptr(), x.data_ptr(),
                                n * hh * ww, c, int(relu), _io_dtype(x), _stream(x.device))
    _check(rc, "gdkvm_bias_act")
    return x


def head_logits(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """1x1 convolution + bias of a channels_last feature [N,C,H,W] to contiguous NCHW class planes [N,classes,H,W] in x's dtype
    (gdkvm_head_logits); weight fp32 [classes, C], bias fp32 [classes]."""
    lib = load()
    if x.dim() != 4 or not x.is_cuda or not x.is_contiguous(memory_format=torch.channels_last):
        raise GdkvmError("head_logits needs a channels_last [N,C,H,W] device tensor (no CPU path)")
    n, c, hh, ww = x.shape
    ncls = weight.shape[0]
    if weight.dtype != torch.float32 or bias.dtype != torch.float32 or tuple(weight.shape) != (ncls, c) or bias.numel() != ncls \
            or not weight.is_contiguous():
        raise GdkvmError("head_logits: weight fp32 [classes, C] (contiguous), bias fp32 [classes]")
    out = torch.empty((n, ncls, hh, ww), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.gdkvm_head_logits(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), out.data_ptr(), n, hh, ww, c, ncls,
                                   _io_dtype(x), _stream(x.device))
    _check(rc, "gdkvm_head_logits")
    return out


class _HeadFunction(torch.autograd.Function):
    """The decoder's 1x1 head in training: gdkvm_head_logits forward (NCHW class planes for the loss kernel), gdkvm_head_bwd backward
    (dx, dW, db in one pass over the feature, fixed summation order)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ncls, c = weight.shape[0], weight.shape[1]
        w2 = weight.detach().reshape(ncls, c).float().contiguous()
        b2 = bias.detach().float().contiguous()
        xc = x.contiguous(memory_format=torch.channels_last)
        ctx.save_for_backward(xc, w2)
        ctx.meta = (weight.dtype, tuple(weight.shape), bias.dtype)
        return head_logits(xc, w2, b2)

    @staticmethod
    def backward(ctx, dz):
        lib = load()
        xc, w2 = ctx.saved_tensors
        n, c, hh, ww = xc.shape
        ncls = w2.shape[0]
        dz = dz.to(xc.dtype).contiguous()
        dx = torch.empty_like(xc)
        dw = torch.empty((ncls, c), dtype=torch.float32, device=xc.device)
        db = torch.empty(ncls, dtype=torch.float32, device=xc.device)
        need = int(lib.gdkvm_head_bwd_workspace_bytes(c, ncls))
        ws = torch.empty(need, dtype=torch.uint8, device=xc.device)
        with torch.cuda.device(xc.device):
            rc = lib.gdkvm_head_bwd(xc.data_ptr(), dz.data_ptr(), w2.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), need,
                                    n, hh, ww, c, ncls, _io_dtype(xc), _stream(xc.device))
        _check(rc, "gdkvm_head_bwd")
        wdt, wshape, bdt = ctx.meta
        return dx, dw.reshape(wshape).to(wdt), db.to(bdt)


def head_served(x: torch.Tensor, conv) -> bool:
    v = 8 if x.dtype == torch.bfloat16 else 4
    g = x.shape[1] // v if x.dim() == 4 else 0
    return (x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float32) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is not None and x.shape[1] % v == 0 and 0 < g <= 64 and g & (g - 1) == 0
            and conv.out_channels <= min(g, 8))


def head(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """conv2d(x, weight [classes, C, 1, 1], bias) as contiguous NCHW class planes, differentiable (training): _HeadFunction."""
    return _HeadFunction.apply(x, weight, bias)


def pack_rows_weight(weight: torch.Tensor) -> torch.Tensor:
    """W [Nout, K] (Nout a multiple of 16, K of 32) -> bf16 in the MFMA fragment order gdkvm_proj_rows streams:
    element ((ot * K/32 + ks) * 64 + lane) * 8 + j = W[16 ot + (lane & 15)][32 ks + 8 (lane >> 4) + j]."""
    nout, k = weight.shape
    if nout % 16 or k % 32:
        raise GdkvmError("pack_rows_weight: [Nout, K] with Nout % 16 == 0 and K % 32 == 0")
    w = weight.detach().reshape(nout // 16, 16, k // 32, 4, 8)            # [ot, i, ks, g, j]
    return w.permute(0, 2, 3, 1, 4).contiguous().to(torch.bfloat16).reshape(-1)       # [ot, ks, g, i, j]: lane = 16 g + i


def proj_rows(x2d: torch.Tensor, wpack: torch.Tensor, bias: torch.Tensor, widths) -> Tuple[torch.Tensor, ...]:
    """The key / query / value projections of token rows x2d [rows, K] (bf16) in one pass (gdkvm_proj_rows): returns one
    contiguous [rows, w] tensor per entry of `widths` (up to three)."""
    lib = load()
    if x2d.dim() != 2 or not x2d.is_cuda or x2d.dtype != torch.bfloat16 or not x2d.is_contiguous():
        raise GdkvmError("proj_rows needs a contiguous bf16 [rows, K] device tensor (no CPU path)")
    rows, k = x2d.shape
    widths = list(widths) + [0] * (3 - len(widths))
    if len(widths) != 3 or bias.dtype != torch.float32 or bias.numel() != sum(widths) or wpack.dtype != torch.bfloat16 \
            or wpack.numel() != sum(widths) * k:
        raise GdkvmError("proj_rows: up to three widths, fp32 bias [sum(widths)], packed bf16 weight [sum(widths) * K]")
    outs = [torch.empty((rows, w), dtype=x2d.dtype, device=x2d.device) if w else None for w in widths]
    with torch.cuda.device(x2d.device):
        rc = lib.gdkvm_proj_rows(x2d.data_ptr(), wpack.data_ptr(), bias.data_ptr(), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]),
                                 rows, k, widths[0], widths[1], widths[2], BF16, _stream(x2d.device))
    _check(rc, "gdkvm_proj_rows")
    return tuple(o for o in outs if o is not None)


def gate_logits(p_tok: torch.Tensor, w_gate: torch.Tensor, b_gate: torch.Tensor, w_decay: torch.Tensor, b_decay: torch.Tensor):
    """(beta_logit [F,N,Hh], alpha_logit [F,Hh]) in fp32 from the pixel feature p_tok [F,N,Cp] in one pass (gdkvm_gate_logits):
    the gate projection per token and the decay projection of the token mean."""
    lib = load()
    if p_tok.dim() != 3 or not p_tok.is_cuda or not p_tok.is_contiguous():
        raise GdkvmError("gate_logits needs a contiguous [frames, N, Cp] device tensor (no CPU path)")
    fr, n, cp = p_tok.shape
    hh = w_gate.shape[0]
    ws = [t.detach().float().contiguous() for t in (w_gate.reshape(hh, -1), b_gate, w_decay.reshape(hh, -1), b_decay)]
    if ws[0].shape != (hh, cp) or ws[2].shape != (hh, cp) or ws[1].numel() != hh or ws[3].numel() != hh:
        raise GdkvmError("gate_logits: weights must be [Hh, Cp], biases [Hh]")
    beta = torch.empty((fr, n, hh), dtype=torch.float32, device=p_tok.device)
    alpha = torch.empty((fr, hh), dtype=torch.float32, device=p_tok.device)
    with torch.cuda.device(p_tok.device):
        rc = lib.gdkvm_gate_logits(p_tok.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ws[3].data_ptr(),
                                   beta.data_ptr(), alpha.data_ptr(), fr, n, cp, hh, _io_dtype(p_tok), _stream(p_tok.device))
    _check(rc, "gdkvm_gate_logits")
    return beta, alpha


def proj_gates(p_tok: torch.Tensor, wpack: torch.Tensor, bias: torch.Tensor, w_gate: torch.Tensor, b_gate: torch.Tensor,
               w_decay: torch.Tensor, b_decay: torch.Tensor, heads: int, key_dim: int, value_dim: int):
    """Everything the memory path derives from the stride-16 pixel feature p_tok [frames, N, Cp] (bf16) in ONE launch
    (gdkvm_proj_gates): returns (k [rows, Hh*Dk], q [rows, Hh*Dk], v [rows, Hh*Dv]) in bf16, (beta_logit [frames, N, Hh],
    alpha_logit [frames, Hh]) in fp32, and norms [rows, Hh, 2] fp32 -- the inverse L2 norms of the stored key / query rows, which
    scan_fwd(..., norms=norms) takes instead of computing them.  wpack / bias: pack_rows_weight of the stacked key, query, value
    weights and their fp32 biases; gate weights fp32 [Hh, Cp]."""
    lib = load()
    if p_tok.dim() != 3 or not p_tok.is_cuda or p_tok.dtype != torch.bfloat16 or not p_tok.is_contiguous():
        raise GdkvmError("proj_gates needs a contiguous bf16 [frames, N, Cp] device tensor (no CPU path)")
    fr, n, cp = p_tok.shape
    wk, wv = heads * key_dim, heads * value_dim
    gw = [t.detach().float().contiguous() for t in (w_gate.reshape(heads, -1), b_gate, w_decay.reshape(heads, -1), b_decay)]
    if gw[0].shape != (heads, cp) or gw[2].shape != (heads, cp) or gw[1].numel() != heads or gw[3].numel() != heads:
        raise GdkvmError("proj_gates: gate weights must be [Hh, Cp], biases [Hh]")
    if bias.dtype != torch.float32 or bias.numel() != 2 * wk + wv or wpack.dtype != torch.bfloat16 or wpack.numel() != (2 * wk + wv) * cp:
        raise GdkvmError("proj_gates: packed bf16 weight [(2 Hh Dk + Hh Dv) * Cp] and fp32 bias [2 Hh Dk + Hh Dv]")
    dev, rows = p_tok.device, fr * n
    k, q = (torch.empty((rows, wk), dtype=p_tok.dtype, device=dev) for _ in range(2))
    v = torch.empty((rows, wv), dtype=p_tok.dtype, device=dev)
    beta = torch.empty((fr, n, heads), dtype=torch.float32, device=dev)
    alpha = torch.empty((fr, heads), dtype=torch.float32, device=dev)
    norms = torch.empty((rows, heads, 2), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.gdkvm_proj_gates(p_tok.data_ptr(), wpack.data_ptr(), bias.data_ptr(), k.data_ptr(), q.data_ptr(), v.data_ptr(),
                                  gw[0].data_ptr(), gw[1].data_ptr(), gw[2].data_ptr(), gw[3].data_ptr(), beta.data_ptr(), alpha.data_ptr(),
                                  norms.data_ptr(), fr, n, cp, heads, key_dim, value_dim, BF16, _stream(dev))
    _check(rc, "gdkvm_proj_gates")
    return (k, q, v), (beta, alpha), norms


CONV_PACKED_WEIGHTS = 32


def conv3x3_pack_weights(weight: torch.Tensor) -> torch.Tensor:
    """The fragment-ordered copy of channels_last bf16 [K,C,3,3] weights that conv_bias_act(..., packed=...) reads
    (gdkvm_conv3x3_pack_weights): K a multiple of 16, C of 64.  Same bytes, another order; keep it next to the weights."""
    lib = load()
    if weight.dim() != 4 or weight.dtype != torch.bfloat16 or tuple(weight.shape[2:]) != (3, 3) or not weight.is_cuda or \
            not weight.is_contiguous(memory_format=torch.channels_last):
        raise GdkvmError("conv3x3_pack_weights: weight must be a channels_last bf16 [K,C,3,3] device tensor")
    k, c = weight.shape[:2]
    packed = torch.empty(k * 9 * c, dtype=torch.bfloat16, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = lib.gdkvm_conv3x3_pack_weights(weight.data_ptr(), packed.data_ptr(), k, c, BF16, _stream(weight.device))
    _check(rc, "gdkvm_conv3x3_pack_weights")
    return packed


CONV_KERNEL_IGEMM = 9         # gdkvm_conv_bias_act's general implicit-GEMM kernel (any R x S / stride / pad; packed weights only)


def conv_igemm_pack_weights(weight: torch.Tensor) -> torch.Tensor:
    """The fragment-ordered copy of channels_last bf16 [K,C,R,S] weights that conv_bias_act(..., tile=CONV_KERNEL_IGEMM, packed=...)
    reads (gdkvm_conv_igemm_pack_weights): K a multiple of 16 (128 for the kernel), C of 32."""
    lib = load()
    if weight.dim() != 4 or weight.dtype != torch.bfloat16 or not weight.is_cuda or not weight.is_contiguous(memory_format=torch.channels_last):
        raise GdkvmError("conv_igemm_pack_weights: weight must be a channels_last bf16 [K,C,R,S] device tensor")
    k, c, r, s = weight.shape
    packed = torch.empty(weight.numel(), dtype=torch.bfloat16, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = lib.gdkvm_conv_igemm_pack_weights(weight.data_ptr(), packed.data_ptr(), k, c, r, s, BF16, _stream(weight.device))
    _check(rc, "gdkvm_conv_igemm_pack_weights")
    return packed


def conv_down_bias_act(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, packed: torch.Tensor, down_weight: torch.Tensor,
                       down_packed: torch.Tensor, down_bias: Optional[torch.Tensor] = None, stride: int = 2, relu: bool = True):
    """A residual block's first convolution and its 1x1 downsample branch in one launch (gdkvm_conv_down_bias_act):
    (act(conv(x, weight, stride, pad = R // 2) + bias), conv(x, down_weight, stride) (+ down_bias)); channels_last bf16, the packs as
    conv_igemm_pack_weights made them."""
    lib = load()
    if x.dim() != 4 or not x.is_cuda or x.dtype != torch.bfloat16 or not x.is_contiguous(memory_format=torch.channels_last):
        raise GdkvmError("conv_down_bias_act needs a channels_last bf16 [N,C,H,W] device tensor (no CPU path)")
    n, c, hh, ww = x.shape
    k, _, r, s = weight.shape
    if tuple(down_weight.shape) != (k, c, 1, 1) or weight.shape[1] != c or packed.numel() != weight.numel() or down_packed.numel() != k * c \
            or packed.dtype != torch.bfloat16 or down_packed.dtype != torch.bfloat16:
        raise GdkvmError("conv_down_bias_act: weight [K,C,R,R], down_weight [K,C,1,1] and their conv_igemm_pack_weights copies")
    if bias.dtype != torch.float32 or bias.numel() != k or (down_bias is not None and (down_bias.dtype != torch.float32 or down_bias.numel() != k)):
        raise GdkvmError("biases must be float32 [K]")
    pad = r // 2
    ho, wo = (hh + 2 * pad - r) // stride + 1, (ww + 2 * pad - s) // stride + 1
    y = torch.empty((n, k, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    yd = torch.empty_like(y)
    with torch.cuda.device(x.device):
        rc = lib.gdkvm_conv_down_bias_act(x.data_ptr(), packed.data_ptr(), bias.data_ptr(), y.data_ptr(), int(relu), down_packed.data_ptr(),
                                          _ptr(down_bias), yd.data_ptr(), n, c, hh, ww, k, r, s, stride, pad, BF16, _stream(x.device))
    _check(rc, "gdkvm_conv_down_bias_act")
    return y, yd


def _conv3x3_packed(x: torch.Tensor, packed: torch.Tensor, k_out: int, bias: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv3x3 / stride 1 / pad 1 of channels_last bf16 x with weights given ONLY as a pack (no epilogue beyond the bias and an optional
    residual [N, k_out, H, W] of x's type added in the kernel's epilogue)."""
    n, c, hh, ww = x.shape
    y = torch.empty((n, k_out, hh, ww), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = load().gdkvm_conv_bias_act(x.data_ptr(), packed.data_ptr(), bias.data_ptr(), _ptr(residual), y.data_ptr(), n, c, hh, ww, k_out, 3, 3, 1, 1,
                                        0, CONV_PACKED_WEIGHTS, BF16, _stream(x.device))
    _check(rc, "gdkvm_conv_bias_act")
    return y


_ZERO_BIAS = {}


def _zero_bias(k: int, device) -> torch.Tensor:
    key = (k, str(device))
    if key not in _ZERO_BIAS:
        _ZERO_BIAS[key] = torch.zeros(k, dtype=torch.float32, device=device)
    return _ZERO_BIAS[key]


# The training step's weight packs live ON the parameter (``w._gdkvm_train_packs``), not in a table keyed by id(): they go when the parameter
# goes, and a new parameter that reuses the id / address of a dead one cannot inherit them.  They are valid only inside the PACK SCOPE they were
# made for (``train_packs``: one forward of one model) -- NOT "until the version counter changes": torch.optim.AdamW(fused=True) writes the
# parameters without bumping ``_version`` (round 5: keyed on the counter, the fourteen stride-1 3x3 layers kept convolving with their initial
# weights under the fused optimiser, and the loss fell ~10x slower than under the default one).
_PACK_SCOPE = [None]
_PACK_TOKENS = [0]


def conv3x3_train_packs(weights) -> int:
    """Both packs (forward, data gradient) of every listed fp32 [K,C,3,3] weight in ONE launch (gdkvm_conv3x3_pack_weights_train), into
    buffers kept on the parameter; EVERY call re-packs every weight (a fused optimiser step leaves no trace a cache could key on) and opens a
    new pack scope: ops.conv3x3 uses a weight's packs only while the scope of the call that made them is open (``end_train_packs`` closes it;
    ``train_packs`` is the context-manager form).  Returns the number of layers packed."""
    lib = load()
    todo = []
    _PACK_TOKENS[0] += 1
    token = _PACK_TOKENS[0]
    for w in weights:
        k, c = w.shape[:2]
        if not (w.is_cuda and w.dtype == torch.float32 and w.dim() == 4 and tuple(w.shape[2:]) == (3, 3) and k % 64 == 0 and c % 64 == 0):
            continue
        ent = w.__dict__.get("_gdkvm_train_packs")
        if ent is None or ent[1].numel() != k * 9 * c or ent[1].device != w.device:
            ent = (None, torch.empty(k * 9 * c, dtype=torch.bfloat16, device=w.device), torch.empty(k * 9 * c, dtype=torch.bfloat16, device=w.device))
        w.__dict__["_gdkvm_train_packs"] = ((token, w._version, w.data_ptr(), tuple(w.stride())), ent[1], ent[2])
        todo.append(w)
    for i0 in range(0, len(todo), 24):
        part = todo[i0:i0 + 24]
        n = len(part)
        P = ctypes.c_void_p * n
        wp = P(*[w.data_ptr() for w in part])
        fp = P(*[w.__dict__["_gdkvm_train_packs"][1].data_ptr() for w in part])
        dp = P(*[w.__dict__["_gdkvm_train_packs"][2].data_ptr() for w in part])
        ks = (ctypes.c_int * n)(*[w.shape[0] for w in part])
        cs = (ctypes.c_int * n)(*[w.shape[1] for w in part])
        st = (ctypes.c_longlong * (4 * n))(*[x for w in part for x in w.stride()])
        dev = part[0].device
        with torch.cuda.device(dev):
            rc = lib.gdkvm_conv3x3_pack_weights_train(n, ctypes.cast(wp, ctypes.c_void_p), ctypes.cast(fp, ctypes.c_void_p), ctypes.cast(dp, ctypes.c_void_p),
                                                      ctypes.cast(ks, ctypes.c_void_p), ctypes.cast(cs, ctypes.c_void_p), ctypes.cast(st, ctypes.c_void_p),
                                                      _stream(dev))
        _check(rc, "gdkvm_conv3x3_pack_weights_train")
    _PACK_SCOPE[0] = token
    return len(todo)


def end_train_packs() -> None:
    """Close the pack scope conv3x3_train_packs opened (the end of the forward it was opened for)."""
    _PACK_SCOPE[0] = None


class train_packs:
    """``with ops.train_packs(weights): ...`` -- conv3x3_train_packs for the block, the scope closed at its end."""

    def __init__(self, weights):
        self.weights = list(weights)

    def __enter__(self):
        self.packed = conv3x3_train_packs(self.weights)
        return self

    def __exit__(self, *exc):
        end_train_packs()
        return False


def drop_train_packs(weights) -> None:
    for w in weights:
        w.__dict__.pop("_gdkvm_train_packs", None)


def _train_packs_of(weight):
    ent = weight.__dict__.get("_gdkvm_train_packs")
    if ent is not None and _PACK_SCOPE[0] is not None and ent[0] == (_PACK_SCOPE[0], weight._version, weight.data_ptr(), tuple(weight.stride())):
        return ent[1], ent[2]
    return None


class _Conv3x3Function(torch.autograd.Function):
    """Training-mode 3x3 / stride 1 / pad 1 convolution (no bias) on the hand-written kernels: forward, the data gradient (the
    same kernel on the flipped, transposed weights: gdkvm_conv3x3_pack_weights_dgrad) and the weight gradient
    (gdkvm_conv3x3_wgrad).  bf16 activations (autocast), fp32 master weights."""

    @staticmethod
    def forward(ctx, x, weight, fork=False):
        lib = load()
        xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        k, c = weight.shape[:2]
        ctx.fork = bool(fork)
        if fork:
            # conv3x3_fork: a second output that IS the input (for the residual branch of the block).  Its gradient then arrives HERE, together
            # with the convolution's, and is added in the data-gradient kernel's epilogue instead of by a separate pass over both tensors
            y, dummy = _Conv3x3Function._fwd(ctx, lib, x, xb, weight, k, c)
            return y, xb.view(xb.shape)
        return _Conv3x3Function._fwd(ctx, lib, x, xb, weight, k, c)[0]

    @staticmethod
    def _fwd(ctx, lib, x, xb, weight, k, c):
        packs = _train_packs_of(weight)                    # (conv3x3_train_packs ran for this version of the weight: nothing to cast or pack here)
        ctx.kc, ctx.wdtype, ctx.xdtype = (k, c), weight.dtype, x.dtype
        ctx.w_cl = (weight.is_contiguous(memory_format=torch.channels_last) and not weight.is_contiguous()
                    and os.environ.get("GDKVM_WGRAD_KRSC", "1") != "0")          # ("0": A/B switch for tools)
        if packs is not None and os.environ.get("GDKVM_CONV_WGRAD") != "framework":
            # the data-gradient pack belongs to THIS version of the weight: a copy would cost what the pre-pack saves, so the backward
            # checks that the weight has not been written since (an optimiser step between forward and backward is not a thing)
            ctx.dgrad_pack, ctx.pack_key = packs[1], (weight._version, weight.data_ptr())
            ctx.save_for_backward(xb, weight)              # (the weight rides along only to be checked in the backward: autograd tracks it)
            return _conv3x3_packed(xb, packs[0], k, _zero_bias(k, xb.device)), None
        wb = weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        packed = torch.empty(k * 9 * c, dtype=torch.bfloat16, device=xb.device)
        with torch.cuda.device(xb.device):
            _check(lib.gdkvm_conv3x3_pack_weights(wb.data_ptr(), packed.data_ptr(), k, c, BF16, _stream(xb.device)), "gdkvm_conv3x3_pack_weights")
        ctx.dgrad_pack = None
        ctx.save_for_backward(xb, wb)
        return _conv3x3_packed(xb, packed, k, _zero_bias(k, xb.device)), None

    @staticmethod
    def backward(ctx, dy, d_alias=None):
        k, c = ctx.kc
        dyb = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        res = None if d_alias is None else d_alias.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dx = dw = None
        if ctx.dgrad_pack is not None:
            (xb, w), wb = ctx.saved_tensors, None           # (an in-place write of the weight since the forward makes saved_tensors itself raise)
            if (w._version, w.data_ptr()) != ctx.pack_key:
                raise RuntimeError("conv3x3: the weight was modified between forward and backward (its pre-packed data-gradient copy is stale)")
        else:
            xb, wb = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            packed = ctx.dgrad_pack
            if packed is None:
                packed = torch.empty(k * 9 * c, dtype=torch.bfloat16, device=dyb.device)
                with torch.cuda.device(dyb.device):
                    _check(load().gdkvm_conv3x3_pack_weights_dgrad(wb.data_ptr(), packed.data_ptr(), k, c, BF16, _stream(dyb.device)),
                           "gdkvm_conv3x3_pack_weights_dgrad")
            dx = _conv3x3_packed(dyb, packed, c, _zero_bias(c, dyb.device), res).to(ctx.xdtype)
        elif res is not None:
            dx = res.to(ctx.xdtype)
        if ctx.needs_input_grad[1]:
            if os.environ.get("GDKVM_CONV_WGRAD") == "framework":          # (A/B switch for tools: the framework's weight gradient)
                dw = torch.ops.aten.convolution_backward(dyb, xb, wb, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False))[1].to(ctx.wdtype)
            else:
                dw = conv3x3_wgrad(xb, dyb, channels_last=ctx.w_cl).to(ctx.wdtype)  # (fp32 sums over all pixels, deterministic; in the weight's own memory order)
        return dx, dw, None


_WGRAD_WS = {}


def conv3x3_wgrad(x: torch.Tensor, dy: torch.Tensor, channels_last: bool = False) -> torch.Tensor:
    """dW [K,C,3,3] fp32 of a 3x3 / stride 1 / pad 1 convolution from channels_last bf16 x [N,C,H,W] and dy [N,K,H,W]
    (gdkvm_conv3x3_wgrad): C, K multiples of 64, rows of at most 64 pixels; deterministic.  channels_last: the result in the memory
    order of a channels_last parameter (gdkvm_conv3x3_wgrad_krsc) -- the same numbers, no re-layout copy when it becomes that
    parameter's .grad."""
    lib = load()
    for t in (x, dy):
        if t.dim() != 4 or not t.is_cuda or t.dtype != torch.bfloat16 or not t.is_contiguous(memory_format=torch.channels_last):
            raise GdkvmError("conv3x3_wgrad needs channels_last bf16 [N,C,H,W] device tensors (no CPU path)")
    n, c, hh, ww = x.shape
    k = dy.shape[1]
    if tuple(dy.shape) != (n, k, hh, ww):
        raise GdkvmError("conv3x3_wgrad: x and dy must agree in batch and size")
    dw = torch.empty((k, c, 3, 3), dtype=torch.float32, device=x.device, memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    # one workspace per device and stream, grown to the largest layer (up to 75 MB of partial blocks): calls on a stream are
    # ordered, so the next layer's gradient may overwrite it
    need = max(16, int(lib.gdkvm_conv3x3_wgrad_workspace_bytes(n, c, hh, ww, k)))
    key = (x.device, _stream(x.device))
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _WGRAD_WS[key] = torch.empty(need, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        fn = lib.gdkvm_conv3x3_wgrad_krsc if channels_last else lib.gdkvm_conv3x3_wgrad
        rc = fn(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), n, c, hh, ww, k, BF16, _stream(x.device))
    _check(rc, "gdkvm_conv3x3_wgrad")
    return dw


def conv3x3_train_served(x: torch.Tensor, weight: torch.Tensor, stride, padding, dilation, groups) -> bool:
    """Does conv3x3 (forward + data gradient on the hand-written kernels) serve this layer and input?"""
    k, c = weight.shape[:2]
    return (x.is_cuda and x.dim() == 4 and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (1, 1) and tuple(padding) == (1, 1)
            and tuple(dilation) == (1, 1) and groups == 1 and c % 64 == 0 and k % 64 == 0 and x.shape[-1] <= 64
            and (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)))


def conv3x3(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """conv2d(x, weight, padding=1) for a 3x3 / stride-1 layer with channel counts in multiples of 64, differentiable, bf16."""
    return _Conv3x3Function.apply(x, weight)


def conv3x3_fork(x: torch.Tensor, weight: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(conv3x3(x, weight), x') where x' is x itself (bf16, channels_last) as a second output of the same autograd node: a residual block
    feeds x' to its skip connection, and the skip's gradient is then added to the convolution's data gradient inside that kernel's
    epilogue (its residual input) instead of by the framework's add over two full tensors."""
    return _Conv3x3Function.apply(x, weight, True)


_S2_WS = {}


def conv_wgrad_strided(x: torch.Tensor, dy: torch.Tensor, weight_like: torch.Tensor, stride: int, pad: int) -> torch.Tensor:
    """dW (fp32, in `weight_like`'s shape [K,C,R,S] AND memory format) of conv2d(x, w, stride, pad) from channels_last bf16 x [N,C,H,W] and
    dy [N,K,Ho,Wo] (gdkvm_conv_wgrad_strided): any window / stride / pad, C a multiple of 64, K of 8; fixed-order sums -- the same bits on
    every run."""
    lib = load()
    for t in (x, dy):
        if t.dim() != 4 or not t.is_cuda or t.dtype != torch.bfloat16 or not t.is_contiguous(memory_format=torch.channels_last):
            raise GdkvmError("conv_wgrad_strided needs channels_last bf16 [N,C,H,W] device tensors (no CPU path)")
    n, c, hh, ww = x.shape
    k, c2, r, s_ = weight_like.shape
    ho, wo = (hh + 2 * pad - r) // stride + 1, (ww + 2 * pad - s_) // stride + 1
    if c2 != c or tuple(dy.shape) != (n, k, ho, wo):
        raise GdkvmError(f"conv_wgrad_strided: x {tuple(x.shape)}, dy {tuple(dy.shape)} and weight {tuple(weight_like.shape)} do not fit stride {stride} pad {pad}")
    dw = torch.empty_strided(tuple(weight_like.shape), tuple(weight_like.stride()), dtype=torch.float32, device=x.device)
    need = max(16, int(lib.gdkvm_conv_wgrad_strided_workspace_bytes(n, c, hh, ww, k, r, s_, stride, pad)))
    key = (x.device, _stream(x.device))
    ws = _S2_WS.get(key)
    if ws is None or ws.numel() < need:                     # one workspace per device and stream, grown to the largest layer (~30 MB)
        ws = _S2_WS[key] = torch.empty(need, dtype=torch.uint8, device=x.device)
    sk, sc, sr, ss = dw.stride()
    with torch.cuda.device(x.device):
        rc = lib.gdkvm_conv_wgrad_strided(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), sk, sc, sr, ss, ws.data_ptr(), ws.numel(),
                                          n, c, hh, ww, k, r, s_, stride, pad, BF16, _stream(x.device))
    _check(rc, "gdkvm_conv_wgrad_strided")
    return dw


def conv_s2_packs(weight: torch.Tensor, down_weight: Optional[torch.Tensor]):
    """(forward pack, forward pack of the 1x1 branch | None, data-gradient pack) of a [K,C,3,3] fp32 weight and its block's [K,C,1,1] branch,
    ONE launch from the master weights in whatever memory format they have (gdkvm_conv_s2_pack_train)."""
    lib = load()
    k, c = weight.shape[:2]
    if weight.dtype != torch.float32 or tuple(weight.shape[2:]) != (3, 3) or not weight.is_cuda:
        raise GdkvmError("conv_s2_packs: fp32 [K,C,3,3] device weight")
    if down_weight is not None and (tuple(down_weight.shape) != (k, c, 1, 1) or down_weight.dtype != torch.float32 or down_weight.device != weight.device):
        raise GdkvmError("conv_s2_packs: the branch weight must be fp32 [K,C,1,1] on the same device")
    dev = weight.device
    fwd = torch.empty(k * 9 * c, dtype=torch.bfloat16, device=dev)
    fwd_d = None if down_weight is None else torch.empty(k * c, dtype=torch.bfloat16, device=dev)
    dg = torch.empty(int(lib.gdkvm_conv_s2_dgrad_pack_bytes(c, k, int(down_weight is not None))) // 2, dtype=torch.bfloat16, device=dev)
    st = (ctypes.c_longlong * 4)(*weight.stride())
    sd = None if down_weight is None else (ctypes.c_longlong * 2)(*down_weight.stride()[:2])
    with torch.cuda.device(dev):
        rc = lib.gdkvm_conv_s2_pack_train(weight.data_ptr(), ctypes.cast(st, ctypes.c_void_p), _ptr(down_weight),
                                          None if sd is None else ctypes.cast(sd, ctypes.c_void_p), fwd.data_ptr(), _ptr(fwd_d), dg.data_ptr(),
                                          k, c, _stream(dev))
    _check(rc, "gdkvm_conv_s2_pack_train")
    return fwd, fwd_d, dg


class _ConvS2BlockFunction(torch.autograd.Function):
    """A residual block's 3x3 / stride-2 / pad-1 convolution and its 1x1 / stride-2 downsample branch as ONE autograd node on the hand-written
    kernels (csrc/conv_s2_train.hip): (y, y_down) forward from one launch, one data-gradient launch for both branches (the framework would add
    two full-size gradients of the block's input), two deterministic weight gradients.  bf16 activations, fp32 master weights."""

    @staticmethod
    def forward(ctx, x, weight, down_weight):
        lib = load()
        xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        n, c, hh, ww = xb.shape
        k = weight.shape[0]
        fwd, fwd_d, dg = conv_s2_packs(weight.detach(), down_weight.detach())
        ho, wo = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
        y = torch.empty((n, k, ho, wo), dtype=torch.bfloat16, device=xb.device, memory_format=torch.channels_last)
        yd = torch.empty_like(y)
        with torch.cuda.device(xb.device):
            rc = lib.gdkvm_conv_down_bias_act(xb.data_ptr(), fwd.data_ptr(), _zero_bias(k, xb.device).data_ptr(), y.data_ptr(), 0, fwd_d.data_ptr(),
                                              None, yd.data_ptr(), n, c, hh, ww, k, 3, 3, 2, 1, BF16, _stream(xb.device))
        _check(rc, "gdkvm_conv_down_bias_act")
        ctx.save_for_backward(xb, dg, weight, down_weight)
        ctx.xdtype = x.dtype
        return y, yd

    @staticmethod
    def backward(ctx, dy, dyd):
        lib = load()
        xb, dg, weight, down_weight = ctx.saved_tensors
        n, c, hh, ww = xb.shape
        k = weight.shape[0]
        cl = torch.channels_last
        dyb = dy.to(torch.bfloat16).contiguous(memory_format=cl)
        dydb = dyd.to(torch.bfloat16).contiguous(memory_format=cl)
        dx = dw = dwd = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(xb)
            with torch.cuda.device(xb.device):
                rc = lib.gdkvm_conv_s2_dgrad(dyb.data_ptr(), dydb.data_ptr(), dg.data_ptr(), dx.data_ptr(), n, c, hh, ww, k, 1, BF16, _stream(xb.device))
            _check(rc, "gdkvm_conv_s2_dgrad")
            dx = dx.to(ctx.xdtype)
        if ctx.needs_input_grad[1]:
            dw = conv_wgrad_strided(xb, dyb, weight, 2, 1).to(weight.dtype)
        if ctx.needs_input_grad[2]:
            dwd = conv_wgrad_strided(xb, dydb, down_weight, 2, 0).to(down_weight.dtype)
        return dx, dw, dwd


def conv_s2_block_served(x: torch.Tensor, conv, down) -> bool:
    """Does conv_s2_block serve this residual block's (3x3 / stride-2 convolution, 1x1 / stride-2 branch) pair on this input?"""
    w = conv.weight
    k, c = w.shape[:2]
    return (x.is_cuda and x.dim() == 4 and conv.bias is None and down.bias is None and tuple(w.shape[2:]) == (3, 3)
            and tuple(conv.stride) == (2, 2) and tuple(conv.padding) == (1, 1) and tuple(conv.dilation) == (1, 1) and conv.groups == 1
            and conv.padding_mode == "zeros" and tuple(down.weight.shape) == (k, c, 1, 1) and tuple(down.stride) == (2, 2)
            and tuple(down.padding) == (0, 0) and down.groups == 1 and c % 64 == 0 and k % 128 == 0
            and w.dtype == torch.float32 and down.weight.dtype == torch.float32
            and (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)))


def conv_s2_block(x: torch.Tensor, weight: torch.Tensor, down_weight: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(conv2d(x, weight, stride 2, padding 1), conv2d(x, down_weight, stride 2)) in bf16, differentiable, deterministic."""
    return _ConvS2BlockFunction.apply(x, weight, down_weight)


def conv_bias_act(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, residual: Optional[torch.Tensor] = None,
                  stride: int = 1, padding: int = 1, relu: bool = True, tile: int = 0,
                  packed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(conv2d(x, weight) + bias[k] (+ residual)) as ONE kernel on channels_last bf16 tensors (gdkvm_conv_bias_act): the
    epilogue runs on the fp32 accumulator inside the implicit-GEMM kernel, no second pass over the output.  packed: the
    conv3x3_pack_weights copy of `weight` (kernel 5 reads it instead: faster, same result)."""
    lib = load()
    if x.dim() != 4 or not x.is_cuda or x.dtype != torch.bfloat16 or not x.is_contiguous(memory_format=torch.channels_last):
        raise GdkvmError("conv_bias_act needs a channels_last bf16 [N,C,H,W] device tensor (no CPU path)")
    if weight.dim() != 4 or weight.dtype != torch.bfloat16 or weight.shape[1] != x.shape[1] or \
            not weight.is_contiguous(memory_format=torch.channels_last):
        raise GdkvmError("conv_bias_act: weight must be channels_last bf16 [K,C,R,S]")
    n, c, hh, ww = x.shape
    k, _, r, s = weight.shape
    if bias.dtype != torch.float32 or bias.numel() != k:
        raise GdkvmError("bias must be float32 [K]")
    ho, wo = (hh + 2 * padding - r) // stride + 1, (ww + 2 * padding - s) // stride + 1
    y = torch.empty((n, k, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if residual is not None and (residual.shape != y.shape or residual.dtype != y.dtype or
                                 not residual.is_contiguous(memory_format=torch.channels_last)):
        raise GdkvmError("residual must match the output (shape, dtype, channels_last)")
    with torch.cuda.device(x.device):
        if packed is not None:
            if packed.dtype != torch.bfloat16 or packed.numel() != weight.numel() or packed.device != x.device:
                raise GdkvmError("conv_bias_act: packed must be conv3x3_pack_weights(weight)")
            tile |= CONV_PACKED_WEIGHTS
        rc = lib.gdkvm_conv_bias_act(x.data_ptr(), (packed if packed is not None else weight).data_ptr(), bias.data_ptr(), _ptr(residual),
                                     y.data_ptr(), n, c, hh, ww, k, r, s, stride, padding, int(relu), tile, BF16, _stream(x.device))
    _check(rc, "gdkvm_conv_bias_act")
    return y


def bias_relu_maxpool(x: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """relu(max_pool2d(x, 3, 2, 1) + bias[c]) on a channels_last conv output in one pass (gdkvm_bias_relu_maxpool);
    equals max_pool2d(relu(x + bias), 3, 2, 1).  x may be a top-left spatial crop (a view) of a larger channels_last tensor."""
    lib = load()
    if x.dim() != 4 or not x.is_cuda:
        raise GdkvmError("bias_relu_maxpool needs a channels_last [N,C,H,W] device tensor (or a spatial crop of one)")
    n, c, hh, ww = x.shape
    if x.is_contiguous(memory_format=torch.channels_last):
        x_rows, x_cols = hh, ww
    else:                                                   # a top-left spatial crop: the strides carry the stored size
        sn, sc, sh, sw = x.stride()
        if sc != 1 or sw != c or sh % c or (n > 1 and sn % sh):
            raise GdkvmError("bias_relu_maxpool needs a channels_last [N,C,H,W] device tensor (or a spatial crop of one)")
        x_cols = sh // c
        x_rows = sn // sh if n > 1 else hh
    if bias.dtype != torch.float32 or bias.numel() != c:
        raise GdkvmError("bias must be float32 [C]")
    out = torch.empty((n, c, (hh - 1) // 2 + 1, (ww - 1) // 2 + 1), device=x.device, dtype=x.dtype,
                      memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = lib.gdkvm_bias_relu_maxpool(x.data_ptr(), bias.data_ptr(), out.data_ptr(), n, hh, ww, c, x_rows, x_cols,
                                         _io_dtype(x), _stream(x.device))
    _check(rc, "gdkvm_bias_relu_maxpool")
    return out


def stem_s2d(x: torch.Tensor, cpad: int) -> torch.Tensor:
    """Space-to-depth of NCHW frames for the stem: [N,C,H,W] contiguous -> channels_last [N,cpad,H/2,W/2] with channel
    (c*2+p)*2+q = x[c, 2i+p, 2j+q] and zeros above 4C (gdkvm_stem_s2d)."""
    lib = load()
    if x.dim() != 4 or not x.is_cuda or not x.is_contiguous():
        raise GdkvmError("stem_s2d needs a contiguous NCHW device tensor")
    n, c, hh, ww = x.shape
    out = torch.empty((n, cpad, hh // 2, ww // 2), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = lib.gdkvm_stem_s2d(x.data_ptr(), out.data_ptr(), n, c, hh, ww, cpad, _io_dtype(x), _stream(x.device))
    _check(rc, "gdkvm_stem_s2d")
    return out


def upsample_bilinear(lo: torch.Tensor, size) -> torch.Tensor:
    """F.interpolate(lo, size, mode="bilinear", align_corners=False) on a channels_last bf16 [N,C,h,w] device tensor
    (gdkvm_upsample_cat without a skip tensor): what conv_cat_bias_act reads next to the skip feature."""
    lib = load()
    if lo.dim() != 4 or not lo.is_cuda or lo.dtype != torch.bfloat16 or not lo.is_contiguous(memory_format=torch.channels_last):
        raise GdkvmError("upsample_bilinear needs a channels_last bf16 [N,C,h,w] device tensor (no CPU path)")
    n, c1, hl, wl = lo.shape
    H, W = int(size[0]), int(size[1])
    out = torch.empty((n, c1, H, W), dtype=lo.dtype, device=lo.device, memory_format=torch.channels_last)
    with torch.cuda.device(lo.device):
        rc = lib.gdkvm_upsample_cat(lo.data_ptr(), None, out.data_ptr(), n, hl, wl, H, W, c1, 0, BF16, _stream(lo.device))
    _check(rc, "gdkvm_upsample_cat")
    return out


def conv_cat_bias_act(x1: torch.Tensor, x2: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor,
                      residual: Optional[torch.Tensor] = None, relu: bool = True, tile: int = 0,
                      packed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv_bias_act(torch.cat([x1, x2], 1), weight, ...) for a 3x3 / stride 1 / pad 1 layer WITHOUT the concatenated tensor
    (gdkvm_conv_cat_bias_act): channel counts in multiples of 64, same result bit for bit."""
    lib = load()
    for t in (x1, x2):
        if t.dim() != 4 or not t.is_cuda or t.dtype != torch.bfloat16 or not t.is_contiguous(memory_format=torch.channels_last):
            raise GdkvmError("conv_cat_bias_act needs channels_last bf16 [N,C,H,W] device tensors (no CPU path)")
    n, c1, hh, ww = x1.shape
    c2 = x2.shape[1]
    if tuple(x2.shape) != (n, c2, hh, ww):
        raise GdkvmError("conv_cat_bias_act: x1 and x2 must agree in batch and size")
    if weight.dim() != 4 or weight.dtype != torch.bfloat16 or tuple(weight.shape[1:]) != (c1 + c2, 3, 3) or \
            not weight.is_contiguous(memory_format=torch.channels_last):
        raise GdkvmError("conv_cat_bias_act: weight must be channels_last bf16 [K,C1+C2,3,3]")
    k = weight.shape[0]
    if bias.dtype != torch.float32 or bias.numel() != k:
        raise GdkvmError("bias must be float32 [K]")
    y = torch.empty((n, k, hh, ww), dtype=x1.dtype, device=x1.device, memory_format=torch.channels_last)
    if residual is not None and (residual.shape != y.shape or residual.dtype != y.dtype or
                                 not residual.is_contiguous(memory_format=torch.channels_last)):
        raise GdkvmError("residual must match the output (shape, dtype, channels_last)")
    if packed is not None:
        if packed.dtype != torch.bfloat16 or packed.numel() != weight.numel() or packed.device != x1.device:
            raise GdkvmError("conv_cat_bias_act: packed must be conv3x3_pack_weights(weight)")
        tile |= CONV_PACKED_WEIGHTS
    with torch.cuda.device(x1.device):
        rc = lib.gdkvm_conv_cat_bias_act(x1.data_ptr(), x2.data_ptr(), (packed if packed is not None else weight).data_ptr(), bias.data_ptr(),
                                         _ptr(residual), y.data_ptr(), n, c1, c2, hh, ww, k, int(relu), tile, BF16, _stream(x1.device))
    _check(rc, "gdkvm_conv_cat_bias_act")
    return y


def _upsample_cat_fwd(lo: torch.Tensor, skip: torch.Tensor) -> torch.Tensor:
    lib = load()
    for t in (lo, skip):
        if t.dim() != 4 or not t.is_cuda or not t.is_contiguous(memory_format=torch.channels_last):
            raise GdkvmError("upsample_cat needs channels_last device tensors")
    if lo.dtype != torch.bfloat16 or skip.dtype != torch.bfloat16 or lo.shape[0] != skip.shape[0]:
        raise GdkvmError("upsample_cat: bf16 tensors with equal batch")
    n, c1, hl, wl = lo.shape
    _, c2, H, W = skip.shape
    out = torch.empty((n, c1 + c2, H, W), dtype=lo.dtype, device=lo.device, memory_format=torch.channels_last)
    with torch.cuda.device(lo.device):
        rc = lib.gdkvm_upsample_cat(lo.data_ptr(), skip.data_ptr(), out.data_ptr(), n, hl, wl, H, W, c1, c2, BF16, _stream(lo.device))
    _check(rc, "gdkvm_upsample_cat")
    return out


def stem_conv_pool(xs: torch.Tensor, w_s2d: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """max_pool2d(relu(conv2d(xs, w_s2d, padding=2)[..., :Hs, :Ws] + bias), 3, 2, 1) in one kernel (gdkvm_stem_conv_pool):
    xs channels_last bf16 [N,16,Hs,Ws] (ops.stem_s2d), w_s2d channels_last bf16 [64,16,4,4], bias fp32 [64]."""
    lib = load()
    if xs.dim() != 4 or not xs.is_cuda or xs.dtype != torch.bfloat16 or xs.shape[1] != 16 or \
            not xs.is_contiguous(memory_format=torch.channels_last):
        raise GdkvmError("stem_conv_pool needs a channels_last bf16 [N,16,Hs,Ws] device tensor (no CPU path)")
    if tuple(w_s2d.shape) != (64, 16, 4, 4) or w_s2d.dtype != torch.bfloat16 or not w_s2d.is_contiguous(memory_format=torch.channels_last):
        raise GdkvmError("stem_conv_pool: weight must be channels_last bf16 [64,16,4,4]")
    if bias.dtype != torch.float32 or bias.numel() != 64:
        raise GdkvmError("bias must be float32 [64]")
    n, _, hs, ws = xs.shape
    y = torch.empty((n, 64, (hs - 1) // 2 + 1, (ws - 1) // 2 + 1), dtype=xs.dtype, device=xs.device, memory_format=torch.channels_last)
    with torch.cuda.device(xs.device):
        rc = lib.gdkvm_stem_conv_pool(xs.data_ptr(), w_s2d.data_ptr(), bias.data_ptr(), y.data_ptr(), n, hs, ws, BF16, _stream(xs.device))
    _check(rc, "gdkvm_stem_conv_pool")
    return y


def stem_conv_pool_nchw(x: torch.Tensor, w_s2d: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """stem_conv_pool(stem_s2d(x, 16), w_s2d, bias) without the space-to-depth copy (gdkvm_stem_conv_pool_nchw): x contiguous NCHW bf16
    [N, C <= 4, H, W] with H, W even; bit-identical to the two-kernel form."""
    lib = load()
    if x.dim() != 4 or not x.is_cuda or x.dtype != torch.bfloat16 or not x.is_contiguous() or x.shape[1] > 4 or x.shape[2] % 2 or x.shape[3] % 2:
        raise GdkvmError("stem_conv_pool_nchw needs a contiguous NCHW bf16 [N, C <= 4, H, W] device tensor with even H, W (no CPU path)")
    if tuple(w_s2d.shape) != (64, 16, 4, 4) or w_s2d.dtype != torch.bfloat16 or not w_s2d.is_contiguous(memory_format=torch.channels_last):
        raise GdkvmError("stem_conv_pool: weight must be channels_last bf16 [64,16,4,4]")
    if bias.dtype != torch.float32 or bias.numel() != 64:
        raise GdkvmError("bias must be float32 [64]")
    n, c, hh, ww = x.shape
    hs, ws = hh // 2, ww // 2
    y = torch.empty((n, 64, (hs - 1) // 2 + 1, (ws - 1) // 2 + 1), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = lib.gdkvm_stem_conv_pool_nchw(x.data_ptr(), w_s2d.data_ptr(), bias.data_ptr(), y.data_ptr(), n, c, hh, ww, BF16, _stream(x.device))
    _check(rc, "gdkvm_stem_conv_pool_nchw")
    return y


class _StemConvFunction(torch.autograd.Function):
    """The training stem's convolution (7x7 / stride 2 / pad 3, no bias, <= 4 input channels -> 64) on the hand-written stem kernel in its
    convolution-only form (gdkvm_stem_pack_s2d + gdkvm_stem_conv_nchw): bf16 NCHW frames in, channels_last bf16 out.  The input is the
    image (no data gradient); the weight gradient is gdkvm_stem_wgrad_nchw (stem_wgrad)."""

    @staticmethod
    def forward(ctx, x, weight):
        lib = load()
        xb = x.detach().to(torch.bfloat16).contiguous()
        n, c, hh, ww = xb.shape
        w4 = torch.empty(64 * 256, dtype=torch.bfloat16, device=xb.device)
        y = torch.empty((n, 64, hh // 2, ww // 2), dtype=torch.bfloat16, device=xb.device, memory_format=torch.channels_last)
        wd = weight.detach()
        with torch.cuda.device(xb.device):
            _check(lib.gdkvm_stem_pack_s2d(wd.data_ptr(), w4.data_ptr(), c, *wd.stride(), _stream(xb.device)), "gdkvm_stem_pack_s2d")
            _check(lib.gdkvm_stem_conv_nchw(xb.data_ptr(), w4.data_ptr(), y.data_ptr(), n, c, hh, ww, BF16, _stream(xb.device)), "gdkvm_stem_conv_nchw")
        ctx.save_for_backward(xb, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, weight = ctx.saved_tensors
        dw = None
        if ctx.needs_input_grad[1]:
            dw = stem_wgrad(xb, dy, like=weight)
        return None, dw


def stem_wgrad(x: torch.Tensor, dy: torch.Tensor, like: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Weight gradient [64, C, 7, 7] fp32 of the stem convolution (7x7 / stride 2 / pad 3) for NCHW bf16 frames x [N, C <= 4, H, W] and
    dy [N, 64, H/2, W/2] (channels_last bf16), on gdkvm_stem_wgrad_nchw: fixed summation order, in the memory format of `like`."""
    lib = load()
    xb = x.detach().to(torch.bfloat16).contiguous()
    dyb = _nhwc(dy.detach(), "stem_wgrad")
    if dyb.dtype != torch.bfloat16:
        dyb = dyb.to(torch.bfloat16)
    n, c, hh, ww = xb.shape
    if tuple(dyb.shape) != (n, 64, hh // 2, ww // 2):
        raise GdkvmError(f"stem_wgrad: dy {tuple(dyb.shape)} does not match x {tuple(xb.shape)}")
    dw = torch.empty_like(like, dtype=torch.float32) if like is not None else torch.empty((64, c, 7, 7), dtype=torch.float32, device=xb.device)
    need = lib.gdkvm_stem_wgrad_workspace_bytes(n, hh, ww)
    ws = torch.empty(need, dtype=torch.uint8, device=xb.device)
    with torch.cuda.device(xb.device):
        _check(lib.gdkvm_stem_wgrad_nchw(xb.data_ptr(), dyb.data_ptr(), dw.data_ptr(), *dw.stride(), ws.data_ptr(), need, n, c, hh, ww, BF16,
                                         _stream(xb.device)), "gdkvm_stem_wgrad_nchw")
    return dw


def stem_conv_served(x: torch.Tensor, conv) -> bool:
    w = conv.weight
    return (x.is_cuda and x.dim() == 4 and x.shape[1] <= 4 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and conv.bias is None
            and tuple(w.shape) == (64, x.shape[1], 7, 7) and w.dtype == torch.float32 and conv.stride == (2, 2) and conv.padding == (3, 3)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros"
            and (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)))


def stem_conv(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """conv2d(x, weight, stride=2, padding=3) for the training stem (bf16 out, channels_last), differentiable in the weight."""
    return _StemConvFunction.apply(x, weight)


def upsample_cat_bwd(dout: torch.Tensor, lo_shape, skip_shape) -> Tuple[torch.Tensor, torch.Tensor]:
    """(d_lo, d_skip) of upsample_cat for d_out [N,C1+C2,H,W] channels_last bf16 (gdkvm_upsample_cat_bwd)."""
    lib = load()
    dout = _nhwc(dout, "upsample_cat_bwd")
    if dout.dtype != torch.bfloat16:
        dout = dout.to(torch.bfloat16)
    n, c1, hl, wl = lo_shape
    _, c2, H, W = skip_shape
    if tuple(dout.shape) != (n, c1 + c2, H, W):
        raise GdkvmError(f"upsample_cat_bwd: d_out {tuple(dout.shape)} does not match {lo_shape} + {skip_shape}")
    dlo = torch.empty((n, c1, hl, wl), dtype=dout.dtype, device=dout.device, memory_format=torch.channels_last)
    dskip = torch.empty((n, c2, H, W), dtype=dout.dtype, device=dout.device, memory_format=torch.channels_last)
    with torch.cuda.device(dout.device):
        rc = lib.gdkvm_upsample_cat_bwd(dout.data_ptr(), dlo.data_ptr(), dskip.data_ptr(), n, hl, wl, H, W, c1, c2, BF16,
                                        _stream(dout.device))
    _check(rc, "gdkvm_upsample_cat_bwd")
    return dlo, dskip


class _UpsampleCatFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lo, skip):
        ctx.shapes = (tuple(lo.shape), tuple(skip.shape))
        return _upsample_cat_fwd(lo, skip)

    @staticmethod
    def backward(ctx, dout):
        return upsample_cat_bwd(dout, *ctx.shapes)


def upsample_cat(lo: torch.Tensor, skip: torch.Tensor) -> torch.Tensor:
    """concat(bilinear_upsample(lo -> skip's H x W, align_corners=False), skip) along channels, both channels_last
    bf16 [N,C,h,w]; returns a channels_last [N,C1+C2,H,W] tensor (gdkvm_upsample_cat; differentiable:
    gdkvm_upsample_cat_bwd)."""
    if torch.is_grad_enabled() and (lo.requires_grad or skip.requires_grad):
        return _UpsampleCatFunction.apply(lo, skip)
    return _upsample_cat_fwd(lo, skip)


def _nhwc(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dim() != 4 or not t.is_cuda:
        raise GdkvmError(f"{what}: needs a [N,C,H,W] device tensor (no CPU path)")
    return t.contiguous(memory_format=torch.channels_last)


def bn_act_fwd(x, weight, bias, running_mean=None, running_var=None, residual=None, momentum: float = 0.1, eps: float = 1e-5,
               relu: bool = True):
    """Batch-statistics BatchNorm + optional residual add + optional ReLU on a channels_last tensor in three streaming
    passes (gdkvm_bn_fwd_train).  Updates running_mean / running_var in place.  Returns (y, stats) with stats fp32 [4, C] =
    (mean, 1/sqrt(var + eps), scale, shift)."""
    lib = load()
    x = _nhwc(x, "bn_act_fwd")
    n, c, hh, ww = x.shape
    if residual is not None:
        residual = _nhwc(residual, "bn_act_fwd")
        if residual.shape != x.shape or residual.dtype != x.dtype:
            raise GdkvmError("bn_act_fwd: residual must match x (shape, dtype)")
    for t in (weight, bias, running_mean, running_var):
        if t is not None and (t.dtype != torch.float32 or t.numel() != c or not t.is_contiguous()):
            raise GdkvmError("bn_act_fwd: weight / bias / running statistics must be contiguous float32 [C]")
    y = torch.empty_like(x)
    stats = torch.empty((4, c), dtype=torch.float32, device=x.device)
    ws = torch.empty(int(lib.gdkvm_bn_workspace_bytes(c)), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.gdkvm_bn_fwd_train(x.data_ptr(), _ptr(residual), weight.data_ptr(), bias.data_ptr(), _ptr(running_mean),
                                    _ptr(running_var), y.data_ptr(), stats.data_ptr(), ws.data_ptr(), ws.numel(),
                                    n * hh * ww, c, float(eps), float(momentum), int(relu), _io_dtype(x), _stream(x.device))
    _check(rc, "gdkvm_bn_fwd_train")
    return y, stats


def bn_act_bwd(x, y, dy, weight, stats, relu: bool = True, want_dres: bool = False):
    """Backward of bn_act_fwd (gdkvm_bn_bwd): returns (dx, dres | None, dweight, dbias).  y = None with relu: the forward had
    no residual and the ReLU mask is recomputed from x (one tensor less to read)."""
    lib = load()
    x, dy = _nhwc(x, "bn_act_bwd"), _nhwc(dy, "bn_act_bwd")
    if dy.dtype != x.dtype:
        dy = dy.to(x.dtype)
    if dy.shape != x.shape or (y is not None and y.shape != x.shape):
        raise GdkvmError("bn_act_bwd: shape mismatch")
    n, c, hh, ww = x.shape
    mode = 0 if not relu else (2 if y is None else 1)
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if (want_dres and relu) else None
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty_like(dgamma)
    ws = torch.empty(int(lib.gdkvm_bn_workspace_bytes(c)), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.gdkvm_bn_bwd(x.data_ptr(), _ptr(y) if mode == 1 else None, dy.data_ptr(), weight.data_ptr(), stats.data_ptr(),
                              dx.data_ptr(), _ptr(dres), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(),
                              ws.numel(), n * hh * ww, c, mode, _io_dtype(x), _stream(x.device))
    _check(rc, "gdkvm_bn_bwd")
    if want_dres and not relu:
        dres = dy                                           # no mask: the residual branch receives dy itself
    return dx, dres, dgamma, dbeta


class _BNActFunction(torch.autograd.Function):
    """y = act(BatchNorm_train(x) (+ residual)) with the HIP passes of csrc/bn.hip in both directions."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, momentum, eps, relu):
        x = _nhwc(x, "bn_act")
        y, stats = bn_act_fwd(x, weight, bias, running_mean, running_var, residual, momentum, eps, relu)
        ctx.relu, ctx.has_res = relu, residual is not None
        ctx.save_for_backward(x, y if (relu and residual is not None) else None, weight, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, stats = ctx.saved_tensors
        dx, dres, dgamma, dbeta = bn_act_bwd(x, y, dy, weight, stats, ctx.relu, ctx.has_res and ctx.needs_input_grad[3])
        return dx, dgamma, dbeta, dres, None, None, None, None, None


def bn_act(x, weight, bias, running_mean=None, running_var=None, residual=None, momentum: float = 0.1, eps: float = 1e-5,
           relu: bool = True):
    """Differentiable fused BatchNorm(batch statistics) (+ residual) (+ ReLU); see bn_act_fwd."""
    return _BNActFunction.apply(x, weight, bias, residual, running_mean, running_var, momentum, eps, relu)


class _BNReluPoolFunction(torch.autograd.Function):
    """maxpool3x3s2(relu(BatchNorm_train(x))) as one op in both directions (gdkvm_bn_pool_fwd_train / gdkvm_bn_pool_bwd): the training
    stem's tail without the full-resolution activation or its gradient ever being written.  Same bits as bn_act + maxpool3x3s2."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps):
        lib = load()
        x = _nhwc(x, "bn_relu_pool")
        n, c, hh, ww = x.shape
        ho, wo = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
        for t in (weight, bias, running_mean, running_var):
            if t is not None and (t.dtype != torch.float32 or t.numel() != c or not t.is_contiguous()):
                raise GdkvmError("bn_relu_pool: weight / bias / running statistics must be contiguous float32 [C]")
        y = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=x.device)
        stats = torch.empty((4, c), dtype=torch.float32, device=x.device)
        ws = torch.empty(int(lib.gdkvm_bn_workspace_bytes(c)), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.gdkvm_bn_pool_fwd_train(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), _ptr(running_mean), _ptr(running_var),
                                             y.data_ptr(), idx.data_ptr(), stats.data_ptr(), ws.data_ptr(), ws.numel(), n, hh, ww, c,
                                             float(eps), float(momentum), _io_dtype(x), _stream(x.device))
        _check(rc, "gdkvm_bn_pool_fwd_train")
        ctx.save_for_backward(x, idx, weight, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = load()
        x, idx, weight, stats = ctx.saved_tensors
        n, c, hh, ww = x.shape
        dy = _nhwc(dy, "bn_relu_pool backward")
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = torch.empty_like(x)
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = torch.empty_like(dgamma)
        ws = torch.empty(int(lib.gdkvm_bn_workspace_bytes(c)), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.gdkvm_bn_pool_bwd(x.data_ptr(), dy.data_ptr(), idx.data_ptr(), weight.data_ptr(), stats.data_ptr(), dx.data_ptr(),
                                       dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), ws.numel(), n, hh, ww, c, _io_dtype(x), _stream(x.device))
        _check(rc, "gdkvm_bn_pool_bwd")
        return dx, dgamma, dbeta, None, None, None, None


def bn_relu_pool_served(x: torch.Tensor) -> bool:
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and x.shape[1] % 8 == 0 and x.shape[1] // 8 <= 256
            and x.shape[0] * x.shape[2] * x.shape[3] < (1 << 22))


def bn_relu_pool(x, weight, bias, running_mean=None, running_var=None, momentum: float = 0.1, eps: float = 1e-5):
    """maxpool3x3s2(bn_act(x, ..., relu=True)) as one differentiable op (the training stem's tail); bf16 channels_last."""
    return _BNReluPoolFunction.apply(x, weight, bias, running_mean, running_var, momentum, eps)


class _MaxPoolFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = load()
        x = _nhwc(x, "maxpool3x3s2")
        n, c, hh, ww = x.shape
        ho, wo = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
        y = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.gdkvm_maxpool_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), n, hh, ww, c, _io_dtype(x), _stream(x.device))
        _check(rc, "gdkvm_maxpool_fwd")
        ctx.save_for_backward(idx)
        ctx.in_shape = (n, c, hh, ww)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = load()
        (idx,) = ctx.saved_tensors
        n, c, hh, ww = ctx.in_shape
        dy = _nhwc(dy, "maxpool3x3s2 backward")
        dx = torch.empty((n, c, hh, ww), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        with torch.cuda.device(dy.device):
            rc = lib.gdkvm_maxpool_bwd(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), n, hh, ww, c, _io_dtype(dy), _stream(dy.device))
        _check(rc, "gdkvm_maxpool_bwd")
        return dx


def maxpool3x3s2(x: torch.Tensor) -> torch.Tensor:
    """max_pool2d(x, 3, 2, 1) on a channels_last tensor, differentiable (gdkvm_maxpool_fwd / _bwd: the backward is a gather
    from the recorded winning taps, PyTorch's tie rule)."""
    return _MaxPoolFunction.apply(x)


def gemm_nt(a: torch.Tensor, bt: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C [M,N] = a [M,K] @ bt [N,K]^T (+ bias [N], fp32) on the hand-written MFMA kernel (gdkvm_gemm_nt); a, bt share one dtype
    (bf16 or fp32), C comes back in it, accumulation is fp32."""
    lib = load()
    if a.dim() != 2 or bt.dim() != 2 or a.shape[1] != bt.shape[1] or a.dtype != bt.dtype:
        raise GdkvmError(f"gemm_nt: a [M,K] and bt [N,K] of one dtype, got {tuple(a.shape)} {a.dtype} / {tuple(bt.shape)} {bt.dtype}")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != bt.shape[0]):
        raise GdkvmError("gemm_nt: bias must be float32 [N]")
    kq = 32 if a.dtype == torch.bfloat16 else 16
    if a.shape[1] % kq:                                     # an odd contraction length: zero columns up to the MFMA's k step
        pad = kq - a.shape[1] % kq
        a, bt = torch.nn.functional.pad(a, (0, pad)), torch.nn.functional.pad(bt, (0, pad))
    dev = _dev(a, bt, bias)
    M, K = a.shape
    N = bt.shape[0]
    c = torch.empty((M, N), dtype=a.dtype, device=dev)
    with torch.cuda.device(dev):
        rc = lib.gdkvm_gemm_nt(_ptr(a), _ptr(bt), _ptr(bias), _ptr(c), M, N, K, _io_dtype(a), _stream(dev))
    _check(rc, "gdkvm_gemm_nt")
    return c


def wgrad(dy2d: torch.Tensor, x2d: torch.Tensor, colsum: bool = False):
    """dW [M,N] (fp32) = dy^T x for token-major operands dy [K,M], x [K,N] with K (the tokens of the batch) >> M, N: the
    reduction over the tokens is split over workgroups into fp32 partial tiles and summed in a fixed order (gdkvm_gemm_tn);
    the sum never passes through bf16.  colsum=True: (dW, db) with db [M] (fp32) = the column sums of dy -- the bias gradient --
    from the same launch (gdkvm_gemm_tn_colsum)."""
    lib = load()
    if dy2d.dim() != 2 or x2d.dim() != 2 or dy2d.shape[0] != x2d.shape[0]:
        raise GdkvmError("wgrad: dy [K,M] and x [K,N] with equal K")
    if dy2d.dtype != x2d.dtype:
        x2d = x2d.to(dy2d.dtype)
    dy2d, x2d = dy2d.contiguous(), x2d.contiguous()
    dev = _dev(dy2d, x2d)
    K, M = dy2d.shape
    N = x2d.shape[1]
    c = torch.empty((M, N), dtype=torch.float32, device=dev)
    if colsum:
        db = torch.empty(M, dtype=torch.float32, device=dev)
        ws = torch.empty(int(lib.gdkvm_gemm_tn_colsum_workspace_bytes(K, M, N)), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = lib.gdkvm_gemm_tn_colsum(_ptr(dy2d), _ptr(x2d), _ptr(c), _ptr(db), ws.data_ptr(), ws.numel(), K, M, N, _io_dtype(dy2d), _stream(dev))
        _check(rc, "gdkvm_gemm_tn_colsum")
        return c, db
    ws = torch.empty(int(lib.gdkvm_gemm_tn_workspace_bytes(K, M, N)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.gdkvm_gemm_tn(_ptr(dy2d), _ptr(x2d), _ptr(c), ws.data_ptr(), ws.numel(), K, M, N, _io_dtype(dy2d), _stream(dev))
    _check(rc, "gdkvm_gemm_tn")
    return c


class _TokenLinear(torch.autograd.Function):
    """y = x W^T + b on token rows [K, Cin] (the 1x1 projections), forward and backward on the hand-written products:
    gdkvm_gemm_nt with the bias in its epilogue; dX = dY W as gdkvm_gemm_nt on the weight with its input index leading; dW
    through wgrad (gdkvm_gemm_tn)."""

    @staticmethod
    def forward(ctx, x2d, weight, bias):
        w = weight.to(x2d.dtype).contiguous()
        x2d = x2d.contiguous()
        ctx.save_for_backward(x2d, w)
        ctx.has_bias, ctx.wdtype = bias is not None, weight.dtype
        return gemm_nt(x2d, w, None if bias is None else bias.detach().float().contiguous())

    @staticmethod
    def backward(ctx, dy):
        x2d, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = gemm_nt(dy, w.t().contiguous()) if ctx.needs_input_grad[0] else None
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        dw = db = None
        if ctx.needs_input_grad[1] and want_b and dy.is_cuda:
            dw, db = wgrad(dy, x2d, colsum=True)            # the bias gradient rides on the weight gradient's launch (one more MFMA per step)
            dw, db = dw.to(ctx.wdtype), db.to(ctx.wdtype)
        else:
            if ctx.needs_input_grad[1]:
                dw = wgrad(dy, x2d).to(ctx.wdtype)
            if want_b:                                      # (column sums with fp32 accumulation straight from the rows)
                db = dy.sum(0, dtype=torch.float32).to(ctx.wdtype)
        return dx, dw, db


def token_linear(x2d: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    return _TokenLinear.apply(x2d, weight, bias)


class _TokenProjections(torch.autograd.Function):
    """Several 1x1 projections of the SAME token rows (key, query, value, write gate) as ONE product against their stacked weights, forward
    and backward: y_i = x W_i^T + b_i.  As four _TokenLinear calls a training step spent ~55 launches on them -- per projection a weight cast, the
    product, a transposed weight copy, the data-gradient product, an add into the feature's gradient, the split-K weight gradient and its
    sum, plus zero-padding of the one-row gate projection both ways; stacked it is: cat + cast of the weights, cat of the biases, one
    gdkvm_gemm_nt, one copy per output (contiguous rows for the scan kernels) -- and backward one cat of the incoming gradients, one
    transposed weight copy, one gdkvm_gemm_nt for dX and one gdkvm_gemm_tn_colsum for every dW and db (returned as row slices of its result).
    The stacked width is padded with zero rows to a multiple of 32 (the MFMA's k step in the dX product)."""

    @staticmethod
    def forward(ctx, x2d, *wb):
        ws, bs = wb[0::2], wb[1::2]
        x2d = x2d.contiguous()
        widths = [int(w.shape[0]) for w in ws]
        total = sum(widths)
        padded = (total + 31) // 32 * 32
        cin = x2d.shape[1]
        parts = [w.detach().reshape(w.shape[0], cin) for w in ws]
        if padded > total:
            parts.append(torch.zeros((padded - total, cin), dtype=parts[0].dtype, device=x2d.device))
        w_all = torch.cat(parts, 0).to(x2d.dtype)
        b_parts = [(b.detach().float() if b is not None else torch.zeros(n, dtype=torch.float32, device=x2d.device)) for b, n in zip(bs, widths)]
        if padded > total:
            b_parts.append(torch.zeros(padded - total, dtype=torch.float32, device=x2d.device))
        y = gemm_nt(x2d, w_all, torch.cat(b_parts, 0))
        ctx.save_for_backward(x2d, w_all)
        ctx.widths, ctx.padded = widths, padded
        ctx.meta = [(w.dtype, tuple(w.shape), b is not None, None if b is None else b.dtype) for w, b in zip(ws, bs)]
        outs, o = [], 0
        for n in widths:
            outs.append(y[:, o:o + n].contiguous())
            o += n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        x2d, w_all = ctx.saved_tensors
        widths, padded = ctx.widths, ctx.padded
        total = sum(widths)
        parts = [dy.to(x2d.dtype) if dy is not None else torch.zeros((x2d.shape[0], n), dtype=x2d.dtype, device=x2d.device) for dy, n in zip(dys, widths)]
        if padded > total:
            parts.append(torch.zeros((x2d.shape[0], padded - total), dtype=x2d.dtype, device=x2d.device))
        dy_all = torch.cat(parts, 1)
        dx = gemm_nt(dy_all, w_all.t().contiguous()) if ctx.needs_input_grad[0] else None
        grads = [dx]
        need_w = any(ctx.needs_input_grad[1 + 2 * i] for i in range(len(widths)))
        need_b = any(ctx.needs_input_grad[2 + 2 * i] and ctx.meta[i][2] for i in range(len(widths)))
        dw_all = db_all = None
        if need_w or need_b:
            dw_all, db_all = wgrad(dy_all, x2d, colsum=True)
        o = 0
        for i, n in enumerate(widths):
            wdt, wshape, has_b, bdt = ctx.meta[i]
            dw = dw_all[o:o + n].reshape(wshape).to(wdt) if ctx.needs_input_grad[1 + 2 * i] else None
            db = db_all[o:o + n].to(bdt) if (has_b and ctx.needs_input_grad[2 + 2 * i]) else None
            grads += [dw, db]
            o += n
        return tuple(grads)


def token_projections(x2d: torch.Tensor, layers) -> Tuple[torch.Tensor, ...]:
    """[conv(x) for conv in layers] for 1x1 convolutions (or Linear layers) applied to token rows x2d [rows, Cin]: one stacked product
    forward, two backward (_TokenProjections).  Each result is a contiguous [rows, out_channels] tensor in x2d's dtype."""
    args = []
    for m in layers:
        args += [m.weight, m.bias]
    return _TokenProjections.apply(x2d, *args)


class _SegLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, target, H, W, dice_weight, eps):
        lib = load()
        if z.dim() != 4 or not z.is_cuda or target.dim() != 3 or target.shape[0] != z.shape[0]:
            raise GdkvmError("seg_loss: logits [images,C,h,w] and labels [images,H,W] on the device (no CPU path)")
        if target.dtype not in (torch.int64, torch.uint8) or tuple(target.shape[1:]) != (H, W):
            raise GdkvmError("seg_loss: labels must be int64 or uint8 [images,H,W]")
        z, target = z.contiguous(), target.contiguous()
        ni, c, h, w = z.shape
        out = torch.empty(3, dtype=torch.float32, device=z.device)
        ws = torch.empty(int(lib.gdkvm_seg_loss_workspace_bytes(c)), dtype=torch.uint8, device=z.device)
        tb = 8 if target.dtype == torch.int64 else 1
        with torch.cuda.device(z.device):
            rc = lib.gdkvm_seg_loss_fwd(z.data_ptr(), target.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), ni, c, h, w, H, W,
                                        float(dice_weight), float(eps), _io_dtype(z), tb, _stream(z.device))
        _check(rc, "gdkvm_seg_loss_fwd")
        ctx.save_for_backward(z, target, ws)
        ctx.dims = (ni, c, h, w, H, W, tb)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        lib = load()
        z, target, ws = ctx.saved_tensors
        ni, c, h, w, H, W, tb = ctx.dims
        g = g.to(torch.float32).reshape(1).contiguous()
        dz = torch.empty_like(z)
        with torch.cuda.device(z.device):
            rc = lib.gdkvm_seg_loss_bwd(z.data_ptr(), target.data_ptr(), ws.data_ptr(), ws.numel(), g.data_ptr(), dz.data_ptr(),
                                        ni, c, h, w, H, W, _io_dtype(z), tb, _stream(z.device))
        _check(rc, "gdkvm_seg_loss_bwd")
        return dz, None, None, None, None, None


def seg_loss(lowres_logits: torch.Tensor, target: torch.Tensor, dice_weight: float = 1.0, eps: float = 1.0) -> torch.Tensor:
    """Cross-entropy + soft Dice of bilinear_upsample(lowres_logits -> target's H x W) against integer labels, without the
    full-resolution logits (gdkvm_seg_loss_fwd / _bwd).  lowres_logits [images,C,h,w], target [images,H,W]; scalar loss."""
    return _SegLossFunction.apply(lowres_logits, target, target.shape[-2], target.shape[-1], dice_weight, eps)


def dice_from_counts(counts: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """Dice_c = (2|AnB| + eps) / (|A| + |B| + eps) from the integer counts of argmax_dice."""
    c = counts.to(torch.float64)
    return (2.0 * c[..., 0] + eps) / (c[..., 1] + c[..., 2] + eps)
