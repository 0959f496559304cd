"""Tensor-level wrappers over the C ABI (include/onet_hip.h).

PyTorch-ROCm tensors are used for STORAGE ONLY: every function here takes
tensors, checks them, and hands raw device pointers + sizes + the current HIP
stream to libonet_hip.so.  No torch compute op is called on the data path.
There is no CPU fallback: CPU tensors raise."""
from __future__ import annotations

import torch

from . import _lib

F32 = torch.float32


# ----------------------------------------------------------------------------- per-model settings
import os as _os


def _flags_from_env():
    """ONET_FLAGS="NAME=VALUE,NAME=VALUE": diagnostic overrides of the module switches below (the values a fresh process starts with;
    tests and tools may also set the attributes directly).  The product has three environment switches of its own -- ONET_HIP_LIB
    (another build of the library), ONET_CONV_ALGO (the default algorithm) and this one; models carry their switches in `Settings`."""
    out = {}
    for item in _os.environ.get("ONET_FLAGS", "").split(","):
        if "=" in item:
            k, v = item.split("=", 1)
            out[k.strip()] = v.strip()
    return out


_FLAGS = _flags_from_env()


def _flag(name, default):
    """Process default of switch `name`: ONET_FLAGS' value if given (bool switches: "0" = off; ints and strings as written)."""
    v = _FLAGS.get(name)
    if v is None:
        return default
    if isinstance(default, bool):
        return v != "0"
    if isinstance(default, int):
        return int(v)
    return v


class Settings:
    """Switches a MODEL carries (`Onet.settings`); a field left at None falls back to the process default (the module
    attribute of the same purpose below, initialised from the environment).  The active record is thread-local: `Onet.forward`
    / `compute_loss` activate their model's record for the calls they make, and every autograd Function of
    onet_amd.functional captures the record active in its forward and re-activates it in its backward (which autograd runs
    on another thread, possibly after another model has run) -- so two models with different settings in one process, or on
    two streams / threads, never see each other's.

      conv          3x3 convolution algorithm: "auto" | "split" | "winograd4" | "direct" | "bf16"         (default CONV_ALGO)
      twin          weight-shared Onet: X and 1-X as ONE batch of 2B                                        (default TWIN)
      convt_bf16    under conv == "bf16": the ConvTranspose2d GEMMs take bf16 operands too                  (default CONVT_BF16)
      lazy_nan      OV:234's NaN assertion deferred to FlatAdam.step()                                      (default LAZY_NAN_CHECK)
      split         under conv == "auto": fp32 3x3 convolutions on the bf16 matrix pipe by operand splitting   (default SPLIT_AUTO)
      bn_on_load    the second convolution of a DoubleConv applies the first unit's BatchNorm + ReLU on load     (default BN_ON_LOAD)
      split_f16     split forward convolution on fp16 parts (22-bit operands); False: bf16 parts (16-bit)         (default SPLIT_F16)
      grad_f16      split input / weight gradients on fp16 parts of power-of-two-scaled operands; False: bf16      (default SPLIT_GRAD_F16)
      split_dgrad   input gradients may take the split kernels (False: diagnostic, fp32-MFMA kernels)             (default SPLIT_DGRAD)
      stem_fused    the stem convolution + its BatchNorm statistics in one streaming pass                          (default STEM_FUSED)
      sync_bn       BatchNorm statistics all-gathered over the process group                                       (default SYNC_BN)
      presplit      the split kernels' operands are written pre-split (fp16 hi | mid slots) by their producers     (default PRESPLIT)
      z_bf16        under conv == "bf16" with pre-split operands: the convolution outputs z are STORED as bf16          (default Z_BF16)"""
    __slots__ = ("conv", "twin", "convt_bf16", "lazy_nan", "split", "bn_on_load", "split_f16", "grad_f16", "split_dgrad",
                 "stem_fused", "sync_bn", "presplit", "z_bf16")

    def __init__(self, conv=None, twin=None, convt_bf16=None, lazy_nan=None, split=None, bn_on_load=None,
                 split_f16=None, grad_f16=None, split_dgrad=None, stem_fused=None, sync_bn=None, presplit=None, z_bf16=None):
        self.presplit = presplit
        self.z_bf16 = z_bf16
        self.conv, self.twin, self.convt_bf16, self.lazy_nan = conv, twin, convt_bf16, lazy_nan
        self.split, self.bn_on_load = split, bn_on_load
        self.split_f16, self.grad_f16, self.split_dgrad, self.stem_fused, self.sync_bn = split_f16, grad_f16, split_dgrad, stem_fused, sync_bn

    def replace(self, **kw):
        out = Settings(*(getattr(self, k) for k in self.__slots__))
        for k, v in kw.items():
            setattr(out, k, v)
        return out

    def __repr__(self):
        return "Settings(%s)" % ", ".join("%s=%r" % (k, getattr(self, k)) for k in self.__slots__)


import threading as _threading
_TLS = _threading.local()


def active_settings():
    return getattr(_TLS, "cur", None)


class using:
    """`with ops.using(settings):` -- make `settings` the active record of this thread (None: leave the current one)."""

    def __init__(self, settings):
        self.settings = settings

    def __enter__(self):
        self.prev = getattr(_TLS, "cur", None)
        if self.settings is not None:
            _TLS.cur = self.settings
        return self.settings

    def __exit__(self, *exc):
        _TLS.cur = self.prev
        return False


def _setting(field, default):
    cur = getattr(_TLS, "cur", None)
    v = None if cur is None else getattr(cur, field)
    return default if v is None else v


def conv_algo():
    return _setting("conv", CONV_ALGO)


def twin_enabled():
    return bool(_setting("twin", TWIN))


def lazy_nan_check():
    return bool(_setting("lazy_nan", LAZY_NAN_CHECK))


def split_f16():
    return bool(_setting("split_f16", SPLIT_F16))


def grad_f16():
    return bool(_setting("grad_f16", SPLIT_GRAD_F16))


def split_dgrad():
    return bool(_setting("split_dgrad", SPLIT_DGRAD))


def stem_fused():
    return bool(_setting("stem_fused", STEM_FUSED))


def sync_bn():
    return bool(_setting("sync_bn", SYNC_BN))


def presplit():
    """Pre-split storage (round 4): activations and BatchNorm-backward gradients whose consumers are the split convolution kernels
    are written by their producers as fp16 (hi, mid) parts in the kernels' slot layout; the MFMA kernels' staging is an LDS-DMA copy."""
    if not bool(_setting("presplit", PRESPLIT)) or sync_bn() or not (FUSE_BN_STATS and FUSE_BN_REDUCE and FUSE_POOL):
        return False
    if conv_algo() == "bf16":           # BASELINE configs[2]: the same machinery with ONE part of plain bf16 operands (p16_parts() == 1)
        return True
    return split_enabled() and conv_algo() in ("auto", "split") and split_f16() and split_dgrad()


Z_BF16 = _flag("Z_BF16", True)       # BASELINE configs[2] (conv == "bf16", pre-split operands): conv outputs stored as bf16 (Settings.z_bf16)


def z16_storage():
    """Round 5: under conv == "bf16" with pre-split (plain bf16) operands the 3x3 convolutions STORE their output z as bf16 -- rounded
    once, to nearest even, in the epilogue; the BatchNorm statistics still come from the fp32 accumulators -- and every pass that
    reads z (normalise + ReLU, both backward passes, the pooling backward) moves half the bytes.  z is an internal tensor: nothing
    the reference's callers can see changes type."""
    return conv_algo() == "bf16" and bool(_setting("z_bf16", Z_BF16)) and presplit() and p16_parts() == 1


def p16_parts():
    """Parts per pre-split operand: 2 = fp16 (hi | mid) of the split kernels (fp32-level results), 1 = plain bf16 (conv == "bf16")."""
    return 1 if conv_algo() == "bf16" else 2


def split_enabled():
    return bool(_setting("split", SPLIT_AUTO))


def convt_operand_bf16(B, h, w, Ct):
    """operand_bf16 argument of the onet_convT2x2_* entry points for the calling model and layer: 1 = bf16 operands (the bf16
    conv path), 2 = split bf16 operands (fp32-level results; with the split 3x3 kernels under "auto", on layers whose forward
    GEMM has at least CONVT_SPLIT_MIN_BLOCKS 128 x 128 tiles -- half the CUs by default: like conv3x3_algo, small problems
    keep the fp32 kernels, where there is nothing to gain), 0 = fp32 MFMA"""
    algo = conv_algo()
    if algo == "bf16":
        return int(bool(_setting("convt_bf16", CONVT_BF16)))
    if not (algo in ("auto", "split") and split_enabled() and CONVT_SPLIT):
        return 0
    min_blocks = CONVT_SPLIT_MIN_BLOCKS if CONVT_SPLIT_MIN_BLOCKS is not None else n_cu() // 2
    return 2 if (B * h * w // 128) * (4 * Ct // 128) >= min_blocks else 0


_CU_COUNT = {}


def n_cu(device=None):
    """compute units of the current device (onet_device_info; 256 on MI355X)"""
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    n = _CU_COUNT.get(idx)
    if n is None:
        import ctypes
        cu, lds, wave = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        arch = ctypes.create_string_buffer(64)
        rc = _lib.load().onet_device_info(ctypes.byref(cu), ctypes.byref(lds), ctypes.byref(wave), arch, 64)
        n = int(cu.value) if rc == 0 and cu.value > 0 else 256
        _CU_COUNT[idx] = n
    return n


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def require_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("onet_amd: the HIP path needs tensors on a ROCm GPU (got a CPU tensor); "
                               "there is no CPU fallback -- move the module and inputs to 'cuda'")
        if t is not None and t.dtype != F32 and t.is_floating_point():
            raise TypeError(f"onet_amd: fp32 tensors expected, got {t.dtype}")


def plane(t):
    """-> (tensor, batch_stride) with the (C,H,W) block of every image contiguous.
    Channel-slices of a concat buffer qualify as they are; anything else is copied."""
    B, C, H, W = t.shape
    if is_placeholder(t) and getattr(t, "_onet_twin_src", None) is not None:
        t = twin_materialize(t)          # a reader other than the stem kernels: the twin batch as a real tensor
    if is_placeholder(t):
        raise RuntimeError("onet_amd: an fp32 placeholder (bf16 storage: tensor kept in bf16 only) reached a kernel that reads fp32")
    ok = (W == 1 or t.stride(3) == 1) and (H == 1 or t.stride(2) == W) and (C == 1 or t.stride(1) == H * W)
    if ok and B > 1 and t.stride(0) < C * H * W:
        ok = False
    if not ok:
        t = t.contiguous()
    return t, (t.stride(0) if B > 1 else C * H * W)


# Optional in-process kernel timing (bench.py): PROFILE = {} enables HIP-event brackets around the
# MFMA launches, on the stream they are launched on; entries: kind -> [(flops, ev_start, ev_end)].
PROFILE = None
# PROFILE_ALL = {} additionally brackets EVERY other launch of the library (streaming kernels: BatchNorm, pooling, head,
# loss, Adam, packs): entry point -> [(algorithmic HBM bytes | None, ev_start, ev_end)].
PROFILE_ALL = None
_IN_MFMA_BRACKET = False


class _AllHook:
    """_lib.PROFILE_HOOK while PROFILE_ALL is collected; launches inside an MFMA bracket are already timed."""

    @staticmethod
    def begin():
        if PROFILE_ALL is None or _IN_MFMA_BRACKET:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return e0

    @staticmethod
    def end(name, nbytes, e0):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROFILE_ALL.setdefault(name, []).append((nbytes, e0, e1))


PROFILE_ONLY = None


def profile_start(everything=True, only=None):
    """everything: bracket every launch of the library (MFMA kernels -> PROFILE, all others -> PROFILE_ALL); else the MFMA kernels
    only -- or, with `only` = a set of kinds, those kinds alone (bench.py's timed region: the dominant kernel)."""
    global PROFILE, PROFILE_ALL, PROFILE_ONLY
    PROFILE = {}
    PROFILE_ALL = {} if everything else None
    PROFILE_ONLY = None if (everything or only is None) else frozenset(only)
    _lib.PROFILE_HOOK = _AllHook if everything else None


def profile_stop():
    """-> (mfma records, all-other records); timing is off afterwards."""
    global PROFILE, PROFILE_ALL
    out = (PROFILE, PROFILE_ALL or {})
    PROFILE = PROFILE_ALL = None
    _lib.PROFILE_HOOK = None
    return out


def _prof_begin(kind=None):
    global _IN_MFMA_BRACKET
    if PROFILE is None or (PROFILE_ONLY is not None and kind not in PROFILE_ONLY):
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    _IN_MFMA_BRACKET = True
    return e0


def _prof_end(kind, flops, e0, nbytes=0.0):
    """flops: direct-algorithm FLOPs of the launch; nbytes: its compulsory HBM bytes (operands read once + result written
    once, fp32)."""
    global _IN_MFMA_BRACKET
    if e0 is None:
        return
    _IN_MFMA_BRACKET = False
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    PROFILE.setdefault(kind, []).append((flops, e0, e1, nbytes))


# OV:234 asserts "jsd is not NaN" inside the loss, which costs a device synchronisation in the MIDDLE of a step (forward
# drained, backward not yet launched: ~1 ms of idle GPU per step at B=32).  A training loop that owns the optimizer
# step (trainer.FlatAdam / trainer.fit, bench.py) may defer the verdict: the flag is copied to pinned host memory behind an
# event, and `check_deferred_nan()` -- called by FlatAdam.step() BEFORE the update is applied, and by the next
# compute_loss -- raises the same AssertionError without draining the queue.  Default: strict (assert in place).
import os as _os

LAZY_NAN_CHECK = _flag("LAZY_NAN_CHECK", False)
_NAN_PENDING = []
_NAN_POOL = []


def defer_nan_check(flag, what="jsd is NaN"):
    host = _NAN_POOL.pop() if _NAN_POOL else torch.empty(1, dtype=torch.bool, pin_memory=True)
    host.copy_(flag.reshape(1), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _NAN_PENDING.append((host, ev, what))


def check_deferred_nan():
    while _NAN_PENDING:
        host, ev, what = _NAN_PENDING.pop(0)
        ev.synchronize()
        bad = bool(host.item())
        _NAN_POOL.append(host)
        assert not bad, what


_WS = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer per device (wgrad split-K slabs)."""
    key = (device.type, device.index)
    cur = _WS.get(key)
    if cur is None or cur.numel() * 4 < nbytes:
        cur = None
        _WS.pop(key, None)
        cur = torch.empty((int(nbytes) + 3) // 4, dtype=F32, device=device)
        _WS[key] = cur
    return cur


# ----------------------------------------------------------------------------- conv
def pack3x3(w):
    require_gpu(w)
    w = w.detach().contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    wf = torch.empty(Cin * 9 * Cout, dtype=F32, device=w.device)
    wd = torch.empty(Cout * 9 * Cin, dtype=F32, device=w.device)
    _lib.call("onet_conv3x3_pack_weights", _p(w), _p(wf), _p(wd), Cout, Cin, _stream())
    return wf, wd


def packT2x2(w):
    require_gpu(w)
    w = w.detach().contiguous()
    Cin, Cout = w.shape[0], w.shape[1]
    wf = torch.empty(Cin * 4 * Cout, dtype=F32, device=w.device)
    wd = torch.empty(4 * Cout * Cin, dtype=F32, device=w.device)
    _lib.call("onet_convT2x2_pack_weights", _p(w), _p(wf), _p(wd), Cin, Cout, _stream())
    return wf, wd


def packT2x2_fused(w):
    """[Cin][4*Ct] with column c*4 + sub-pixel: the layout of convT2x2_fwd's shuffle epilogue."""
    require_gpu(w)
    w = w.detach().contiguous()
    Cin, Cout = w.shape[0], w.shape[1]
    wq = torch.empty(Cin * 4 * Cout, dtype=F32, device=w.device)
    _lib.call("onet_convT2x2_pack_weights_fused", _p(w), _p(wq), Cin, Cout, _stream())
    return wq


# BASELINE configs[2]: under the bf16 conv path the ConvTranspose2d GEMMs take bf16 MFMA operands too (what torch.autocast does to
# nn.ConvTranspose2d); CONVT_BF16=0 (ONET_FLAGS) keeps them fp32.  Passed to the library per call (`operand_bf16`).
CONVT_BF16 = _flag("CONVT_BF16", True)


def convT2x2_fwd(x, wq, bias, out, Ct, pt, pl):
    """out[:, c, pt + 2i + di, pl + 2j + dj] = sum_ci x[:, ci, i, j] * W[ci, c, di, dj] + bias[c]: ConvTranspose2d(k=2, s=2)
    written straight into `out`, a plane-contiguous [B, Ct, Ho, Wo] view (e.g. the second half of a concat buffer)."""
    require_gpu(x, wq, out)
    x, xbs = plane(x)
    B, Cin, h, w = x.shape
    Ho, Wo = out.shape[2], out.shape[3]
    obs = out.stride(0) if B > 1 else Ct * Ho * Wo
    e0 = _prof_begin("convt_gemm_kernel")
    _lib.call("onet_convT2x2_fwd", _p(x), xbs, _p(wq), _p(bias), _p(out), obs, B, Cin, Ct, h, w, Ho, Wo, pt, pl,
              convt_operand_bf16(B, h, w, Ct), _stream())
    _prof_end("convt_gemm_kernel", 2.0 * B * h * w * Cin * 4 * Ct, e0, 4.0 * (B * h * w * (Cin + 4 * Ct) + 4 * Cin * Ct))


def convT2x2_out_bound(weight, bias, x_amax):
    """Magnitude slots bounding |ConvTranspose2d(x) + bias| from the layer's weights and the exact max |x| its producer recorded."""
    w = weight.detach()
    w = w if w.is_contiguous() else w.contiguous()
    slots = new_amax(w.device)
    _lib.call("onet_convT2x2_out_bound", _p(w), _p(bias), w.shape[0], w.shape[1], _p(x_amax), _p(slots), _stream())
    return slots


def packT2x2_slots(w, parts=2, dgrad=False):
    """nn.ConvTranspose2d weight [Cin, Ct, 2, 2] -> the slot pack of the slot-operand forward GEMM: [Cin/8, parts, 4 Ct, 8] fp16 (hi | mid)
    parts of 2^k w (+ (2^k, 2^-k) behind the pack), or one part of bf16(w).  dgrad: -> (that, the K-slot pack of the input gradient
    [(Ct/8) 4, parts, Cin, 8]) from the same launch."""
    require_gpu(w)
    w = w.detach().contiguous()
    Cin, Ct = w.shape[0], w.shape[1]
    dt = torch.float16 if parts == 2 else BF
    wP = torch.empty(Cin * 4 * Ct * parts + 4, dtype=dt, device=w.device)
    wdP = torch.empty(Cin * 4 * Ct * parts + 4, dtype=dt, device=w.device) if dgrad else None
    ws = torch.empty(2048, dtype=torch.int32, device=w.device) if parts == 2 else None
    _lib.call("onet_convT2x2_pack_weights_slots", _p(w), _p(wP), _p(wdP), _p(ws), Cin, Ct, parts, _stream())
    return (wP, wdP) if dgrad else wP


def convT2x2_dgrad_slots(dyP, wdP, Cin, dy_slots=None):
    """dx1 [B, Cin, h, w] fp32 = input gradient of ConvTranspose2d(k=2, s=2) from the pre-split up-sampled gradient dyP [B, Ct/8, 2h, parts,
    2w, 8] and the K-slot pack wdP; None where the kernel does not take the shape."""
    B, C8, Ho, parts, Wo, _ = dyP.shape
    Ct, h, w = C8 * 8, Ho // 2, Wo // 2
    dx = torch.empty((B, Cin, h, w), dtype=F32, device=dyP.device)
    e0 = _prof_begin("convt_gemm_kernel")
    rc = _lib.load().onet_convT2x2_dgrad_slots(_p(dyP), _pbs(dyP), _p(dy_slots), _p(wdP), _p(dx), Cin * h * w, parts, B, Cin, Ct, h, w, _stream())
    flops, nb = 2.0 * B * h * w * Cin * 4 * Ct, B * h * w * (4.0 * Cin + 2.0 * parts * 4 * Ct)
    _prof_end("convt_gemm_kernel", flops if rc == 0 else 0.0, e0, nb if rc == 0 else 0.0)
    if rc < 0:
        raise _lib.OnetHipError(f"onet_convT2x2_dgrad_slots failed ({rc}): {_lib.last_error()}")
    return dx if rc == 0 else None


def convT2x2_wgrad_slots(xP, dyP, dw_shape, x_slots=None, dy_slots=None, want_dbias=False, out=None, db_out=None):
    """(dw [Cin, Ct, 2, 2], dbias | None) from the pre-split input xP and the pre-split up-sampled gradient dyP; None where the kernel does
    not take the shape."""
    B, C8, h, parts, w, _ = xP.shape
    Cin, Ct = C8 * 8, dyP.shape[1] * 8
    assert tuple(dw_shape) == (Cin, Ct, 2, 2) and dyP.shape[3] == parts and dyP.shape[2] == 2 * h and xP.dtype == dyP.dtype
    need = int(_lib.load().onet_convT2x2_wgrad_slots_ws_bytes(B, Cin, Ct, h, w))
    if need <= 0:
        return None
    ws = workspace(need, xP.device)
    dw = out if out is not None else torch.empty(dw_shape, dtype=F32, device=xP.device)
    db = (db_out if db_out is not None else torch.empty(Ct, dtype=F32, device=xP.device)) if want_dbias else None
    e0 = _prof_begin("convt_wgrad_gemm_kernel")
    rc = _lib.load().onet_convT2x2_wgrad_slots(_p(xP), _pbs(xP), _p(x_slots), _p(dyP), _pbs(dyP), _p(dy_slots), _p(dw), _p(db), _p(ws),
                                               ws.numel() * 4, parts, B, Cin, Ct, h, w, _stream())
    flops, nb = 2.0 * B * h * w * Cin * 4 * Ct, 2.0 * parts * B * h * w * (Cin + 4 * Ct)
    _prof_end("convt_wgrad_gemm_kernel", flops if rc == 0 else 0.0, e0, nb if rc == 0 else 0.0)
    if rc < 0:
        raise _lib.OnetHipError(f"onet_convT2x2_wgrad_slots failed ({rc}): {_lib.last_error()}")
    return (dw, db) if rc == 0 else None


def conv3x3_dgrad_bound(weight, dz_slots, ci0):
    """Magnitude slots bounding the channels >= ci0 of a 3x3 convolution's input gradient: max |dz| (dz_slots) x the largest L1 norm of
    the weights feeding one input channel."""
    w = weight.detach()
    w = w if w.is_contiguous() else w.contiguous()
    slots = new_amax(w.device)
    _lib.call("onet_conv3x3_dgrad_bound", _p(w), w.shape[0], w.shape[1], int(ci0), _p(dz_slots), _p(slots), _stream())
    return slots


def conv3x3_split_dgrad_pre_slots(dzP, wq, Cout, ch0, daP_slots, slots=None, always=False):
    """Input gradient of a decoder block's first convolution from pre-split dz: -> (da [B, Cout, H, W] fp32 whose channels >= ch0 are NOT
    written, daP [B, (Cout - ch0)/8, H, parts, W, 8] = those channels pre-split: fp16 (hi | mid) parts scaled by daP_slots' `always` rule,
    or -- plain bf16 operands -- one part of bf16(da))."""
    B, C8, H, parts, W, _ = dzP.shape
    assert wq.dtype == dzP.dtype == (torch.float16 if parts == 2 else BF)
    da = torch.empty((B, Cout, H, W), dtype=F32, device=dzP.device)
    daP = torch.empty((B, (Cout - ch0) // 8, H, parts, W, 8), dtype=dzP.dtype, device=dzP.device)
    e0 = _prof_begin("conv3x3_split_pre_kernel")
    _lib.call("onet_conv3x3_split_dgrad_pre_slots", _p(dzP), _pbs(dzP), _p(slots), int(always), _p(wq), 2 if parts == 1 else 1, _p(da), Cout * H * W,
              _p(daP), _pbs(daP), int(ch0), _p(daP_slots), B, C8 * 8, Cout, H, W, _stream())
    eb = 2.0 * parts
    _prof_end("conv3x3_split_pre_kernel", 2.0 * B * H * W * C8 * 8 * Cout * 9, e0,
              B * H * W * (eb * C8 * 8 + 4.0 * ch0 + eb * (Cout - ch0)) + eb * 9 * C8 * 8 * Cout)
    return da, daP


CONVT_SLOTS = _flag("CONVT_SLOTS", True)     # 0: the ConvTranspose2d forward reads the fp32 activation (round 4's kernels)


def convt_slots_ok(B, Cin, Ct, h, w):
    """Does the slot-operand ConvTranspose2d forward take this layer (onet_convT2x2_fwd_slots' predicate)?  The block below then writes
    its output pre-split for it (UNet._forward)."""
    return bool(CONVT_SLOTS and presplit() and (p16_parts() == 2 or CONVT_BF16) and Cin % 32 == 0 and Cin >= 128 and Ct % 32 == 0 and (h * w) % 128 == 0 and w % 2 == 0
                and Cin * h * w * 2 * p16_parts() < 2 ** 31)


CONVT_BWD_SLOTS = _flag("CONVT_BWD_SLOTS", True)     # 0: the ConvTranspose2d backward GEMMs read fp32 operands (round 4's kernels)


def convt_bwd_slots_ok(B, Cin, Ct, C2, h, w):
    """Do the slot-operand ConvTranspose2d BACKWARD GEMMs take this layer -- and the input gradient of the decoder block's first
    convolution (C2 skip + Ct up-sampled channels on a 2h x 2w map) the slot-writing form that feeds them?"""
    return bool(CONVT_BWD_SLOTS and convt_slots_ok(B, Cin, Ct, h, w) and Cin % 128 == 0 and (w & (w - 1)) == 0
                and C2 % 64 == 0 and (C2 + Ct) % 64 == 0 and (2 * h) % 16 == 0 and (2 * w) % 32 == 0 and Ct * 4 * h * w * 4 < 2 ** 31)


class _LazyPack:
    """One of PackedT's two classic packs, packed when something actually touches the tensor (get())."""

    def __init__(self, owner, i):
        self.owner, self.i = owner, i

    def get(self):
        return self.owner[self.i]


def pack_of(p):
    """The tensor behind a pack handed through autograd as a non-tensor argument: a tensor, or a lazy handle of PackedT."""
    return p.get() if isinstance(p, _LazyPack) else p


class PackedT:
    """Packed forms of one ConvTranspose2d weight: (fused forward pack, input-gradient pack) -- unpacks like that tuple -- plus the slot
    packs of the slot-operand forward, built on first use per number of parts."""

    def __init__(self, w):
        self.w = w.detach()
        self._t = [None, None]          # built on first use: a step on the slot-operand forward never asks for the fused fp32 pack
        self._s = {}

    def __getitem__(self, i):
        if self._t[i] is None:
            self._t[i] = packT2x2_fused(self.w) if i == 0 else packT2x2(self.w)[1]
        return self._t[i]

    def __iter__(self):
        return iter((_LazyPack(self, 0), _LazyPack(self, 1)))

    def __len__(self):
        return 2

    def slots(self, parts):
        if parts not in self._s:       # (the backward GEMMs' pack comes out of the same launch -- a training step wants both)
            self._s[parts] = packT2x2_slots(self.w, parts, dgrad=bool(CONVT_BWD_SLOTS))
        v = self._s[parts]
        return v[0] if isinstance(v, tuple) else v

    def dgrad_slots(self, parts):
        v = self._s.get(parts)
        if not isinstance(v, tuple):
            self._s[parts] = v = packT2x2_slots(self.w, parts, dgrad=True)
        return v[1]


def convT2x2_fwd_slots(xP, wP, bias, outP, Ct, x_slots=None, slots=None):
    """ConvTranspose2d(k=2, s=2) + bias from the PRE-SPLIT input xP [B, Cin/8, h, parts, w, 8] and the slot pack wP (packT2x2_slots),
    written pre-split into outP [B, Ct/8, 2h, parts, 2w, 8]: both GEMM operands are LDS-DMA copies.  x_slots / slots: the magnitude slots
    of the input / output.  -> False where the kernel does not take the shape (nothing written)."""
    B, C8, h, parts, w, _ = xP.shape
    assert outP.shape[3] == parts and outP.shape[2] == 2 * h and outP.shape[4] == 2 * w and xP.dtype == wP.dtype == outP.dtype
    Cin = C8 * 8
    e0 = _prof_begin("convt_gemm_kernel")
    rc = _lib.load().onet_convT2x2_fwd_slots(_p(xP), _pbs(xP), _p(x_slots), _p(wP), _p(bias), _p(outP), _pbs(outP), _p(slots), parts, B, Cin, Ct,
                                             h, w, _stream())
    eb = 2.0 * parts
    flops, nb = 2.0 * B * h * w * Cin * 4 * Ct, eb * (B * h * w * (Cin + 4 * Ct) + 4 * Cin * Ct)
    _prof_end("convt_gemm_kernel", flops if rc == 0 else 0.0, e0, nb if rc == 0 else 0.0)
    if rc < 0:
        raise _lib.OnetHipError(f"onet_convT2x2_fwd_slots failed ({rc}): {_lib.last_error()}")
    return rc == 0


def convT2x2_fwd_p(x, wq, bias, outP, Ct, pt, pl, slots=None):
    """ConvTranspose2d(k=2, s=2) + bias written PRE-SPLIT into outP [B, Ct/8, Ho, 2, Wo, 8] (the up-sampled channel groups of a pre-split
    concat buffer): no fp32 tensor.  -> False where the GEMM fast path does not take the shape (nothing written)."""
    require_gpu(x, wq)
    x, xbs = plane(x)
    B, Cin, h, w = x.shape
    Ho, Wo = outP.shape[2], outP.shape[4]
    e0 = _prof_begin("convt_gemm_kernel")
    rc = _lib.load().onet_convT2x2_fwd_p(_p(x), xbs, _p(wq), _p(bias), _p(outP), _pbs(outP), _p(slots), outP.shape[3], B, Cin, Ct, h, w, Ho, Wo, pt, pl,
                                         convt_operand_bf16(B, h, w, Ct), _stream())
    flops, nb = 2.0 * B * h * w * Cin * 4 * Ct, 4.0 * (B * h * w * (Cin + 4 * Ct) + 4 * Cin * Ct)
    _prof_end("convt_gemm_kernel", flops if rc == 0 else 0.0, e0, nb if rc == 0 else 0.0)
    if rc < 0:
        raise _lib.OnetHipError(f"onet_convT2x2_fwd_p failed ({rc}): {_lib.last_error()}")
    return rc == 0


# 3x3 convolution algorithm for fwd/dgrad, chosen per call from the layer shape (ONET_CONV_ALGO overrides):
#   "auto" (default)  the split kernels (fp32-level results on the 16-bit matrix cores, conv_split.hip) on maps at least 32 pixels wide
#                     with Cin % 16 == 0 and enough tiles to fill the chip (pre-split operands: also the 16-pixel level at batches that
#                     fill it, pre_layer_ok); else Winograd F(4x4,3x3) where its 64-channel x 32-tile blocks fill the chip, else the
#                     direct implicit-GEMM kernel
#   "split"           the split kernels on every legal layer (parity tests)
#   "winograd4"       F(4x4,3x3) on every legal layer with maps >= 8x8 (parity tests; the fp32-MFMA reference dispatch of bench.py)
#   "direct"          implicit GEMM only
#   "bf16"            BASELINE configs[2]: ONE part of plain bf16 operands through the pre-split kernels (forward, input gradient,
#                     weight gradient) on every layer pre_layer_ok takes -- 32-channel chunks, maps made of full 16 x 32 tiles; "auto"
#                     (fp32 results) elsewhere (round 5 dropped the round-2 bf16 kernels that served the other shapes)
# Layers no Winograd kernel takes (stem Cin < 16, channel counts not multiples of 4) always run direct.
import os as _os
CONV_ALGO = _os.environ.get("ONET_CONV_ALGO", "auto")
# Weight-shared Onet (dwnu is topu): run the X and the 1-X pass as ONE batch of 2B through every convolution
# (BatchNorm keeps the two halves as separate statistics groups).  TWIN=0 (ONET_FLAGS) runs the two passes one after the
# other, as the reference does.
TWIN = _flag("TWIN", True)


BF = torch.bfloat16


_SENTINEL = {}


def _sentinel(device):
    """The 4-byte storage every fp32 placeholder of `device` points at: what identifies one (an expanded gradient or a user's
    `.expand()`ed input has zero strides too, but its own storage -- those are ordinary tensors and are copied by `plane`)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    s = _SENTINEL.get(key)
    if s is None:
        s = _SENTINEL[key] = torch.zeros(1, dtype=F32, device=device)
    return s


def is_placeholder(t):
    return bool(_SENTINEL) and t.is_cuda and t.dtype == F32 and \
        t.untyped_storage().data_ptr() == _sentinel(t.device).untyped_storage().data_ptr()


def fp32_placeholder(shape, device):
    """A zero-stride fp32 tensor of the given shape over the device's sentinel storage (4 bytes): what autograd passes around
    in place of an activation that exists only in its consumers' operand form (pre-split slots).  Never read by a kernel (`plane` raises).
    Not a view: it has its own version counter, which is what validates the bf16 copy riding on it."""
    t = torch.empty(0, dtype=F32, device=device)
    t.set_(_sentinel(t.device).untyped_storage(), 0, tuple(shape), (0,) * len(shape))
    return t


TWIN_VIRTUAL = _flag("TWIN_VIRTUAL", True)      # False (tests): TwinInputFn materialises the twin batch, the stem kernels read it like any batch


def twin_virtual(x, bias):
    """The twin batch [X ; clip(1 - X + bias, 0, 1)] (OV:178-180 as one batch of 2B) WITHOUT its tensor: a placeholder of that shape
    carrying (X, bias).  The stem kernels (forward + statistics, weight gradient) form the complement half while they load X
    (clip(1 - X + bias) fused into the stem's load); any other reader gets the real tensor from `plane`, which
    materialises it once (twin_materialize)."""
    t = fp32_placeholder((2 * x.shape[0],) + tuple(x.shape[1:]), x.device)
    t._onet_twin_src = [x, float(bias), None]
    return t


def twin_src_of(t):
    """(X, bias) if `t` is a virtual twin batch whose tensor does not exist, else None."""
    tag = getattr(t, "_onet_twin_src", None)
    return None if tag is None else (tag[0], tag[1])


def twin_materialize(t=None, src=None):
    """The real [2B, C, H, W] tensor of a virtual twin batch (made once per forward, cached on the tag)."""
    tag = getattr(t, "_onet_twin_src", None) if t is not None else None
    if tag is not None and tag[2] is not None:
        return tag[2]
    x, bias = (tag[0], tag[1]) if tag is not None else src
    B = x.shape[0]
    xx = torch.empty((2 * B,) + tuple(x.shape[1:]), dtype=F32, device=x.device)
    xx[:B].copy_(x)
    complement_clip(x, bias, out=xx[B:])
    if tag is not None:
        tag[2] = xx
    return xx


def _z16(z):
    """1 if the convolution output z is stored as bf16 (BASELINE configs[2] with Settings.z_bf16), else 0 (fp32): what the entry points
    that read z take as `z_bf16`; batch strides stay in elements."""
    return int(z.dtype == torch.bfloat16)


def _wino_legal(Cin, Cout):
    return Cin >= 16 and Cin % 4 == 0 and Cout % 4 == 0


def _in_buffer_range(Cin, Cout, H, W):
    """The F(4x4) and bf16 kernels address ONE image (inside a concat buffer of up to twice its channels) through a
    32-bit buffer resource: their entry points reject operands beyond 2 GiB, so the dispatch does not select them there."""
    return (2 * max(Cin, Cout) + 16) * H * W * 4 < 2 ** 31 and (max(Cin, Cout) + 16) * 36 * (max(Cin, Cout) + 63) * 4 < 2 ** 31


def _split_legal(Cin, Cout, H, W):
    return Cin % 16 == 0 and W > 16 and W % 4 == 0 and H >= 8


def conv3x3_algo(B, Cin, Cout, H, W):
    """-> "split" | "winograd4" | "direct" | "bf16" for a conv with Cin inputs and Cout outputs on B maps of H x W."""
    algo = conv_algo()
    if algo in ("bf16", "winograd4", "auto", "split") and not _in_buffer_range(Cin, Cout, H, W):
        algo = "direct"               # (operands beyond the 2 GiB buffer-resource range: the direct kernel addresses with 64-bit pointers)
    if algo == "split":
        if _split_legal(Cin, Cout, H, W):
            return "split"
        algo = "auto_nosplit"
    if algo == "auto" and split_enabled() and _split_legal(Cin, Cout, H, W):
        # persistent 8-wave blocks, one per CU, tiles of 64 channels x 16 rows x 32 pixels: worth it once every CU gets a tile
        if B * -(-H // 16) * -(-W // 32) * -(-Cout // 64) >= (n_cu() * 3) // 4:
            return "split"
    if algo == "auto_nosplit":
        algo = "auto"
    if algo == "bf16":
        if Cin % 16 == 0 and Cout % 4 == 0 and W >= 16 and H >= 8:
            return "bf16"
        algo = "auto"
    if algo == "direct" or not _wino_legal(Cin, Cout):
        return "direct"
    # on maps smaller than 8x8 the tile quantisation wastes most of an MFMA tile and the direct kernel's
    # two-level accumulation is the more accurate one (the 1x1 .. 4x4 bottleneck maps of small inputs feed
    # BatchNorm batches of a few values per channel, which amplify any rounding difference)
    if min(H, W) < 8:
        return "direct"
    if algo == "winograd4":
        return "winograd4"
    if min(H, W) >= 16:
        img = 1 if W > 16 else 2                       # conv_wino4.hip block: 32 tiles of 4x4 px, 64 channels
        blocks = -(-B // img) * -(-W // (32 if W > 16 else 16)) * -(-H // 16) * -(-Cout // 64)
        if blocks >= (n_cu() * 7) // 8:
            return "winograd4"
    return "direct"                   # (round 5: the F(2x2,3x3) kernels that used to take the layers too small to fill the chip are gone)


def use_winograd(Cin, Cout, H, W):
    """May a Winograd weight-gradient kernel (F(3x3,4x4)) take this layer?"""
    return conv_algo() != "direct" and _wino_legal(Cin, Cout) and min(H, W) >= 8


class Packed3x3(dict):
    """Packed forms of one 3x3 weight, built on first use per algorithm: [algo] -> (fwd, dgrad)."""

    def __init__(self, w):
        super().__init__(Cin=w.shape[1], Cout=w.shape[0])
        self.w = w.detach()

    def get_pack(self, algo):
        if algo not in self:
            self[algo] = {"direct": pack3x3, "winograd4": pack3x3_winograd4, "split": pack3x3_split, "plain16": pack3x3_plain16}[algo](self.w)
        return self[algo]


def pack3x3_auto(w):
    """-> lazily packed weights for conv3x3_auto: each algorithm's transform runs on first use after a weight
    update, so a layer only pays for the packs its shapes actually select."""
    return Packed3x3(w)


def _fp32_algo(B, Cin, Cout, H, W):
    """conv3x3_algo for a layer that runs on fp32 tensors whatever the model's setting: under conv == "bf16" the layers outside the
    pre-split plain-bf16 kernels' reach (the stem, the smallest maps, odd shapes) take the fp32 dispatch -- round 5 dropped the
    round-2 bf16 kernels that used to serve them."""
    algo = conv3x3_algo(B, Cin, Cout, H, W)
    if algo != "bf16":
        return algo
    cur = active_settings()
    with using(cur.replace(conv="auto") if cur is not None else Settings(conv="auto")):
        return conv3x3_algo(B, Cin, Cout, H, W)


def conv3x3_auto(x, pk, direction, out=None, amax=None):
    """direction 0: forward (Cin -> Cout); 1: dgrad (Cout -> Cin) with the flipped/transposed pack.
    amax: the 64 magnitude slots of x (int32 [64], written by its producer): what the fp16 split kernels scale x by."""
    Ci, Co = (pk["Cin"], pk["Cout"]) if direction == 0 else (pk["Cout"], pk["Cin"])
    shp = x.shape
    algo = _fp32_algo(shp[0], Ci, Co, shp[2], shp[3])
    if algo == "split" and direction == 1 and not split_dgrad():     # diagnostic: input gradients on the fp32-MFMA kernels
        with using(active_settings().replace(split=False) if active_settings() is not None else Settings(split=False)):
            algo = _fp32_algo(shp[0], Ci, Co, shp[2], shp[3])
    wq = pk.get_pack(algo)[direction]
    if algo == "winograd4":
        return conv3x3_winograd4(x, wq, Co, out=out)
    if algo == "split":
        return conv3x3_split(x, wq, Co, out=out, amax=amax, always=direction == 1)
    return conv_fwd(x, wq, Co, 3, out=out)


FUSE_BN_STATS = _flag("FUSE_BN_STATS", True)
BN_ON_LOAD = _flag("BN_ON_LOAD", True)        # 0: every BatchNorm + ReLU output is materialised
BN_ON_LOAD_MAX_COUT = _flag("BN_ON_LOAD_MAX_COUT", 128)
CONVT_SPLIT_MIN_BLOCKS = (int(_FLAGS["CONVT_SPLIT_MIN_BLOCKS"]) if "CONVT_SPLIT_MIN_BLOCKS" in _FLAGS else None)
CONVT_SPLIT = _flag("CONVT_SPLIT", True)     # 0: the ConvTranspose2d GEMMs stay on the fp32 MFMA pipe
SPLIT_F16 = _flag("SPLIT_F16", True)         # 0: the forward split kernel takes bf16 parts like the gradients
SPLIT_DGRAD = _flag("SPLIT_DGRAD", True)     # 0 (diagnostic): input gradients stay on the fp32-MFMA kernels
PRESPLIT = _flag("PRESPLIT", True)           # 1: pre-split operand storage (Settings.presplit)
PRESPLIT_W16 = _flag("PRESPLIT_W16", True)    # ... also the 16-pixel level (0: fp32 Winograd F(4x4) there, as in round 3)
# diagnostic / tests: every activation written pre-split ALSO leaves its fp32 tensor (same values: the parts are split from them), so
# that a harness can read each unit's output; the kernels that consume the pre-split forms are unchanged
PRESPLIT_KEEP_FP32 = _flag("PRESPLIT_KEEP_FP32", False)
# 1: split input / weight gradients on fp16 parts of power-of-two-scaled operands (22-bit operands; 15x lower per-layer error than
# the bf16 parts, but no change in the model-level worst gradient error -- 9.4e-5 either way on b4_c1_256 -- and +1.3 ms/step: the
# fp16 MFMAs hold a lower clock); default 0: bf16 parts, as in round 3
SPLIT_GRAD_F16 = _flag("SPLIT_GRAD_F16", False)
SPLIT_WGRAD_MINW = _flag("SPLIT_WGRAD_MINW", 16)   # 64: the 32- and 16-pixel levels keep the Winograd weight gradients
SPLIT_AUTO = _flag("SPLIT", True)          # 0: "auto" never selects the split-bf16 kernel (round-2 dispatch)
STEM_FUSED = _flag("STEM_FUSED", True)      # 0: the stem takes the direct MFMA kernel + a statistics pass


def conv3x3_fwd_bn_partials(x, pk, norm=None, amax=None):
    """Forward 3x3 convolution of a Conv-BatchNorm pair (OV:47-48, 51-52): -> (z, cm).  cm = the channel-major
    BatchNorm records [Cout, nparts, 3] the F(4x4) kernel's epilogue emits (image-major: nparts / B per image), or None
    where the selected kernel does not emit them (then `bn_train_coeffs` runs its own statistics pass).
    norm = (z_prev, save [G, 4, Cin]): normalise on load -- x is NOT read; the input is relu(bn(z_prev)) of the unit below,
    applied in the split kernel's staging (the caller has checked norm_on_load_ok)."""
    Ci, Co = pk["Cin"], pk["Cout"]
    if norm is not None:
        z_prev, save = norm
        require_gpu(z_prev, save)
        zs, zbs = plane(z_prev)
        B, _, H, W = zs.shape
        nparts = int(_lib.load().onet_conv3x3_split_nparts(B, H, W)) if (FUSE_BN_STATS and not sync_bn()) else 0
        out = torch.empty((B, Co, H, W), dtype=F32, device=zs.device)
        cm = torch.empty((Co, nparts, 3), dtype=F32, device=zs.device) if nparts > 0 else None
        e0 = _prof_begin("conv3x3_split_kernel")
        wq = pk.get_pack("split")[0]
        _lib.call("onet_conv3x3_split_fwd_norm", _p(zs), zbs, _p(save), save.shape[0], _p(wq), int(wq.dtype == torch.float16), _p(out),
                  Co * H * W, _p(cm), B, Ci, Co, H, W, _stream())
        _prof_end("conv3x3_split_kernel", 2.0 * B * H * W * Ci * Co * 9, e0, 4.0 * (B * H * W * (Ci + Co) + 9 * Ci * Co))
        return out, cm
    B, _, H, W = x.shape
    if Ci <= 4 and stem_fused() and FUSE_BN_STATS and not sync_bn() and hasattr(pk, "w"):
        # the stem (Cin = n_channels): one streaming pass writes z and its statistics records (stem.hip)
        nparts = int(_lib.load().onet_conv3x3_stem_nparts(B, Ci, Co, H, W))
        if nparts > 0:
            require_gpu(x)
            tw = twin_src_of(x)          # virtual twin batch: the kernel reads X and forms the complement half on load (K7)
            xs, xbs = plane(tw[0] if tw is not None else x)
            w = pk.w if pk.w.is_contiguous() else pk.w.contiguous()
            out = torch.empty((B, Co, H, W), dtype=F32, device=x.device)
            cm = torch.empty((Co, nparts, 3), dtype=F32, device=x.device)
            e0 = _prof_begin("stem_conv_stats_kernel")
            _lib.call("onet_conv3x3_stem_fwd_stats", _p(xs), xbs, _p(w), _p(out), Co * H * W, _p(cm), B, Ci, Co, H, W,
                      B // 2 if tw is not None else 0, tw[1] if tw is not None else 0.0, _stream())
            _prof_end("stem_conv_stats_kernel", 2.0 * B * H * W * Ci * Co * 9, e0, 4.0 * (B * H * W * (Ci + Co) + 9 * Ci * Co))
            return out, cm
    algo = _fp32_algo(B, Ci, Co, H, W)
    nparts = 0
    if algo == "winograd4" and FUSE_BN_STATS and not sync_bn():
        nparts = int(_lib.load().onet_conv3x3_winograd4_nparts(B, H, W))
    if algo == "split" and FUSE_BN_STATS and not sync_bn():
        nparts = int(_lib.load().onet_conv3x3_split_nparts(B, H, W))
        if nparts > 0:
            wq = pk.get_pack(algo)[0]
            require_gpu(x)
            xs, xbs = plane(x)
            if xs.data_ptr() % 16 or xbs % 4:
                xs = xs.contiguous()
                xbs = xs.stride(0) if B > 1 else xs[0].numel()
            out = torch.empty((B, Co, H, W), dtype=F32, device=x.device)
            cm = torch.empty((Co, nparts, 3), dtype=F32, device=x.device)
            e0 = _prof_begin("conv3x3_split_kernel")
            if wq.dtype == torch.float16:
                _lib.call("onet_conv3x3_split_conv_amax", _p(xs), xbs, _p(amax), 0, _p(wq), _p(out), Co * H * W, _p(cm), B, Ci, Co, H, W,
                          _stream())
            else:
                _lib.call("onet_conv3x3_split_fwd_stats", _p(xs), xbs, _p(wq), 0, _p(out), Co * H * W, _p(cm), B, Ci, Co, H, W, _stream())
            _prof_end("conv3x3_split_kernel", 2.0 * B * H * W * Ci * Co * 9, e0, 4.0 * (B * H * W * (Ci + Co) + 9 * Ci * Co))
            return out, cm
    if nparts <= 0:
        return conv3x3_auto(x, pk, 0, amax=amax), None
    wq = pk.get_pack(algo)[0]
    require_gpu(x, wq)
    x, xbs = plane(x)
    out = torch.empty((B, Co, H, W), dtype=F32, device=x.device)
    cm = torch.empty((Co, nparts, 3), dtype=F32, device=x.device)
    e0 = _prof_begin("conv_wino4_kernel")
    _lib.call("onet_conv3x3_winograd4_fwd_stats", _p(x), xbs, _p(wq), _p(out), Co * H * W, _p(cm), B, Ci, Co, H, W,
              _stream())
    _prof_end("conv_wino4_kernel", 2.0 * B * H * W * Ci * Co * 9, e0, 4.0 * (B * H * W * (Ci + Co) + 9 * Ci * Co))
    return out, cm


FUSE_BN_REDUCE = _flag("FUSE_BN_REDUCE", True)


def conv3x3_dgrad_bnreduce(dz, pk, z_prev, save_prev):
    """Input gradient of the second convolution of a Conv-BN-ReLU-Conv chain (OV:47-53) with the first BatchNorm-backward
    pass of the layer BELOW folded into the epilogue: -> (da, records [Cin, nparts, 2]) or None where the F(4x4) kernel
    is not the one selected / the map is not made of full blocks.  save_prev: [G, 4, Cin] coefficients of the G
    statistics groups (consecutive batch slices) of the layer below; z_prev its pre-activation."""
    if not FUSE_BN_REDUCE or sync_bn():
        return None
    Ci, Co = pk["Cout"], pk["Cin"]                       # dgrad: Cout -> Cin
    B, _, H, W = dz.shape
    if conv3x3_algo(B, Ci, Co, H, W) != "winograd4":
        return None
    nparts = int(_lib.load().onet_conv3x3_winograd4_nparts(B, H, W))
    G = save_prev.shape[0]
    if nparts <= 0 or B % G or tuple(z_prev.shape) != (B, Co, H, W) or not save_prev.is_contiguous():
        return None
    z_prev, zbs = plane(z_prev)
    if (zbs & 3) or (z_prev.data_ptr() & 15):
        return None
    wq = pk.get_pack("winograd4")[1]
    require_gpu(dz, wq, z_prev, save_prev)
    dz, dbs = plane(dz)
    da = torch.empty((B, Co, H, W), dtype=F32, device=dz.device)
    rec = torch.empty((Co, nparts, 2), dtype=F32, device=dz.device)
    e0 = _prof_begin("conv_wino4_kernel")
    _lib.call("onet_conv3x3_winograd4_dgrad_bnreduce", _p(dz), dbs, _p(wq), _p(da), Co * H * W, _p(z_prev), zbs,
              _p(save_prev), B // G, _p(rec), B, Ci, Co, H, W, _stream())
    _prof_end("conv_wino4_kernel", 2.0 * B * H * W * Ci * Co * 9, e0, 4.0 * (B * H * W * (Ci + Co) + 9 * Ci * Co))
    return da, rec






def pack3x3_winograd4(w):
    """F(4x4,3x3) transformed weights: (fwd [Cin][36][CoutP], dgrad [Cout][36][CinP]) with the row's channel dimension padded to
    a multiple of 64 and permuted inside 64-blocks as conv_wino4.hip's A operand wants it (written in full by the pack kernel)."""
    require_gpu(w)
    w = w.detach().contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    wf = torch.empty(Cin * 36 * ((Cout + 63) // 64 * 64), dtype=F32, device=w.device)
    wd = torch.empty(Cout * 36 * ((Cin + 63) // 64 * 64), dtype=F32, device=w.device)
    _lib.call("onet_conv3x3_pack_weights_winograd4", _p(w), _p(wf), _p(wd), Cout, Cin, _stream())
    return wf, wd


def conv3x3_winograd4(x, wq, Cout, out=None):
    """z = conv3x3(x) by Winograd F(4x4,3x3) with transformed weights wq ([Cin][36][Cout]); fwd and dgrad."""
    require_gpu(x, wq)
    x, xbs = plane(x)
    B, Cin, H, W = x.shape
    if out is None:
        out = torch.empty((B, Cout, H, W), dtype=F32, device=x.device)
    zbs = out.stride(0) if B > 1 else Cout * H * W
    e0 = _prof_begin("conv_wino4_kernel")
    _lib.call("onet_conv3x3_winograd4_fwd", _p(x), xbs, _p(wq), _p(out), zbs, B, Cin, Cout, H, W, _stream())
    _prof_end("conv_wino4_kernel", 2.0 * B * H * W * Cin * Cout * 9, e0, 4.0 * (B * H * W * (Cin + Cout) + 9 * Cin * Cout))
    return out


def pack3x3_plain16(w):
    """Plain bf16 packs (ONE part, bf16(w)) in the split kernels' slot order, for the pre-split kernels under conv == "bf16":
    (fwd [Cin/16][9][2][Cout][8], dgrad [ceil(Cout/16)][9][2][Cin][8])."""
    require_gpu(w)
    w = w.detach().contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    wf = torch.empty(Cin * 9 * Cout + 8, dtype=BF, device=w.device) if Cin % 16 == 0 else None
    wd = torch.empty((-(-Cout // 16) * 16) * 9 * Cin + 8, dtype=BF, device=w.device)
    _lib.call("onet_conv3x3_split_pack_weights", _p(w), _p(wf), _p(wd), None, Cout, Cin, 2, 2, _stream())
    return wf, wd


def pack3x3_split(w):
    """bf16 (hi, mid) packs of a 3x3 weight for conv_split.hip: (fwd [Cin/16][2][9][2][Cout][8], dgrad [ceil(Cout/16)][2][9][2][Cin][8]);
    a pack whose reduction dimension is not a multiple of 16 on the forward side is None."""
    require_gpu(w)
    w = w.detach().contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    # a pack's parts are fp16 (of 2^k w, k from the tensor's largest magnitude; (2^k, 2^-k) stored behind the pack: + 8 elements)
    # or bf16 -- its dtype tells conv3x3_split which arithmetic the pack is for
    f_fwd, f_dg = split_f16(), grad_f16() or presplit()
    wf = torch.empty(2 * Cin * 9 * Cout + 8, dtype=torch.float16 if f_fwd else BF, device=w.device) if Cin % 16 == 0 else None
    wd = torch.empty(2 * (-(-Cout // 16) * 16) * 9 * Cin + 8, dtype=torch.float16 if f_dg else BF, device=w.device)
    ws = torch.empty(AMAX_SLOTS, dtype=torch.int32, device=w.device) if (f_fwd or f_dg) else None
    _lib.call("onet_conv3x3_split_pack_weights", _p(w), _p(wf), _p(wd), _p(ws), Cout, Cin, int(f_fwd), int(f_dg), _stream())
    return wf, wd


COUNT_FOREACH = _flag("COUNT_FOREACH", True)    # 0 (diagnostic): one add_ launch per BatchNorm counter


class counting_batches:
    """Within the block BatchNorm's num_batches_tracked increments (count_batches) are collected and applied in ONE multi-tensor launch
    at the end (also when the block raises: the units that ran have updated their running statistics)."""

    def __enter__(self):
        self.outer = getattr(_TLS, "counters", None)
        _TLS.counters = [] if COUNT_FOREACH else None
        return self

    def __exit__(self, *exc):
        pending, _TLS.counters = _TLS.counters or [], self.outer
        by_inc = {}
        for t, n in pending:
            by_inc.setdefault(n, []).append(t)
        for n, ts in by_inc.items():
            if len(ts) > 1 and all(t.is_cuda for t in ts):
                torch._foreach_add_(ts, n)
            else:
                for t in ts:
                    t.add_(n)
        return False


def count_batches(counter, n):
    pending = getattr(_TLS, "counters", None)
    if pending is None:
        counter.add_(n)
    else:
        pending.append((counter, n))


AMAX_SLOTS = 64 * 32       # 64 magnitude slots, one per 128-byte line (bn.hip: amax_commit)


_AMAX_ARENA = {}            # device -> [tensor [N, AMAX_SLOTS] int32, next free row]
AMAX_ARENA_ROWS = 192


def amax_arena_reset(device):
    """One fill zeroes a step's worth of magnitude slots (Onet.forward calls this; ~70 sets per training step): new_amax then hands
    out rows instead of launching a fill each."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    # a FRESH block per forward (1.5 MB from the caching allocator: no device malloc in steady state): the rows a forward handed out are
    # read again by its backward (ctx.x_slots: the weight gradient undoes the activation's scale), so a second forward before that backward
    # -- gradient accumulation, another model on the device, an eval pass -- must not zero or re-issue them; the rows keep their block alive
    ar = _AMAX_ARENA[key] = [torch.empty((AMAX_ARENA_ROWS, AMAX_SLOTS), dtype=torch.int32, device=device), 0]
    fill(ar[0].view(torch.float32), 0.0)


def new_amax(device):
    """64 zeroed magnitude slots (8 KB: a cache line each; conv_split.hip: amax_read / amax_scale) for a tensor its producer is
    about to write."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    ar = _AMAX_ARENA.get(key)
    if ar is not None and ar[1] < AMAX_ARENA_ROWS:
        ar[1] += 1
        return ar[0][ar[1] - 1]
    return torch.zeros(AMAX_SLOTS, dtype=torch.int32, device=device)


def absmax_slots(x):
    """Magnitude slots of an existing tensor (one extra pass over it: for callers whose producer did not record them)."""
    require_gpu(x)
    xc = x.contiguous()
    slots = new_amax(x.device)
    _lib.call("onet_absmax_slots", _p(xc), xc.numel(), _p(slots), _stream())
    return slots


def conv3x3_split(x, wq, Cout, out=None, norm=None, amax=None, always=False):
    """z = conv3x3(x) in fp32 accuracy on the 16-bit matrix cores (operands split into two fp16 / bf16 parts, three MFMAs per term).
    norm = save [G, 4, Cin]: x is a pre-activation; the kernel convolves relu(bn(x)), applied in its staging.
    fp16 pack: amax = the magnitude slots of x; always = True (input gradients): x is scaled so that its amax lands in [2^13, 2^14)
    -- the slots are then REQUIRED and computed here when the caller has none; always = False: overflow guard only."""
    if wq is None or not wq.is_cuda or wq.dtype not in (torch.bfloat16, torch.float16):
        raise TypeError("conv3x3_split: wq must be a split pack on the GPU (pack3x3_split)")
    f16 = int(wq.dtype == torch.float16)
    require_gpu(x)
    if norm is not None:
        require_gpu(norm)
        x, xbs = plane(x)
        B, Cin, H, W = x.shape
        if out is None:
            out = torch.empty((B, Cout, H, W), dtype=F32, device=x.device)
        e0 = _prof_begin("conv3x3_split_kernel")
        _lib.call("onet_conv3x3_split_fwd_norm", _p(x), xbs, _p(norm), norm.shape[0], _p(wq), f16, _p(out),
                  out.stride(0) if B > 1 else Cout * H * W, None, B, Cin, Cout, H, W, _stream())
        _prof_end("conv3x3_split_kernel", 2.0 * B * H * W * Cin * Cout * 9, e0, 4.0 * (B * H * W * (Cin + Cout) + 9 * Cin * Cout))
        return out
    x, xbs = plane(x)
    if x.data_ptr() % 16 or xbs % 4:                 # the kernel stages whole float4s: 16-byte aligned image rows
        x = x.contiguous()
        xbs = x.stride(0) if x.shape[0] > 1 else x[0].numel()
    B, Cin, H, W = x.shape
    if out is None:
        out = torch.empty((B, Cout, H, W), dtype=F32, device=x.device)
    zbs = out.stride(0) if B > 1 else Cout * H * W
    if f16 and always and amax is None:
        amax = absmax_slots(x)
    e0 = _prof_begin("conv3x3_split_kernel")
    if f16:
        _lib.call("onet_conv3x3_split_conv_amax", _p(x), xbs, _p(amax), int(always), _p(wq), _p(out), zbs, None, B, Cin, Cout, H, W, _stream())
    else:
        _lib.call("onet_conv3x3_split_fwd", _p(x), xbs, _p(wq), 0, _p(out), zbs, B, Cin, Cout, H, W, _stream())
    _prof_end("conv3x3_split_kernel", 2.0 * B * H * W * Cin * Cout * 9, e0, 4.0 * (B * H * W * (Cin + Cout) + 9 * Cin * Cout))
    return out


def split_pack_act(x, f16=True, scale=1.0, out=None, parts=2, slots=None):
    """fp32 NCHW -> the pre-split slot layout of conv_split.hip: [B, C/8, H, 2 (hi | mid), W, 8] fp16 (or bf16) parts of
    scale * x; 4 bytes per element, the fp32 tensor's footprint.  parts = 1: plain bf16, [B, C/8, H, 1, W, 8].
    slots: the tensor's magnitude slots -- x is additionally scaled by their guard scale (2^0 unless the bound reaches 2^15), which
    the consumer undoes from the same slots (what the fused producers do)."""
    require_gpu(x)
    x, xbs = plane(x)
    B, C, H, W = x.shape
    if out is not None:
        parts = out.shape[3]
    if out is None:
        out = torch.empty((B, C // 8, H, parts, W, 8), dtype=torch.float16 if (f16 and parts == 2) else BF, device=x.device)
    _lib.call("onet_split_pack_act", _p(x), xbs, _p(out), _pbs(out), B, C, H, W, 2 if parts == 1 else int(f16), float(scale), _p(slots),
              _stream())
    return out


def conv3x3_split_pre(xs, wq, Cout, out=None, slots=None, always=False, stats=None, slots2=None, split_ch=0, z16=False):
    """z = conv3x3 of a PRE-SPLIT activation xs [B, Cin/8, H, 2, W, 8] (split_pack_act / the producers' fused variants) with the split
    weight pack wq: the arithmetic of conv3x3_split, staging by LDS-DMA.  slots / always: the magnitude slots and rule the producer
    scaled xs by (undone by the kernel; None: unscaled)."""
    if wq is None or not wq.is_cuda or wq.dtype not in (torch.bfloat16, torch.float16) or xs.dtype != wq.dtype:
        raise TypeError("conv3x3_split_pre: xs and wq must be split packs of the same 16-bit type on the GPU")
    B, C8, H, two, W, eight = xs.shape
    f16 = 2 if two == 1 else int(wq.dtype == torch.float16)            # 2: plain bf16, one part
    Cin = C8 * 8
    if out is None:           # z16 (plain bf16 operands only): the output is stored as bf16, rounded once in the epilogue
        out = torch.empty((B, Cout, H, W), dtype=BF if (z16 and two == 1) else F32, device=xs.device)
    e0 = _prof_begin("conv3x3_split_pre_kernel")
    _lib.call("onet_conv3x3_split_fwd_pre", _p(xs), _pbs(xs), _p(slots), int(always), _p(slots2), int(split_ch if slots2 is not None or slots is not None else 0),
              _p(wq), f16, _p(out), _z16(out),
              out.stride(0) if B > 1 else Cout * H * W, _p(stats), B, Cin, Cout, H, W, _stream())
    _prof_end("conv3x3_split_pre_kernel", 2.0 * B * H * W * Cin * Cout * 9, e0,
              B * H * W * (2.0 * two * Cin + out.element_size() * Cout) + 2.0 * two * 9 * Cin * Cout)
    return out


def conv3x3_split_dgrad_pre_bnreduce(dzP, wq, Cout, z_prev, save_prev, slots=None, always=False, want_amax=False):
    """da = input gradient of a DoubleConv's second convolution from pre-split dz (conv3x3_split_pre's arithmetic) with the
    BatchNorm-backward REDUCE of the first unit -- whose output gradient da is -- in the kernel's epilogue: -> (da, rec4, da_amax | None),
    or None where the kernel does not take the shape.  z_prev / save_prev [G, 4, Cout]: the first unit's pre-activation and coefficients
    (G statistics groups = consecutive batch slices).  rec4 [nparts, Cout, 4] is bn_relu_bwd_reduce's record format (_bn_bwd_groups)."""
    if not FUSE_BN_REDUCE or sync_bn() or not FUSE_DGRAD_REDUCE:
        return None
    B, C8, H, two, W, _ = dzP.shape
    Cin = C8 * 8
    G = save_prev.shape[0]
    if tuple(z_prev.shape) != (B, Cout, H, W) or B % G or not save_prev.is_contiguous() or wq.dtype != dzP.dtype:
        return None
    lib = _lib.load()
    nparts = int(lib.onet_conv3x3_split_pre_nparts(B, H, W))
    z_prev, zbs = plane(z_prev)
    if nparts <= 0 or nparts % G or (zbs & 3) or (z_prev.data_ptr() & 15):
        return None
    f16 = 2 if two == 1 else int(wq.dtype == torch.float16)
    da = torch.empty((B, Cout, H, W), dtype=F32, device=dzP.device)
    rec4 = torch.empty((nparts, Cout, 4), dtype=F32, device=dzP.device)
    am = new_amax(dzP.device) if want_amax else None
    e0 = _prof_begin("conv3x3_split_pre_kernel")
    rc = lib.onet_conv3x3_split_dgrad_pre_bnreduce(_p(dzP), _pbs(dzP), _p(slots), int(always), _p(wq), f16, _p(da), Cout * H * W, _p(z_prev), _z16(z_prev), zbs,
                                                   _p(save_prev), B // G if G > 1 else 0, _p(rec4), _p(am), B, Cin, Cout, H, W, _stream())
    _prof_end("conv3x3_split_pre_kernel", 2.0 * B * H * W * Cin * Cout * 9 if rc == 0 else 0.0, e0,
              (B * H * W * (2.0 * two * Cin + (4.0 + z_prev.element_size()) * Cout) + 2.0 * two * 9 * Cin * Cout) if rc == 0 else 0.0)
    if rc < 0:
        raise _lib.OnetHipError(f"onet_conv3x3_split_dgrad_pre_bnreduce failed ({rc}): {_lib.last_error()}")
    return (da, rec4, am) if rc == 0 else None


FUSE_DGRAD_REDUCE = _flag("FUSE_DGRAD_REDUCE", True)      # False (tests / A-B): the first unit of a DoubleConv runs its own BatchNorm-backward reduce pass


def conv3x3_split_wgrad_pre(xs, dzs, dw_shape, out=None, x_slots=None, dz_slots=None, x_slots2=None, split_ch=0):
    """Weight gradient of a 3x3 convolution from PRE-SPLIT x and dz (split_pack_act layout, same 16-bit type):
    conv3x3_split_wgrad's arithmetic with LDS-DMA staging and transposed fragment reads; x_slots / dz_slots: the magnitude slots the
    producers scaled the operands by (undone on the result)."""
    if xs.dtype != dzs.dtype or xs.dtype not in (torch.bfloat16, torch.float16):
        raise TypeError("conv3x3_split_wgrad_pre: xs and dzs must be pre-split tensors of the same 16-bit type")
    B, C8, H, two, W, eight = xs.shape
    Cin, Cout = C8 * 8, dzs.shape[1] * 8
    assert tuple(dw_shape) == (Cout, Cin, 3, 3) and dzs.shape[3] == two
    dw = out if out is not None else torch.empty(dw_shape, dtype=F32, device=xs.device)
    need = _lib.load().onet_conv3x3_split_wgrad_ws_bytes(B, Cin, Cout, H, W)
    ws = workspace(need, xs.device)
    e0 = _prof_begin("conv3x3_split_wgrad_pre_kernel")
    _lib.call("onet_conv3x3_split_wgrad_pre", _p(xs), _pbs(xs), _p(x_slots), _p(x_slots2), int(split_ch), _p(dzs),
              _pbs(dzs), _p(dz_slots), 2 if two == 1 else int(xs.dtype == torch.float16), _p(dw), _p(ws),
              ws.numel() * 4, B, Cin, Cout, H, W, 0, _stream())
    _prof_end("conv3x3_split_wgrad_pre_kernel", 2.0 * B * H * W * Cin * Cout * 9, e0,
              2.0 * two * B * H * W * (Cin + Cout) + 4.0 * 9 * Cin * Cout)
    return dw


# ----------------------------------------------------------------------------- pre-split storage (Settings.presplit)
def pre_layer_ok(B, Cin, Cout, H, W):
    """Can a 3x3 convolution layer (Cin -> Cout on B maps of H x W) run ALL THREE of its kernels on pre-split operands -- forward and
    input gradient by conv3x3_split_pre_kernel, weight gradient by conv3x3_split_wgrad_pre_kernel?  Producers write the pre-split
    form of a tensor only where this says yes for its consumer (the decision is a function of shapes and settings, so producer and
    consumer agree without talking)."""
    if not presplit() or Cin % 16 or Cout % 16 or W < 16 or (W % 32 and W != 16) or H % 16 or H * W >= 2 ** 24:
        return False
    np_ = p16_parts()
    if np_ == 1 and (Cin % 32 or Cout % 32):                         # plain bf16: 32-channel chunks
        return False
    if W <= 32 and B * max(Cin, Cout) * H * W * 2 * np_ >= 2 ** 31:  # weight-gradient units pair images: one buffer resource over the batch
        return False
    lib = _lib.load()
    if W == 16:
        # the 16-pixel level: two images side by side per forward tile, four per weight-gradient unit; worth it once the tiles fill
        # the chip (the dispatch of the fp32-operand kernels has no split kernel for this width, so the decision is made here)
        return PRESPLIT_W16 and B % 4 == 0 and (B // 2) * -(-min(Cin, Cout) // 64) >= (n_cu() * 3) // 4 and \
            bool(lib.onet_conv3x3_split_wgrad_pre_ok(B, Cin, Cout, H, W)) and int(lib.onet_maxpool2_bwd_bn_bands(H, W)) >= 0
    want = "bf16" if np_ == 1 else "split"
    return conv3x3_algo(B, Cin, Cout, H, W) == want and conv3x3_algo(B, Cout, Cin, H, W) == want and \
        bool(lib.onet_conv3x3_split_wgrad_pre_ok(B, Cin, Cout, H, W)) and int(lib.onet_conv3x3_split_nparts(B, H, W)) > 0 and \
        int(lib.onet_maxpool2_bwd_bn_bands(H, W)) > 0


def p16_empty(B, C, H, W, device, parts=None):
    """An uninitialised pre-split tensor in the slot layout [B, C/8, H, parts, W, 8]: fp16 (hi | mid) parts (4 bytes per element) or,
    under conv == "bf16", one part of plain bf16 (2 bytes per element)."""
    parts = p16_parts() if parts is None else parts
    return torch.empty((B, C // 8, H, parts, W, 8), dtype=torch.float16 if parts == 2 else BF, device=device)


def p16_of(t):
    """The valid pre-split form riding on tensor `t` (tag_p16), or None."""
    tag = getattr(t, "_onet_p16", None)
    if tag is None:
        return None
    return tag[0] if tag[1] == t._version else None


def p16_slots(t):
    """The magnitude slots the pre-split form of `t` was scaled by: None (unscaled), one set, or (skip set, up-sampled set, first
    channel of the second group) for a concat buffer with two producers."""
    tag = getattr(t, "_onet_p16", None)
    if tag is None or tag[1] != t._version:
        return None
    return tag[2]


def tag_p16(t, P, slots=None):
    if P is not None and t is not None:
        t._onet_p16 = (P, t._version, slots)
    return t


def _slots3(info):
    """-> (slots of the first group | None, slots of the second group | None, first channel of the second group or 0)"""
    if isinstance(info, tuple):
        return info
    return info, None, 0


def _pbs(P):
    """batch stride of a pre-split tensor (or of a channel-group slice of one) in 4-byte units"""
    return P.stride(0) // 2 if P.shape[0] > 1 else P.shape[1] * P.shape[2] * P.shape[3] * P.shape[4] * 4


def bn_relu_apply_split(z, save, xs, a=None, slots=None, group_images=0):
    """relu(bn(z)) written pre-split into xs [B, C/8, H, 2, W, 8] (a whole tensor or the leading channel groups of a concat buffer)
    and, when `a` is given, in fp32 too.  The same values, bit for bit, as bn_relu_apply (times the power of two the magnitude
    slots select -- 1 unless the activation's bound reaches 2^15).  group_images > 0: `save` is [G][4][C], one set of coefficients
    per group of that many consecutive images (the statistics groups of a twin batch in one launch)."""
    z, zbs = plane(z)
    B, C, H, W = z.shape
    assert group_images == 0 or (B % group_images == 0 and save.numel() == (B // group_images) * 4 * C and save.is_contiguous())
    _lib.call("onet_bn_relu_apply_split", _p(z), _z16(z), zbs, _p(xs), _pbs(xs), _p(a), 0 if a is None else (a.stride(0) if B > 1 else C * H * W),
              _p(save), _p(slots), xs.shape[3], group_images, B, C, H, W, _stream(),
              nbytes=(z.element_size() + 2 * xs.shape[3] + 4 * (a is not None)) * z.numel())


def bn_relu_apply_pool_split(z, save, xs, a, ys, y, slots=None, group_images=0):
    """relu(bn(z)) and its 2 x 2 max-pooling in one pass: the activation pre-split (xs) and / or fp32 (a), the pooled tensor pre-split
    (ys) or fp32 (y).  -> False where the kernel does not take the shape.  group_images: as bn_relu_apply_split."""
    z, zbs = plane(z)
    B, C, H, W = z.shape
    assert group_images == 0 or (B % group_images == 0 and save.numel() == (B // group_images) * 4 * C and save.is_contiguous())
    n, m = C * H * W, C * (H // 2) * (W // 2)
    rc = _lib.load().onet_bn_relu_apply_pool_split(_p(z), _z16(z), zbs, _p(xs), 0 if xs is None else _pbs(xs), _p(a),
                                                  0 if a is None else (a.stride(0) if B > 1 else n), _p(ys), 0 if ys is None else _pbs(ys),
                                                  _p(y), 0 if y is None else (y.stride(0) if B > 1 else m), _p(save), _p(slots),
                                                  (xs if xs is not None else ys).shape[3] if (xs is not None or ys is not None) else 2,
                                                  group_images, B, C, H, W, _stream())
    if rc < 0:
        raise _lib.OnetHipError(f"onet_bn_relu_apply_pool_split failed ({rc}): {_lib.last_error()}")
    return rc == 0


def conv3x3_pre_bn_partials(xP, pk, slots=None):
    """conv3x3_fwd_bn_partials for a pre-split input: -> (z, cm) with the BatchNorm statistics records of the epilogue.
    slots: what xP's producer(s) scaled it by (p16_slots)."""
    B, C8, H, np_, W, _ = xP.shape
    Co = pk["Cout"]
    wq = pk.get_pack("split" if np_ == 2 else "plain16")[0]
    nparts = int(_lib.load().onet_conv3x3_split_pre_nparts(B, H, W))
    cm = torch.empty((Co, nparts, 3), dtype=F32, device=xP.device) if nparts > 0 else None
    s1, s2, sc = _slots3(slots)
    # (z stored as bf16 only where the statistics come from the epilogue: a separate statistics pass would read the rounded values)
    return conv3x3_split_pre(xP, wq, Co, slots=s1, stats=cm, slots2=s2, split_ch=sc, z16=cm is not None and np_ == 1 and z16_storage()), cm


def _bn_bwd_groups(da, dabs, z, zbs, save_all, training, affine_out, rec4, da_amax, dz_slots):
    """Reduce + finalize of a BatchNorm + ReLU backward over ALL statistics groups of the tensor (consecutive batch slices; save_all
    [G][4][C]) in one launch each: -> (coef [G][4][C] | None, dgamma, dbeta).  rec4: the reduce records [G * np][C][4] a producer of
    da already wrote; da_amax: magnitude slots the reduce pass records max |da| in; dz_slots: slots that receive the bound of |dz|."""
    B, C, H, W = z.shape
    HW = H * W
    G = save_all.shape[0]
    Bg = B // G
    dev = z.device
    assert B % G == 0 and save_all.is_contiguous()
    if rec4 is None:
        nparts = _bn_nparts(Bg, HW)
        rec4 = torch.empty((G * nparts, C, 4), dtype=F32, device=dev)
        _lib.call("onet_bn_relu_bwd_reduce", _p(da), dabs, _p(z), _z16(z), zbs, _p(save_all), _p(rec4), G * nparts, _p(da_amax), Bg if G > 1 else 0,
                  B, C, HW, _stream(), nbytes=(4 + z.element_size()) * z.numel())
    else:
        assert rec4.shape[0] % G == 0 and rec4.shape[1] == C and rec4.shape[2] == 4 and rec4.is_contiguous()
    og, ob = affine_out if affine_out is not None else (None, None)
    dgamma = torch.empty(C, dtype=F32, device=dev) if og is None else og
    dbeta = torch.empty(C, dtype=F32, device=dev) if ob is None else ob
    coef = torch.empty((G, 4, C), dtype=F32, device=dev) if training else None
    _lib.call("onet_bn_bwd_finalize", _p(rec4), rec4.shape[0] // G, Bg * HW, _p(dgamma), _p(dbeta), _p(coef), 0, G, C, _p(save_all),
              _p(da_amax), _p(dz_slots), _stream())
    return coef, dgamma, dbeta


def bn_relu_bwd_split(da, z, save_all, training, need_affine_grads=True, affine_out=None, rec=None, rec4=None, da_amax=None):
    """BatchNorm + ReLU backward of a layer whose dz is consumed by the pre-split kernels, all statistics groups of the tensor per
    launch: the reduce pass (also recording max |da|, unless the records -- and da's magnitude slots -- came fused from da's producer),
    then the finalize pass, which also writes the bound of |dz| into fresh magnitude slots, then dz written pre-split, scaled by the
    power of two those slots select.  Plain bf16 operands (conv == "bf16", one part): no magnitudes, no scale.
    -> (dzP, dz_slots | None, dgamma, dbeta)."""
    da, dabs = plane(da)
    z, zbs = plane(z)
    B, C, H, W = z.shape
    G = save_all.shape[0]
    dev = z.device
    np_ = p16_parts()
    dzP = p16_empty(B, C, H, W, dev, np_)
    scaled = np_ == 2
    dz_slots = new_amax(dev) if scaled else None
    if scaled and rec4 is not None and da_amax is None:
        da_amax = absmax_slots(da)                  # records fused by a producer that did not record the magnitude: one extra pass
    if rec4 is None:
        da_amax = new_amax(dev) if scaled else None
    coef, dgamma, dbeta = _bn_bwd_groups(da, dabs, z, zbs, save_all, training, affine_out, rec4, da_amax, dz_slots)
    _lib.call("onet_bn_relu_bwd_apply_split", _p(da), dabs, _p(z), _z16(z), zbs, _p(save_all), _p(coef), _p(dzP), _pbs(dzP), _p(dz_slots), np_,
              B // G if G > 1 else 0, B, C, H, W, _stream(), nbytes=(4 + z.element_size() + 2 * np_) * z.numel())
    return dzP, dz_slots, dgamma, dbeta


def stem_wgrad_bn(x, da, z, save_all, training, dw_shape, affine_out=None, rec4=None, out=None, twin=None):
    """BatchNorm + ReLU backward and weight gradient of the stem unit (Cin <= 4; its input has no gradient, so dz has no other reader):
    reduce + finalize as bn_relu_bwd_groups, then ONE pass over (da, z) that forms dz per element and accumulates dW -- the apply pass
    and its dz tensor are gone.  Same bits as bn_relu_bwd_groups + conv_wgrad.  -> (dw, dgamma, dbeta)."""
    require_gpu(x, da, z)
    if twin is None:
        twin = twin_src_of(x)            # (X, bias) of a virtual twin batch: x is then only its placeholder
    B, Cin, H, W = x.shape
    x, xbs = plane(twin[0] if twin is not None else x)
    da, dabs = plane(da)
    z, zbs = plane(z)
    Cout = z.shape[1]
    G = save_all.shape[0]
    coef, dgamma, dbeta = _bn_bwd_groups(da, dabs, z, zbs, save_all, training, affine_out, rec4, None, None)
    dw = torch.empty(dw_shape, dtype=F32, device=x.device) if out is None else out
    ws = workspace(_lib.load().onet_conv_wgrad_ws_bytes(B, Cin, Cout, H, W, 3), x.device)
    e0 = _prof_begin("conv_wgrad_kernel")
    _lib.call("onet_conv3x3_stem_wgrad_bn", _p(x), xbs, _p(da), dabs, _p(z), zbs, _p(save_all), _p(coef), B // G if G > 1 else 0, _p(dw),
              _p(ws), ws.numel() * 4, B, Cin, Cout, H, W, 0, B // 2 if twin is not None else 0, twin[1] if twin is not None else 0.0, _stream())
    _prof_end("conv_wgrad_kernel", 2.0 * B * H * W * Cin * Cout * 9, e0, 4.0 * (B * H * W * (Cin + 2 * Cout) + 9 * Cin * Cout))
    return dw, dgamma, dbeta


STEM_WGRAD_BN = _flag("STEM_WGRAD_BN", True)     # 0 (diagnostic): the stem's dz is materialised by the apply pass


def bn_relu_bwd_groups(da, z, save_all, training, affine_out=None, rec4=None):
    """The same for a layer whose dz stays fp32 (the stem under pre-split storage): -> (dz, dgamma, dbeta)."""
    da, dabs = plane(da)
    z, zbs = plane(z)
    B, C, H, W = z.shape
    G = save_all.shape[0]
    coef, dgamma, dbeta = _bn_bwd_groups(da, dabs, z, zbs, save_all, training, affine_out, rec4, None, None)
    dz = torch.empty((B, C, H, W), dtype=F32, device=z.device)
    _lib.call("onet_bn_relu_bwd_apply", _p(da), dabs, _p(z), zbs, _p(save_all), _p(coef), _p(dz), C * H * W, None, B // G if G > 1 else 0,
              B, C, H * W, _stream(), nbytes=12 * z.numel())
    return dz, dgamma, dbeta


def split_wgrad_ok(x, dz):
    B, Cin, H, W = x.shape
    ok = bool(_lib.load().onet_conv3x3_split_wgrad_ok(B, Cin, dz.shape[1], H, W)) and \
        x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and dz.is_contiguous() and dz.data_ptr() % 16 == 0 and \
        max(Cin, dz.shape[1]) * H * W * 4 < 2 ** 31
    if ok and W < 64:        # 64 / W images side by side per unit: one buffer resource over the whole batch
        ok = ((B - 1) * x.stride(0) + Cin * H * W) * 4 < 2 ** 31 and B * dz.shape[1] * H * W * 4 < 2 ** 31
    return ok


def norm_on_load_ok(B, Cmid, Cout, H, W, groups):
    """May the second convolution of a DoubleConv (Cmid -> Cout on B maps of H x W, `groups` BatchNorm statistics groups) take
    the first unit's BatchNorm + ReLU into its own operand staging -- forward AND weight gradient -- so that the first unit's
    activation is never written?  Both must be the split-bf16 kernels (fp32 model, default dispatch)."""
    if not _setting("bn_on_load", BN_ON_LOAD) or sync_bn() or conv_algo() not in ("auto", "split") or groups not in (1, 2) or B % groups:
        return False
    if conv3x3_algo(B, Cmid, Cout, H, W) != "split" or W < SPLIT_WGRAD_MINW:
        return False
    # every 64-channel output tile of the forward kernel normalises the input tile again: the extra staging work grows with
    # Cout / 64 while the saved pass does not -- measured worth it up to two output tiles (the 256- and 128-pixel levels)
    if Cout > BN_ON_LOAD_MAX_COUT:
        return False
    if not _lib.load().onet_conv3x3_split_wgrad_ok(B, Cmid, Cout, H, W) or max(Cmid, Cout) * H * W * 4 >= 2 ** 31:
        return False
    return W >= 64 or B * max(Cmid, Cout) * H * W * 4 < 2 ** 31


def conv3x3_split_wgrad(x, dz, dw_shape, out=None, norm=None, dz_amax=None, x_amax=None):
    """dW of a 3x3 convolution in fp32 accuracy on the 16-bit matrix cores (conv_split.hip: both operands split, three MFMAs per term).
    norm = save [G, 4, Cin]: x is the pre-activation of the unit below, normalised (BatchNorm + ReLU) on load.
    dz_amax (magnitude slots of dz, from onet_bn_relu_bwd_apply): fp16 parts of the scaled operands (22-bit operands; x_amax:
    the overflow guard for x); None: bf16 parts (16-bit operands), the round-3 arithmetic."""
    if dz_amax is not None:
        require_gpu(x, dz)
        x, xbs = plane(x)
        dz, dzbs = plane(dz)
        B, Cin, H, W = x.shape
        Cout = dz.shape[1]
        dw = torch.empty(dw_shape, dtype=F32, device=x.device) if out is None else out
        ws = workspace(_lib.load().onet_conv3x3_split_wgrad_ws_bytes(B, Cin, Cout, H, W), x.device)
        e0 = _prof_begin("conv3x3_split_wgrad_kernel")
        _lib.call("onet_conv3x3_split_wgrad_f16", _p(x), xbs, _p(x_amax), _p(norm), 0 if norm is None else norm.shape[0], _p(dz), dzbs,
                  _p(dz_amax), _p(dw), _p(ws), ws.numel() * 4, B, Cin, Cout, H, W, 0, _stream())
        _prof_end("conv3x3_split_wgrad_kernel", 2.0 * B * H * W * Cin * Cout * 9, e0, 4.0 * (B * H * W * (Cin + Cout) + 9 * Cin * Cout))
        return dw
    if norm is not None:
        require_gpu(x, dz, norm)
        x, xbs = plane(x)
        dz, dzbs = plane(dz)
        B, Cin, H, W = x.shape
        Cout = dz.shape[1]
        dw = torch.empty(dw_shape, dtype=F32, device=x.device) if out is None else out
        ws = workspace(_lib.load().onet_conv3x3_split_wgrad_ws_bytes(B, Cin, Cout, H, W), x.device)
        e0 = _prof_begin("conv3x3_split_wgrad_kernel")
        _lib.call("onet_conv3x3_split_wgrad_norm", _p(x), xbs, _p(norm), norm.shape[0], _p(dz), dzbs, _p(dw), _p(ws), ws.numel() * 4,
                  B, Cin, Cout, H, W, 0, _stream())
        _prof_end("conv3x3_split_wgrad_kernel", 2.0 * B * H * W * Cin * Cout * 9, e0, 4.0 * (B * H * W * (Cin + Cout) + 9 * Cin * Cout))
        return dw
    require_gpu(x, dz)
    x, xbs = plane(x)
    dz, dzbs = plane(dz)
    B, Cin, H, W = x.shape
    Cout = dz.shape[1]
    dw = torch.empty(dw_shape, dtype=F32, device=x.device) if out is None else out
    need = _lib.load().onet_conv3x3_split_wgrad_ws_bytes(B, Cin, Cout, H, W)
    ws = workspace(need, x.device)
    e0 = _prof_begin("conv3x3_split_wgrad_kernel")
    _lib.call("onet_conv3x3_split_wgrad", _p(x), xbs, _p(dz), dzbs, _p(dw), _p(ws), ws.numel() * 4, B, Cin, Cout, H, W, 0, _stream())
    _prof_end("conv3x3_split_wgrad_kernel", 2.0 * B * H * W * Cin * Cout * 9, e0, 4.0 * (B * H * W * (Cin + Cout) + 9 * Cin * Cout))
    return dw


def grad_slot_if_free(param):
    """FlatAdam registers, per parameter, its slice of the flat gradient buffer.  When the parameter has no .grad yet
    (FlatAdam.zero_grad sets it to None) a backward kernel may write its result straight into that slice and hand the
    view to autograd, which then adopts it as .grad without an accumulation kernel (62 + 72 tiny adds per step
    otherwise).  -> a fresh contiguous view, or None (no slot, or a gradient is already there: regular path)."""
    slot = getattr(param, "_onet_gslot", None)
    if slot is None or param.grad is not None or getattr(param, "_onet_gslot_taken", False):
        return None
    # one taker per zero_grad: with shared weights used twice in a graph (two-pass mode) autograd sums the two
    # contributions in its input buffer BEFORE .grad is set, so ".grad is None" alone would hand the slice out twice
    param._onet_gslot_taken = True
    flat, off, n, shape = slot
    return flat[off:off + n].view(shape)




def conv3x3_winograd4_wgrad(x, dz, dw_shape, out=None):
    """dW of a 3x3 convolution by Winograd F(3x3,4x4) (conv_wino4w.hip); see `winograd4_wgrad_ok` for its shapes."""
    require_gpu(x, dz)
    x, xbs = plane(x)
    dz, dzbs = plane(dz)
    B, Cin, H, W = x.shape
    Cout = dz.shape[1]
    dw = torch.empty(dw_shape, dtype=F32, device=x.device) if out is None else out
    need = _lib.load().onet_conv3x3_winograd4_wgrad_ws_bytes(B, Cin, Cout, H, W)
    ws = workspace(need, x.device)
    e0 = _prof_begin("conv_wino4_wgrad_kernel")
    _lib.call("onet_conv3x3_winograd4_wgrad", _p(x), xbs, _p(dz), dzbs, _p(dw), _p(ws), ws.numel() * 4, B, Cin, Cout, H, W,
              0, _stream())
    _prof_end("conv_wino4_wgrad_kernel", 2.0 * B * H * W * Cin * Cout * 9, e0, 4.0 * (B * H * W * (Cin + Cout) + 9 * Cin * Cout))
    return dw


# WGRAD4 (ONET_FLAGS): "auto" (default) = the F(3x3,4x4) weight gradient where it is the faster one today: layers with at least
# 256 input channels (or 128 -> >= 256), whose strips are re-read from L2 by many tiles (d2.*, d3.*, d4.*, up1.*, up2.*,
# up3.c1 of the 256x256 U-Net: 224-298 TF against 205-238); below that its one-unit prefetch does not cover HBM latency
# (64-channel layers: 130 TF against 200) or it only ties (128 -> 128).
# "1": wherever legal; "0": never.
WGRAD4 = _flag("WGRAD4", "auto")


def winograd4_wgrad_ok(x, dz):
    B, Cin, H, W = x.shape
    return bool(_lib.load().onet_conv3x3_winograd4_wgrad_ok(B, Cin, dz.shape[1], H, W)) and \
        x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and dz.is_contiguous()


def conv3x3_wgrad_auto(x, dz, dw_shape, out=None, dz_amax=None, x_amax=None):
    Cout, Cin = dw_shape[0], dw_shape[1]
    # fp32 tensors, maps 16 / 32 / >= 64 pixels wide: the split-bf16 row kernel (the stem, Cin < 16, keeps its own VALU kernel)
    if conv_algo() in ("auto", "split", "bf16") and (split_enabled() or conv_algo() == "split") and Cin >= 16 and \
            x.shape[3] >= SPLIT_WGRAD_MINW and split_wgrad_ok(x, dz):
        return conv3x3_split_wgrad(x, dz, dw_shape, out=out, dz_amax=dz_amax if grad_f16() else None, x_amax=x_amax)
    if use_winograd(Cin, Cout, x.shape[2], x.shape[3]):
        if WGRAD4 != "0" and winograd4_wgrad_ok(x, dz):
            return conv3x3_winograd4_wgrad(x, dz, dw_shape, out=out)
    return conv_wgrad(x, dz, dw_shape, 3, out=out)


def conv_fwd(x, wp, Cout, ks, out=None):
    """z = conv_ks(x) with packed weights wp ([Cin][ks*ks][Cout]); also used for dgrad."""
    require_gpu(x, wp)
    x, xbs = plane(x)
    B, Cin, H, W = x.shape
    if out is None:
        out = torch.empty((B, Cout, H, W), dtype=F32, device=x.device)
    zbs = out.stride(0) if B > 1 else Cout * H * W
    e0 = _prof_begin("conv_fwd_kernel")
    _lib.call("onet_conv_fwd", _p(x), xbs, _p(wp), _p(out), zbs, None, B, Cin, Cout, H, W, ks, _stream())
    _prof_end("conv_fwd_kernel", 2.0 * B * H * W * Cin * Cout * ks * ks, e0,
              4.0 * (B * H * W * (Cin + Cout) + ks * ks * Cin * Cout))
    return out


def conv_wgrad(x, dz, dw_shape, ks, out_layout=0, out=None):
    require_gpu(x, dz)
    x, xbs = plane(x)
    dz, dzbs = plane(dz)
    B, Cin, H, W = x.shape
    Cout = dz.shape[1]
    dw = torch.empty(dw_shape, dtype=F32, device=x.device) if out is None else out
    need = _lib.load().onet_conv_wgrad_ws_bytes(B, Cin, Cout, H, W, ks)
    ws = workspace(need, x.device)
    e0 = _prof_begin("conv_wgrad_kernel")
    _lib.call("onet_conv_wgrad", _p(x), xbs, _p(dz), dzbs, _p(dw), _p(ws), ws.numel() * 4, B, Cin, Cout, H, W, ks,
              out_layout, 0, _stream())
    _prof_end("conv_wgrad_kernel", 2.0 * B * H * W * Cin * Cout * ks * ks, e0,
              4.0 * (B * H * W * (Cin + Cout) + ks * ks * Cin * Cout))
    return dw


# ----------------------------------------------------------------------------- batch norm + relu
# SyncBN (opt-in): the per-rank Welford partials (forward) and (sum dy, sum dy*xhat)
# partials (backward) are all-gathered over the data-parallel group before the finalize kernels, so
# every rank normalises with the statistics of the GLOBAL batch and running stats stay identical.
SYNC_BN = False


def set_sync_bn(flag: bool):
    global SYNC_BN
    SYNC_BN = bool(flag)


def _gather_partials(part):
    """[nparts, C, k] -> [world*nparts, C, k] over the default process group (RCCL)."""
    import torch.distributed as dist
    if not (sync_bn() and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return part, 1
    world = dist.get_world_size()
    out = torch.empty((world * part.shape[0],) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, part.contiguous())
    else:       # gloo (CPU tests, the two-ranks-on-one-GPU rehearsal) has no flat all-gather: per-rank views of `out`
        dist.all_gather(list(out.chunk(world, dim=0)), part.contiguous())
    return out, world


def _bn_nparts(B, HW):
    return B * max(1, (HW + 16383) // 16384)


def bn_train_coeffs(z, gamma, beta, running_mean, running_var, momentum, eps, cm=None, save=None, act_slots=None, groups=1):
    """batch statistics -> save [4][C] = (mean, invstd, scale, shift); updates running stats in place.
    `cm` = (records [C, NP, 3], first, count): the convolution already produced this batch's statistics records
    (`conv3x3_fwd_bn_partials`), records first .. first+count-1 of every channel belong to `z`.
    groups > 1 (with `cm`): z is ONE statistics group of a twin batch and the records of the others follow (count each): all groups in
    one launch, save [groups][4][C], the running statistics updated group after group."""
    z, zbs = plane(z)
    B, C, H, W = z.shape
    if B * H * W <= 1:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {list(z.shape)}")
    if cm is not None:
        rec, first, count = cm
        assert rec.shape[0] == C and rec.shape[2] == 3 and 0 <= first and first + count <= rec.shape[1]
        if save is None:
            save = torch.empty((4, C), dtype=F32, device=z.device)
        assert groups == 1 or (first + groups * count <= rec.shape[1] and save.shape[0] == groups and save.is_contiguous())
        # (act_slots: also the bound of relu(bn(z)) into the activation's magnitude slots -- pre-split storage)
        _lib.call("onet_bn_finalize_cm", rec.data_ptr() + first * 12, count, rec.shape[1] * 3, _p(gamma), _p(beta),
                  _p(running_mean), _p(running_var), float(momentum), float(eps), _p(save), _p(act_slots), groups, C, _stream())
        return save
    assert groups == 1
    nparts = _bn_nparts(B, H * W)
    part = torch.empty((nparts, C, 3), dtype=F32, device=z.device)
    _lib.call("onet_bn_stats_partial", _p(z), zbs, _p(part), nparts, B, C, H * W, _stream(), nbytes=4 * z.numel())
    part, world = _gather_partials(part)
    nparts *= world
    if save is None:
        save = torch.empty((4, C), dtype=F32, device=z.device)
    _lib.call("onet_bn_finalize", _p(part), nparts, B * H * W * world, _p(gamma), _p(beta), _p(running_mean),
              _p(running_var), float(momentum), float(eps), _p(save), _p(act_slots), C, _stream())
    return save


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps, save=None):
    C = running_mean.numel()
    if save is None:
        save = torch.empty((4, C), dtype=F32, device=running_mean.device)
    _lib.call("onet_bn_eval_coeffs", _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(eps), _p(save),
              C, _stream())
    return save


def amax_of(t):
    """The valid magnitude slots riding on activation `t` (tag_amax), or None."""
    tag = getattr(t, "_onet_amax", None)
    if tag is None:
        return None
    slots, ver = tag
    return slots if ver == t._version else None


def tag_amax(t, slots):
    if slots is not None and t is not None:
        t._onet_amax = (slots, t._version)
    return t


def bn_relu_apply(z, save, out=None, amax=None, group_images=0):
    """a = relu(bn(z)) in fp32.  amax: magnitude slots that receive max a.  group_images > 0: save is [G][4][C], one set per group of
    that many consecutive images.  (z may be stored as bf16: ops.z16_storage.)"""
    z, zbs = plane(z)
    B, C, H, W = z.shape
    assert group_images == 0 or (B % group_images == 0 and save.numel() == (B // group_images) * 4 * C and save.is_contiguous())
    if out is None:
        out = torch.empty((B, C, H, W), dtype=F32, device=z.device)
    abs_ = out.stride(0) if B > 1 else C * H * W
    _lib.call("onet_bn_relu_apply", _p(z), _z16(z), zbs, _p(out), abs_, _p(save), _p(amax), group_images,
              B, C, H * W, _stream(), nbytes=(4 + z.element_size()) * z.numel())
    return out


FUSE_POOL = _flag("FUSE_POOL", True)       # 0: separate max-pool pass after BatchNorm + ReLU


def bn_relu_apply_pool(z, save, out, y, amax=None):
    """a = relu(bn(z)) and y = maxpool2(a) in one pass (an encoder block's output that is pooled next); out, y: plane-contiguous fp32
    destinations.  -> False (nothing done) where the fused kernel does not take the shape.  amax: magnitude slots that receive
    max a = max y."""
    z, zbs = plane(z)
    B, C, H, W = z.shape
    if H % 2 or W % 4:
        return False

    def bs(t, n):
        return 0 if t is None else (t.stride(0) if B > 1 else n)
    n, m = C * H * W, C * (H // 2) * (W // 2)
    rc = _lib.load().onet_bn_relu_apply_pool(_p(z), zbs, _p(out), bs(out, n), _p(y), bs(y, m), _p(save), _p(amax), B, C, H, W, _stream())
    if rc < 0:
        raise _lib.OnetHipError(f"onet_bn_relu_apply_pool failed ({rc}): {_lib.last_error()}")
    return rc == 0


def bn_bwd_coefs(da, z, save, training, need_affine_grads=True, acc=None, affine_out=None, red=None, red4=None, da_amax=None):
    """The reduce + finalize passes of bn_relu_bwd for one statistics group: -> (coef [4, C] | None, dgamma, dbeta).  da_amax: magnitude
    slots in which the (standalone) reduce pass records max |da|."""
    da, dabs = plane(da)
    z, zbs = plane(z)
    B, C, H, W = z.shape
    HW = H * W
    nparts = _bn_nparts(B, HW)
    dev = z.device
    dgamma = dbeta = coef = None
    if red is not None and (training or need_affine_grads):
        rec, first, count = red
        assert rec.shape[0] == C and rec.shape[2] == 2 and 0 <= first and first + count <= rec.shape[1]
        if acc is None:
            og, ob = affine_out if affine_out is not None else (None, None)
            dgamma = torch.empty(C, dtype=F32, device=dev) if og is None else og
            dbeta = torch.empty(C, dtype=F32, device=dev) if ob is None else ob
        else:
            dgamma, dbeta = acc
        coef = torch.empty((4, C), dtype=F32, device=dev) if training else None
        _lib.call("onet_bn_bwd_finalize_cm", rec.data_ptr() + first * 8, count, rec.shape[1] * 2, B * HW, _p(dgamma),
                  _p(dbeta), _p(coef), 0 if acc is None else 1, C, _stream())
    elif training or need_affine_grads:
        if red4 is not None:
            rec4, first, count = red4
            assert rec4.shape[1] == C and rec4.shape[2] == 4 and 0 <= first and first + count <= rec4.shape[0]
            part2, nparts = rec4[first:first + count], count
        else:
            part2 = torch.empty((nparts, C, 4), dtype=F32, device=dev)
            _lib.call("onet_bn_relu_bwd_reduce", _p(da), dabs, _p(z), _z16(z), zbs, _p(save), _p(part2), nparts, _p(da_amax), 0, B, C, HW,
                      _stream(), nbytes=(4 + z.element_size()) * z.numel())
        if acc is None:
            og, ob = affine_out if affine_out is not None else (None, None)
            dgamma = torch.empty(C, dtype=F32, device=dev) if og is None else og
            dbeta = torch.empty(C, dtype=F32, device=dev) if ob is None else ob
        else:
            dgamma, dbeta = acc
        accf = 0 if acc is None else 1
        coef = torch.empty((4, C), dtype=F32, device=dev) if training else None
        gathered, world = _gather_partials(part2) if training else (part2, 1)
        if world == 1:
            _lib.call("onet_bn_bwd_finalize", _p(part2), nparts, B * HW, _p(dgamma), _p(dbeta), _p(coef), accf, 1, C, None, None, None, _stream())
        else:
            # SyncBN: dgamma/dbeta are the LOCAL sums (the gradient all-reduce adds the ranks, as in
            # torch.nn.SyncBatchNorm); c1/c2 in `coef` are means over the GLOBAL batch.
            _lib.call("onet_bn_bwd_finalize", _p(part2), nparts, B * HW, _p(dgamma), _p(dbeta), None, accf, 1, C, None, None, None, _stream())
            _lib.call("onet_bn_bwd_finalize", _p(gathered), nparts * world, B * HW * world, None, None, _p(coef), 0, 1, C, None, None, None,
                      _stream())
    return coef, dgamma, dbeta


def bn_relu_bwd(da, z, save, training, need_affine_grads=True, out=None, acc=None, affine_out=None, red=None, red4=None, amax=None):
    """-> dz, dgamma, dbeta.  `out`: plane-contiguous destination for dz (a batch slice of a larger buffer);
    `acc` = (dgamma, dbeta) of another statistics group of the same layer to accumulate into; `affine_out` =
    (dgamma, dbeta) destinations to overwrite (None entries are allocated); `red` = (records [C, NP, 2], first, count):
    the (sum dy, sum dy*xhat) records of this batch slice were already written by the dgrad launch that produced
    `da` (`conv3x3_dgrad_bnreduce`), so the reduce pass over (da, z) is skipped; `red4` = (records [NP, C, 4], first,
    count): the same in the reduce kernel's own record format (written by the pooling-backward kernel).
    `amax`: 64 zeroed magnitude slots (new_amax) in which the apply pass
    records max |dz| for the fp16-split gradient kernels (several statistics groups of one tensor share them)."""
    coef, dgamma, dbeta = bn_bwd_coefs(da, z, save, training, need_affine_grads, acc, affine_out, red, red4)
    da, dabs = plane(da)
    z, zbs = plane(z)
    B, C, H, W = z.shape
    HW = H * W
    dev = z.device
    dz = torch.empty((B, C, H, W), dtype=F32, device=dev) if out is None else out
    dzbs = dz.stride(0) if B > 1 else C * HW
    _lib.call("onet_bn_relu_bwd_apply", _p(da), dabs, _p(z), zbs, _p(save), _p(coef), _p(dz), dzbs, _p(amax), 0, B, C, HW,
              _stream(), nbytes=12 * z.numel())
    return dz, dgamma, dbeta


# ----------------------------------------------------------------------------- pool / up / cat
def maxpool2_fwd(x):
    """y = MaxPool2d(2)(x)"""
    x, xbs = plane(x)
    B, C, H, W = x.shape
    y = torch.empty((B, C, H // 2, W // 2), dtype=F32, device=x.device)
    _lib.call("onet_maxpool2_fwd", _p(x), xbs, _p(y), C * (H // 2) * (W // 2), B, C, H, W, _stream(), nbytes=5 * x.numel())
    return y


def maxpool2_bwd(x, dy, add=None, add2=None, bn=None, dx_amax=None):
    """dx of MaxPool2d(2); `add`, `add2`: other gradients of x (plane-contiguous, e.g. a slice of a concat gradient)
    summed in the same pass.  `bn` = (z, save_all [G, 4, C]): x is the output relu(bn(z)) of a Conv-BN-ReLU unit with G
    statistics groups -- the unit's BatchNorm-backward reduce records are taken on the way: -> (dx, part2 [B * bands, C, 4])
    (or (dx, None) where the fused form is not available).  Pre-split storage: x may be a placeholder (the activation exists only
    pre-split; with `bn` the kernel recomputes it from z) and dx_amax = magnitude slots that receive max |dx|."""
    if is_placeholder(x):
        if bn is None:
            raise RuntimeError("onet_amd: max-pooling backward of an activation kept only pre-split needs its (z, coefficients)")
        B, C, H, W = x.shape
        x, xbs = None, C * H * W
    else:
        x, xbs = plane(x)
        B, C, H, W = x.shape
    dy, dybs = plane(dy)
    dx = torch.empty((B, C, H, W), dtype=F32, device=dy.device)
    if add is None and add2 is not None:
        add, add2 = add2, None
    a1, a1bs = plane(add) if add is not None else (None, 0)
    a2, a2bs = plane(add2) if add2 is not None else (None, 0)
    if bn is not None:
        z, save_all = bn
        bands = int(_lib.load().onet_maxpool2_bwd_bn_bands(H, W)) if (FUSE_BN_REDUCE and not sync_bn()) else 0
        z, zbs = plane(z)
        G = save_all.shape[0]
        aligned = all(t is None or (t.data_ptr() & 15) == 0 for t in (x, a1, a2)) and (z.data_ptr() & (7 if _z16(z) else 15)) == 0 and \
            (dy.data_ptr() & 7) == 0 and all((v & 3) == 0 for v in (xbs, zbs, a1bs, a2bs)) and (dybs & 1) == 0
        if bands > 0 and aligned and B % G == 0 and tuple(z.shape) == (B, C, H, W) and save_all.is_contiguous():
            part2 = torch.empty((B * bands, C, 4), dtype=F32, device=dy.device)
            _lib.call("onet_maxpool2_bwd_add_bnreduce", _p(x), xbs, _p(dy), dybs, _p(a1), a1bs, _p(a2), a2bs, _p(dx),
                      C * H * W, _p(z), _z16(z), zbs, _p(save_all), B // G, _p(part2), _p(dx_amax), B, C, H, W, _stream(),
                      nbytes=(5 + z.element_size() + 4 * (x is not None) + 4 * (a1 is not None) + 4 * (a2 is not None)) * dx.numel())
            return dx, part2
        if x is None:
            raise RuntimeError("onet_amd: max-pooling backward of an activation kept only pre-split: the fused kernel does not take this shape")
    _lib.call("onet_maxpool2_bwd", _p(x), xbs, _p(dy), dybs, _p(a1), a1bs, _p(a2), a2bs, _p(dx), C * H * W,
              B, C, H, W, 0, _stream(), nbytes=(9 + 4 * (a1 is not None) + 4 * (a2 is not None)) * x.numel())
    return dx if bn is None else (dx, None)


def copy_strided(src, dst):
    """dst[b] = src[b] for plane-contiguous [B,C,H,W] views."""
    src, sbs = plane(src)
    B, C, H, W = src.shape
    dbs = dst.stride(0) if B > 1 else C * H * W
    _lib.call("onet_copy_strided", _p(src), sbs, _p(dst), dbs, B, C * H * W, _stream())
    return dst


def pixel_shuffle2_bias(sub, bias, out, pt, pl):
    """sub [B,4C,h,w] -> out [B,C,Ho,Wo] (a plane-contiguous view, e.g. the 2nd half of a cat buffer)."""
    B, C4, h, w = sub.shape
    C = C4 // 4
    Ho, Wo = out.shape[2], out.shape[3]
    obs = out.stride(0) if B > 1 else C * Ho * Wo
    _lib.call("onet_pixel_shuffle2_bias", _p(sub), _p(bias), _p(out), obs, B, C, h, w, Ho, Wo, pt, pl, _stream())
    return out


def space_to_depth2(dy, h, w, pt, pl, want_dbias):
    dy, dybs = plane(dy)
    B, C, Ho, Wo = dy.shape
    sub = torch.empty((B, 4 * C, h, w), dtype=F32, device=dy.device)
    dbias = torch.empty(C, dtype=F32, device=dy.device) if want_dbias else None
    scratch = torch.empty(B * C, dtype=torch.float64, device=dy.device) if want_dbias else None
    _lib.call("onet_space_to_depth2", _p(dy), dybs, _p(sub), _p(dbias), _p(scratch), 0, B, C, h, w, Ho, Wo, pt, pl,
              _stream())
    return sub, dbias


def convT2x2_bwd_fusable(Ct):
    """The gather-fused ConvTranspose2d backward needs a block's 64 GEMM rows inside one sub-pixel plane."""
    return Ct % 64 == 0


def convT2x2_dgrad(dy, wp_dgrad, Cin, h, w, pt, pl, want_dbias=False, db_out=None):
    """dx1 of ConvTranspose2d(k=2, s=2) straight from the [B, Ct, Ho, Wo] window `dy` of the concat gradient.
    want_dbias: -> (dx1, dbias | None): where the 128 x 128 GEMM path takes the shape the bias gradient is summed from the
    dy rows that launch stages anyway; None = not taken (the caller then runs the separate dbias pass)."""
    require_gpu(dy, wp_dgrad)
    dy, dybs = plane(dy)
    B, Ct, Ho, Wo = dy.shape
    dx = torch.empty((B, Cin, h, w), dtype=F32, device=dy.device)
    flops, nbytes = 2.0 * B * h * w * Cin * 4 * Ct, 4.0 * (B * h * w * (Cin + 4 * Ct) + 4 * Cin * Ct)
    if want_dbias:
        lib = _lib.load()
        need = int(lib.onet_convT2x2_dgrad_dbias_ws_bytes(B, Ct, h, w))
        if need > 0:
            db = torch.empty(Ct, dtype=F32, device=dy.device) if db_out is None else db_out
            ws = torch.empty(need // 4, dtype=F32, device=dy.device)
            e0 = _prof_begin("convt_gemm_kernel")
            rc = lib.onet_convT2x2_dgrad_dbias(_p(dy), dybs, _p(wp_dgrad), _p(dx), Cin * h * w, _p(db), _p(ws), need, B, Cin, Ct, h,
                                               w, Ho, Wo, pt, pl, convt_operand_bf16(B, h, w, Ct), _stream())
            if rc == 0:
                _prof_end("convt_gemm_kernel", flops, e0, nbytes)
                return dx, db
            _prof_end("convt_gemm_kernel", 0.0, e0, 0.0)
            if rc < 0:
                raise _lib.OnetHipError(f"onet_convT2x2_dgrad_dbias failed ({rc}): {_lib.last_error()}")
    e0 = _prof_begin("convt_gemm_kernel")
    _lib.call("onet_convT2x2_dgrad", _p(dy), dybs, _p(wp_dgrad), _p(dx), Cin * h * w, B, Cin, Ct, h, w, Ho, Wo, pt, pl,
              convt_operand_bf16(B, h, w, Ct), _stream())
    _prof_end("convt_gemm_kernel", flops, e0, nbytes)
    return (dx, None) if want_dbias else dx


def convT2x2_wgrad(x, dy, dw_shape, pt, pl, want_dbias, out=None, db_out=None):
    """(dW [Cin, Ct, 2, 2], dbias | None) of ConvTranspose2d(k=2, s=2) from x1 and the concat-gradient window."""
    require_gpu(x, dy)
    x, xbs = plane(x)
    dy, dybs = plane(dy)
    B, Cin, h, w = x.shape
    Ct, Ho, Wo = dy.shape[1], dy.shape[2], dy.shape[3]
    dw = torch.empty(dw_shape, dtype=F32, device=x.device) if out is None else out
    need = _lib.load().onet_convT2x2_wgrad_ws_bytes(B, Cin, Ct, h, w)
    ws = workspace(need, x.device)
    e0 = _prof_begin("convt_wgrad_gemm_kernel")
    _lib.call("onet_convT2x2_wgrad", _p(x), xbs, _p(dy), dybs, _p(dw), _p(ws), ws.numel() * 4, B, Cin, Ct, h, w, Ho, Wo,
              pt, pl, convt_operand_bf16(B, h, w, Ct), _stream())
    _prof_end("convt_wgrad_gemm_kernel", 2.0 * B * h * w * Cin * 4 * Ct, e0, 4.0 * (B * h * w * (Cin + 4 * Ct) + 4 * Cin * Ct))
    db = None
    if want_dbias:
        db = torch.empty(Ct, dtype=F32, device=x.device) if db_out is None else db_out
        scratch = torch.empty(B * Ct, dtype=torch.float64, device=x.device)
        _lib.call("onet_convT2x2_dbias", _p(dy), dybs, _p(db), _p(scratch), 0, B, Ct, h, w, Ho, Wo, pt, pl, _stream())
    return dw, db


def bilinear2x_fwd(x, out, pt, pl):
    x, xbs = plane(x)
    B, C, h, w = x.shape
    Ho, Wo = out.shape[2], out.shape[3]
    obs = out.stride(0) if B > 1 else C * Ho * Wo
    _lib.call("onet_bilinear2x_fwd", _p(x), xbs, _p(out), obs, B, C, h, w, Ho, Wo, pt, pl, _stream())
    return out


def bilinear2x_bwd(dy, h, w, pt, pl):
    dy, dybs = plane(dy)
    B, C, Ho, Wo = dy.shape
    dx = torch.empty((B, C, h, w), dtype=F32, device=dy.device)      # (gather kernel: every element written, no atomics)
    _lib.call("onet_bilinear2x_bwd", _p(dy), dybs, _p(dx), C * h * w, B, C, h, w, Ho, Wo, pt, pl, _stream())
    return dx


# ----------------------------------------------------------------------------- elementwise
def complement_clip(x, bias, out=None):
    require_gpu(x)
    x = x.contiguous()
    y = torch.empty_like(x) if out is None else out          # `out`: a contiguous tensor of x's size
    _lib.call("onet_complement_clip", _p(x), _p(y), float(bias), x.numel(), _stream())
    return y


def fill(t, value):
    _lib.call("onet_fill", _p(t), float(value), t.numel(), _stream())
    return t


# ----------------------------------------------------------------------------- head / loss
HEAD_NORM = _flag("HEAD_NORM", True)      # 0: the last unit's BatchNorm + ReLU pass writes its activation for the head to read


def head_softmax_fwd(Lt, Ht, Ld, Hd, want_sums=False, h_norm=None):
    """-> Vt, Vd, S  (+ sLt, sLd = per-pixel channel sums of Lt / Ld, [B,1,H,W], when want_sums).
    h_norm = (save_t, save_d): Ht / Hd are the pre-activations of the last unit, normalised + rectified on load."""
    require_gpu(Lt, Ht, Ld, Hd)
    Lt, a = plane(Lt)
    Ht, b = plane(Ht)
    Ld, c = plane(Ld)
    Hd, d = plane(Hd)
    B, C, H, W = Lt.shape
    dev = Lt.device
    Vt = torch.empty((B, 1, H, W), dtype=F32, device=dev)
    Vd = torch.empty((B, 1, H, W), dtype=F32, device=dev)
    S = torch.empty((B, 2, H, W), dtype=F32, device=dev)
    sLt = torch.empty((B, 1, H, W), dtype=F32, device=dev) if want_sums else None
    sLd = torch.empty((B, 1, H, W), dtype=F32, device=dev) if want_sums else None
    nt, nd = h_norm if h_norm is not None else (None, None)
    _lib.call("onet_head_softmax_fwd", _p(Lt), a, _p(Ht), b, _p(Ld), c, _p(Hd), d, _p(Vt), _p(Vd), _p(S), _p(sLt), _p(sLd), _p(nt), _p(nd),
              B, C, H * W, _stream(), nbytes=16 * Lt.numel())
    return (Vt, Vd, S, sLt, sLd) if want_sums else (Vt, Vd, S)


def head_softmax_bwd(dVt, dVd, dS, S, Lt, Ht, Ld, Hd, twin=False, gsums=(None, None), h_norm=None):
    """-> dLt, dHt, dLd, dHd; twin=True: -> (dL, dH) of shape [2B, C, H, W], the top and down halves adjacent.
    gsums = (d loss / d sLt, d loss / d sLd), [B,1,H,W] each or None: added to every channel of dLt / dLd."""
    Lt, a = plane(Lt)
    Ht, b = plane(Ht)
    Ld, c = plane(Ld)
    Hd, d = plane(Hd)
    B, C, H, W = Lt.shape
    dev = Lt.device
    dVt = None if dVt is None else dVt.contiguous()
    dVd = None if dVd is None else dVd.contiguous()
    dS = None if dS is None else dS.contiguous()
    if twin:
        dL = torch.empty((2 * B, C, H, W), dtype=F32, device=dev)
        dH = torch.empty((2 * B, C, H, W), dtype=F32, device=dev)
        outs = [dL[:B], dH[:B], dL[B:], dH[B:]]
    else:
        outs = [torch.empty((B, C, H, W), dtype=F32, device=dev) for _ in range(4)]
    gst = None if gsums[0] is None else gsums[0].contiguous()
    gsd = None if gsums[1] is None else gsums[1].contiguous()
    nt, nd = h_norm if h_norm is not None else (None, None)
    _lib.call("onet_head_softmax_bwd", _p(dVt), _p(dVd), _p(dS), _p(gst), _p(gsd), _p(S), _p(Lt), a, _p(Ht), b,
              _p(Ld), c, _p(Hd), d, _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(outs[3]), _p(nt), _p(nd), B, C, H * W, _stream(),
              nbytes=32 * Lt.numel())
    return (dL, dH) if twin else outs


def _rows(t):
    """[B,1,H,W] slice of S -> (tensor, batch stride) with the H*W block contiguous."""
    B, C, H, W = t.shape
    assert C == 1
    ok = (W == 1 or t.stride(3) == 1) and (H == 1 or t.stride(2) == W)
    if not ok:
        t = t.contiguous()
    return t, (t.stride(0) if B > 1 else H * W)


def jsd_fwd(L, Si, Sp, sums=None):
    """one jsd term; `sums` given ([B,1,H,W] contiguous channel sums from the head kernel): L is not read (may be None)."""
    require_gpu(Si, Sp)
    Si, ibs = _rows(Si)
    Sp, pbs = _rows(Sp)
    if sums is not None:
        B, _, H, W = sums.shape
        C, lbs, L = 1, 0, None
        dev = sums.device
        sums = sums.contiguous()
    else:
        require_gpu(L)
        L, lbs = plane(L)
        B, C, H, W = L.shape
        dev = L.device
        sums = torch.empty(B * H * W, dtype=F32, device=dev)
    part = torch.empty(_lib.load().onet_jsd_nparts(), dtype=torch.float64, device=dev)
    out = torch.empty((), dtype=F32, device=dev)
    _lib.call("onet_jsd_fwd", _p(L), lbs, _p(Si), ibs, _p(Sp), pbs, _p(sums), _p(part), _p(out), B, C, H * W,
              _stream())
    return out, sums


def jsd_bwd(g, sums, Si, Sp, shape):
    B, C, H, W = shape
    Si, ibs = _rows(Si)
    Sp, pbs = _rows(Sp)
    dev = sums.device
    g = g.contiguous()
    gL = torch.empty((B, 1, H, W), dtype=F32, device=dev)
    dSi = torch.empty((B, 1, H, W), dtype=F32, device=dev)
    dSp = torch.empty((B, 1, H, W), dtype=F32, device=dev)
    _lib.call("onet_jsd_bwd", _p(g), _p(sums), _p(Si), ibs, _p(Sp), pbs, _p(gL), _p(dSi), _p(dSp), B, H * W,
              _stream())
    return gL, dSi, dSp


def log1pexp_(x):
    require_gpu(x)
    if not x.is_contiguous():
        raise ValueError("log1pexp_: contiguous tensor required (it is mutated in place)")
    _lib.call("onet_log1pexp_inplace", _p(x), x.numel(), _stream())
    return x


def log1pexp_bwd(x_orig, g):
    g = g.contiguous()
    out = torch.empty_like(x_orig)
    _lib.call("onet_log1pexp_bwd", _p(x_orig), _p(g), _p(out), x_orig.numel(), _stream())
    return out


def argmax2(S):
    require_gpu(S)
    S = S.contiguous()
    B, C, H, W = S.shape
    if C != 2:
        raise ValueError("argmax2: S must have 2 channels")
    Y = torch.empty((B, H, W), dtype=torch.int64, device=S.device)
    _lib.call("onet_argmax2", _p(S), _p(Y), B, H * W, _stream())
    return Y


# ----------------------------------------------------------------------------- optimizer
def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    require_gpu(p, g, m, v)
    _lib.call("onet_adam_step", _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2),
              float(eps), float(weight_decay), int(step), float(grad_scale), _stream(), nbytes=28 * p.numel())
