"""ctypes binding of libedtts_hip.so (C ABI declared in include/edtts.h).

This is the only door between the Python host classes and the HIP kernels.  There is NO fallback: if the
shared library is missing, or a tensor is not a contiguous fp32/int64 tensor on a HIP device, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EDTTS_LIB", os.path.join(os.path.dirname(_PKG_DIR), "lib", "libedtts_hip.so"))

# every symbol include/edtts.h declares (tests check that the built library exports all of them)
EXPORTED_SYMBOLS = (
    "edtts_version", "edtts_last_error", "edtts_num_global_slots", "edtts_num_layer_slots", "edtts_global_slot_name",
    "edtts_layer_slot_name", "edtts_packed_bytes", "edtts_pack_weights", "edtts_workspace_bytes", "edtts_decoder_forward",
    "edtts_ddim_step", "edtts_ddpm_step", "edtts_generate", "edtts_sample_ddpm", "edtts_sample_multistep", "edtts_dsconv_forward", "edtts_profile_enable",
    "edtts_profile_collect", "edtts_randn", "edtts_index_errors", "edtts_sample_inpaint",
    "edtts_mel_to_spec", "edtts_griffin_lim_scratch_floats", "edtts_griffin_lim", "edtts_set_substreams", "edtts_set_coop", "edtts_dsconv_scratch_floats", "edtts_substreams_for",
)


class EdttsDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "hidden", "layers", "heads", "n_mels", "ffn_mult", "codebook_size", "semantic_dim", "window", "max_pos",
        "max_ctx_pos", "n_step_emb", "compute_dtype")]


COMPUTE_DTYPES = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}


class EdttsError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built -- there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EdttsError(
            f"HIP extension not found at {LIB_PATH}; build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The sampler path has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, sz, f32 = C.c_void_p, C.c_int, C.c_size_t, C.c_float
    L.edtts_version.restype = i32
    L.edtts_last_error.restype = C.c_char_p
    L.edtts_num_global_slots.restype = i32
    L.edtts_num_layer_slots.restype = i32
    L.edtts_global_slot_name.restype = C.c_char_p
    L.edtts_global_slot_name.argtypes = [i32]
    L.edtts_layer_slot_name.restype = C.c_char_p
    L.edtts_layer_slot_name.argtypes = [i32]
    L.edtts_packed_bytes.argtypes = [C.POINTER(EdttsDims), C.POINTER(sz)]
    L.edtts_pack_weights.argtypes = [C.POINTER(EdttsDims), C.POINTER(vp), i32, vp, vp]
    L.edtts_workspace_bytes.argtypes = [C.POINTER(EdttsDims), i32, i32, i32, i32, C.POINTER(sz)]
    L.edtts_decoder_forward.argtypes = [C.POINTER(EdttsDims), vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.edtts_ddim_step.argtypes = [vp, i32, vp, vp, vp, vp, i32, sz, f32, vp, vp, vp, vp]
    L.edtts_ddpm_step.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, i32, sz, vp, vp, vp]
    L.edtts_generate.argtypes = [C.POINTER(EdttsDims), vp, vp, i32, i32, vp, vp, i32, C.POINTER(C.c_int64),
                                 C.POINTER(f32), vp, vp, vp]
    L.edtts_sample_ddpm.argtypes = [C.POINTER(EdttsDims), vp, vp, i32, i32, vp, vp, i32, vp, C.POINTER(f32), vp, C.c_uint64, C.c_int64, vp, vp]
    L.edtts_randn.argtypes = [vp, sz, C.c_uint64, C.c_uint32, C.c_uint64, f32, vp]
    L.edtts_index_errors.argtypes = [vp, C.POINTER(i32), vp]
    L.edtts_mel_to_spec.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
    L.edtts_griffin_lim_scratch_floats.argtypes = [i32, i32, i32, i32, C.POINTER(sz)]
    L.edtts_griffin_lim.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32, f32, f32, vp, C.c_uint64, vp, vp, vp]
    L.edtts_sample_inpaint.argtypes = [C.POINTER(EdttsDims), vp, vp, vp, i32, i32, i32, vp, vp, vp, i32, vp, vp, C.POINTER(f32), vp, i32,
                                       vp, C.c_uint64, f32, vp, vp]
    L.edtts_sample_multistep.argtypes = [C.POINTER(EdttsDims), vp, vp, i32, i32, i32, vp, vp, vp, i32, C.POINTER(C.c_int64),
                                         C.POINTER(f32), vp, vp, vp, vp]
    L.edtts_dsconv_forward.argtypes = [vp] * 6 + [i32] * 7 + [vp, vp, vp]
    L.edtts_dsconv_scratch_floats.argtypes = [i32] * 7 + [C.POINTER(sz)]
    L.edtts_profile_enable.argtypes = [i32]
    L.edtts_set_substreams.argtypes = [i32]
    L.edtts_set_substreams.restype = i32
    L.edtts_set_coop.argtypes = [i32]
    L.edtts_set_coop.restype = i32
    L.edtts_substreams_for.argtypes = [C.POINTER(EdttsDims), i32, i32]
    L.edtts_substreams_for.restype = i32
    L.edtts_profile_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(i32)]  # arrays of 2
    for name in EXPORTED_SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("edtts_version", "edtts_num_global_slots", "edtts_num_layer_slots", "edtts_set_substreams", "edtts_set_coop", "edtts_substreams_for"):
            fn.errcheck = _errcheck
    _lib = L
    return L


def _errcheck(rc, func, args):
    if rc != 0:
        msg = _lib.edtts_last_error().decode() if _lib is not None else "?"
        # -2 = argument errors: the reference raises ValueError / IndexError / RuntimeError for these
        if rc == -2 and "Either sem_idx or sem_features" in msg:
            raise ValueError(msg)
        raise EdttsError(f"{func.__name__} failed (code {rc}): {msg}")
    return rc


# ---------------------------------------------------------------------------------------------- helpers
def _dev_ptr(t: Optional[torch.Tensor], dtype: torch.dtype, name: str) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise EdttsError(f"{name}: expected a tensor on the HIP device, got {t.device} -- the MI355X sampler path has no CPU fallback")
    if t.dtype != dtype:
        raise EdttsError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise EdttsError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def slot_names(n_layers: int) -> List[str]:
    L = lib()
    names = [L.edtts_global_slot_name(i).decode() for i in range(L.edtts_num_global_slots())]
    per = [L.edtts_layer_slot_name(i).decode() for i in range(L.edtts_num_layer_slots())]
    for l in range(n_layers):
        names += [f"layers.{l}.{n}" for n in per]
    return names


def packed_bytes(dims: EdttsDims) -> int:
    out = C.c_size_t(0)
    lib().edtts_packed_bytes(C.byref(dims), C.byref(out))
    return out.value


def workspace_bytes(dims: EdttsDims, B: int, T: int, S: int, cond_rows: int) -> int:
    out = C.c_size_t(0)
    lib().edtts_workspace_bytes(C.byref(dims), B, T, S, cond_rows, C.byref(out))
    return out.value


def pack_weights(dims: EdttsDims, tensors: Sequence[torch.Tensor], packed: torch.Tensor) -> None:
    ptrs = (C.c_void_p * len(tensors))(*[_dev_ptr(t, torch.float32, f"weight[{i}]") for i, t in enumerate(tensors)])
    lib().edtts_pack_weights(C.byref(dims), ptrs, len(tensors), _dev_ptr(packed, torch.uint8, "packed"), _stream(packed.device))


def decoder_forward(dims: EdttsDims, packed: torch.Tensor, workspace: torch.Tensor, x: torch.Tensor, t: torch.Tensor,
                    step_idx: Optional[torch.Tensor], sem_idx: Optional[torch.Tensor], sem_features: Optional[torch.Tensor],
                    S: int) -> torch.Tensor:
    B, T, M = x.shape
    eps = torch.empty_like(x)
    lib().edtts_decoder_forward(
        C.byref(dims), packed.data_ptr(), workspace.data_ptr(), B, T, S, _dev_ptr(x, torch.float32, "x_t"),
        _dev_ptr(t, torch.int64, "t"), _dev_ptr(step_idx, torch.int64, "step_idx"), _dev_ptr(sem_idx, torch.int64, "sem_idx"),
        _dev_ptr(sem_features, torch.float32, "sem_features"), eps.data_ptr(), _stream(x.device))
    check_indices(workspace)
    return eps


def generate(dims: EdttsDims, packed: torch.Tensor, workspace: torch.Tensor, sem_idx: torch.Tensor, x_T: torch.Tensor,
             timesteps: Sequence[int], coefs: Sequence[Tuple[float, float, float, float]]) -> torch.Tensor:
    B, S = sem_idx.shape
    n = len(timesteps)
    ts = (C.c_int64 * n)(*[int(v) for v in timesteps])
    flat = [float(v) for c in coefs for v in c]
    cf = (C.c_float * (4 * n))(*flat)
    x_work = torch.empty_like(x_T)
    x0 = torch.empty_like(x_T)
    lib().edtts_generate(C.byref(dims), packed.data_ptr(), workspace.data_ptr(), B, S, _dev_ptr(sem_idx, torch.int64, "sem_idx"),
                         _dev_ptr(x_T, torch.float32, "x_T"), n, ts, cf, x_work.data_ptr(), x0.data_ptr(), _stream(x_T.device))
    check_indices(workspace)
    return x0


def sample_ddpm(dims: EdttsDims, packed: torch.Tensor, workspace: torch.Tensor, sem_idx: torch.Tensor, x_T: torch.Tensor,
                t_all: torch.Tensor, coefs: Sequence[Tuple[float, float, float]], noise_all: Optional[torch.Tensor], seed: int,
                batch_offset: int = 0) -> torch.Tensor:
    B, S = sem_idx.shape
    n = t_all.numel()
    flat = [float(v) for c in coefs for v in c]
    cf = (C.c_float * (3 * n))(*flat)
    out = torch.empty_like(x_T)
    lib().edtts_sample_ddpm(C.byref(dims), packed.data_ptr(), workspace.data_ptr(), B, S, _dev_ptr(sem_idx, torch.int64, "sem_idx"),
                            _dev_ptr(x_T, torch.float32, "x_T"), n, _dev_ptr(t_all, torch.int64, "t_all"), cf,
                            _dev_ptr(noise_all, torch.float32, "noise"), C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), C.c_int64(int(batch_offset)),
                            out.data_ptr(), _stream(x_T.device))
    check_indices(workspace)
    return out


def sample_multistep(dims: EdttsDims, packed: torch.Tensor, workspace: torch.Tensor, sem_idx: Optional[torch.Tensor],
                     sem_features: Optional[torch.Tensor], S: int, x_T: torch.Tensor, timesteps: Sequence[int],
                     coefs: Sequence[Sequence[float]], want_intermediates: bool):
    B, T, M = x_T.shape
    n = len(timesteps)
    ts = (C.c_int64 * n)(*[int(v) for v in timesteps])
    cf = (C.c_float * (8 * n))(*[float(v) for c in coefs for v in c])
    hist = torch.empty((2, B, T, M), dtype=torch.float32, device=x_T.device)
    x0_all = torch.empty((n, B, T, M), dtype=torch.float32, device=x_T.device) if want_intermediates else None
    out = torch.empty_like(x_T)
    lib().edtts_sample_multistep(C.byref(dims), packed.data_ptr(), workspace.data_ptr(), B, T, S, _dev_ptr(sem_idx, torch.int64, "sem_idx"),
                                 _dev_ptr(sem_features, torch.float32, "sem_features"), _dev_ptr(x_T, torch.float32, "x_T"), n, ts, cf,
                                 hist.data_ptr(), None if x0_all is None else x0_all.data_ptr(), out.data_ptr(), _stream(x_T.device))
    check_indices(workspace)
    return out, x0_all


def ddim_step(alpha_bar: torch.Tensor, x_t: torch.Tensor, t: torch.Tensor, t_prev: torch.Tensor, eps: torch.Tensor, eta: float,
              noise: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    if x_t.shape != eps.shape:
        raise EdttsError(f"x_t {tuple(x_t.shape)} and eps_pred {tuple(eps.shape)} differ")
    B = x_t.shape[0]
    check_table_index(t, alpha_bar.numel(), "t")
    check_table_index(t_prev, alpha_bar.numel(), "t_prev", lo=-alpha_bar.numel())  # negative = "before the first step"
    x_prev, x0 = torch.empty_like(x_t), torch.empty_like(x_t)
    lib().edtts_ddim_step(_dev_ptr(alpha_bar, torch.float32, "alpha_bar"), alpha_bar.numel(), _dev_ptr(x_t, torch.float32, "x_t"),
                          _dev_ptr(eps, torch.float32, "eps_pred"), _dev_ptr(t, torch.int64, "t"), _dev_ptr(t_prev, torch.int64, "t_prev"),
                          B, x_t.numel() // B, float(eta), _dev_ptr(noise, torch.float32, "noise"), x_prev.data_ptr(), x0.data_ptr(),
                          _stream(x_t.device))
    return x_prev, x0


def ddpm_step(alphas, alpha_bar, betas, post_var, x_t, t, eps, noise) -> torch.Tensor:
    B = x_t.shape[0]
    check_table_index(t, alpha_bar.numel(), "t")
    out = torch.empty_like(x_t)
    lib().edtts_ddpm_step(_dev_ptr(alphas, torch.float32, "alphas"), _dev_ptr(alpha_bar, torch.float32, "alpha_bar"),
                          _dev_ptr(betas, torch.float32, "betas"), _dev_ptr(post_var, torch.float32, "posterior_variance"),
                          alpha_bar.numel(), _dev_ptr(x_t, torch.float32, "x_t"), _dev_ptr(eps, torch.float32, "eps_pred"),
                          _dev_ptr(t, torch.int64, "t"), B, x_t.numel() // B, _dev_ptr(noise, torch.float32, "noise"), out.data_ptr(),
                          _stream(x_t.device))
    return out


def dsconv_forward(x, dw, pw, pb, gn_w, gn_b, groups: int, stride: int = 1) -> torch.Tensor:
    B, Ci, T = x.shape
    Co, ks = pw.shape[0], dw.shape[-1]
    To = (T + 2 * (ks // 2) - ks) // stride + 1
    if To < 1:
        raise EdttsError(f"dsconv: no output frames for T={T}, kernel_size={ks}, stride={stride}")
    y = torch.empty(B, Co, To, device=x.device, dtype=torch.float32)
    need = C.c_size_t(0)
    lib().edtts_dsconv_scratch_floats(B, Ci, Co, T, ks, int(stride), groups, C.byref(need))  # 0: the one-kernel path
    scratch = torch.empty(need.value, device=x.device, dtype=torch.float32) if need.value else None
    f = torch.float32
    lib().edtts_dsconv_forward(_dev_ptr(x, f, "x"), _dev_ptr(dw, f, "depthwise.weight"), _dev_ptr(pw, f, "pointwise.weight"),
                               _dev_ptr(pb, f, "pointwise.bias"), _dev_ptr(gn_w, f, "norm.weight"), _dev_ptr(gn_b, f, "norm.bias"),
                               B, Ci, Co, T, ks, int(stride), groups, None if scratch is None else scratch.data_ptr(), y.data_ptr(),
                               _stream(x.device))
    return y


def randn(shape, device, seed: int = 0, stream_id: int = 0, elem_offset: int = 0, scale: float = 1.0) -> torch.Tensor:
    """scale * N(0, 1) from the library's Philox stream (seed, stream_id) at global element offset `elem_offset`: the start
    noise of a batch shard, identical to what one GPU would draw for the same rows (include/edtts.h: edtts_randn)."""
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    if not out.is_cuda:
        raise EdttsError(f"randn: expected a HIP device, got {out.device} -- the MI355X sampler path has no CPU fallback")
    if out.numel() == 0:  # the empty shard of a rank without utterances: nothing to draw (data_ptr() is NULL)
        return out
    lib().edtts_randn(out.data_ptr(), out.numel(), C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), C.c_uint32(stream_id & 0xFFFFFFFF),
                      C.c_uint64(int(elem_offset)), float(scale), _stream(out.device))
    return out


CHECK_INDICES = os.environ.get("EDTTS_CHECK_INDICES", "0") == "1"


def index_errors(workspace: torch.Tensor) -> int:
    """EDTTS_IDX_* bits recorded since the last call (synchronises the current stream)."""
    flags = C.c_int(0)
    lib().edtts_index_errors(workspace.data_ptr(), C.byref(flags), _stream(workspace.device))
    return flags.value


def check_indices(workspace: torch.Tensor) -> None:
    """Debug mode (EDTTS_CHECK_INDICES=1 or native.CHECK_INDICES = True): raise IndexError where the reference would have --
    the kernels clamp out-of-range token / step indices instead of faulting (include/edtts.h: edtts_index_errors)."""
    if not CHECK_INDICES or torch.cuda.is_current_stream_capturing():
        return
    flags = index_errors(workspace)
    if flags:
        what = [n for b, n in ((1, "sem_idx outside [0, codebook_size)"), (2, "step_idx outside [0, n_step_emb)")) if flags & b]
        raise IndexError("index out of range in the decoder call: " + "; ".join(what))


def check_table_index(t: torch.Tensor, n: int, name: str, lo: int = 0) -> None:
    if CHECK_INDICES and not torch.cuda.is_current_stream_capturing() and bool(((t < lo) | (t >= n)).any()):
        raise IndexError(f"{name}: index out of range [{lo}, {n})")


def set_substreams(n: int) -> int:
    """1: every sampler call runs its batch in one piece; n >= 2 (default 4): large batches are cut into up to n sub-batches on as many
    streams (include/edtts.h: edtts_set_substreams).  Returns the previous setting."""
    return int(lib().edtts_set_substreams(int(n)))


def substreams_for(dims: EdttsDims, B: int, T: int) -> int:
    """Sub-batches a sampler call of this shape makes under the current setting."""
    return int(lib().edtts_substreams_for(C.byref(dims), int(B), int(T)))


def set_coop(mode: int) -> int:
    """-1: the cooperative layer kernel is chosen automatically for small grids (default); 0: never; 14 / 24 / 22: force an
    instance (include/edtts.h: edtts_set_coop).  Returns the previous mode."""
    return int(lib().edtts_set_coop(int(mode)))


def profile_enable(max_records: int) -> None:
    lib().edtts_profile_enable(int(max_records))


def profile_collect():
    """((ms, launches) of the fused-layer / attention-half kernels, (ms, launches) of the FFN + tail half kernels) recorded
    since the last call."""
    ms, n = (C.c_double * 2)(), (C.c_int * 2)()
    lib().edtts_profile_collect(ms, n)
    return (ms[0], n[0]), (ms[1], n[1])
