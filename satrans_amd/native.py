"""ctypes binding of libsatrans_hip.so (the C ABI declared in include/satrans_hip.h).

There is no CPU fallback: if the library cannot be loaded, or a call returns an error code, this
module raises.  Tensors cross the boundary as raw device pointers (`tensor.data_ptr()`) plus sizes;
the stream is torch's current HIP stream so that every kernel is ordered with the surrounding
torch plumbing (allocation, tiny scenario-encoder ops, RCCL collectives).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import torch

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue serialise.  The
# data-parallel step runs five (launch stream, next-batch stream, tail stream, two RCCL communicator streams): on four queues
# its tail chains - gradient rows to their owners beside the slab reduction and the all-reduce - ran one after the other
# (measured, one rank through RCCL: 1.324 -> 1.214 ms/step with 8 queues; the one-GPU step is unchanged).  A default only: it has
# no effect when the caller's environment sets it, or when the HIP runtime was initialised before this import.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SATRANS_LIB_PATH") or os.path.join(_HERE, "libsatrans_hip.so")   # (override: kernel experiments)
ABI_VERSION = 6

ID_F32, ID_I32, ID_I64 = 0, 1, 2
META_Q, META_K, RELU_OUT, NO_RES, TRAIN, GATE, BILINEAR = 1, 2, 4, 8, 16, 32, 64
X_SORTED, Y_SORTED = 128, 256      # general path: activations stay in scenario-sorted order between the layers of a stack

_c_f32p = C.c_void_p
_vp = C.c_void_p


class LayerDesc(C.Structure):
    """Mirror of `satrans_layer_desc`."""
    _fields_ = [
        ("B", C.c_int32), ("F", C.c_int32), ("D", C.c_int32), ("H", C.c_int32),
        ("U", C.c_int32), ("S", C.c_int32), ("flags", C.c_int32), ("layer", C.c_int32),
        ("drop_p", C.c_float), ("seed", C.c_uint32), ("step", C.c_uint32),
        ("tab_stride", C.c_int64),
        ("x", _vp), ("sid", _vp), ("order", _vp), ("seg", _vp),
        ("w_query", _vp), ("w_key", _vp), ("w_value", _vp), ("w_out", _vp),
        ("ln_g", _vp), ("ln_b", _vp),
        ("lnq_g", _vp), ("lnq_b", _vp), ("lnk_g", _vp), ("lnk_b", _vp),
        ("tab_q", _vp), ("tab_k", _vp),
        ("x_rows", _vp),
        ("attn_save", _vp),
    ]


class HeadDesc(C.Structure):
    """Mirror of `satrans_head_desc` (the head operands of satrans_layer_bwd_head)."""
    _fields_ = [("w", _vp), ("bias", _vp), ("labels", _vp), ("dense", _vp), ("dense_stride", C.c_int64),
                ("h_dense_cols", C.POINTER(C.c_int32)), ("n_dense", C.c_int32), ("loss_kind", C.c_int32),
                ("prob", _vp), ("logit", _vp), ("loss_sum", _vp), ("g_w", _vp), ("g_b", _vp), ("scratch", _vp)]


class LayerGrads(C.Structure):
    """Mirror of `satrans_layer_grads`."""
    _fields_ = [(k, _vp) for k in ("g_wq", "g_wk", "g_wv", "g_wo", "g_ln", "g_lnq", "g_lnk", "g_tab_q", "g_tab_k")]


class SelfAttDesc(C.Structure):
    """Mirror of `satrans_selfatt_desc`."""
    _fields_ = [("B", C.c_int32), ("F", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("flags", C.c_int32),
                ("layer", C.c_int32), ("drop_p", C.c_float), ("seed", C.c_uint32), ("step", C.c_uint32),
                ("x", _vp), ("w_query", _vp), ("w_key", _vp), ("w_value", _vp), ("w_res", _vp), ("ln_g", _vp), ("ln_b", _vp)]


class MetaNetDesc(C.Structure):
    """Mirror of `satrans_metanet_desc`."""
    _fields_ = [("B", C.c_int32), ("F", C.c_int32), ("D", C.c_int32), ("U", C.c_int32), ("S", C.c_int32), ("flags", C.c_int32),
                ("layer", C.c_int32), ("drop_p", C.c_float), ("seed", C.c_uint32), ("step", C.c_uint32),
                ("tab_stride", C.c_int64), ("x", _vp), ("order", _vp), ("seg", _vp), ("tab", _vp), ("ln_g", _vp), ("ln_b", _vp)]


class AdamHParams(C.Structure):
    """Mirror of `satrans_adam_hparams`."""
    _fields_ = [("lr_over_bc1", C.c_float), ("bc2_sqrt", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("l2", C.c_float), ("arith", C.c_int32)]


ADAM_EXACT, ADAM_FAST = 0, 1      # satrans_adam_hparams.arith (SATRANS_ADAM_EXACT / SATRANS_ADAM_FAST)


# name -> (restype, argtypes); tests/test_host_cpu.py::test_library_exports_every_declared_symbol checks this table against the header
SIGNATURES = {
    "satrans_last_error": (C.c_char_p, []),
    "satrans_abi_version": (C.c_int, []),
    "satrans_stream_create_low_priority": (C.c_int, [C.POINTER(C.c_void_p)]),
    "satrans_stream_destroy": (C.c_int, [C.c_void_p]),
    "satrans_host_randperm": (C.c_int, [C.c_uint64, C.c_int64, _vp, _vp]),
    "satrans_kernel_timing": (C.c_int, [C.c_int]),
    "satrans_kernel_timing_read": (C.c_int, [_vp, _vp, C.c_int]),
    "satrans_bucket_workspace_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "satrans_bucket_scenarios": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp,
                                           _vp, C.c_int64, _vp]),
    "satrans_gather_fwd": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, _vp, _vp,
                                     _vp, _vp]),
    "satrans_gather_read_probe_floats": (C.c_int64, []),
    "satrans_gather_read_probe": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp]),
    "satrans_set_layer_impl": (C.c_int, [C.c_int]),
    "satrans_layer_fused_supported": (C.c_int, [C.POINTER(LayerDesc)]),
    "satrans_layer_fwd": (C.c_int, [C.POINTER(LayerDesc), _vp, _vp, _vp]),
"satrans_layer_fwd_bf16_supported": (C.c_int, [C.POINTER(LayerDesc)]),
    "satrans_layer_fwd_bf16": (C.c_int, [C.POINTER(LayerDesc), _vp, _vp]),
    "satrans_stack_fwd_bf16_supported": (C.c_int, [C.c_int, C.POINTER(C.POINTER(LayerDesc))]),
    "satrans_stack_fwd_bf16": (C.c_int, [C.c_int, C.POINTER(C.POINTER(LayerDesc)), _vp, _vp]),
    "satrans_stack_fwd_bf16_head": (C.c_int, [C.c_int, C.POINTER(C.POINTER(LayerDesc)), C.POINTER(HeadDesc), _vp]),
"satrans_layer_generic_supported": (C.c_int, [C.POINTER(LayerDesc)]),
    "satrans_layer_generic_saved_floats": (C.c_int64, [C.POINTER(LayerDesc)]),
    "satrans_layer_generic_scratch_floats": (C.c_int64, [C.POINTER(LayerDesc)]),
    "satrans_layer_fwd_generic": (C.c_int, [C.POINTER(LayerDesc), _vp, _vp, _vp, _vp]),
    "satrans_layer_bwd_generic": (C.c_int, [C.POINTER(LayerDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                            _vp, _vp]),
    "satrans_set_generic_attention": (C.c_int, [C.c_int]),
"satrans_selfatt_saved_floats": (C.c_int64, [C.POINTER(SelfAttDesc)]),
    "satrans_selfatt_scratch_floats": (C.c_int64, [C.POINTER(SelfAttDesc)]),
    "satrans_selfatt_fwd": (C.c_int, [C.POINTER(SelfAttDesc), _vp, _vp, _vp, _vp]),
    "satrans_selfatt_bwd": (C.c_int, [C.POINTER(SelfAttDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "satrans_metanet_saved_floats": (C.c_int64, [C.POINTER(MetaNetDesc)]),
    "satrans_metanet_scratch_floats": (C.c_int64, [C.POINTER(MetaNetDesc)]),
    "satrans_metanet_fwd": (C.c_int, [C.POINTER(MetaNetDesc), _vp, _vp, _vp]),
    "satrans_metanet_bwd": (C.c_int, [C.POINTER(MetaNetDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "satrans_layer_bwd_slab_floats": (C.c_int64, [C.POINTER(LayerDesc)]),
    "satrans_layer_attn_save_floats": (C.c_int64, [C.POINTER(LayerDesc)]),
    "satrans_batch_metrics": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "satrans_layer_bwd": (C.c_int, [C.POINTER(LayerDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp]),
    "satrans_layer_bwd_head_supported": (C.c_int, [C.POINTER(LayerDesc), C.POINTER(HeadDesc)]),
    "satrans_layer_bwd_head_scratch_floats": (C.c_int64, [C.POINTER(LayerDesc), C.c_int]),
    "satrans_layer_bwd_head": (C.c_int, [C.POINTER(LayerDesc), C.POINTER(HeadDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                         _vp, _vp, _vp]),
    "satrans_layer_bwd_deferred_supported": (C.c_int, [C.POINTER(LayerDesc)]),
    "satrans_layer_bwd_launch": (C.c_int, [C.POINTER(LayerDesc), _vp, _vp, _vp, _vp]),
    "satrans_layer_bwd_head_launch": (C.c_int, [C.POINTER(LayerDesc), C.POINTER(HeadDesc), _vp, _vp, _vp]),
    "satrans_layer_bwd_reduce": (C.c_int, [C.c_int, C.POINTER(C.POINTER(LayerDesc)), C.POINTER(_vp), C.POINTER(LayerGrads),
                                           C.POINTER(HeadDesc), _vp]),
    "satrans_head_scratch_floats": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "satrans_head": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, _vp, _vp, _vp]),
"satrans_head_loss": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "satrans_optim_flat": (C.c_int, [C.c_int, _vp, _vp, _vp, C.c_int64, C.c_float, C.c_float, C.c_float, _vp]),
    "satrans_adam_flat": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.POINTER(AdamHParams), _vp]),
    "satrans_adam_flat_sum": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.POINTER(AdamHParams), _vp, C.c_int64, _vp, _vp]),
    "satrans_embed_sort_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int64]),
    "satrans_embed_sort": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp]),
    "satrans_embed_reg_partials": (C.c_int64, [C.c_int64, C.c_int64, C.c_int]),
    "satrans_embed_partial_ws_floats": (C.c_int64, [C.c_int64, C.c_int]),
    "satrans_scenario_table_fwd": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "satrans_scenario_table_bwd_ws_floats": (C.c_int64, [C.c_int, C.c_int]),
    "satrans_scenario_table_bwd": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "satrans_scenario_inputs_fwd": (C.c_int, [C.POINTER(_vp), _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp]),
    "satrans_scenario_inputs_bwd": (C.c_int, [C.POINTER(_vp), C.POINTER(C.c_int32), _vp, C.c_int, C.c_int, C.c_int, _vp, _vp,
                                              _vp, C.c_int, _vp]),
    "satrans_scenario_relu_fwd": (C.c_int, [_vp, C.c_int64, _vp, _vp]),
    "satrans_scenario_relu_bwd": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp]),
    "satrans_embed_adam_touched": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int64, _vp, _vp,
                                             C.POINTER(AdamHParams), _vp, _vp, C.c_int, _vp]),
    "satrans_embed_adam_untouched": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int, _vp, C.POINTER(AdamHParams),
                                               _vp, C.c_int, _vp]),
    "satrans_embed_mark_touched": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp, _vp]),
    "satrans_embed_segment_sums": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "satrans_embed_adam_rows_partials": (C.c_int64, [C.c_int64, C.c_int]),
    "satrans_embed_adam_rows": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int, _vp, C.POINTER(AdamHParams),
                                          C.c_int, _vp, _vp]),
    "satrans_embed_pack_rows": (C.c_int, [_vp, C.c_int64, _vp, C.c_int, _vp, _vp]),
    "satrans_embed_rows_sort_fields": (C.c_int, [_vp, C.c_int, C.c_int64, _vp, _vp, _vp, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32), _vp, _vp, _vp, _vp]),
    "satrans_embed_merge_runs": (C.c_int, [_vp, C.c_int64, C.POINTER(C.c_int64), C.c_int, _vp, _vp, _vp]),
    "satrans_embed_inverse_positions": (C.c_int, [_vp, C.c_int64, _vp, _vp]),
    "satrans_embed_sort_fields": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                            _vp, _vp, _vp]),
    "satrans_embed_lazy_reg_partials": (C.c_int64, [C.c_int64, C.c_int]),
    "satrans_embed_lazy_replay": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int64, C.c_int, _vp,
                                            C.POINTER(AdamHParams), _vp, _vp]),
    "satrans_embed_lazy_flush": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, _vp, C.POINTER(AdamHParams),
                                           C.c_int64, _vp, _vp]),
    "satrans_embed_lazy_mark": (C.c_int, [_vp, C.c_int64, _vp, C.c_int, _vp]),
    "satrans_debug_check_packed_math": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), _vp]),
    "satrans_embed_grad_dense": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp, C.c_int64, C.c_int, C.c_float, _vp, _vp]),
    "satrans_sum_f64": (C.c_int, [_vp, C.c_int64, _vp, C.c_int, _vp]),
}

_lib: Optional[C.CDLL] = None


class NativeError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU).  Returns the .so path."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    if force:
        for f in os.listdir(os.path.join(_HERE, "csrc", "build")) if os.path.isdir(os.path.join(_HERE, "csrc", "build")) else []:
            os.remove(os.path.join(_HERE, "csrc", "build", f))
    subprocess.run(["bash", script], check=True)
    return LIB_PATH


def source_hash() -> str:
    """sha256 over the kernel sources the library is built from (csrc/*.hip, csrc/*.h, include/*.h; names and bytes, sorted).
    Profiles under profiles/ record it, so a counter summary can be matched to the code it was taken on - the hash of the
    .so would differ between two builds of identical sources."""
    import hashlib
    h = hashlib.sha256()
    inc = os.path.join(os.path.dirname(_HERE), "include")
    files = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc")) if f.endswith((".hip", ".h"))]
    files += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    for path in sorted(files, key=os.path.basename):
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()


def lib() -> C.CDLL:
    """Load (once) and type the shared library.  Raises NativeError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} is missing: the SATrans hot path has no CPU/eager fallback. "
            f"Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `bash satrans_amd/csrc/build.sh`.")
    try:
        handle = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the box
        raise NativeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise NativeError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype, fn.argtypes = res, args
    if handle.satrans_abi_version() != ABI_VERSION:
        raise NativeError(f"ABI version mismatch: library {handle.satrans_abi_version()}, binding {ABI_VERSION}")
    _lib = handle
    return handle


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().satrans_last_error()
        raise NativeError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous tensor (None stays NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise NativeError("non-contiguous tensor handed to the native library")
    return t.data_ptr()


def stream_handle(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise NativeError(
            f"{what}: tensor is on '{t.device}'. The SATrans hot path runs only as HIP kernels on an MI355X; "
            f"construct the model with device='cuda:0' (there is no CPU fallback; the CPU restatement lives in "
            f"oracle/ and is test infrastructure).")


def id_dtype_of(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return ID_F32
    if t.dtype == torch.int32:
        return ID_I32
    if t.dtype == torch.int64:
        return ID_I64
    raise NativeError(f"input matrix dtype {t.dtype} is not one of float32/int32/int64")
