"""ctypes binding of csrc/libcnrma_hip.so (the C-ABI declared in include/cnrma.h).

The product path has NO fallback: if the library is missing or a symbol is absent this module raises.

Two libraries are built from the same sources: libcnrma_hip.so (the product: shipped kernels only, no cnrma_debug_* entry point,
no tuning state) and libcnrma_hip_exp.so (-DCNRMA_EXPERIMENTS: + the measured-and-rejected kernel forms behind
cnrma_debug_conv_tuning / cnrma_debug_dense_tuning).  Product code only ever sees the first; `experiments(True)` -- called by
sparse.conv_tuning(...) / rma.dense_tuning(...) with arguments, i.e. by scripts/ and by the bit-identity tests -- routes this
process's calls through the second until `experiments(False)`.
"""
import ctypes
import threading
import os
from ctypes import c_double, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcnrma_hip.so")
EXP_LIB_PATH = os.path.join(_HERE, "csrc", "libcnrma_hip_exp.so")

P, I, L, F = c_void_p, c_int, c_int64, c_float

# name -> (restype, argtypes); mirrors include/cnrma.h one to one
SIGNATURES = {
    "cnrma_fill_bytes_u8": (c_int, [P, I, c_size_t, P]),
    "cnrma_abi_version": (c_int, []),
    "cnrma_range_violations_i32": (c_int, [P, P, P, I, P, P]),
    "cnrma_nchw_to_nhwc_f32": (c_int, [P, P, I, I, I, I, P]),
    "cnrma_backproject_accum_f32": (c_int, [P, P, I, I, I, I, I, I, I, F, F, F, F, P, P, P, L, P]),
    "cnrma_backproject_accum_ref_f32": (c_int, [P, P, I, I, I, I, I, I, I, F, F, F, F, P, P, P, L, P]),
    "cnrma_backproject_backward_f32": (c_int, [P, P, P, I, I, I, I, I, I, I, F, F, F, F, P, P]),
    "cnrma_backproject_index_f32": (c_int, [P, I, I, I, I, I, F, F, F, F, P, P, P, P]),
    "cnrma_ray_params_f32": (c_int, [P, I, I, I, P, P, P]),
    "cnrma_rma_neus_count_f32": (c_int, [P, P, I, I, I, I, I, I, F, F, F, F, I, F, F, P, P, P]),
    "cnrma_rma_neus_emit_f32": (c_int, [P, P, P, I, I, I, I, I, I, I, F, F, F, F, I, F, F, P, P, P, F, F, F,
                                        P, I, P, I, P, I, P, P]),
    "cnrma_rma_neus_rows_backward_f32": (c_int, [P, I, I, I, I, I, P, P, I, P, P, P, P]),
    "cnrma_rma_sigmoid_table_f32": (c_int, [P, L, P, P]),
    "cnrma_rma_skip_table_bytes": (c_size_t, [I, I, I]),
    "cnrma_rma_march_tables_f32": (c_int, [P, I, I, I, P, P, P]),
    "cnrma_rma_neus_march_f32": (c_int, [P, P, P, I, I, I, I, I, I, F, F, F, F, I, F, F, P, P, P, I, P, P, P]),
    "cnrma_nchw_to_nhwc_march_f32": (c_int, [P, P, I, P, P, P, I, I, I, I, I, I, F, F, F, F, I, F, F, P, P, P, I, P, P, P]),
    "cnrma_rma_neus_emit_rows_f32": (c_int, [P, P, I, I, I, I, I, F, P, L, P, P, I, P, L, P, P, F, F, F, P, I, P, I, P, I, P, P]),
    "cnrma_rma_neus_emit_rows_ref_f32": (c_int, [P, P, I, I, I, I, I, F, P, L, P, P, I, P, L, P, P, F, F, F, P, I, P, I, P, I, P, P]),
    "cnrma_sample_workspace_bytes": (c_size_t, []),
    "cnrma_sample_mask": (c_int, [P, L, I, ctypes.c_uint32, P, P, P, P]),
    "cnrma_topk_mask_f32": (c_int, [P, P, L, I, P, P, P]),
    "cnrma_topk_indices_f32": (c_int, [P, P, L, I, P, P, P]),
    "cnrma_rma_select_records": (c_int, [P, L, P, I, P, L, I, ctypes.c_uint32, P, P, P, P, P, L, P, P, P]),
    "cnrma_rma_depth_count_f32": (c_int, [P, P, I, I, I, I, I, I, F, F, F, F, I, F, I, P, P, P]),
    "cnrma_rma_depth_emit_f32": (c_int, [P, P, P, I, I, I, I, I, I, I, F, F, F, F, I, F, I, P, P, P, F, F, F,
                                         P, I, P, I, P, I, L, L, P]),
    "cnrma_rma_drop_single_sample_views": (c_int, [P, P, I, L, P]),
    "cnrma_rma_mean_weight": (c_int, [P, P, P, P]),
    "cnrma_scan_workspace_bytes": (c_size_t, [L]),
    "cnrma_exclusive_scan_i32": (c_int, [P, P, L, P, P]),
    "cnrma_sum_f64": (c_int, [P, P, L, P, P]),
    "cnrma_mask_to_index": (c_int, [P, P, P, L, P, P]),
    "cnrma_select_rows_f32": (c_int, [P, L, I, P, F, F, F, P, P, P]),
    "cnrma_voxelize_workspace_bytes": (c_size_t, [L]),
    "cnrma_voxelize_f32": (c_int, [P, P, L, P, I, F, I, I, P, P, L, P, P, P, L, P, P, P]),
    "cnrma_sparse_build_map": (c_int, [P, L, P, P, P, L, I, P]),
    "cnrma_sparse_stride_coords": (c_int, [P, L, P, I, P, P, L, P, L, P, P, P]),
    "cnrma_sparse_stride_coords_sorted": (c_int, [P, L, P, I, P, L, P, P, P]),
    "cnrma_sparse_kernel_map": (c_int, [P, L, P, P, P, L, P, I, P, P]),
    "cnrma_sparse_kernel_map_symmetric": (c_int, [P, L, P, P, P, L, P, I, P, I, P]),
    "cnrma_sparse_kernel_map_children": (c_int, [P, L, P, P, P]),
    "cnrma_sparse_kernel_map_strided": (c_int, [P, L, P, I, I, P, P, L, P, L, I, P]),
    "cnrma_sparse_conv_workspace_bytes": (c_size_t, [L, I, I]),
    "cnrma_sparse_conv_plan": (c_int, [L, I, I, I, I, I, c_size_t, P]),
    "cnrma_sparse_conv_f32": (c_int, [P, I, P, I, P, I, P, P, P, I, P, L, P, P, c_size_t, P]),
    "cnrma_sparse_conv_weight_bytes": (c_size_t, [I, I, I]),
    "cnrma_sparse_conv_prepare_weights": (c_int, [P, I, I, I, P, P]),
    "cnrma_sparse_split_features": (c_int, [P, L, P, I, P, P]),
    "cnrma_sparse_conv_bf16x6": (c_int, [P, P, L, I, P, I, P, I, P, P, P, I, P, P, L, P, P, c_size_t, P]),
    "cnrma_sparse_convtr_gen_bf16x6": (c_int, [P, P, P, L, P, I, I, P, I, P, P, I, P, P, P, P]),
    "cnrma_amax_bytes": (c_size_t, []),
    "cnrma_absmax_f32": (c_int, [P, L, P, I, P, P]),
    "cnrma_rma_emit_features_f32": (c_int, [P, P, I, P, L, P, P, P, I, P, P]),
    "cnrma_sparse_conv_f16_weight_bytes": (c_size_t, [I, I, I]),
    "cnrma_sparse_conv_prepare_weights_f16": (c_int, [P, I, I, I, P, P]),
    "cnrma_sparse_conv_f16x3": (c_int, [P, P, I, P, I, P, I, P, P, P, I, P, P, L, P, P, c_size_t, P]),
    "cnrma_sparse_tile_union_bytes": (c_size_t, [L]),
    "cnrma_sparse_tile_union_build": (c_int, [P, L, P, I, P, P]),
    "cnrma_sparse_conv_prepare_weights_f16_frag": (c_int, [P, I, I, I, P, P]),
    "cnrma_sparse_conv_f32_frag_weight_bytes": (c_size_t, [I, I, I]),
    "cnrma_sparse_conv_prepare_weights_f32_frag": (c_int, [P, I, I, I, P, P]),
    "cnrma_sparse_conv_go_f32": (c_int, [P, I, P, P, I, P, P, P, I, P, L, P, P, c_size_t, P]),
    "cnrma_sparse_conv_go_plan": (c_int, [L, I, I, c_size_t, I, P]),
    "cnrma_sparse_conv_go_f16x3": (c_int, [P, P, I, P, P, I, P, P, P, I, P, P, L, P, P, c_size_t, P, P]),
    "cnrma_sparse_conv_pairs_workspace_bytes": (c_size_t, [L, I, I, L]),
    "cnrma_sparse_conv_pairs_f32": (c_int, [P, I, P, I, P, I, P, P, P, I, P, L, P, L, P, c_size_t, P]),
    "cnrma_sparse_conv_pairs_f16x3": (c_int, [P, P, I, P, I, P, I, P, P, P, I, P, P, L, P, L, P, c_size_t, P]),
    "cnrma_sparse_conv_bf16_weight_bytes": (c_size_t, [I, I, I]),
    "cnrma_sparse_conv_prepare_weights_bf16": (c_int, [P, I, I, I, P, P]),
    "cnrma_sparse_conv_prepare_weights_bf16_t": (c_int, [P, I, I, I, I, P, P]),
    "cnrma_sparse_conv_bf16": (c_int, [P, I, P, I, P, I, P, P, P, I, P, L, P, P, c_size_t, P]),
    "cnrma_sparse_convtr_gen_f16x3": (c_int, [P, P, P, L, P, I, I, P, I, P, P, I, P, P, P, P]),
    "cnrma_sparse_kernel_map_transpose": (c_int, [P, L, P, I, L, P, P]),
    "cnrma_sparse_conv_wgrad_chunks": (c_int, [L, I]),
    "cnrma_sparse_conv_wgrad_f32": (c_int, [P, I, P, I, P, I, L, P, I, P, P]),
    "cnrma_sparse_conv_wgrad_bf16": (c_int, [P, I, P, I, P, I, L, P, I, P, P]),
    "cnrma_sparse_conv_wgrad_go_bf16": (c_int, [P, I, P, P, I, L, P, I, P, P]),
    "cnrma_sparse_conv_bf16_frag_weight_bytes": (c_size_t, [I, I, I]),
    "cnrma_sparse_conv_prepare_weights_bf16_frag": (c_int, [P, I, I, I, I, I, P, P]),
    "cnrma_sparse_conv_prepare_weights_bf16_frag_pair": (c_int, [P, I, I, I, I, P, P, P]),
    "cnrma_sparse_conv_go_bf16": (c_int, [P, I, P, P, I, P, L, P, P, c_size_t, P]),
    "cnrma_sparse_convtr_gen_f32": (c_int, [P, P, L, P, I, I, P, I, P, P, I, P, P, P]),
    "cnrma_sparse_maxpool_f32": (c_int, [P, I, P, I, P, L, P, P]),
    "cnrma_instnorm_workspace_bytes": (c_size_t, [I]),
    "cnrma_sparse_instnorm_f32": (c_int, [P, L, P, P, I, P, P, F, I, P, P, P]),
    "cnrma_sparse_instnorm_maxpool_f32": (c_int, [P, I, P, P, P, F, I, P, I, P, L, P, P, P]),
    "cnrma_bn_backward_f32": (c_int, [P, P, L, I, P, P, F, P, P, P, P, P]),
    "cnrma_bn_train_forward_f32": (c_int, [P, L, I, P, P, F, P, I, F, P, P, P, P, P, P]),
    "cnrma_bn_train_backward_f32": (c_int, [P, P, P, I, L, I, P, P, F, P, P, P, P, P, P]),
    "cnrma_union_workspace_bytes": (c_size_t, [L]),
    "cnrma_sparse_union_add_f32": (c_int, [P, P, L, P, P, P, L, P, I, P, P, L, P, P, L, P, P, P]),
    "cnrma_sparse_interp_f32": (c_int, [P, L, P, P, P, P, L, I, P, P]),
    "cnrma_sparse_prune_f32": (c_int, [P, P, L, P, I, P, P, P, P]),
    "cnrma_rowmax_f32": (c_int, [P, L, P, I, P, P]),
    "cnrma_fcaf3d_decode_f32": (c_int, [P, P, I, L, I, P, P]),
    "cnrma_fcaf3d_head_post_f32": (c_int, [P, I, P, L, I, I, P, F, P, P, P, P, P, P]),
    "cnrma_fcaf3d_max_score_f32": (c_int, [P, P, L, I, P, P]),
    "cnrma_fcaf3d_select_decode_f32": (c_int, [P, L, P, P, P, P, I, I, I, P, P, P]),
    "cnrma_nms_mask_f32": (c_int, [P, I, F, I, P, P]),
    "cnrma_box_iou_f32": (c_int, [P, I, P, I, I, I, P, P]),
    "cnrma_fcaf3d_scores_f32": (c_int, [P, P, L, I, P, P, P]),
}

# libcnrma_hip_exp.so exports these on top of SIGNATURES (include/cnrma.h: the prototypes under #ifdef CNRMA_EXPERIMENTS)
EXPERIMENT_SIGNATURES = {
    "cnrma_debug_dense_tuning": (c_int, [P, I]),
    "cnrma_debug_div_by_voxel_size_f32": (c_int, [P, L, F, P, P, P]),
    "cnrma_debug_conv_tuning": (c_int, [P, I]),
}

_libs = {}            # False: product library, True: experiments library
_active = False       # which of the two load() / call() use
_exp_users = set()    # who asked for the experiments library ("conv", "dense", a test): product again when nobody is left
ABI_VERSION = 6


class CnrmaError(RuntimeError):
    pass


def load(experiments=None):
    """The library this process's calls go to, loaded once; raises loudly when it is not built (no CPU fallback exists).
    experiments: None = the active one (the product library unless experiments(True) was called), False / True = that one."""
    which = _active if experiments is None else bool(experiments)
    lib = _libs.get(which)
    if lib is not None:
        return lib
    path = EXP_LIB_PATH if which else LIB_PATH
    if not os.path.exists(path):
        raise CnrmaError(f"{path} not found: build it with `python __graft_entry__.py build` "
                         f"(or `make -C cn-rma_amd/csrc`). There is no CPU fallback for the product path.")
    lib = ctypes.CDLL(path)
    table = dict(SIGNATURES, **EXPERIMENT_SIGNATURES) if which else SIGNATURES
    for name, (res, args) in table.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.cnrma_abi_version() != ABI_VERSION:
        raise CnrmaError("ABI version mismatch")
    _libs[which] = lib
    return lib


def experiments(on=True, who="caller"):
    """route this process's C-ABI calls through libcnrma_hip_exp.so (True) or back through the product library (False, once
    every `who` that asked for it has let go).  The two libraries are the same sources; the experiments one adds kernels and the
    cnrma_debug_* switches.  Never called by product code."""
    global _active
    if on:
        load(True)
        _exp_users.add(who)
    else:
        _exp_users.discard(who)
    _active = bool(_exp_users)


def experiments_active():
    return _active


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "C-ABI buffers must be contiguous"
    if not t.is_cuda:                    # a host pointer handed to a kernel is a GPU memory fault, not an exception
        raise CnrmaError(f"C-ABI buffers must live on the GPU (got a {t.device} tensor of shape {tuple(t.shape)}): "
                         "move the module / tensor with .to(device) first")
    return t.data_ptr()


def stream():
    """raw hipStream_t of torch's current stream on the current device (fast path: no Stream object is built)"""
    import torch
    try:
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
    except AttributeError:
        return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise CnrmaError(f"{name} failed with code {rc}" + (" (invalid argument)" if rc == -22 else " (-hipError_t)"))
    return rc


_tls = threading.local()


def read_ints(t):
    """device int tensor -> list of host ints, as an asynchronous copy into this thread's pinned buffer followed by a
    wait on the CURRENT stream only.  (Tensor.item()/.tolist() issue a blocking hipMemcpy into pageable memory, during
    which other host threads' launches stall: with several scenes in flight that drained the whole GPU at every
    read-back.)"""
    import torch
    t = t.reshape(-1)
    n = t.numel()
    buf = getattr(_tls, "pinned", None)
    if buf is None or buf.numel() < n or buf.dtype != t.dtype:
        buf = torch.empty(max(64, n), dtype=t.dtype, pin_memory=True)
        _tls.pinned = buf
    buf[:n].copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return buf[:n].tolist()


_gpu_ok = False


def require_gpu():
    global _gpu_ok
    if _gpu_ok:
        return
    import torch
    if not torch.cuda.is_available():
        raise CnrmaError("cn-rma_amd needs a HIP device (MI355X); the product path has no CPU implementation")
    _gpu_ok = True
