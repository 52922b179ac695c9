"""ctypes binding of libcosa_hip.so -- the only way the Python host code reaches the HIP kernels.

The C ABI is declared in include/cosa_hip.h.  Loading FAILS LOUDLY when the shared library is
missing: there is no CPU or eager-PyTorch fallback anywhere in the product path.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcosa_hip.so")
_lib = None

c_void_p, c_int, c_float, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t


class CosaError(RuntimeError):
    pass


_SIGS = {
    "cosa_abi_version": (c_int, []),
    "cosa_last_error": (ctypes.c_char_p, []),
    "cosa_denormalize_img": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "cosa_cam_minmax_norm": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "cosa_cam_minmax_norm_ws": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "cosa_cam_flip_merge_upsample": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                     c_void_p]),
    "cosa_cam_flip_merge_upsample_reuse": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                           c_void_p]),
    "cosa_cam2mask_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "cosa_cam2mask": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float,
                              c_int, c_int, ctypes.POINTER(c_int), c_int, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "cosa_cam2mask_multi_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "cosa_cam2mask_multi": (c_int, [c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_void_p, ctypes.POINTER(c_void_p),
                                    ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                    ctypes.POINTER(c_int), c_int, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "cosa_augment_record_bytes": (c_int, []),
    "cosa_augment_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "cosa_augment_batch": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_size_t, c_void_p]),
    "cosa_gmm_workspace_bytes": (c_size_t, []),
    "cosa_gmm_fit_thresholds": (c_int, [c_void_p, c_void_p, ctypes.c_longlong, c_int, ctypes.c_double, ctypes.c_double, c_int,
                                        c_void_p, c_void_p, c_size_t, c_void_p]),
    "cosa_par_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "cosa_par_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int), c_int,
                                 c_int, c_void_p, c_size_t, c_void_p]),
    "bilateralfilter": (None, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_float]),
    "bilateralfilter_batch": (None, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                     c_float, c_float]),
    "cosa_bilateral_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "cosa_bilateralfilter_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float,
                                               c_void_p, c_void_p, c_size_t, c_void_p]),
    "cosa_dense_energy_prepare": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_size_t, c_void_p]),
    "cosa_dense_energy_forward_prepared": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                                   c_float, c_float, c_void_p, c_size_t, c_void_p]),
    "cosa_dense_energy_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                          c_int, c_float, c_float, c_void_p, c_size_t, c_void_p]),
    "cosa_dense_energy_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "cosa_seg_loss_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "cosa_seg_loss_forward": (c_int, [c_void_p] * 10 + [c_int] * 5 + [c_void_p, c_size_t, c_void_p]),
    "cosa_seg_loss_backward": (c_int, [c_void_p] * 9 + [c_int] * 5 + [c_void_p, c_size_t, c_void_p]),
    "cosa_cam_loss_targets": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int, c_void_p,
                              c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "cosa_optim_record_bytes": (c_size_t, []),
    "cosa_optim_chunk_elems": (c_int, []),
    "cosa_fused_adamw_ema": (c_int, [c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_int, c_float, c_void_p]),
    "cosa_add_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                       c_float, c_void_p]),
    "cosa_layernorm_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "cosa_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                   c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "cosa_layernorm_bwd_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                       c_float, c_void_p, c_size_t, c_void_p]),
    "cosa_eval_labels": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                 c_void_p]),
    "cosa_cam_to_label": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_float, c_float,
                                  ctypes.c_longlong, c_void_p, c_void_p, c_void_p]),
    "cosa_confusion_hist": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p, c_void_p]),
    "cosa_attn_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "cosa_attn_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "cosa_attn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p,
                      c_size_t, c_void_p]),
    "cosa_gemm_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "cosa_gemm_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "cosa_gemm_wgrad_batched": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "cosa_gemm_wgrad_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "cosa_conv3x3_dilated_nhwc": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "cosa_head_gemm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.c_longlong, c_int, c_int, c_int, c_int,
                               c_int, c_void_p]),
    "cosa_im2col_flip": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cosa_im2col_flip_c8_tokens": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cosa_im2col_flip_split_tokens": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cosa_im2col_flip_split_tokens_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cosa_msm_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "cosa_softmax_halfres_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "cosa_softmax_halfres_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "cosa_embed_finish": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "cosa_head_gemm_dgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "cosa_head_gemm_wgrad_workspace": (c_size_t, [c_int, c_int]),
    "cosa_head_gemm_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "cosa_conv3x3_dilated_wgrad": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p, c_size_t, c_void_p]),
    "cosa_gemm_set_variant": (None, [c_int]),
    "cosa_gemm_set_grid_policy": (None, [c_int]),
    "cosa_gemm_set_grid_policy_f16": (None, [c_int]),
    "cosa_token_junction_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "cosa_gemm_set_stamp_slot": (None, [c_void_p]),
    "cosa_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "cosa_attn_prepare_vt": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "cosa_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p,
                      c_size_t, c_void_p]),
}

_SIGS.update({
    "cosa_gemm_bf16_dual_gelu": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "cosa_pos_resize": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "cosa_gelu_backward": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_void_p]),
    "cosa_transpose_record_bytes": (c_size_t, []),
    "cosa_transpose_cast_batched": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "cosa_broadcast_rows": (c_int, [c_void_p, c_void_p, c_int, ctypes.c_longlong, c_void_p]),
    "cosa_resize_bilinear": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cosa_split_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_longlong, c_int, c_void_p]),
    "cosa_layernorm_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "cosa_gemm_bf16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cosa_attn_fwd_bf16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p, c_void_p]),
    "cosa_lattice_filter_d2_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "cosa_lattice_filter_d2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "cosa_c8_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_longlong, c_int, c_void_p]),
    "cosa_c8_record_bytes": (c_size_t, []),
    "cosa_c8_rows_batched": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "cosa_layernorm_c8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "cosa_gemm_f16c8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cosa_attn_fwd_f16c8": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "cosa_radix_sort_workspace_bytes": (c_size_t, [ctypes.c_longlong]),
    "cosa_radix_sort_pairs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_void_p, c_size_t, c_void_p]),
    "cosa_c4_scale_bytes": (c_size_t, [c_int, c_int]),
    "cosa_c4_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_longlong, c_int, c_int, c_void_p]),
    "cosa_c4_record_bytes": (c_size_t, []),
    "cosa_c4_rows_batched": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "cosa_layernorm_c4": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "cosa_attn_fwd_f16c4": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "cosa_gemm_f16c4": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                c_int, c_void_p]),
})

# the fp16-operand builds of the GEMM / attention translation units export the same signatures under *_f16 names
for _bf, _f in (("cosa_gemm_bf16", "cosa_gemm_f16"), ("cosa_gemm_wgrad_bf16", "cosa_gemm_wgrad_f16"), ("cosa_layernorm", "cosa_layernorm_f16"),
                ("cosa_conv3x3_dilated_nhwc", "cosa_conv3x3_dilated_nhwc_f16"), ("cosa_conv3x3_dilated_wgrad", "cosa_conv3x3_dilated_wgrad_f16"),
                ("cosa_gemm_set_variant", "cosa_gemm_set_variant_f16"), ("cosa_gemm_set_stamp_slot", "cosa_gemm_set_stamp_slot_f16"),
                ("cosa_attn_workspace_bytes", "cosa_attn_workspace_bytes_f16"), ("cosa_attn_prepare_vt", "cosa_attn_prepare_vt_f16"),
                ("cosa_attn_fwd", "cosa_attn_fwd_f16"), ("cosa_attn_bwd_workspace_bytes", "cosa_attn_bwd_workspace_bytes_f16"),
                ("cosa_attn_bwd", "cosa_attn_bwd_f16"),
                # fp16x3 (round 6): the split-row producers and the three-term GEMM / attention with fp16 halves
                ("cosa_split_rows", "cosa_split_rows_f16"), ("cosa_layernorm_split", "cosa_layernorm_split_f16"),
                ("cosa_gemm_bf16x3", "cosa_gemm_f16x3"), ("cosa_attn_fwd_bf16x3", "cosa_attn_fwd_f16x3")):
    _SIGS[_f] = _SIGS[_bf]

# entry points added by later translation units register themselves here (vit / gemm / attention)
EXTRA_SIGS = {}


def fn16(name, dtype):
    """the entry point `name` for 16-bit operands of torch dtype `dtype`: bf16 -> name, fp16 -> its *_f16 twin"""
    if dtype == torch.bfloat16:
        return getattr(lib(), name)
    if dtype == torch.float16:
        return getattr(lib(), {"cosa_gemm_bf16": "cosa_gemm_f16", "cosa_gemm_wgrad_bf16": "cosa_gemm_wgrad_f16"}.get(name, name + "_f16"))
    raise CosaError(f"{name}: operands must be bfloat16 or float16, got {dtype}")


def lib():
    """Load libcosa_hip.so (once).  Raises CosaError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CosaError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -m cosa_amd.build). "
                "cosa_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in list(_SIGS.items()) + list(EXTRA_SIGS.items()):
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise CosaError(f"libcosa_hip.so does not export {name} (stale build?)") from e
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def declared_symbols():
    return list(_SIGS) + list(EXTRA_SIGS)


def check(rc, what=""):
    if rc != 0:
        msg = lib().cosa_last_error().decode("utf-8", "replace")
        raise CosaError(f"{what} failed (code {rc}): {msg}")


def stream_ptr():
    """Raw hipStream_t of torch's current stream (so kernels order with torch ops)."""
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise CosaError("cosa_amd operators need device (HIP) tensors; there is no CPU path")


_ws_cache = {}


def workspace(nbytes, device, tag="default"):
    """Grow-only per-(device,tag) byte workspace (never freed inside the step: graph-safe)."""
    key = (str(device), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# ---- optional per-kernel HIP-event timing (bench.py's roofline leg) ------------------------------------
_prof = None


def profile_start():
    """Record a HIP event pair around every profiled() region on torch's current stream (= the launch stream)."""
    global _prof
    _prof = {}


def profile_stop():
    """-> {name: (n_launches, total_ms)}; synchronises."""
    global _prof
    p, _prof = _prof, None
    torch.cuda.synchronize()
    out = {}
    for k, v in (p or {}).items():
        n, ms = 0, 0.0
        for a, b in v:
            try:
                ms += a.elapsed_time(b)
                n += 1
            except RuntimeError:                 # an event pair that cannot be timed must not take the measurement down
                pass
        out[k] = (n, ms)
    return out


class profiled:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.on = _prof is not None and not torch.cuda.is_current_stream_capturing()    # events inside a graph capture cannot be timed
        if self.on:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.on and _prof is not None:
            self.b.record()
            _prof.setdefault(self.name, []).append((self.a, self.b))
        return False


def int_array(values):
    arr = (c_int * len(values))(*[int(v) for v in values])
    return arr
