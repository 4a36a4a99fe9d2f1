"""ctypes binding of libhifihr.so (the C ABI in include/hifihr.h).

`get_lib()` loads the in-tree HIP build and FAILS LOUDLY when it is missing -- there is no CPU or
PyTorch fallback for the hot path.  `HifihrLib` itself only marshals pointers, so the test-suite can also
point it at tests/hostsim/libhifihr_hostsim.so (the same kernel sources compiled against a HIP
execution-model emulator) to exercise the kernels without a GPU; the package never does that.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_long, c_size_t, c_void_p

import torch  # noqa: F401  (must be imported first: the library binds to torch's libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhifihr.so")

_c_float_p = POINTER(c_float)


class _LinearDesc(ctypes.Structure):          # include/hifihr.h: hifihr_linear_desc
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("b", c_void_p), ("y", c_void_p), ("B", c_int), ("I", c_int), ("O", c_int), ("act", c_int),
                ("dy", c_void_p), ("dz_scratch", c_void_p), ("dW_acc", c_void_p), ("db_acc", c_void_p), ("dx", c_void_p)]

_c_int_p = POINTER(c_int32)


class HifihrError(RuntimeError):
    pass


def _fp(t):
    """Device (or host) pointer of a contiguous fp32 tensor, or NULL for None."""
    if t is None:
        return None
    dense = t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
    assert t.dtype == torch.float32 and dense, (t.dtype, t.shape, t.stride())
    return ctypes.cast(t.data_ptr(), _c_float_p)


def _ip(t):
    if t is None:
        return None
    assert t.dtype == torch.int32 and t.is_contiguous()
    return ctypes.cast(t.data_ptr(), _c_int_p)


def _stream_of(t):
    if t.is_cuda:
        return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return c_void_p(0)


def _np_fp(a):
    import numpy as np
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_c_float_p)


class HifihrLib:
    def __init__(self, path: str = LIB_PATH):
        if not os.path.exists(path):
            raise HifihrError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C hifihr_amd/csrc`). The HIP extension is mandatory; there is no fallback path.")
        self.path = path
        self.c = ctypes.CDLL(path)
        self._zero_page = set()                # devices whose zero page exists (zero_page_ready)
        self._pair_ok = {}                     # (N, H, W, C, K) -> hifihr_wino4_bwd_gemm_pair_supported
        c = self.c
        c.hifihr_last_error.restype = c_char_p
        c.hifihr_version.restype = c_int
        c.hifihr_device_count.restype = c_int
        c.hifihr_mano_create.argtypes = [POINTER(c_void_p)] + [_c_float_p] * 7
        c.hifihr_mano_destroy.argtypes = [c_void_p]
        c.hifihr_mano_lbs_fwd.argtypes = [c_void_p, _c_float_p, _c_float_p, c_int, _c_float_p, _c_float_p, _c_float_p, c_void_p]
        c.hifihr_mano_lbs_bwd.argtypes = [c_void_p] + [_c_float_p] * 5 + [c_int, _c_float_p, _c_float_p, c_void_p]
        c.hifihr_mano_joints_fwd.argtypes = [c_void_p, _c_float_p, c_int, c_int, _c_float_p, _c_float_p, _c_float_p, c_void_p]
        c.hifihr_mano_joints_bwd.argtypes = [c_void_p, _c_float_p, _c_float_p, _c_float_p, c_int, c_int, _c_float_p, c_void_p]
        c.hifihr_mano_full_fwd.argtypes = [c_void_p, _c_float_p, _c_float_p, c_int, c_int, _c_float_p] + [_c_float_p] * 6 + [c_void_p]
        c.hifihr_mano_full_bwd.argtypes = [c_void_p] + [_c_float_p] * 9 + [c_int, c_int, _c_float_p, _c_float_p, c_void_p]
        c.hifihr_lbs_create.argtypes = [POINTER(c_void_p), c_int, c_int, c_int] + [_c_float_p] * 4 + [_c_int_p]
        c.hifihr_lbs_destroy.argtypes = [c_void_p]
        c.hifihr_lbs_fwd.argtypes = [c_void_p, _c_float_p, _c_float_p, c_int, _c_float_p, _c_float_p, c_void_p]
        c.hifihr_lbs_bwd.argtypes = [c_void_p] + [_c_float_p] * 4 + [c_int] + [_c_float_p] * 3 + [c_void_p]
        self._bind_optional()

    def _bind_optional(self):
        c = self.c
        c.hifihr_renderer_create.argtypes = [POINTER(c_void_p), _c_int_p, c_int, c_int, c_int, c_int, _c_float_p, _c_float_p,
                                             _c_float_p, c_float, _c_float_p]
        c.hifihr_ssim_partial_count.argtypes = [c_int, c_int, c_int]
        c.hifihr_ssim_partial_count.restype = c_int
        c.hifihr_ssim_fwd.argtypes = [_c_float_p, _c_float_p, _c_float_p, c_int, c_int, c_int, _c_float_p, _c_float_p, _c_float_p,
                                      _c_float_p, c_void_p]
        c.hifihr_ssim_bwd.argtypes = [_c_float_p] * 7 + [c_int, c_int, c_int, _c_float_p, c_void_p]
        c.hifihr_ssim_bwd_scaled.argtypes = [_c_float_p] * 7 + [c_float, c_int, c_int, c_int, _c_float_p, c_void_p]
        c.hifihr_ssim_finish.argtypes = [_c_float_p, c_int, c_float, c_float, _c_float_p, c_void_p]
        ci = [c_int] * 9
        c.hifihr_conv2d_fwd.argtypes = [_c_float_p] * 3 + [c_int, _c_float_p] + ci + [c_void_p, c_size_t, c_void_p]
        c.hifihr_bias_relu_bwd.argtypes = [_c_float_p, _c_float_p, c_long, c_int, _c_float_p, _c_float_p, c_void_p]
        c.hifihr_conv2d_workspace_bytes.argtypes = [c_int] * 10
        c.hifihr_conv2d_workspace_bytes.restype = c_size_t
        c.hifihr_conv2d_bwd_data.argtypes = [_c_float_p] * 4 + ci + [c_void_p, c_size_t, c_void_p]
        c.hifihr_conv2d_fwd_bnstats.argtypes = [_c_float_p] * 4 + ci + [c_void_p, c_size_t, c_void_p]
        c.hifihr_conv2d_fwd_bnstats_pair_supported.argtypes = [c_int] * 11
        c.hifihr_conv2d_fwd_bnstats_pair.argtypes = [_c_float_p] * 4 + [c_int] * 3 + [_c_float_p] * 3 + [c_int] * 8 + [c_void_p]
        for fn in (c.hifihr_dwconv2d_bwd_data, c.hifihr_dwconv2d_bwd_weight):
            fn.argtypes = [_c_float_p] * 3 + [c_int] * 10 + [c_void_p]
        c.hifihr_dwconv2d_fwd.argtypes = [_c_float_p] * 4 + [c_int] * 10 + [c_void_p]
        c.hifihr_dwconv2d_fwd_bnswish.argtypes = [_c_float_p] * 8 + [c_int] * 10 + [c_void_p]
        c.hifihr_dwconv2d_bwd_weight_bnswish.argtypes = [_c_float_p] * 7 + [c_int] * 10 + [c_void_p]
        c.hifihr_bn_finalize_fwd.argtypes = [_c_float_p, c_long, c_int, c_float, c_float] + [_c_float_p] * 4 + [c_void_p]
        c.hifihr_bn_stats_floats.argtypes = [c_int]
        c.hifihr_bn_stats_floats.restype = c_int
        c.hifihr_bn_stats.argtypes = [_c_float_p, c_long, c_int, _c_float_p, c_void_p]
        c.hifihr_bn_act_fwd.argtypes = [_c_float_p] * 5 + [c_int, c_long, c_int, c_float, c_float] + [_c_float_p] * 5 + [c_void_p]
        c.hifihr_bn_act_eval.argtypes = [_c_float_p] * 6 + [c_int, c_long, c_int, c_float, _c_float_p, c_void_p]
        c.hifihr_bn_act_bwd.argtypes = [_c_float_p] * 7 + [c_int, c_long, c_int] + [_c_float_p] * 5 + [c_void_p]
        c.hifihr_bn_relu_maxpool_supported.argtypes = [c_int] * 4
        c.hifihr_bn_relu_maxpool_fwd.argtypes = [_c_float_p] * 4 + [c_int] * 4 + [c_float, c_float, _c_float_p, c_void_p] + [_c_float_p] * 4 + [c_void_p]
        c.hifihr_bn_relu_maxpool_bwd.argtypes = [_c_float_p, c_void_p] + [_c_float_p] * 5 + [c_int] * 4 + [_c_float_p] * 4 + [c_void_p]
        c.hifihr_bn_relu_maxpool_bwd_y.argtypes = [_c_float_p, _c_float_p, c_void_p] + [_c_float_p] * 5 + [c_int] * 4 + [_c_float_p] * 4 + [c_void_p]
        c.hifihr_conv2d_bwd_weight.argtypes = [_c_float_p] * 3 + ci + [c_void_p]
        c.hifihr_conv2d_bwd_weight_ws.argtypes = [_c_float_p] * 3 + ci + [c_void_p, c_size_t, c_void_p]
        c.hifihr_conv2d_wgrad_workspace_bytes.argtypes = ci
        c.hifihr_conv2d_bwd_weight_c3_supported.argtypes = [c_int] * 8
        c.hifihr_conv2d_bwd_weight_c3.argtypes = [_c_float_p] * 3 + [c_int] * 8 + [c_void_p, c_size_t, c_void_p]
        c.hifihr_conv2d_wgrad_workspace_bytes.restype = c_size_t
        c.hifihr_image_to_nhwc4.argtypes = [_c_float_p, _c_float_p, c_int, c_int, c_int, c_void_p]
        c.hifihr_image_to_nhwc4_padded.argtypes = [_c_float_p, _c_float_p] + [c_int] * 8 + [c_void_p]
        c.hifihr_geom_loss_fwd.argtypes = [_c_float_p] * 6 + [_c_int_p] + [c_int] * 7 + [_c_float_p] * 3 + [c_void_p]
        c.hifihr_geom_loss_bwd.argtypes = [_c_float_p] * 6 + [_c_int_p] * 3 + [c_int] * 7 + [_c_float_p] * 6 + [c_void_p]
        c.hifihr_photo_loss_partial_floats.argtypes = []
        c.hifihr_photo_loss_partial_floats.restype = c_int
        c.hifihr_photo_loss_fwd.argtypes = [_c_float_p, _c_float_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float] + \
            [_c_float_p] * 4 + [c_void_p]
        c.hifihr_photo_loss_bwd.argtypes = [_c_float_p] * 6 + [c_int, c_int, c_int, c_float, c_float, _c_float_p, c_void_p]
        c.hifihr_sil_post.argtypes = [_c_float_p, _c_float_p, c_int, c_int, c_int, _c_float_p, _c_float_p, c_void_p]
        c.hifihr_linear_fwd.argtypes = [_c_float_p] * 3 + [c_int] * 4 + [_c_float_p] * 2 + [c_float, c_float] + [_c_float_p] * 6 + [c_void_p]
        c.hifihr_linear_bwd.argtypes = [_c_float_p] * 4 + [c_int] * 4 + [_c_float_p] * 10 + [c_void_p]
        c.hifihr_wino_gemm_workspace_bytes.argtypes = [c_int] * 5
        c.hifihr_wino_gemm_workspace_bytes.restype = c_size_t
        c.hifihr_wino_weight_transform.argtypes = [_c_float_p, _c_float_p, c_int, c_int, c_int, c_void_p]
        c.hifihr_wino_input_transform.argtypes = [_c_float_p, _c_float_p, c_int, c_int, c_int, c_int, c_void_p]
        c.hifihr_wino_gemm.argtypes = [_c_float_p] * 3 + [c_int] * 5 + [c_void_p, c_size_t, c_void_p]
        c.hifihr_wino_output_transform.argtypes = [_c_float_p] * 3 + [c_int] * 4 + [c_void_p]
        c.hifihr_linear_fwd_group.argtypes = [POINTER(_LinearDesc), c_int, c_void_p]
        c.hifihr_linear_bwd_group.argtypes = [POINTER(_LinearDesc), c_int, c_void_p]
        c.hifihr_wino_input_dy_transform.argtypes = [_c_float_p] * 3 + [c_int] * 4 + [c_void_p]
        c.hifihr_conv2d_bwd_data_pre.argtypes = [_c_float_p] * 3 + [c_int] * 9 + [c_void_p, c_size_t, c_void_p]
        c.hifihr_conv2d_bwd_data_pre_res.argtypes = [_c_float_p] * 4 + [c_int] * 9 + [c_void_p, c_size_t, c_void_p]
        c.hifihr_conv2d_bwd_data_pre_plus1x1.argtypes = [_c_float_p] * 5 + [c_int] * 9 + [c_void_p]
        c.hifihr_conv2d_bwd_data_pre_plus1x1_supported.argtypes = [c_int] * 9
        c.hifihr_conv2d_bwd_weight_plus1x1.argtypes = [_c_float_p] * 5 + [c_int] * 9 + [c_void_p]
        c.hifihr_conv2d_bwd_weight_plus1x1_supported.argtypes = [c_int] * 9
        c.hifihr_weight_prep.argtypes = [c_void_p, c_int, c_int, c_void_p]
        c.hifihr_freihand_augment.argtypes = [c_void_p, c_void_p, _c_int_p, _c_int_p, c_int, c_int, c_int, _c_float_p, _c_float_p, c_void_p]
        c.hifihr_ho3d_workspace_bytes.argtypes = [c_int, c_int]
        c.hifihr_ho3d_workspace_bytes.restype = c_size_t
        c.hifihr_ho3d_batch.argtypes = ([c_void_p, c_void_p] + [_c_float_p] * 3 + [c_int, c_int, _c_int_p, c_int, c_int, c_void_p, c_size_t] +
                                        [_c_float_p] * 5 + [c_void_p])
        c.hifihr_freihand_batch.argtypes = ([c_void_p, c_void_p] + [_c_float_p] * 4 + [c_int, c_int, _c_int_p, c_int, c_int, c_int, _c_float_p,
                                            _c_float_p, c_void_p] + [_c_float_p] * 6 + [c_void_p, c_void_p])
        c.hifihr_freihand_batch_step.argtypes = ([c_void_p, c_void_p] + [_c_float_p] * 4 + [c_int, c_int, _c_int_p, c_int, c_int, c_int, _c_float_p,
                                                 _c_float_p, c_void_p] + [_c_float_p] * 6 + [c_void_p, c_int, c_float] + [_c_float_p] * 4 + [c_void_p])
        c.hifihr_light_split_fwd.argtypes = [_c_float_p, c_int, _c_float_p, _c_float_p, c_void_p]
        c.hifihr_light_split_bwd.argtypes = [_c_float_p, _c_float_p, _c_float_p, c_int, _c_float_p, c_void_p]
        c.hifihr_wino4_bwd_gemm_pair_supported.argtypes = [c_int] * 5
        c.hifihr_wino4_bwd_gemm_pair.argtypes = [_c_float_p] * 6 + [c_int] * 6 + [c_void_p]
        c.hifihr_loss_total_fwd.argtypes = [POINTER(_c_float_p), POINTER(c_int), c_int, _c_float_p, c_void_p]
        c.hifihr_loss_total_bwd.argtypes = [_c_float_p, POINTER(_c_float_p), POINTER(c_int), POINTER(c_int), c_int, c_void_p]
        c.hifihr_procrustes_error.argtypes = [_c_float_p, _c_float_p, c_int, c_int, _c_float_p, _c_float_p, c_void_p]
        c.hifihr_wino_output_transform_act.argtypes = [_c_float_p] * 3 + [c_int] * 5 + [c_void_p]
        c.hifihr_wino_dy_transform.argtypes = [_c_float_p, _c_float_p, c_int, c_int, c_int, c_int, c_void_p]
        c.hifihr_wino_wgrad_gemm.argtypes = [_c_float_p] * 3 + [c_int] * 5 + [c_void_p]
        c.hifihr_wino_dw_transform.argtypes = [_c_float_p, _c_float_p, c_int, c_int, c_int, c_void_p]
        c.hifihr_weight_transpose.argtypes = [_c_float_p, _c_float_p, c_int, c_int, c_int, c_void_p]
        c.hifihr_bgemm_nt.argtypes = [_c_float_p] * 3 + [c_int] * 4 + [c_void_p, c_size_t, c_void_p]
        c.hifihr_bgemm_nt_workspace_bytes.argtypes = [c_int] * 4
        c.hifihr_bgemm_nt_workspace_bytes.restype = c_size_t
        c.hifihr_renderer_set_light_mode.argtypes = [c_void_p, c_int]
        c.hifihr_texture_pca_fwd.argtypes = [_c_float_p] * 3 + [c_int, c_int, c_long, _c_float_p, c_void_p]
        c.hifihr_texture_pca_bwd.argtypes = [_c_float_p] * 2 + [c_int, c_int, c_long, _c_float_p, c_void_p]
        c.hifihr_comm_last_error.restype = c_char_p
        c.hifihr_comm_get_unique_id.argtypes = [c_void_p]
        c.hifihr_comm_init.argtypes = [POINTER(c_void_p), c_int, c_int, c_void_p]
        c.hifihr_comm_allreduce_f32.argtypes = [c_void_p, _c_float_p, c_size_t, c_void_p]
        c.hifihr_comm_broadcast_f32.argtypes = [c_void_p, _c_float_p, c_size_t, c_int, c_void_p]
        c.hifihr_comm_destroy.argtypes = [c_void_p]
        c.hifihr_wino_tile.argtypes = [c_int] * 5
        c.hifihr_wino_gemm_workspace_bytes_m.argtypes = [c_int] * 6
        c.hifihr_wino_gemm_workspace_bytes_m.restype = c_size_t
        c.hifihr_wino_weight_transform_m.argtypes = [_c_float_p, _c_float_p, c_int, c_int, c_int, c_int, c_void_p]
        c.hifihr_wino_input_transform_m.argtypes = [_c_float_p, _c_float_p] + [c_int] * 5 + [c_void_p]
        c.hifihr_wino_gemm_m.argtypes = [_c_float_p] * 3 + [c_int] * 6 + [c_void_p, c_size_t, c_void_p]
        c.hifihr_wino_output_transform_m.argtypes = [_c_float_p] * 3 + [c_int] * 5 + [c_void_p]
        c.hifihr_wino_output_transform_act_m.argtypes = [_c_float_p] * 3 + [c_int] * 6 + [c_void_p]
        c.hifihr_conv3x3_c64_wino_supported.argtypes = [c_int] * 5
        c.hifihr_conv3x3_c64_wino.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
        c.hifihr_conv3x3_c64_wino_res.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
        c.hifihr_conv3x3_c64_bwd_pair_supported.argtypes = [c_int] * 3
        c.hifihr_conv3x3_c64_bwd_pair.argtypes = [c_void_p] * 7 + [c_size_t] + [c_int] * 3 + [c_void_p]
        c.hifihr_conv3x3_c64_bwd_pair_slabs.argtypes = [c_void_p] * 6 + [c_size_t] + [c_int] * 3 + [c_void_p, c_void_p]
        c.hifihr_conv_halo_wgrad_reduce_multi.argtypes = [c_void_p, c_int, c_void_p]
        c.hifihr_wino_bn_input_supported.argtypes = [c_int, c_int]
        c.hifihr_wino_bn_input_transform.argtypes = [_c_float_p] * 7 + [c_int] * 5 + [c_float, c_float] + [_c_float_p] * 4 + [c_void_p]
        c.hifihr_wino_output_transform_bnred.argtypes = [_c_float_p] * 10 + [c_int] * 5 + [c_void_p]
        c.hifihr_wino_bn_bwd_dual_transform.argtypes = [_c_float_p] * 8 + [c_int] * 5 + [_c_float_p] * 2 + [c_void_p]
        c.hifihr_bn_bwd_apply.argtypes = [_c_float_p] * 5 + [c_long, c_int] + [_c_float_p] * 4 + [c_void_p]
        c.hifihr_joint_terms_fwd.argtypes = [_c_float_p] * 4 + [c_int] * 3 + [POINTER(c_float), _c_float_p, c_void_p]
        c.hifihr_joint_terms_bwd.argtypes = [_c_float_p] * 4 + [c_int] * 3 + [POINTER(c_float)] + [_c_float_p] * 3 + [c_void_p]
        c.hifihr_wino_dy_transform_m.argtypes = [_c_float_p] * 2 + [c_int] * 5 + [c_void_p]
        c.hifihr_wino_input_dy_transform_m.argtypes = [_c_float_p] * 3 + [c_int] * 5 + [c_void_p]
        c.hifihr_wino_wgrad_parts_m.argtypes = [c_int] * 6
        c.hifihr_wino_wgrad_gemm_parts_m.argtypes = [_c_float_p] * 3 + [c_int] * 7 + [c_void_p]
        c.hifihr_wino_dw_transform_parts_m.argtypes = [_c_float_p, c_int, _c_float_p, c_int, c_int, c_int, c_void_p]
        c.hifihr_bgemm_describe.argtypes = [c_int] * 4 + [ctypes.c_char_p, c_int]
        c.hifihr_bgemm_describe_batch.argtypes = [c_int] * 5 + [ctypes.c_char_p, c_int]
        c.hifihr_conv2d_describe.argtypes = [c_int] * 10 + [ctypes.c_char_p, c_int]
        c.hifihr_bgemm_tn_parts.argtypes = [c_int] * 4
        c.hifihr_bgemm_tn.argtypes = [_c_float_p] * 3 + [c_int] * 5 + [c_void_p]
        c.hifihr_wino_wgrad_parts.argtypes = [c_int] * 5
        c.hifihr_wino_wgrad_gemm_parts.argtypes = [_c_float_p] * 3 + [c_int] * 6 + [c_void_p]
        c.hifihr_wino_dw_transform_parts.argtypes = [_c_float_p, c_int, _c_float_p, c_int, c_int, c_void_p]
        c.hifihr_wino4_dw_transform_multi.argtypes = [c_void_p, c_int, c_void_p]
        c.hifihr_se_pool.argtypes = [_c_float_p, c_int, c_int, c_int, _c_float_p, c_void_p]
        c.hifihr_se_scale.argtypes = [_c_float_p, _c_float_p, _c_float_p, c_float, c_int, c_int, c_int, _c_float_p, c_void_p]
        c.hifihr_se_bwd_gate.argtypes = [_c_float_p, _c_float_p, c_int, c_int, c_int, _c_float_p, c_void_p]
        c.hifihr_se_mlp_supported.argtypes = [c_int, c_int]
        c.hifihr_wino_tiles.argtypes = [c_int] * 4
        c.hifihr_wino_tiles.restype = c_long
        c.hifihr_wino_tiles_computed.argtypes = [c_int] * 4
        c.hifihr_wino_tiles_computed.restype = c_long
        c.hifihr_zero_page_ready.argtypes = [c_void_p]
        c.hifihr_drop_connect_add.argtypes = [_c_float_p, _c_float_p, _c_float_p, c_float, c_int, c_size_t, _c_float_p, c_void_p]
        c.hifihr_se_mlp_fwd.argtypes = [_c_float_p] * 5 + [c_int] * 3 + [_c_float_p] * 4 + [c_void_p]
        c.hifihr_se_mlp_bwd.argtypes = [_c_float_p] * 7 + [c_int] * 3 + [_c_float_p] * 7 + [c_void_p]
        c.hifihr_mmpool_fwd.argtypes = [_c_float_p, _c_float_p, c_int, c_int, c_int, _c_float_p, _c_int_p, _c_float_p, _c_float_p, c_void_p]
        c.hifihr_mmpool_bwd.argtypes = [_c_float_p, _c_float_p, _c_int_p, _c_float_p, _c_float_p, c_int, c_int, c_int, _c_float_p,
                                        _c_float_p, c_void_p]
        c.hifihr_maxpool2d_fwd.argtypes = [_c_float_p] + [c_int] * 7 + [_c_float_p, c_void_p, c_void_p]
        c.hifihr_maxpool2d_bwd.argtypes = [_c_float_p, c_void_p] + [c_int] * 7 + [_c_float_p, c_void_p]
        c.hifihr_maxpool2d_fwd_flat.argtypes = [_c_float_p] + [c_int] * 7 + [_c_float_p, c_void_p, c_void_p]
        c.hifihr_maxpool2d_bwd_flat.argtypes = [_c_float_p, c_void_p] + [c_int] * 7 + [_c_float_p, c_void_p]
        c.hifihr_maxpool2d_bwd_relu.argtypes = [_c_float_p, c_void_p, _c_float_p] + [c_int] * 7 + [_c_float_p, c_void_p]
        c.hifihr_wino_output_transform_mask_m.argtypes = [_c_float_p] * 3 + [c_int] * 5 + [c_void_p]
        c.hifihr_adam_step.argtypes = [_c_float_p, _c_float_p, _c_float_p, _c_float_p, c_size_t, c_float, c_float, c_float,
                                       c_float, c_float, c_float, c_int, c_void_p]
        c.hifihr_adam_state_bytes.restype = c_size_t
        c.hifihr_adam_state_bytes.argtypes = []
        c.hifihr_adam_step_counted.argtypes = [_c_float_p, _c_float_p, _c_float_p, _c_float_p, c_size_t, c_float, c_float, c_float, c_void_p, c_void_p]
        c.hifihr_adam_step_dyn.argtypes = [_c_float_p, _c_float_p, _c_float_p, _c_float_p, c_size_t, c_float, c_float, c_float,
                                           c_float, c_float, _c_float_p, c_void_p]
        c.hifihr_renderer_destroy.argtypes = [c_void_p]
        c.hifihr_render_workspace_bytes.argtypes = [c_void_p, c_int]
        c.hifihr_render_workspace_bytes.restype = c_size_t
        c.hifihr_render_fwd.argtypes = [c_void_p, _c_float_p, _c_float_p, c_int, _c_float_p, _c_float_p, _c_float_p, c_int,
                                        _c_float_p, _c_int_p, c_void_p, c_void_p]
        c.hifihr_render_bwd.argtypes = [c_void_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_int_p, _c_float_p, c_int,
                                        _c_float_p, _c_float_p, _c_float_p, _c_float_p, c_void_p, c_void_p]
        c.hifihr_renderer_set_uv.argtypes = [c_void_p, _c_int_p, _c_float_p, c_int]
        c.hifihr_render_uv_scratch_bytes.argtypes = [c_void_p, c_int]
        c.hifihr_render_uv_scratch_bytes.restype = c_size_t
        c.hifihr_render_fwd_uv.argtypes = [c_void_p, _c_float_p, _c_float_p, c_int, c_int, _c_float_p, _c_float_p, _c_float_p, c_int, _c_float_p,
                                           _c_int_p, _c_float_p, c_void_p, c_void_p]
        c.hifihr_render_bwd_uv.argtypes = [c_void_p, _c_float_p, _c_float_p, c_int, c_int, _c_float_p, _c_float_p, _c_float_p, _c_int_p, _c_float_p,
                                           c_int] + [_c_float_p] * 6 + [c_void_p, c_void_p]

    # ------------------------------------------------------------------
    def check(self, rc: int, what: str):
        if rc != 0:
            raise HifihrError(f"{what} failed ({rc}): {self.c.hifihr_last_error().decode()}")

    # ---- MANO --------------------------------------------------------
    def mano_create(self, tables) -> c_void_p:
        h = c_void_p()
        keep = [_np_fp(a) for a in (tables.v_template, tables.shapedirs, tables.posedirs, tables.J_regressor,
                                    tables.weights, tables.hands_components, tables.hands_mean)]
        self.check(self.c.hifihr_mano_create(ctypes.byref(h), *[k[1] for k in keep]), "hifihr_mano_create")
        return h

    def mano_destroy(self, h):
        self.c.hifihr_mano_destroy(h)

    def mano_lbs_fwd(self, h, pose, beta, verts, jtr, saved):
        B = pose.shape[0]
        assert pose.shape == (B, 48) and beta.shape == (B, 10) and verts.shape == (B, 778, 3)
        self.check(self.c.hifihr_mano_lbs_fwd(h, _fp(pose), _fp(beta), B, _fp(verts), _fp(jtr), _fp(saved),
                                              _stream_of(pose)), "hifihr_mano_lbs_fwd")

    def mano_lbs_bwd(self, h, pose, beta, saved, gverts, gjtr, gpose, gbeta):
        B = pose.shape[0]
        self.check(self.c.hifihr_mano_lbs_bwd(h, _fp(pose), _fp(beta), _fp(saved), _fp(gverts), _fp(gjtr), B,
                                              _fp(gpose), _fp(gbeta), _stream_of(pose)), "hifihr_mano_lbs_bwd")

    def mano_joints_fwd(self, h, verts, root_id, joints_rel, verts_rel, root):
        B = verts.shape[0]
        self.check(self.c.hifihr_mano_joints_fwd(h, _fp(verts), B, root_id, _fp(joints_rel), _fp(verts_rel), _fp(root),
                                                 _stream_of(verts)), "hifihr_mano_joints_fwd")

    def mano_joints_bwd(self, h, gjoints_rel, gverts_rel, groot, root_id, gverts):
        B = gverts.shape[0]
        self.check(self.c.hifihr_mano_joints_bwd(h, _fp(gjoints_rel), _fp(gverts_rel), _fp(groot), B, root_id,
                                                 _fp(gverts), _stream_of(gverts)), "hifihr_mano_joints_bwd")

    def mano_full_fwd(self, h, pose, beta, root_id, root_xyz, verts, joints_rel, verts_rel, verts_cam, root, saved):
        """hifihr_mano_full_fwd: the layer, then joint regression + root-relative step + camera-space offset."""
        B = pose.shape[0]
        self.check(self.c.hifihr_mano_full_fwd(h, _fp(pose), _fp(beta), B, int(root_id), _fp(root_xyz), _fp(verts), _fp(joints_rel), _fp(verts_rel),
                                               _fp(verts_cam), _fp(root), _fp(saved), _stream_of(pose)), "hifihr_mano_full_fwd")

    def mano_full_bwd(self, h, pose, beta, saved, gjoints_rel, gverts_rel, gverts_cam, groot, root_id, gpose, gbeta, gpose_add=None,
                      gbeta_add=None):
        B = pose.shape[0]
        self.check(self.c.hifihr_mano_full_bwd(h, _fp(pose), _fp(beta), _fp(saved), _fp(gjoints_rel), _fp(gverts_rel), _fp(gverts_cam),
                                               _fp(groot), _fp(gpose_add), _fp(gbeta_add), B, int(root_id), _fp(gpose), _fp(gbeta),
                                               _stream_of(pose)), "hifihr_mano_full_bwd")

    # ---- generic LBS (NIMBLE-shaped layer) -----------------------------
    def lbs_create(self, v_template, shapedirs, j_regressor, weights, parents) -> c_void_p:
        V, J, S = int(v_template.shape[0]), int(weights.shape[1]), int(shapedirs.shape[2])
        assert shapedirs.shape == (V, 3, S) and j_regressor.shape == (J, V) and weights.shape == (V, J) and len(parents) == J
        h = c_void_p()
        keep = [_np_fp(a) for a in (v_template, shapedirs, j_regressor, weights)]
        import numpy as np
        par = np.ascontiguousarray(np.asarray(parents, dtype=np.int32))
        self.check(self.c.hifihr_lbs_create(ctypes.byref(h), V, J, S, *[k[1] for k in keep], par.ctypes.data_as(_c_int_p)), "hifihr_lbs_create")
        return h

    def lbs_destroy(self, h):
        self.c.hifihr_lbs_destroy(h)

    def lbs_fwd(self, h, theta, beta, verts, joints):
        self.check(self.c.hifihr_lbs_fwd(h, _fp(theta), _fp(beta), theta.shape[0], _fp(verts), _fp(joints), _stream_of(theta)), "hifihr_lbs_fwd")

    def lbs_bwd(self, h, theta, beta, gverts, gjoints, scratch, gtheta, gbeta):
        self.check(self.c.hifihr_lbs_bwd(h, _fp(theta), _fp(beta), _fp(gverts), _fp(gjoints), theta.shape[0], _fp(scratch), _fp(gtheta),
                                         _fp(gbeta), _stream_of(theta)), "hifihr_lbs_bwd")

    # ---- convolution (NHWC, f32 MFMA implicit GEMM) --------------------
    @staticmethod
    def _ws(ws):
        """(pointer, bytes) of an optional zero-initialised, self-cleaning workspace tensor (any dtype, contiguous)."""
        if ws is None:
            return c_void_p(0), c_size_t(0)
        assert ws.is_contiguous()
        return c_void_p(ws.data_ptr()), c_size_t(ws.numel() * ws.element_size())

    def conv2d_workspace_bytes(self, N, H, W, C, K, R, S, stride, pad, bwd_data=False):
        return int(self.c.hifihr_conv2d_workspace_bytes(N, H, W, C, K, R, S, stride, pad, int(bool(bwd_data))))

    def conv2d_fwd(self, x, w, bias, y, N, H, W, C, K, R, S, stride, pad, ws=None, act=0):
        self.check(self.c.hifihr_conv2d_fwd(_fp(x), _fp(w), _fp(bias), int(act), _fp(y), N, H, W, C, K, R, S, stride, pad,
                                            *self._ws(ws), _stream_of(x)), "hifihr_conv2d_fwd")

    def bias_relu_bwd(self, dy, y, M, C, g, db_acc):
        self.check(self.c.hifihr_bias_relu_bwd(_fp(dy), _fp(y), c_long(M), C, _fp(g), _fp(db_acc), _stream_of(dy)),
                   "hifihr_bias_relu_bwd")

    def conv2d_fwd_bnstats(self, x, w, y, stats, N, H, W, C, K, R, S, stride, pad, ws=None):
        self.check(self.c.hifihr_conv2d_fwd_bnstats(_fp(x), _fp(w), _fp(y), _fp(stats), N, H, W, C, K, R, S, stride, pad,
                                                    *self._ws(ws), _stream_of(x)), "hifihr_conv2d_fwd_bnstats")

    def conv2d_fwd_bnstats_pair_supported(self, N, H, W, C, stride, K1, R1, pad1, K2, R2, pad2):
        return bool(self.c.hifihr_conv2d_fwd_bnstats_pair_supported(N, H, W, C, stride, K1, R1, pad1, K2, R2, pad2))

    def conv2d_fwd_bnstats_pair(self, x, w1, y1, stats1, K1, R1, pad1, w2, y2, stats2, K2, R2, pad2, N, H, W, C, stride):
        """two convolutions of the same input (+ their batch-norm statistics) in one launch (include/hifihr.h)"""
        self.check(self.c.hifihr_conv2d_fwd_bnstats_pair(_fp(x), _fp(w1), _fp(y1), _fp(stats1), K1, R1, pad1, _fp(w2), _fp(y2), _fp(stats2), K2, R2, pad2,
                                                         N, H, W, C, stride, _stream_of(x)), "hifihr_conv2d_fwd_bnstats_pair")

    def bn_stats_floats(self, C):
        return int(self.c.hifihr_bn_stats_floats(int(C)))

    def bn_stats(self, x, M, C, stats):
        self.check(self.c.hifihr_bn_stats(_fp(x), c_long(M), C, _fp(stats), _stream_of(x)), "hifihr_bn_stats")

    def bn_act_fwd(self, x, stats, gamma, beta, residual, act, M, C, eps, momentum, y, save_mean, save_invstd, rmean, rvar):
        self.check(self.c.hifihr_bn_act_fwd(_fp(x), _fp(stats), _fp(gamma), _fp(beta), _fp(residual), int(act), c_long(M), C,
                                            c_float(eps), c_float(momentum), _fp(y), _fp(save_mean), _fp(save_invstd), _fp(rmean),
                                            _fp(rvar), _stream_of(x)), "hifihr_bn_act_fwd")

    def bn_act_eval(self, x, rmean, rvar, gamma, beta, residual, act, M, C, eps, y):
        self.check(self.c.hifihr_bn_act_eval(_fp(x), _fp(rmean), _fp(rvar), _fp(gamma), _fp(beta), _fp(residual), int(act), c_long(M), C,
                                             c_float(eps), _fp(y), _stream_of(x)), "hifihr_bn_act_eval")

    def bn_act_bwd(self, dy, y, x, save_mean, save_invstd, gamma, beta, act, M, C, red, dx, dres, dgamma_acc, dbeta_acc):
        self.check(self.c.hifihr_bn_act_bwd(_fp(dy), _fp(y), _fp(x), _fp(save_mean), _fp(save_invstd), _fp(gamma), _fp(beta), int(act),
                                            c_long(M), C, _fp(red), _fp(dx), _fp(dres), _fp(dgamma_acc), _fp(dbeta_acc),
                                            _stream_of(dy)), "hifihr_bn_act_bwd")

    def bn_relu_maxpool_supported(self, N, H, W, C):
        return bool(self.c.hifihr_bn_relu_maxpool_supported(N, H, W, C))

    def bn_relu_maxpool_fwd(self, x, stats, gamma, beta, N, H, W, C, eps, momentum, pooled, tap, save_mean, save_invstd, running_mean,
                            running_var):
        self.check(self.c.hifihr_bn_relu_maxpool_fwd(_fp(x), _fp(stats), _fp(gamma), _fp(beta), N, H, W, C, c_float(eps), c_float(momentum),
                                                     _fp(pooled), c_void_p(tap.data_ptr()), _fp(save_mean), _fp(save_invstd),
                                                     _fp(running_mean), _fp(running_var), _stream_of(x)), "hifihr_bn_relu_maxpool_fwd")

    def bn_relu_maxpool_bwd_y(self, gy, pooled, tap, x, save_mean, save_invstd, gamma, beta, N, H, W, C, red, dx, dgamma_acc, dbeta_acc):
        self.check(self.c.hifihr_bn_relu_maxpool_bwd_y(_fp(gy), _fp(pooled), c_void_p(tap.data_ptr()), _fp(x), _fp(save_mean), _fp(save_invstd),
                                                       _fp(gamma), _fp(beta), N, H, W, C, _fp(red), _fp(dx), _fp(dgamma_acc), _fp(dbeta_acc),
                                                       _stream_of(gy)), "hifihr_bn_relu_maxpool_bwd_y")

    def bn_relu_maxpool_bwd(self, gy, tap, x, save_mean, save_invstd, gamma, beta, N, H, W, C, red, dx, dgamma_acc, dbeta_acc):
        self.check(self.c.hifihr_bn_relu_maxpool_bwd(_fp(gy), c_void_p(tap.data_ptr()), _fp(x), _fp(save_mean), _fp(save_invstd), _fp(gamma),
                                                     _fp(beta), N, H, W, C, _fp(red), _fp(dx), _fp(dgamma_acc), _fp(dbeta_acc),
                                                     _stream_of(gy)), "hifihr_bn_relu_maxpool_bwd")

    def conv2d_bwd_weight_c3_supported(self, N, H, W, K, R, S, stride, pad):
        return bool(self.c.hifihr_conv2d_bwd_weight_c3_supported(N, H, W, K, R, S, stride, pad))

    def conv2d_bwd_weight_c3(self, x4, dy, dw3, N, H, W, K, R, S, stride, pad, ws):
        self.check(self.c.hifihr_conv2d_bwd_weight_c3(_fp(x4), _fp(dy), _fp(dw3), N, H, W, K, R, S, stride, pad, c_void_p(ws.data_ptr()),
                                                      c_size_t(ws.numel() * ws.element_size()), _stream_of(x4)), "hifihr_conv2d_bwd_weight_c3")

    def dwconv2d_fwd(self, x, w, y, N, H, W, C, OH, OW, K, stride, pt, pl, stats=None):
        self.check(self.c.hifihr_dwconv2d_fwd(_fp(x), _fp(w), _fp(y), _fp(stats), N, H, W, C, OH, OW, K, stride, pt, pl,
                                              _stream_of(x)), "hifihr_dwconv2d_fwd")

    def dwconv2d_fwd_bnswish(self, x, mean, invstd, gamma, beta, w, y, N, H, W, C, OH, OW, K, stride, pt, pl, stats=None):
        self.check(self.c.hifihr_dwconv2d_fwd_bnswish(_fp(x), _fp(mean), _fp(invstd), _fp(gamma), _fp(beta), _fp(w), _fp(y), _fp(stats), N, H, W, C,
                                                      OH, OW, K, stride, pt, pl, _stream_of(x)), "hifihr_dwconv2d_fwd_bnswish")

    def dwconv2d_bwd_weight_bnswish(self, x, mean, invstd, gamma, beta, dy, dw, N, H, W, C, OH, OW, K, stride, pt, pl):
        self.check(self.c.hifihr_dwconv2d_bwd_weight_bnswish(_fp(x), _fp(mean), _fp(invstd), _fp(gamma), _fp(beta), _fp(dy), _fp(dw), N, H, W, C,
                                                             OH, OW, K, stride, pt, pl, _stream_of(x)), "hifihr_dwconv2d_bwd_weight_bnswish")

    def bn_finalize_fwd(self, stats, M, C, eps, momentum, save_mean, save_invstd, rmean, rvar):
        self.check(self.c.hifihr_bn_finalize_fwd(_fp(stats), c_long(M), C, c_float(eps), c_float(momentum), _fp(save_mean), _fp(save_invstd),
                                                 _fp(rmean), _fp(rvar), _stream_of(stats)), "hifihr_bn_finalize_fwd")

    def dwconv2d_bwd_data(self, dy, w, dx, N, H, W, C, OH, OW, K, stride, pt, pl):
        self.check(self.c.hifihr_dwconv2d_bwd_data(_fp(dy), _fp(w), _fp(dx), N, H, W, C, OH, OW, K, stride, pt, pl, _stream_of(dy)),
                   "hifihr_dwconv2d_bwd_data")

    def dwconv2d_bwd_weight(self, x, dy, dw, N, H, W, C, OH, OW, K, stride, pt, pl):
        self.check(self.c.hifihr_dwconv2d_bwd_weight(_fp(x), _fp(dy), _fp(dw), N, H, W, C, OH, OW, K, stride, pt, pl, _stream_of(x)),
                   "hifihr_dwconv2d_bwd_weight")

    def light_split_fwd(self, lights, colors, directions):
        self.check(self.c.hifihr_light_split_fwd(_fp(lights), lights.shape[0], _fp(colors), _fp(directions), _stream_of(lights)),
                   "hifihr_light_split_fwd")

    def light_split_bwd(self, lights, gcolors, gdirections, glights):
        self.check(self.c.hifihr_light_split_bwd(_fp(lights), _fp(gcolors), _fp(gdirections), lights.shape[0], _fp(glights),
                                                 _stream_of(lights)), "hifihr_light_split_bwd")

    def loss_total_fwd(self, parts, counts, total):
        n = len(parts)
        arr = (_c_float_p * n)(*[_fp(t) for t in parts])
        self.check(self.c.hifihr_loss_total_fwd(arr, (c_int * n)(*[int(c) for c in counts]), n, _fp(total), _stream_of(total)),
                   "hifihr_loss_total_fwd")

    def loss_total_bwd(self, gtotal, grads, counts):
        n = len(grads)
        arr = (_c_float_p * n)(*[_fp(t) for t in grads])
        self.check(self.c.hifihr_loss_total_bwd(_fp(gtotal), arr, (c_int * n)(*[int(c) for c in counts]),
                                                (c_int * n)(*[int(t.numel()) for t in grads]), n, _stream_of(gtotal)), "hifihr_loss_total_bwd")

    @staticmethod
    def _lambda5(lam):
        return (c_float * 5)(*[float(v) for v in lam])

    def joint_terms_fwd(self, j2d, j2d_gt, joints, joints_gt, mse, lam3, out):
        ref = j2d if j2d is not None else joints
        self.check(self.c.hifihr_joint_terms_fwd(_fp(j2d), _fp(j2d_gt), _fp(joints), _fp(joints_gt), ref.shape[0], ref.shape[1], int(mse),
                                                 (c_float * 3)(*[float(v) for v in lam3]), _fp(out), _stream_of(ref)), "hifihr_joint_terms_fwd")

    def joint_terms_bwd(self, j2d, j2d_gt, joints, joints_gt, mse, lam3, gout, g_j2d, g_joints):
        ref = j2d if j2d is not None else joints
        self.check(self.c.hifihr_joint_terms_bwd(_fp(j2d), _fp(j2d_gt), _fp(joints), _fp(joints_gt), ref.shape[0], ref.shape[1], int(mse),
                                                 (c_float * 3)(*[float(v) for v in lam3]), _fp(gout), _fp(g_j2d), _fp(g_joints),
                                                 _stream_of(ref)), "hifihr_joint_terms_bwd")

    def geom_loss_fwd(self, joints, joints_gt, verts, verts_gt, shape, pose, faces, mse, lam, partial, out):
        B, J, V = joints.shape[0], joints.shape[1], verts.shape[1]
        F = 0 if faces is None else faces.shape[0]
        NS = 0 if shape is None else shape.shape[1]
        NP = 0 if pose is None else pose.shape[1]
        self.check(self.c.hifihr_geom_loss_fwd(_fp(joints), _fp(joints_gt), _fp(verts), _fp(verts_gt), _fp(shape), _fp(pose), _ip(faces),
                                               B, J, V, F, NS, NP, int(mse), self._lambda5(lam), _fp(partial), _fp(out),
                                               _stream_of(joints)), "hifihr_geom_loss_fwd")

    def geom_loss_bwd(self, joints, joints_gt, verts, verts_gt, shape, pose, faces, vf_off, vf_idx, mse, lam, gout, gj, gv, gshape,
                      gpose):
        B, J, V = joints.shape[0], joints.shape[1], verts.shape[1]
        F = 0 if faces is None else faces.shape[0]
        NS = 0 if shape is None else shape.shape[1]
        NP = 0 if pose is None else pose.shape[1]
        self.check(self.c.hifihr_geom_loss_bwd(_fp(joints), _fp(joints_gt), _fp(verts), _fp(verts_gt), _fp(shape), _fp(pose), _ip(faces),
                                               _ip(vf_off), _ip(vf_idx), B, J, V, F, NS, NP, int(mse), self._lambda5(lam), _fp(gout),
                                               _fp(gj), _fp(gv), _fp(gshape), _fp(gpose), _stream_of(joints)), "hifihr_geom_loss_bwd")

    def photo_loss_partial_floats(self):
        return int(self.c.hifihr_photo_loss_partial_floats())

    def photo_loss_fwd(self, rgba, imgs, seg, l_tex, l_mrgb, l_sil, re_img_m, mask_rgbs, partial, out):
        B, _, H, W = rgba.shape
        assert seg.dtype == torch.int64 and seg.is_contiguous() and rgba.is_contiguous() and imgs.is_contiguous()
        self.check(self.c.hifihr_photo_loss_fwd(_fp(rgba), _fp(imgs), c_void_p(seg.data_ptr()), B, H, W, float(l_tex), float(l_mrgb),
                                                float(l_sil), _fp(re_img_m), _fp(mask_rgbs), _fp(partial), _fp(out),
                                                _stream_of(rgba)), "hifihr_photo_loss_fwd")

    def photo_loss_bwd(self, rgba, re_img_m, mask_rgbs, g_re_img, gout, fwd_out, l_tex, l_mrgb, grad_rgba):
        B, _, H, W = rgba.shape
        self.check(self.c.hifihr_photo_loss_bwd(_fp(rgba), _fp(re_img_m), _fp(mask_rgbs), _fp(g_re_img), _fp(gout), _fp(fwd_out), B, H, W,
                                                float(l_tex), float(l_mrgb), _fp(grad_rgba), _stream_of(rgba)), "hifihr_photo_loss_bwd")

    def sil_post(self, rgba, imgs, re_sil, mask_rgbs):
        B, _, H, W = rgba.shape
        assert rgba.is_contiguous() and (imgs is None or imgs.is_contiguous())
        self.check(self.c.hifihr_sil_post(_fp(rgba), _fp(imgs), B, H, W, _fp(re_sil), _fp(mask_rgbs), _stream_of(rgba)), "hifihr_sil_post")

    # ---- Winograd F(2x2, 3x3) ------------------------------------------
    # Winograd: m = output-tile edge (2: F(2x2, 3x3), 16 positions; 4: F(4x4, 3x3), 36 positions); wino_tile() = the library's choice
    def wino_tile(self, N, H, W, C, K):
        return int(self.c.hifihr_wino_tile(N, H, W, C, K))

    def wino_gemm_workspace_bytes(self, N, H, W, C, K, m=2):
        return int(self.c.hifihr_wino_gemm_workspace_bytes_m(N, H, W, C, K, m))

    def wino4_bwd_gemm_pair_supported(self, N, H, W, C, K):
        key = (N, H, W, C, K)
        hit = self._pair_ok.get(key)
        if hit is None:
            hit = self._pair_ok[key] = bool(self.c.hifihr_wino4_bwd_gemm_pair_supported(int(N), int(H), int(W), int(C), int(K)))
        return hit

    def wino4_bwd_gemm_pair(self, V2, U2, M2, Vx, Yt, dU_parts, N, H, W, C, K, parts):
        """Backward-data and backward-weight products of one F(4x4, 3x3) layer (C -> K channels) in one launch."""
        self.check(self.c.hifihr_wino4_bwd_gemm_pair(_fp(V2), _fp(U2), _fp(M2), _fp(Vx), _fp(Yt), _fp(dU_parts), N, H, W, C, K, int(parts),
                                                     _stream_of(V2)), "hifihr_wino4_bwd_gemm_pair")

    def wino_weight_transform(self, w, U, K, C, flip, m=2):
        self.check(self.c.hifihr_wino_weight_transform_m(_fp(w), _fp(U), K, C, int(flip), m, _stream_of(w)), "hifihr_wino_weight_transform")

    def wino_input_transform(self, x, V, N, H, W, C, m=2):
        self.check(self.c.hifihr_wino_input_transform_m(_fp(x), _fp(V), N, H, W, C, m, _stream_of(x)), "hifihr_wino_input_transform")

    def zero_page_ready(self, device=None):
        """The 256 zero bytes the halo / Winograd kernels read out-of-image pixels from exist on `device` (allocated at the first call
        OUTSIDE a stream capture; the steppers call this before they capture).  A positive answer is cached per device."""
        import torch
        idx = torch.cuda.current_device() if device is None else torch.device(device).index
        idx = torch.cuda.current_device() if idx is None else idx
        if idx in self._zero_page:
            return True
        with torch.cuda.device(idx):
            ok = bool(self.c.hifihr_zero_page_ready(c_void_p(torch.cuda.current_stream(idx).cuda_stream)))
        if ok:
            self._zero_page.add(idx)
        return ok

    def conv3x3_c64_wino_supported(self, N, H, W, C, K):
        return bool(self.c.hifihr_conv3x3_c64_wino_supported(int(N), int(H), int(W), int(C), int(K)))

    def conv3x3_c64_wino(self, x, U, bias, relu, y, stats, N, H, W):
        """64 -> 64 channel 3x3 / stride 1 / pad 1 convolution as register-resident Winograd F(2x2, 3x3) (include/hifihr.h)."""
        self.check(self.c.hifihr_conv3x3_c64_wino(_fp(x), _fp(U), _fp(bias), int(bool(relu)), _fp(y), _fp(stats), N, H, W, _stream_of(x)),
                   "hifihr_conv3x3_c64_wino")

    def conv3x3_c64_wino_res(self, x, U, res, y, N, H, W):
        """conv3x3_c64_wino with y = product + res (backward-data + the gradient of the input's other consumer; include/hifihr.h)."""
        self.check(self.c.hifihr_conv3x3_c64_wino_res(_fp(x), _fp(U), _fp(res), _fp(y), N, H, W, _stream_of(x)), "hifihr_conv3x3_c64_wino_res")

    def conv3x3_c64_bwd_pair_supported(self, N, H, W):
        return bool(self.c.hifihr_conv3x3_c64_bwd_pair_supported(int(N), int(H), int(W)))

    def conv3x3_c64_bwd_pair(self, dy, U_bwd, res, dx, x, dw, N, H, W, ws=None):
        """dx = conv3x3_c64_wino[_res](dy, U_bwd[, res]) and dw += conv2d_bwd_weight(x, dy) in ONE launch + the slab sum (include/hifihr.h)."""
        self.check(self.c.hifihr_conv3x3_c64_bwd_pair(_fp(dy), _fp(U_bwd), _fp(res), _fp(dx), _fp(x), _fp(dw), _fp(ws),
                                                      0 if ws is None else ws.numel() * ws.element_size(), N, H, W, _stream_of(dy)),
                   "hifihr_conv3x3_c64_bwd_pair")

    def conv3x3_c64_bwd_pair_slabs(self, dy, U_bwd, res, dx, x, slabs, N, H, W):
        """hifihr_conv3x3_c64_bwd_pair without its slab sum: -> number of weight-gradient slabs left in `slabs` (hifihr_conv_halo_wgrad_reduce_multi)."""
        import ctypes
        n = ctypes.c_int(0)
        self.check(self.c.hifihr_conv3x3_c64_bwd_pair_slabs(_fp(dy), _fp(U_bwd), _fp(res), _fp(dx), _fp(x), _fp(slabs), slabs.numel() * slabs.element_size(),
                                                            N, H, W, ctypes.cast(ctypes.pointer(n), c_void_p), _stream_of(dy)), "hifihr_conv3x3_c64_bwd_pair_slabs")
        return n.value

    def conv_halo_wgrad_reduce_multi(self, jobs):
        """jobs: [(slabs tensor, nslab, dw_acc tensor)] -- the slab sums of several 64 -> 64 layers in one launch."""
        import ctypes
        import struct
        raw = b"".join(struct.pack("<QQii", sl.data_ptr(), dw.data_ptr(), int(n), 0) for sl, n, dw in jobs)
        buf = ctypes.create_string_buffer(raw, len(raw))
        self.check(self.c.hifihr_conv_halo_wgrad_reduce_multi(ctypes.cast(buf, c_void_p), len(jobs), _stream_of(jobs[0][0])),
                   "hifihr_conv_halo_wgrad_reduce_multi")

    def wino_bn_input_supported(self, C, m):
        return bool(self.c.hifihr_wino_bn_input_supported(int(C), int(m)))

    def wino_bn_input_transform(self, x, stats, gamma, beta, residual, out, V, N, H, W, C, m, eps, momentum, save_mean, save_invstd,
                                running_mean, running_var):
        self.check(self.c.hifihr_wino_bn_input_transform(_fp(x), _fp(stats), _fp(gamma), _fp(beta), _fp(residual), _fp(out), _fp(V), N, H, W, C, m,
                                                         eps, momentum, _fp(save_mean), _fp(save_invstd), _fp(running_mean), _fp(running_var),
                                                         _stream_of(x)), "hifihr_wino_bn_input_transform")

    def wino_output_transform_bnred(self, Mm, x, out, gadd, save_mean, save_invstd, gamma, beta, red, g, N, H, W, C, m):
        self.check(self.c.hifihr_wino_output_transform_bnred(_fp(Mm), _fp(x), _fp(out), _fp(gadd), _fp(save_mean), _fp(save_invstd), _fp(gamma),
                                                             _fp(beta), _fp(red), _fp(g), N, H, W, C, m, _stream_of(Mm)),
                   "hifihr_wino_output_transform_bnred")

    def wino_bn_bwd_dual_transform(self, g, y, save_mean, save_invstd, gamma, red, V, Yt, N, H, W, K, m, dgamma_acc, dbeta_acc):
        self.check(self.c.hifihr_wino_bn_bwd_dual_transform(_fp(g), _fp(y), _fp(save_mean), _fp(save_invstd), _fp(gamma), _fp(red), _fp(V), _fp(Yt),
                                                            N, H, W, K, m, _fp(dgamma_acc), _fp(dbeta_acc), _stream_of(g)),
                   "hifihr_wino_bn_bwd_dual_transform")

    def bn_bwd_apply(self, g, x, save_mean, save_invstd, gamma, M, C, red, dx, dgamma_acc, dbeta_acc):
        self.check(self.c.hifihr_bn_bwd_apply(_fp(g), _fp(x), _fp(save_mean), _fp(save_invstd), _fp(gamma), M, C, _fp(red), _fp(dx),
                                              _fp(dgamma_acc), _fp(dbeta_acc), _stream_of(g)), "hifihr_bn_bwd_apply")

    def wino_gemm(self, V, U, M, N, H, W, C, K, ws=None, m=2):
        self.check(self.c.hifihr_wino_gemm_m(_fp(V), _fp(U), _fp(M), N, H, W, C, K, m, *self._ws(ws), _stream_of(V)), "hifihr_wino_gemm")

    def wino_input_dy_transform(self, dy, V, Yt, N, H, W, K, m=2):
        self.check(self.c.hifihr_wino_input_dy_transform_m(_fp(dy), _fp(V), _fp(Yt), N, H, W, K, m, _stream_of(dy)), "hifihr_wino_input_dy_transform")

    def conv2d_bwd_data_pre(self, dy, wt, dx, N, H, W, C, K, R, S, stride, pad, ws=None):
        wsp, wsb = self._ws(ws)
        self.check(self.c.hifihr_conv2d_bwd_data_pre(_fp(dy), _fp(wt), _fp(dx), N, H, W, C, K, R, S, stride, pad, wsp, wsb, _stream_of(dy)),
                   "hifihr_conv2d_bwd_data_pre")

    def conv2d_bwd_data_pre_res(self, dy, wt, res, dx, N, H, W, C, K, R, S, stride, pad, ws=None):
        """conv2d_bwd_data_pre with dx = product + res (include/hifihr.h)."""
        wsp, wsb = self._ws(ws)
        self.check(self.c.hifihr_conv2d_bwd_data_pre_res(_fp(dy), _fp(wt), _fp(res), _fp(dx), N, H, W, C, K, R, S, stride, pad, wsp, wsb,
                                                         _stream_of(dy)), "hifihr_conv2d_bwd_data_pre_res")

    def conv2d_bwd_data_pre_plus1x1_supported(self, N, H, W, C, K, R, S, stride, pad):
        return bool(self.c.hifihr_conv2d_bwd_data_pre_plus1x1_supported(int(N), int(H), int(W), int(C), int(K), int(R), int(S), int(stride), int(pad)))

    def conv2d_bwd_data_pre_plus1x1(self, dy, wt, dy2, wt2, dx, N, H, W, C, K, R, S, stride, pad):
        """dx = backward-data of the strided convolution (dy, wt) + backward-data of a 1x1 / same stride / pad 0 convolution of the same input
        (dy2 [N][OH][OW][K], wt2 [C][K]) in one launch (include/hifihr.h)."""
        self.check(self.c.hifihr_conv2d_bwd_data_pre_plus1x1(_fp(dy), _fp(wt), _fp(dy2), _fp(wt2), _fp(dx), N, H, W, C, K, R, S, stride, pad,
                                                             _stream_of(dy)), "hifihr_conv2d_bwd_data_pre_plus1x1")

    def conv2d_bwd_weight_plus1x1_supported(self, N, H, W, C, K, R, S, stride, pad):
        return bool(self.c.hifihr_conv2d_bwd_weight_plus1x1_supported(int(N), int(H), int(W), int(C), int(K), int(R), int(S), int(stride), int(pad)))

    def conv2d_bwd_weight_plus1x1(self, x, dy, dw, dy2, dw2, N, H, W, C, K, R, S, stride, pad):
        """dw += weight gradient of the strided convolution, dw2 += that of the 1x1 / same stride / pad 0 convolution of the same input, one launch."""
        self.check(self.c.hifihr_conv2d_bwd_weight_plus1x1(_fp(x), _fp(dy), _fp(dw), _fp(dy2), _fp(dw2), N, H, W, C, K, R, S, stride, pad,
                                                           _stream_of(x)), "hifihr_conv2d_bwd_weight_plus1x1")

    @staticmethod
    def prep_jobs(jobs, device):
        """[(src tensor, dst tensor, K, C, RS, kind)] -> device table of hifihr_prep_job (two pointers + four ints = 32 bytes)."""
        import struct
        raw = b"".join(struct.pack("<QQiiii", s.data_ptr(), d.data_ptr(), K, C, RS, kind) for s, d, K, C, RS, kind in jobs)
        return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)

    def weight_prep(self, table, njobs, blocks_per_job=64):
        self.check(self.c.hifihr_weight_prep(c_void_p(table.data_ptr()), njobs, blocks_per_job, _stream_of(table)), "hifihr_weight_prep")

    def freihand_batch(self, img_rgbx, mask, Ks, joints, verts, scales, packed, B, out, root_id=None, image_size=None):
        """out: dict with any of imgs, masks, segms_gt, Ks, Ps, joints, verts, j2d_gt, scales, idxs (device tensors, contiguous);
        root_id given: also root_xyz [B,1,3], joints_rel, verts_rel, cam_ndc [B,4] (hifihr_freihand_batch_step)."""
        n, H, W = img_rgbx.shape
        J, V = joints.shape[1], verts.shape[1]
        vp = lambda t: c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
        g = out.get
        for k in ("segms_gt", "idxs"):
            assert g(k) is None or g(k).dtype == torch.int64
        for t in out.values():
            assert t.is_contiguous()
        if root_id is None:
            self.check(self.c.hifihr_freihand_batch(vp(img_rgbx), vp(mask), _fp(Ks), _fp(joints), _fp(verts), _fp(scales), J, V, _ip(packed), B, H, W,
                                                    _fp(g("imgs")), _fp(g("masks")), vp(g("segms_gt")), _fp(g("Ks")), _fp(g("Ps")), _fp(g("joints")),
                                                    _fp(g("verts")), _fp(g("j2d_gt")), _fp(g("scales")), vp(g("idxs")), _stream_of(img_rgbx)),
                       "hifihr_freihand_batch")
            return
        # + the terms every training iteration derives from the batch: root_xyz, root-relative ground truth, the NDC camera
        self.check(self.c.hifihr_freihand_batch_step(vp(img_rgbx), vp(mask), _fp(Ks), _fp(joints), _fp(verts), _fp(scales), J, V, _ip(packed), B, H, W,
                                                     _fp(g("imgs")), _fp(g("masks")), vp(g("segms_gt")), _fp(g("Ks")), _fp(g("Ps")), _fp(g("joints")),
                                                     _fp(g("verts")), _fp(g("j2d_gt")), _fp(g("scales")), vp(g("idxs")), int(root_id), float(image_size or H),
                                                     _fp(g("root_xyz")), _fp(g("joints_rel")), _fp(g("verts_rel")), _fp(g("cam_ndc")), _stream_of(img_rgbx)),
                   "hifihr_freihand_batch_step")

    def ho3d_workspace_bytes(self, B, out_size):
        return int(self.c.hifihr_ho3d_workspace_bytes(B, out_size))

    def ho3d_batch(self, img_rgbx, hand_mask, Ks, uv21, xyz21, packed, B, out_size, ws, out):
        """out: dict with any of img_crop [B,3,S,S], hand_mask_crop [B,1,S,S], K_crop [B,3,3], uv21_crop [B,21,2], xyz21 [B,21,3]."""
        n, FH, FW = img_rgbx.shape
        vp = lambda t: c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
        g = out.get
        for t in out.values():
            assert t.is_contiguous() and t.dtype == torch.float32
        self.check(self.c.hifihr_ho3d_batch(vp(img_rgbx), vp(hand_mask), _fp(Ks), _fp(uv21), _fp(xyz21), FH, FW, _ip(packed), B, out_size,
                                            vp(ws), c_size_t(ws.numel() * ws.element_size()), _fp(g("img_crop")), _fp(g("hand_mask_crop")),
                                            _fp(g("K_crop")), _fp(g("uv21_crop")), _fp(g("xyz21")), _stream_of(img_rgbx)), "hifihr_ho3d_batch")

    def freihand_augment(self, img_rgbx, mask, idx, coef_fix, out_img, out_mask):
        """img_rgbx int32/uint8x4 [n,H,W], mask uint8 [n,H,W] (either None with its output), idx int32 [B], coef_fix int32 [B,6]."""
        ref = img_rgbx if img_rgbx is not None else mask
        H, W = ref.shape[1], ref.shape[2]
        ptr = lambda t: None if t is None else c_void_p(t.data_ptr())
        self.check(self.c.hifihr_freihand_augment(ptr(img_rgbx), ptr(mask), _ip(idx), _ip(coef_fix), idx.shape[0], H, W, _fp(out_img),
                                                  _fp(out_mask), _stream_of(idx)), "hifihr_freihand_augment")

    def procrustes_error(self, pred, gt, aligned, err_sum):
        B, N = pred.shape[0], pred.shape[1]
        self.check(self.c.hifihr_procrustes_error(_fp(pred), _fp(gt), B, N, _fp(aligned), _fp(err_sum), _stream_of(pred)),
                   "hifihr_procrustes_error")

    def wino_output_transform(self, M, y, stats, N, H, W, K, bias=None, act=0, m=2, mask=None):
        if mask is not None:
            assert stats is None and bias is None and not act and m == 4
            self.check(self.c.hifihr_wino_output_transform_mask_m(_fp(M), _fp(y), _fp(mask), N, H, W, K, m, _stream_of(M)),
                       "hifihr_wino_output_transform_mask_m")
            return
        if bias is not None or act:
            assert stats is None
            self.check(self.c.hifihr_wino_output_transform_act_m(_fp(M), _fp(y), _fp(bias), act, N, H, W, K, m, _stream_of(M)),
                       "hifihr_wino_output_transform_act")
            return
        self.check(self.c.hifihr_wino_output_transform_m(_fp(M), _fp(y), _fp(stats), N, H, W, K, m, _stream_of(M)), "hifihr_wino_output_transform")

    def wino_dy_transform(self, dy, Y, N, H, W, K, m=2):
        self.check(self.c.hifihr_wino_dy_transform_m(_fp(dy), _fp(Y), N, H, W, K, m, _stream_of(dy)), "hifihr_wino_dy_transform")

    def wino_wgrad_gemm(self, V, Y, dU_zeroed, N, H, W, C, K):
        self.check(self.c.hifihr_wino_wgrad_gemm(_fp(V), _fp(Y), _fp(dU_zeroed), N, H, W, C, K, _stream_of(V)), "hifihr_wino_wgrad_gemm")

    def wino_dw_transform(self, dU, dw_acc, K, C, clear=True):
        self.check(self.c.hifihr_wino_dw_transform(_fp(dU), _fp(dw_acc), K, C, int(bool(clear)), _stream_of(dU)), "hifihr_wino_dw_transform")

    def bgemm_nt_workspace_bytes(self, M, N, K, batch):
        return int(self.c.hifihr_bgemm_nt_workspace_bytes(M, N, K, batch))

    def bgemm_nt(self, A, B, C, M, N, K, batch, ws=None):
        self.check(self.c.hifihr_bgemm_nt(_fp(A), _fp(B), _fp(C), M, N, K, batch, *self._ws(ws), _stream_of(A)), "hifihr_bgemm_nt")

    # ---- RCCL wrapper (csrc/comm.hip); torch.distributed drives the exchange in this package, these bindings serve tests / other hosts
    def comm_unique_id(self) -> bytes:
        buf = ctypes.create_string_buffer(128)
        if self.c.hifihr_comm_get_unique_id(buf) != 0:
            raise HifihrError("hifihr_comm_get_unique_id: " + (self.c.hifihr_comm_last_error() or b"").decode())
        return buf.raw

    def comm_init(self, rank, world, uid: bytes):
        h = c_void_p()
        buf = ctypes.create_string_buffer(uid, 128)
        if self.c.hifihr_comm_init(ctypes.byref(h), rank, world, buf) != 0:
            raise HifihrError("hifihr_comm_init: " + (self.c.hifihr_comm_last_error() or b"").decode())
        return h

    def comm_allreduce(self, h, buf):
        if self.c.hifihr_comm_allreduce_f32(h, _fp(buf), buf.numel(), _stream_of(buf)) != 0:
            raise HifihrError("hifihr_comm_allreduce_f32: " + (self.c.hifihr_comm_last_error() or b"").decode())

    def comm_broadcast(self, h, buf, root=0):
        if self.c.hifihr_comm_broadcast_f32(h, _fp(buf), buf.numel(), root, _stream_of(buf)) != 0:
            raise HifihrError("hifihr_comm_broadcast_f32: " + (self.c.hifihr_comm_last_error() or b"").decode())

    def comm_destroy(self, h):
        self.c.hifihr_comm_destroy(h)

    def conv2d_describe(self, N, H, W, C, K, R, S, stride, pad, direction):
        """direction: 0 / False forward, 1 / True backward-data, 2 backward-weight."""
        buf = ctypes.create_string_buffer(64)
        self.check(self.c.hifihr_conv2d_describe(N, H, W, C, K, R, S, stride, pad, int(direction), buf, 64), "hifihr_conv2d_describe")
        return buf.value.decode()

    def bgemm_describe(self, tn, M, N, K, batch=16):
        buf = ctypes.create_string_buffer(96)
        self.check(self.c.hifihr_bgemm_describe_batch(int(bool(tn)), M, N, K, batch, buf, 96), "hifihr_bgemm_describe_batch")
        return buf.value.decode()

    def bgemm_tn_parts(self, M, N, T, batch):
        return int(self.c.hifihr_bgemm_tn_parts(M, N, T, batch))

    def bgemm_tn(self, A, B, Cparts, M, N, T, batch, parts):
        self.check(self.c.hifihr_bgemm_tn(_fp(A), _fp(B), _fp(Cparts), M, N, T, batch, parts, _stream_of(A)), "hifihr_bgemm_tn")

    def wino_wgrad_parts(self, N, H, W, C, K, m=2):
        return int(self.c.hifihr_wino_wgrad_parts_m(N, H, W, C, K, m))

    def wino_wgrad_gemm_parts(self, V, Y, dU_parts, N, H, W, C, K, parts, m=2):
        self.check(self.c.hifihr_wino_wgrad_gemm_parts_m(_fp(V), _fp(Y), _fp(dU_parts), N, H, W, C, K, parts, m, _stream_of(V)),
                   "hifihr_wino_wgrad_gemm_parts")

    def wino4_dw_transform_multi(self, jobs):
        """jobs: [(dU_parts tensor, parts, dw_acc tensor, K, C)] -- the F(4x4) weight-gradient transforms of several layers in one launch
        (hifihr_wino4_dw_transform_multi: a host array of hifihr_wino_dw_job, 32 bytes each, passed in the kernel arguments)."""
        import ctypes
        import struct
        raw = b"".join(struct.pack("<QQiiii", du.data_ptr(), dw.data_ptr(), int(parts), int(K), int(C), 0) for du, parts, dw, K, C in jobs)
        buf = ctypes.create_string_buffer(raw, len(raw))
        self.check(self.c.hifihr_wino4_dw_transform_multi(ctypes.cast(buf, c_void_p), len(jobs), _stream_of(jobs[0][0])),
                   "hifihr_wino4_dw_transform_multi")

    def wino_dw_transform_parts(self, dU_parts, parts, dw_acc, K, C, m=2):
        self.check(self.c.hifihr_wino_dw_transform_parts_m(_fp(dU_parts), parts, _fp(dw_acc), K, C, m, _stream_of(dU_parts)),
                   "hifihr_wino_dw_transform_parts")

    def weight_transpose(self, w, wt, K, RS, C):
        self.check(self.c.hifihr_weight_transpose(_fp(w), _fp(wt), K, RS, C, _stream_of(w)), "hifihr_weight_transpose")

    def se_pool(self, x, B, HW, C, mean_zeroed):
        self.check(self.c.hifihr_se_pool(_fp(x), B, HW, C, _fp(mean_zeroed), _stream_of(x)), "hifihr_se_pool")

    def se_scale(self, x, gate, add, add_scale, B, HW, C, y):
        self.check(self.c.hifihr_se_scale(_fp(x), _fp(gate), _fp(add), float(add_scale), B, HW, C, _fp(y), _stream_of(x)), "hifihr_se_scale")

    def drop_connect_add(self, x, skip, u, keep, B, per_sample, out):
        self.check(self.c.hifihr_drop_connect_add(_fp(x), _fp(skip), _fp(u), float(keep), B, int(per_sample), _fp(out), _stream_of(x)),
                   "hifihr_drop_connect_add")

    def wino_tiles(self, N, H, W, m):
        return int(self.c.hifihr_wino_tiles(int(N), int(H), int(W), int(m)))

    def wino_tiles_computed(self, N, H, W, m):
        return int(self.c.hifihr_wino_tiles_computed(int(N), int(H), int(W), int(m)))

    def se_mlp_supported(self, C, SQ):
        return bool(self.c.hifihr_se_mlp_supported(int(C), int(SQ)))

    def se_mlp_fwd(self, mean_acc, w1, b1, w2t, b2, B, C, SQ, mean, z1, h1, gate):
        self.check(self.c.hifihr_se_mlp_fwd(_fp(mean_acc), _fp(w1), _fp(b1), _fp(w2t), _fp(b2), B, C, SQ, _fp(mean), _fp(z1), _fp(h1), _fp(gate),
                                            _stream_of(mean_acc)), "hifihr_se_mlp_fwd")

    def se_mlp_bwd(self, dgate_acc, gate, z1, h1, mean, w1, w2t, B, C, SQ, dz2, dz1, dmean, dw1, db1, dw2, db2):
        self.check(self.c.hifihr_se_mlp_bwd(_fp(dgate_acc), _fp(gate), _fp(z1), _fp(h1), _fp(mean), _fp(w1), _fp(w2t), B, C, SQ, _fp(dz2), _fp(dz1),
                                            _fp(dmean), _fp(dw1), _fp(db1), _fp(dw2), _fp(db2), _stream_of(dgate_acc)), "hifihr_se_mlp_bwd")

    def se_bwd_gate(self, dy, x, B, HW, C, dgate_zeroed):
        self.check(self.c.hifihr_se_bwd_gate(_fp(dy), _fp(x), B, HW, C, _fp(dgate_zeroed), _stream_of(x)), "hifihr_se_bwd_gate")

    def linear_fwd(self, x, w, b, act, y, bn=None, z=None):
        """bn: None or (gamma, beta, eps, momentum, running_mean, running_var, z, save_mean, save_invstd); z: pre-activation
        buffer for act 2 (swish) without batch-norm."""
        B, I = x.shape
        O = w.shape[0]
        gamma, beta, eps, mom, rm, rv, z_bn, sm, si = bn if bn is not None else (None, None, 0.0, 0.0, None, None, None, None, None)
        z = z_bn if bn is not None else z
        self.check(self.c.hifihr_linear_fwd(_fp(x), _fp(w), _fp(b), B, I, O, int(act), _fp(gamma), _fp(beta), float(eps), float(mom),
                                            _fp(rm), _fp(rv), _fp(y), _fp(z), _fp(sm), _fp(si), _stream_of(x)), "hifihr_linear_fwd")

    def linear_bwd(self, dy, y, x, w, act, dz, dW_acc, db_acc, dx, bn=None, z=None):
        """bn: None or (gamma, z, save_mean, save_invstd, dgamma_acc, dbeta_acc); z: pre-activation of an act-2 layer."""
        B, I = x.shape
        O = w.shape[0]
        gamma, z_bn, sm, si, dg, dbt = bn if bn is not None else (None,) * 6
        z = z_bn if bn is not None else z
        self.check(self.c.hifihr_linear_bwd(_fp(dy), _fp(y), _fp(x), _fp(w), B, I, O, int(act), _fp(gamma), _fp(z), _fp(sm), _fp(si),
                                            _fp(dz), _fp(dW_acc), _fp(db_acc), _fp(dg), _fp(dbt), _fp(dx), _stream_of(x)),
                   "hifihr_linear_bwd")

    @staticmethod
    def _descs(members):
        """members: list of dicts with tensors x, w, b, y and act (+ dy, dz, dW, db, dx for the backward) -> ctypes array."""
        arr = (_LinearDesc * len(members))()
        p = lambda t: None if t is None else t.data_ptr()
        for d, m in zip(arr, members):
            d.x, d.w, d.b, d.y = p(m["x"]), p(m["w"]), p(m.get("b")), p(m["y"])
            d.B, d.I, d.O, d.act = m["x"].shape[0], m["x"].shape[1], m["w"].shape[0], int(m["act"])
            d.dy, d.dz_scratch, d.dW_acc, d.db_acc, d.dx = p(m.get("dy")), p(m.get("dz")), p(m.get("dW")), p(m.get("db")), p(m.get("dx"))
        return arr

    def linear_fwd_group(self, members):
        self.check(self.c.hifihr_linear_fwd_group(self._descs(members), len(members), _stream_of(members[0]["x"])), "hifihr_linear_fwd_group")

    def linear_bwd_group(self, members):
        self.check(self.c.hifihr_linear_bwd_group(self._descs(members), len(members), _stream_of(members[0]["x"])), "hifihr_linear_bwd_group")

    def mmpool_fwd(self, x, p, B, HW, C, y, argmax, xmax, xavg):
        self.check(self.c.hifihr_mmpool_fwd(_fp(x), _fp(p), B, HW, C, _fp(y), _ip(argmax), _fp(xmax), _fp(xavg), _stream_of(x)),
                   "hifihr_mmpool_fwd")

    def mmpool_bwd(self, gy, p, argmax, xmax, xavg, B, HW, C, dx, dp_acc):
        self.check(self.c.hifihr_mmpool_bwd(_fp(gy), _fp(p), _ip(argmax), _fp(xmax), _fp(xavg), B, HW, C, _fp(dx), _fp(dp_acc),
                                            _stream_of(gy)), "hifihr_mmpool_bwd")

    def maxpool2d_fwd(self, x, N, H, W, C, k, s, p, y, tap):
        assert tap.dtype == torch.uint8 and tap.is_contiguous()
        self.check(self.c.hifihr_maxpool2d_fwd(_fp(x), N, H, W, C, k, s, p, _fp(y), c_void_p(tap.data_ptr()), _stream_of(x)),
                   "hifihr_maxpool2d_fwd")

    def maxpool2d_fwd_flat(self, x, N, H, W, C, k, s, p, y_flat, tap):
        """maxpool2d_fwd with the output as the [N, C * OH * OW] matrix of `y.view(N, -1)` (NCHW order; include/hifihr.h)."""
        self.check(self.c.hifihr_maxpool2d_fwd_flat(_fp(x), N, H, W, C, k, s, p, _fp(y_flat), c_void_p(tap.data_ptr()), _stream_of(x)),
                   "hifihr_maxpool2d_fwd_flat")

    def maxpool2d_bwd_flat(self, gy_flat, tap, N, H, W, C, k, s, p, dx):
        self.check(self.c.hifihr_maxpool2d_bwd_flat(_fp(gy_flat), c_void_p(tap.data_ptr()), N, H, W, C, k, s, p, _fp(dx), _stream_of(gy_flat)),
                   "hifihr_maxpool2d_bwd_flat")

    def maxpool2d_bwd(self, gy, tap, N, H, W, C, k, s, p, dx, relu_y=None):
        if relu_y is not None:
            self.check(self.c.hifihr_maxpool2d_bwd_relu(_fp(gy), c_void_p(tap.data_ptr()), _fp(relu_y), N, H, W, C, k, s, p, _fp(dx),
                                                        _stream_of(gy)), "hifihr_maxpool2d_bwd_relu")
            return
        self.check(self.c.hifihr_maxpool2d_bwd(_fp(gy), c_void_p(tap.data_ptr()), N, H, W, C, k, s, p, _fp(dx), _stream_of(gy)),
                   "hifihr_maxpool2d_bwd")

    def conv2d_bwd_data(self, dy, w, dx, scratch, N, H, W, C, K, R, S, stride, pad, ws=None):
        self.check(self.c.hifihr_conv2d_bwd_data(_fp(dy), _fp(w), _fp(dx), _fp(scratch), N, H, W, C, K, R, S, stride, pad,
                                                 *self._ws(ws), _stream_of(dy)), "hifihr_conv2d_bwd_data")

    def conv2d_wgrad_workspace_bytes(self, N, H, W, C, K, R, S, stride, pad):
        return int(self.c.hifihr_conv2d_wgrad_workspace_bytes(N, H, W, C, K, R, S, stride, pad))

    def conv2d_bwd_weight(self, x, dy, dw, N, H, W, C, K, R, S, stride, pad, ws=None):
        p, n = self._ws(ws)
        self.check(self.c.hifihr_conv2d_bwd_weight_ws(_fp(x), _fp(dy), _fp(dw), N, H, W, C, K, R, S, stride, pad, p, n, _stream_of(x)),
                   "hifihr_conv2d_bwd_weight")

    def image_to_nhwc4(self, images, out):
        B, _, H, W = images.shape
        self.check(self.c.hifihr_image_to_nhwc4(_fp(images), _fp(out), B, H, W, _stream_of(images)), "hifihr_image_to_nhwc4")

    def image_to_nhwc4_padded(self, images, out, pad4, normalize):
        """pad4 = (left, right, top, bottom) like F.pad."""
        B, _, H, W = images.shape
        pl, pr, pt, pb = pad4
        self.check(self.c.hifihr_image_to_nhwc4_padded(_fp(images), _fp(out), B, H, W, pt, pl, pb, pr, int(bool(normalize)),
                                                       _stream_of(images)), "hifihr_image_to_nhwc4_padded")

    # ---- SSIM ----------------------------------------------------------
    def ssim_partial_count(self, planes, H, W):
        return int(self.c.hifihr_ssim_partial_count(planes, H, W))

    def ssim_fwd(self, window, img1, img2, partial, dA, dB, dC):
        planes, H, W = img1.shape[0] * img1.shape[1], img1.shape[2], img1.shape[3]
        self.check(self.c.hifihr_ssim_fwd(window, _fp(img1), _fp(img2), planes, H, W, _fp(partial), _fp(dA), _fp(dB), _fp(dC),
                                          _stream_of(img1)), "hifihr_ssim_fwd")

    def ssim_finish(self, partial, scale, offset, out):
        self.check(self.c.hifihr_ssim_finish(_fp(partial), partial.numel(), c_float(scale), c_float(offset), _fp(out), _stream_of(partial)),
                   "hifihr_ssim_finish")

    def ssim_bwd_scaled(self, window, img1, img2, dA, dB, dC, grad_out, out_scale, gimg1):
        planes, H, W = img1.shape[0] * img1.shape[1], img1.shape[2], img1.shape[3]
        self.check(self.c.hifihr_ssim_bwd_scaled(window, _fp(img1), _fp(img2), _fp(dA), _fp(dB), _fp(dC), _fp(grad_out), c_float(out_scale),
                                                 planes, H, W, _fp(gimg1), _stream_of(img1)), "hifihr_ssim_bwd_scaled")

    def ssim_bwd(self, window, img1, img2, dA, dB, dC, grad_out, gimg1):
        planes, H, W = img1.shape[0] * img1.shape[1], img1.shape[2], img1.shape[3]
        self.check(self.c.hifihr_ssim_bwd(window, _fp(img1), _fp(img2), _fp(dA), _fp(dB), _fp(dC), _fp(grad_out), planes, H, W,
                                          _fp(gimg1), _stream_of(img1)), "hifihr_ssim_bwd")

    # ---- optimizer ---------------------------------------------------
    def adam_step(self, p, g, m, v, grad_scale, lr, beta1, beta2, eps, weight_decay, step):
        n = p.numel()
        assert g.numel() == n and m.numel() == n and v.numel() == n
        self.check(self.c.hifihr_adam_step(_fp(p), _fp(g), _fp(m), _fp(v), c_size_t(n), c_float(grad_scale), c_float(lr),
                                           c_float(beta1), c_float(beta2), c_float(eps), c_float(weight_decay), int(step),
                                           _stream_of(p)), "hifihr_adam_step")

    @staticmethod
    def adam_state_image(lr, beta1, beta2, step):
        """The 48 bytes of hifihr_adam_step_counted's state (include/hifihr.h) as a uint8 CPU tensor:
        f64 lr, beta1, beta2, beta1^step, beta2^step, i32 step, i32 0."""
        import struct
        b1, b2 = float(beta1), float(beta2)
        return torch.frombuffer(bytearray(struct.pack("<dddddii", float(lr), b1, b2, b1 ** int(step), b2 ** int(step), int(step), 0)),
                                dtype=torch.uint8).clone()

    def adam_step_counted(self, p, g, m, v, grad_scale, eps, weight_decay, state):
        n = p.numel()
        assert state.dtype == torch.uint8 and state.numel() == int(self.c.hifihr_adam_state_bytes()) and state.device == p.device
        self.check(self.c.hifihr_adam_step_counted(_fp(p), _fp(g), _fp(m), _fp(v), c_size_t(n), c_float(grad_scale), c_float(eps),
                                                   c_float(weight_decay), c_void_p(state.data_ptr()), _stream_of(p)), "hifihr_adam_step_counted")

    def adam_step_dyn(self, p, g, m, v, grad_scale, beta1, beta2, eps, weight_decay, dyn):
        n = p.numel()
        self.check(self.c.hifihr_adam_step_dyn(_fp(p), _fp(g), _fp(m), _fp(v), c_size_t(n), c_float(grad_scale), c_float(beta1),
                                               c_float(beta2), c_float(eps), c_float(weight_decay), _fp(dyn), _stream_of(p)),
                   "hifihr_adam_step_dyn")

    # ---- renderer ----------------------------------------------------
    def renderer_create(self, faces, V, image_size=224, aa=3, ambient=(0.5, 0.5, 0.5), mat_diffuse=(0.8, 0.8, 0.8),
                        specular=(0.04, 0.04, 0.04), shininess=30.0, background=(1.0, 1.0, 1.0)) -> c_void_p:
        import numpy as np
        f = np.ascontiguousarray(faces, dtype=np.int32)
        h = c_void_p()
        a, m, s, b = (_np_fp(x) for x in (ambient, mat_diffuse, specular, background))
        self.check(self.c.hifihr_renderer_create(ctypes.byref(h), f.ctypes.data_as(_c_int_p), int(V), int(f.shape[0]),
                                                 int(image_size), int(aa), a[1], m[1], s[1], c_float(shininess), b[1]),
                   "hifihr_renderer_create")
        return h

    def texture_pca_fwd(self, coef, basis, mean, out):
        B, K = coef.shape
        n = basis.shape[1]
        self.check(self.c.hifihr_texture_pca_fwd(_fp(coef), _fp(basis), _fp(mean), B, K, c_long(n), _fp(out), _stream_of(coef)), "hifihr_texture_pca_fwd")

    def texture_pca_bwd(self, gtex, basis, dcoef_zeroed):
        B, K = dcoef_zeroed.shape
        n = basis.shape[1]
        self.check(self.c.hifihr_texture_pca_bwd(_fp(gtex), _fp(basis), B, K, c_long(n), _fp(dcoef_zeroed), _stream_of(gtex)), "hifihr_texture_pca_bwd")

    def renderer_set_light_mode(self, h, point_lights):
        self.check(self.c.hifihr_renderer_set_light_mode(h, int(bool(point_lights))), "hifihr_renderer_set_light_mode")

    def renderer_destroy(self, h):
        self.c.hifihr_renderer_destroy(h)

    def render_workspace_bytes(self, h, B) -> int:
        return int(self.c.hifihr_render_workspace_bytes(h, int(B)))

    def render_fwd(self, h, verts, vcolors, cam, light_color, light_dir, rgba, face_id, ws):
        B = verts.shape[0]
        batched = 1 if vcolors.dim() == 3 else 0
        self.check(self.c.hifihr_render_fwd(h, _fp(verts), _fp(vcolors), batched, _fp(cam), _fp(light_color), _fp(light_dir), B,
                                            _fp(rgba), _ip(face_id), c_void_p(ws.data_ptr()), _stream_of(verts)),
                   "hifihr_render_fwd")

    def renderer_set_uv(self, h, faces_uvs, verts_uvs):
        import numpy as np
        fu = np.ascontiguousarray(faces_uvs, dtype=np.int32)
        vu = np.ascontiguousarray(verts_uvs, dtype=np.float32)
        self.check(self.c.hifihr_renderer_set_uv(h, fu.ctypes.data_as(_c_int_p), vu.ctypes.data_as(_c_float_p), vu.shape[0]),
                   "hifihr_renderer_set_uv")

    def render_uv_scratch_bytes(self, h, B):
        return int(self.c.hifihr_render_uv_scratch_bytes(h, B))

    def render_fwd_uv(self, h, verts, maps, cam, light_color, light_dir, rgba, face_id, texels, ws):
        B, TH, TW = maps.shape[0], maps.shape[1], maps.shape[2]
        self.check(self.c.hifihr_render_fwd_uv(h, _fp(verts), _fp(maps), TH, TW, _fp(cam), _fp(light_color), _fp(light_dir), B, _fp(rgba),
                                               _ip(face_id), _fp(texels), c_void_p(ws.data_ptr()), _stream_of(verts)), "hifihr_render_fwd_uv")

    def render_bwd_uv(self, h, verts, maps, cam, light_color, light_dir, face_id, grad_rgba, texels, gtexels, gverts, gmaps, glc, gld, ws):
        B, TH, TW = maps.shape[0], maps.shape[1], maps.shape[2]
        self.check(self.c.hifihr_render_bwd_uv(h, _fp(verts), _fp(maps), TH, TW, _fp(cam), _fp(light_color), _fp(light_dir), _ip(face_id),
                                               _fp(grad_rgba), B, _fp(texels), _fp(gtexels), _fp(gverts), _fp(gmaps), _fp(glc), _fp(gld),
                                               c_void_p(ws.data_ptr()), _stream_of(verts)), "hifihr_render_bwd_uv")

    def render_bwd(self, h, verts, cam, light_color, light_dir, face_id, grad_rgba, gverts, gvcolors, glc, gld, ws):
        B = verts.shape[0]
        self.check(self.c.hifihr_render_bwd(h, _fp(verts), _fp(cam), _fp(light_color), _fp(light_dir), _ip(face_id),
                                            _fp(grad_rgba), B, _fp(gverts), _fp(gvcolors), _fp(glc), _fp(gld),
                                            c_void_p(ws.data_ptr()), _stream_of(verts)), "hifihr_render_bwd")


_LIB = None


def get_lib() -> HifihrLib:
    """The product library (HIP, in-tree).  Raises HifihrError when it has not been built."""
    global _LIB
    if _LIB is None:
        _LIB = HifihrLib(LIB_PATH)
    return _LIB


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HifihrError("hifihr_amd ops run on the GPU only (HIP kernels); got a CPU tensor. "
                              "There is no CPU fallback: use oracle/ only from tests.")
