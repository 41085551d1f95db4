"""ctypes binding of librfx.so (the C ABI in include/rfx_api.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C ransac-flow_amd/csrc``.  There is
no fallback: if the shared object is missing or a symbol cannot be resolved the import of any op fails
loudly (RuntimeError), and every op refuses tensors that are not on a HIP device.
"""
import ctypes
import os
from ctypes import c_int, c_int32, c_int64, c_uint64, c_float, c_double, c_void_p, c_size_t, c_longlong, c_char_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RFX_LIB") or os.path.normpath(os.path.join(_HERE, "..", "librfx.so"))  # RFX_LIB: experiments

# name -> (restype, argtypes); mirrors include/rfx_api.h one to one
SIGNATURES = {
    "rfx_version": (c_char_p, []),
    "rfx_abi_version": (c_int, []),
    "rfx_group_begin": (c_int, []),
    "rfx_group_end": (c_int, [c_void_p]),
    "rfx_group_abort": (c_int, []),
    "rfx_group_side_streams": (c_int, [c_int]),
    "rfx_conv2d_f32": (c_int, [c_void_p] * 7 + [c_int] * 10 + [c_void_p]),
    "rfx_conv2d_dilated_f32": (c_int, [c_void_p] * 7 + [c_int] * 11 + [c_void_p]),
    "rfx_conv1x1_split_f32": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p]),
    "rfx_conv3x3_split_f32": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p]),
    "rfx_conv3x3_split_s2_f32": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p]),
    "rfx_conv1x1_split_strided_f32": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p]),
    "rfx_adaptive_avgpool2d_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "rfx_softmax_accum_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_longlong, c_float, c_int, c_void_p]),
    "rfx_argmax_mask_f32": (c_int, [c_void_p, c_int, c_int, c_longlong, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rfx_conv2d_tile_variant": (c_int, [c_int] * 4),
    "rfx_conv2d_kernel_id": (c_int, [c_int] * 9),
    "rfx_conv3x3_f32": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p]),
    "rfx_conv3x3_kernel_id": (c_int, [c_int] * 6),
    "rfx_conv3x3_s2_f32": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p]),
    "rfx_conv3x3_conv1x1_kernel_id": (c_int, [c_int] * 4),
    "rfx_conv3x3_conv1x1_f32": (c_int, [c_void_p] * 4 + [c_int] + [c_void_p] * 4 + [c_int, c_void_p] + [c_int] * 6 + [c_void_p]),
    "rfx_maxpool2d_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "rfx_blurpool2d_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "rfx_maxblurpool2d_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "rfx_stem_conv3x3_maxblur_f32": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "rfx_stem_conv7x7_maxpool_f32": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "rfx_l2norm_nchw_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_longlong, c_longlong, c_void_p]),
    "rfx_flow_head_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "rfx_resize_bilinear_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "rfx_lanczos_pass_u8": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p, c_void_p] + [c_int] * 3 + [c_void_p]),
    "rfx_u8_to_f32_chw": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p] * 3),
    "rfx_copy_cols_f32": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p]),
    "rfx_corr_neigh_f32": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p]),
    "rfx_corr_neigh_variant_f32": (c_int, [c_void_p] * 3 + [c_int] * 6 + [c_void_p]),
    "rfx_corr_neigh_bidir_f32": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "rfx_corr_timing": (c_int, [c_int]),
    "rfx_corr_timing_collect": (c_int, [c_void_p, c_int]),
    "rfx_warp_grid_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 3 + [c_void_p]),
    "rfx_grid_sample_f32": (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_void_p]),
    "rfx_compose_flow_f32": (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p]),
    "rfx_flow_grad_clamp_f32": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "rfx_merge_multi_h_f32": (c_int, [c_void_p, c_void_p, c_longlong, c_void_p, c_void_p, c_int, c_longlong, c_float, c_int,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "rfx_match_score_f32": (c_int, [c_void_p, c_longlong, c_void_p, c_void_p, c_int, c_longlong, c_void_p, c_void_p]),
    "rfx_remove_small_cc_ws_bytes": (c_size_t, [c_int] * 3),
    "rfx_remove_small_cc_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p]),
    "rfx_mutual_nn_ws_bytes": (c_size_t, [c_int, c_int]),
    "rfx_mutual_nn_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int] + [c_void_p] * 5 + [c_int, c_void_p]),
    "rfx_mutual_nn_batched_f32": (c_int, [c_void_p, c_int, c_int, c_longlong, c_void_p, c_int, c_int, c_longlong, c_int]
                                  + [c_void_p] * 5 + [c_int, c_int, c_void_p]),
    "rfx_dlt4_homography": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "rfx_prediction_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "rfx_score_hypotheses": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_float] + [c_void_p] * 5),
    "rfx_ransac_ws_bytes": (c_size_t, [c_int, c_int]),
    "rfx_ransac_h4": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_float] + [c_void_p] * 5),
    "rfx_ransac_batched_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "rfx_ransac_h4_batched_stage": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_float] + [c_void_p] * 4
                                    + [c_int, c_int, c_void_p]),
    "rfx_ransac_degenerate_list": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "rfx_ransac_patch_h": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "rfx_ransac_degenerate_gather": (c_int, [c_void_p] * 4 + [c_int, c_void_p, c_int, c_int] + [c_void_p] * 5 + [c_int, c_void_p]),
    "rfx_ransac_patch_h_rows": (c_int, [c_void_p, c_int, c_int, c_int] + [c_void_p] * 4 + [c_int, c_void_p]),
    "rfx_dlt4_homography_flags": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "rfx_ransac_h4_batched": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_float] + [c_void_p] * 4
                              + [c_int, c_void_p]),
    "rfx_gather_matches_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int] + [c_void_p] * 6 + [c_int, c_void_p]),
    "rfx_draw_samples_i64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_uint64, c_uint64, c_void_p, c_void_p]),
    "rfx_filter_matches_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 4
                               + [c_void_p] * 9),
    "rfx_keep_mask_f32": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p, c_void_p]),
    "rfx_multih_accept_ws_bytes": (c_size_t, [c_int]),
    "rfx_multih_accept_f32": (c_int, [c_void_p] * 4 + [c_int] * 3 + [c_void_p] * 3 + [c_double, c_int] + [c_void_p] * 7
                              + [c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_longlong] + [c_int] * 5 + [c_void_p]),
}

ABI_VERSION = 10    # RFX_ABI_VERSION of the include/rfx_api.h these prototypes mirror

_lib = None


def load():
    """Load librfx.so and attach prototypes.  Raises RuntimeError when the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must load its bundled HIP runtime first: librfx.so then binds to the SAME libamdhip64 instance
    # (two runtimes in one process -> hipErrorNoDevice on launch).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "librfx.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError("librfx.so is missing symbol %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    if lib.rfx_abi_version() != ABI_VERSION:
        raise RuntimeError("librfx.so has ABI revision %d, this binding was written against %d (stale build?): rebuild with "
                           "`make -C ransac-flow_amd/csrc`" % (lib.rfx_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


class RfxError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        kind = "bad argument" if rc == -1 else ("limit exceeded" if rc == -2 else "hipError_t %d" % rc)
        raise RfxError("%s failed: %s" % (what, kind))
