"""ctypes binding of the C-ABI kernel library (include/surf_hip.h -> surf_amd/libsurf_hip.so).

The library is the product path: there is no CPU fallback.  Importing this module without the
built library raises; calling a kernel without a GPU fails inside HIP.
"""
import ctypes
import os
import subprocess

import torch  # noqa: F401  (load PyTorch's HIP runtime first: one runtime per process, whichever import order the caller uses)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SURF_HIP_LIB", os.path.join(_HERE, "libsurf_hip.so"))

# must equal SURF_ABI_VERSION of include/surf_hip.h (tests/test_host_modules.py compares the two texts); lib() refuses a
# library built from another header
ABI_VERSION = 39

c_f32p = ctypes.c_void_p
c_ptr = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_float = ctypes.c_float

# name -> (restype, argtypes); mirrors include/surf_hip.h one to one
SIGNATURES = {
    "surf_abi_version": (c_int, []),
    "surf_pack_texel4": (c_int, [c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "surf_ray_setup": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr,
                               c_int, c_ptr, c_float, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_sdf_packed_floats": (c_i64, []),
    "surf_sdf_pack_weights": (c_int, [c_ptr, c_ptr, c_ptr]),
    "surf_sdf_scratch_bytes": (c_i64, [c_i64]),
    "surf_sdf_mlp": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_sdf_bf16_packed_bytes": (c_i64, []),
    "surf_sdf_bf16_scratch_bytes": (c_i64, [c_i64]),
    "surf_sdf_pack_weights_bf16": (c_int, [c_ptr, c_ptr, c_ptr]),
    "surf_sdf_mlp_bf16x3": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_sdf_mlp_bf16x3_dn": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_sdf_lattice_bf16x3": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_float, c_ptr]),
    "surf_sdf_lattice_f16x2": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_float, c_ptr]),
    "surf_sdf_mlp_f16x2_dn": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_blend_split_dn": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_int,
                                    c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_sdf_f16_packed_bytes": (c_i64, []),
    "surf_sdf_f16_scratch_bytes": (c_i64, [c_i64]),
    "surf_sdf_pack_weights_f16": (c_int, [c_ptr, c_ptr, c_ptr]),
    "surf_sdf_mlp_f16x2": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_ptloss_terms": (c_int, [c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_composite_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_float, c_float, c_ptr,
                                        c_ptr, c_ptr, c_float, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_composite_backward_s": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_float, c_float, c_ptr,
                                          c_ptr, c_ptr, c_float, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_sdf_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_blend_backward_row_floats": (c_int, []),
    "surf_blend_backward": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_colgram_workspace_floats": (c_i64, [c_i64, c_int, c_int]),
    "surf_colgram": (c_int, [c_ptr, c_int, c_int, c_ptr, c_int, c_int, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr]),
    "surf_colgram_p": (c_int, [c_ptr, c_int, c_int, c_ptr, c_int, c_int, c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr]),
    "surf_patch_warp_tangent": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_lncc_jvp": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr]),
    "surf_crossing_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_lncc": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr]),
    "surf_lncc_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr]),
    "surf_sdf_smooth_packed_floats": (c_i64, []),
    "surf_sdf_smooth_pack_weights": (c_int, [c_ptr, c_ptr, c_ptr]),
    "surf_sdf_smooth": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_blend_raw_floats": (c_int, []),
    "surf_blend_packed_floats": (c_int, []),
    "surf_blend_pack_weights": (c_int, [c_ptr, c_ptr]),
    "surf_blend": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                           c_ptr, c_ptr]),
    "surf_blend_split_packed_bytes": (c_i64, [c_int]),
    "surf_blend_pack_weights_split": (c_int, [c_ptr, c_ptr, c_int]),
    "surf_blend_split_scratch_bytes": (c_i64, [c_i64, c_int]),
    "surf_blend_split": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_int,
                                 c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_mc_classify": (c_int, [c_ptr, c_int, c_int, c_int, ctypes.c_double, c_ptr, c_ptr]),
    "surf_mc_workspace_ints": (c_i64, [c_i64]),
    "surf_mc_count": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr]),
    "surf_mc_emit": (c_int, [c_ptr, c_int, c_int, c_int, ctypes.c_double, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_raster_first_hit": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "surf_composite": (c_int, [c_ptr] * 9 + [c_int, c_int, c_float, c_float] + [c_ptr] * 13),
    "surf_upsample_bilinear_t4": (c_int, [c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "surf_surface_points": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr]),
    "surf_patch_warp": (c_int, [c_ptr, c_ptr, c_int, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr]),
    "surf_upsample_filter": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_float, c_ptr, c_ptr]),
    "surf_costvol": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                             c_ptr]),
    "surf_compact_workspace_ints": (c_i64, [c_i64]),
    "surf_compact": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_gather_rows": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "surf_compose_index": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    "surf_densify": (c_int, [c_ptr, c_ptr, c_int, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_spconv": (c_int, [c_ptr, c_int, c_ptr, c_int, c_ptr, c_i64, c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_spconv_rows16": (c_int, [c_ptr, c_int, c_ptr, c_int, c_ptr, c_i64, c_int, c_ptr, c_int, c_ptr, c_ptr]),
    "surf_rows_to_bf16": (c_int, [c_ptr, c_i64, c_ptr, c_ptr]),
    "surf_bn_relu_apply16": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_bn_relu_backward16": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_bn_workspace_bytes": (c_i64, [c_int]),
    "surf_bn_train_affine": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr, ctypes.c_float, ctypes.c_float, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_matching_depth_backward": (c_int, [c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr,
                                              c_int, c_ptr, ctypes.c_float, ctypes.c_float, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr,
                                              c_ptr]),
    "surf_densify_backward": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr]),
    "surf_scatter_rows_add": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "surf_costvol_backward_workspace_floats": (c_i64, [c_i64, c_int, c_int, c_int]),
    "surf_costvol_backward_workspace_floats_for": (c_i64, [c_i64, c_int, c_ptr]),
    "surf_costvol_backward": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_conv3x3_wgrad_workspace_floats": (c_i64, [c_int, c_int, c_int, c_int, c_int]),
    "surf_conv3x3_wgrad": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr]),
    "surf_conv3x3_wgrad_p": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_int, c_ptr]),
    "surf_ptloss_backward": (c_int, [c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_sdf_smooth_backward": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_bn_relu_backward": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_inorm_backward_workspace_bytes": (c_i64, [c_int, c_int]),
    "surf_inorm_relu_backward": (c_int, [c_ptr, c_ptr, c_int, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_bn_relu_apply": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_spconv_wgrad": (c_int, [c_ptr, c_int, c_ptr, c_int, c_ptr, c_i64, c_int, c_ptr, c_int, c_ptr, c_ptr]),
    "surf_spconv_wgrad_mfma_supported": (c_int, [c_int, c_int]),
    "surf_spconv_wgrad_mfma": (c_int, [c_ptr, c_int, c_ptr, c_int, c_ptr, c_i64, c_int, c_ptr, c_int, c_ptr, c_ptr]),
    "surf_spconv_packed_bytes": (c_i64, [c_int, c_int]),
    "surf_spconv_pack_weights": (c_int, [c_ptr, c_int, c_int, c_ptr, c_ptr]),
    "surf_spconv_mfma": (c_int, [c_ptr, c_int, c_ptr, c_int, c_ptr, c_i64, c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr]),
    "surf_coords_bbox": (c_int, [c_ptr, c_i64, c_ptr, c_ptr]),
    "surf_mark_down_sites": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr, c_int, c_ptr]),
    "surf_sites_from_keys": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr]),
    "surf_table_from_coords": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr]),
    "surf_row_linear8": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    "surf_conv3x3": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "surf_conv3x3_p": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr, c_int, c_ptr]),
    "surf_deconv3x3_s2": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "surf_deconv3x3_s2_p": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr, c_int, c_ptr]),
    "surf_inorm_workspace_doubles": (c_i64, [c_int, c_int, c_int, c_int]),
    "surf_inorm_relu": (c_int, [c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_inorm_relu_out": (c_int, [c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_occupied_any": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_int, c_ptr, c_ptr]),
    "surf_masked_l1_workspace_bytes": (c_i64, []),
    "surf_masked_l1": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_masked_l1_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_weight_norm_backward": (c_int, [c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "surf_matching_depth": (c_int, [c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr,
                                    c_ptr, c_int, c_ptr, c_float, c_float, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
}

_lib = None


def build(verbose=False):
    """Compile the HIP sources for gfx950 (cross-compiles without a GPU)."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    res = subprocess.run(["bash", script], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise RuntimeError("building libsurf_hip.so failed")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with surf_amd/csrc/build.sh (or __graft_entry__.build()). "
                "There is no CPU fallback for the SuRF hot path.")
        cand = ctypes.CDLL(LIB_PATH)
        rebuild = "rebuild with surf_amd/csrc/build.sh (or __graft_entry__.build())"
        try:                                  # the version first: a stale library lacks newer symbols and would otherwise
            cand.surf_abi_version.restype = c_int         # die with a bare AttributeError in the binding loop below
            cand.surf_abi_version.argtypes = []
            got = cand.surf_abi_version()
        except AttributeError:
            raise RuntimeError(f"{LIB_PATH} exports no surf_abi_version: not a surf_hip library; {rebuild}") from None
        if got != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} reports ABI version {got}, this binding is written for {ABI_VERSION}: {rebuild}")
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(cand, name)
            except AttributeError:
                raise RuntimeError(f"{LIB_PATH} (ABI {got}) lacks {name}, which include/surf_hip.h declares: {rebuild}") from None
            fn.restype = res
            fn.argtypes = args
        _lib = cand                           # cached only when fully bound
    return _lib


class SurfHipError(RuntimeError):
    pass


def check(code, what):
    if code != 0:
        kind = {-1: "invalid argument", -2: "exceeds a SURF_MAX_* limit"}.get(code, f"hipError {code}")
        raise SurfHipError(f"{what}: {kind}")
