"""ctypes binding of libdnmf_hip.so (the C ABI declared in include/dnmf_hip.h).

There is no CPU fallback: if the library is missing or a call fails, the error is raised here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DNMF_LIB selects another build of the library (kernel-variant timing, ablations); the product default is in-tree
LIB_PATH = os.environ.get("DNMF_LIB") or os.path.join(_HERE, "libdnmf_hip.so")
ABI_VERSION = 6

_vp, _i, _l, _sz, _d = C.c_void_p, C.c_int, C.c_long, C.c_size_t, C.c_double

# name -> (restype, argtypes); mirrors include/dnmf_hip.h line by line
SIGNATURES = {
    "dnmf_version": (_i, []),
    "dnmf_last_error": (C.c_char_p, []),
    "dnmf_build_stamp": (C.c_char_p, []),
    "dnmf_padded_k": (_i, [_i]),
    "dnmf_pack_footprints": (_i, [_vp, _l, _i, _vp, _i, _vp]),
    "dnmf_warp_gather": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "dnmf_halo_voxels": (_l, [_i, _i, _i]),
    "dnmf_halo_row": (_i, [_i, _i]),
    "dnmf_recon_image": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _l, _vp, _i, _vp, _l, _vp]),
    "dnmf_warp_recon_grad_workspace": (_sz, [_i, _i, _i, _i]),
    "dnmf_warp_recon_grad": (_i, [_vp, _l, _vp, _vp, _l, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp,
                                  _vp, _vp, _vp, _sz, _vp]),
    "dnmf_warp_gram_rhs_workspace": (_sz, [_l, _i, _i]),
    "dnmf_warp_gram_rhs": (_i, [_vp, _i, _i, _l, _i, _i, _i, _vp, _i, _vp, _i, _vp, _l, _vp, _vp, _vp, _vp, _sz,
                                _vp]),
    "dnmf_warp_gram_rhs_bf16": (_i, [_vp, _i, _i, _l, _i, _i, _i, _vp, _i, _vp, _i, _vp, _l, _vp, _vp, _vp, _vp, _sz,
                                     _vp]),
    "dnmf_sparse_k": (_i, [_i]),
    "dnmf_pack_footprints_sparse": (_i, [_vp, _l, _i, _vp, _vp, _i, _vp, _vp]),
    "dnmf_warp_gram_rhs_sparse_workspace": (_sz, [_l, _i, _i]),
    "dnmf_warp_gram_rhs_sparse": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _l, _vp, _vp, _vp, _vp,
                                       _sz, _vp, _vp]),
    "dnmf_warp_gram_rhs_sparse_lt_workspace": (_sz, [_l, _i, _i]),
    "dnmf_warp_gram_rhs_sparse_lt": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _l, _vp, _vp, _vp, _vp,
                                          _sz, _vp, _vp]),
    "dnmf_mu_temporal": (_i, [_vp, _vp, _vp, _l, _i, _i, _i, _vp]),
    "dnmf_mu_temporal_nbr": (_i, [_vp, _vp, _vp, _l, _i, _i, _i, _vp, _i, _vp]),
    "dnmf_mu_temporal_slots": (_i, [_vp, _i, _i, _vp, _vp, _l, _i, _i, _i, _vp, _i, _vp]),
    "dnmf_warp_gram_rhs_lists_chunks": (_i, [_i, _i, _i, _i]),
    "dnmf_mu_temporal_step": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _d, _vp, _vp, _vp]),
    "dnmf_spatial_accum": (_i, [_vp, _l, _vp, _vp, _l, _vp, _l, _vp, _i, _l, _i, _vp, _vp, _i, _vp]),
    "dnmf_mu_spatial": (_i, [_vp, _vp, _vp, _vp, _d, _l, _i, _vp]),
    "dnmf_image_iwarp_workspace": (_sz, [_i, _i, _i, _i]),
    "dnmf_image_iwarp": (_i, [_vp, _l, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _l, _vp, _sz, _i, _vp, _vp]),
    "dnmf_adam_epoch_workspace": (_sz, [_i]),
    "dnmf_adam_epoch": (_i, [_vp, _vp, _vp, _vp, _i, _l, _vp, _vp, _i, _d, _d, _d, _d, _i, _vp, _sz, _vp]),
    "dnmf_render_frames": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _d, _d, _vp, _l, _vp]),
    "dnmf_lists_axis_masks_bytes": (_sz, [_i, _i, _i, _i]),
    "dnmf_pack_footprints_lists": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dnmf_warp_gram_rhs_lists_workspace": (_sz, [_i, _i, _i, _i, _i, _i]),
    "dnmf_warp_gram_rhs_lists": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _l, _vp, _vp, _vp,
                                      _vp, _sz, _vp, _vp]),
    "dnmf_recon_image_lists": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _l, _vp, _i, _vp, _l, _vp]),
    "dnmf_recon_image_lists_ex": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _l, _vp, _i, _vp, _l, _i, _vp]),
    "dnmf_motion_grad_lists_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "dnmf_motion_grad_lists": (_i, [_vp, _vp, _i, _vp, _l, _vp, _l, _vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _i,
                                    _vp, _sz, _vp]),
    "dnmf_spatial_lists_tiles": (_l, [_i, _i, _i]),
    "dnmf_spatial_lists_setup": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "dnmf_spatial_accum_lists_workspace": (_sz, [_i, _i, _i, _l, _i]),
    "dnmf_spatial_accum_lists": (_i, [_vp, _l, _vp, _vp, _l, _vp, _i, _i, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _sz, _vp]),
    "dnmf_mu_spatial_lists": (_i, [_vp, _vp, _vp, _vp, _vp, _d, _i, _i, _i, _i, _vp, _vp]),
    "dnmf_register_patches_grid": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "dnmf_register_patches_workspace": (_sz, [_i, _i, _i, _vp, _vp, _i]),
    "dnmf_register_patches": (_i, [_vp, _l, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, C.c_float, _vp, _vp, _vp, _sz, _vp]),
    "dnmf_rigid_correct_workspace": (_sz, [_i, _i, _i, _i]),
    "dnmf_rigid_correct": (_i, [_vp, _l, _vp, _i, _vp, _i, _i, _i, _vp, _i, C.c_float, _i, _vp, _vp, _l, _vp, _vp, _vp, _sz, _vp]),
    "dnmf_apply_shifts_points": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "dnmf_comm_unique_id": (_i, [_vp]),
    "dnmf_comm_init": (_i, [_vp, _vp, _i, _i]),
    "dnmf_allreduce_sum_f32": (_i, [_vp, _vp, _sz, _vp]),
    "dnmf_comm_destroy": (_i, [_vp]),
}

_lib = None


class DnmfHipError(RuntimeError):
    pass


def load():
    """Load the shared library once and attach the prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DnmfHipError(
            f"{LIB_PATH} is missing: build it with `python -m dnmf_amd.build` (hipcc, gfx950). "
            "dnmf_amd has no CPU fallback.")
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process BEFORE this library
    # is opened so that libdnmf_hip.so binds to the same runtime that owns torch's device pointers and streams;
    # opened the other way round the process ends up with two runtimes and every launch fails with
    # hipErrorNoDevice.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = stale library
        fn.restype, fn.argtypes = res, args
    if lib.dnmf_version() != ABI_VERSION:
        raise DnmfHipError(f"{LIB_PATH}: ABI {lib.dnmf_version()} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().dnmf_last_error().decode(errors="replace")
        kind = "argument error" if rc < 0 else "hipError_t"
        raise DnmfHipError(f"{what} failed ({kind} {rc}): {msg}")
