"""ctypes binding of libcdml_hip.so (the C ABI declared in include/cdml.h).

There is deliberately NO fallback: if the HIP library is missing or a call
fails, the product raises.  PyTorch is used only for device memory and streams.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libcdml_hip.so"


class CdmlError(RuntimeError):
    """A C-ABI call returned a negative cdml_status."""

    def __init__(self, code, message):
        super().__init__(f"cdml status {code}: {message}")
        self.code = code


def lib_path():
    """Path of the built library; CDML_LIB_PATH overrides it (kernel A/B builds)."""
    return os.environ.get("CDML_LIB_PATH") or os.path.join(_HERE, "lib", _LIB_NAME)


def source_id(csrc=None, header=None):
    """sha256[:16] over the sources the library is built from -- csrc/*.hip, csrc/*.h (sorted by name) and
    include/cdml.h.  __graft_entry__.build() compiles it into the library (``cdml_build_id()``); ``load_library``
    compares the two, so a prebuilt .so that travelled with a tree it was not built from is refused instead of
    silently passing for it (file times say nothing after a copy).  None when the tree has no sources."""
    import glob
    import hashlib
    csrc = csrc or os.path.join(_HERE, "csrc")
    header = header or os.path.join(os.path.dirname(_HERE), "include", "cdml.h")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")))
    if not files or not os.path.exists(header):
        return None
    h = hashlib.sha256()
    for f in files + [header]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def embedded_id(path):
    """The build id compiled into the library file at ``path`` (read from the file, nothing is loaded), or None."""
    import re
    try:
        m = re.search(rb"CDML_BUILD_ID=([0-9a-f]{16})", open(path, "rb").read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_u64 = C.c_uint64
_f = C.c_float
_sz = C.c_size_t

# name -> (restype, argtypes); must list every symbol include/cdml.h declares
SIGNATURES = {
    "cdml_version": (_i, []),
    "cdml_build_id": (C.c_char_p, []),
    "cdml_last_error": (C.c_char_p, []),
    "cdml_fill_uniform_table": (_i, [_p, _i64, _i64, _i, _i64, _u64, _p]),
    "cdml_sample_uniform": (_i, [_p, _i64, _i64, _u64, _u64, _p, _i, _i64, _i64, _p, _p]),
    "cdml_sample_inbatch": (_i, [_p, _i64, _u64, _u64, _p, _i, _i64, _i64, _p, _p, _p]),
    "cdml_step_advance": (_i, [_p, _p]),
    "cdml_gather_rows": (_i, [_p, _i64, _i64, _i64, _p, _i, _i, _i, _p, _i64, _p, _p, _p]),
    "cdml_sample_gather": (_i, [_i, _p, _i64, _u64, _u64, _p, _i, _i64, _i64, _p, _i64, _i64, _i,
                                _p, _p, _p, _i64, _i, _i64, _i64, _p, _p]),
    "cdml_route_rows": (_i, [_p, _i, _i64, _i, _i, _p, _p, _p, _p]),
    "cdml_scatter_rows": (_i, [_p, _i64, _p, _i, _i, _p, _i64, _p]),
    "cdml_sample_gather_f16": (_i, [_i, _p, _i64, _u64, _u64, _p, _i, _i64, _i64, _p, _i64, _i64, _i,
                                    _p, _p, _p, _i64, _i, _i64, _i64, _p, _p]),
    "cdml_l2norm_fwd": (_i, [_p, _i64, _i, _i, _p, _i64, _p, _p]),
    "cdml_l2norm_bwd": (_i, [_p, _i64, _p, _i64, _i, _i, _f, _p, _i64, _p]),
    "cdml_fc_lrelu_fwd": (_i, [_p, _i64, _p, _i64, _p, _f, _i, _i, _i, _p, _i64, _p]),
    "cdml_fc_bwd_data": (_i, [_p, _i64, _p, _i64, _p, _i64, _f, _i, _i, _i, _p, _i64, _p]),
    "cdml_fc_bwd_weight_workspace": (_sz, [_i, _i, _i]),
    "cdml_fc_bwd_weight": (_i, [_p, _i64, _p, _i64, _i, _i, _i, _p, _i64, _p, _p, _sz, _p]),
    "cdml_fc_bwd_weight2_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "cdml_fc_bwd_weight2": (_i, [_p, _i64, _p, _i64, _i, _i, _p, _i64, _p, _p, _i64, _p, _i64, _i, _i, _p, _i64, _p,
                                 _i, _p, _sz, _p]),
    "cdml_triplet_hinge": (_i, [_p, _i64, _i, _i, _f, _p, _p, _p, _p, _p, _i64, _p]),
    "cdml_triplet_hinge_inbatch": (_i, [_p, _i64, _p, _p, _i, _i, _f, _p, _p, _p, _p, _p, _p,
                                        _i64, _p]),
    "cdml_vnet_tail_workspace": (_sz, [_i, _i]),
    "cdml_vnet_tail": (_i, [_i, _p, _i64, _p, _p, _i, _i, _f, _f, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _i64,
                            _p, _p, _p]),
    "cdml_vnet_tail_planes": (_i, [_i, _p, _i64, _p, _p, _i, _i, _f, _f, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _i64, _i64,
                                   _p, _p, _p]),
    "cdml_vnet_tail_h2": (_i, [_i, _p, _i64, _p, _p, _i, _i, _f, _f, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _f,
                               _p, _p, _p]),
    "cdml_semihard_select": (_i, [_p, _i64, _p, _i64, _p, _i, _i, _p, _p, _p]),
    "cdml_semihard_mine_x3_workspace": (_sz, [_i]),
    "cdml_semihard_mine_x3": (_i, [_p, _i64, _p, _i, _i, _p, _i64, _i64, _p, _p, _p, _sz, _p, _p]),
    "cdml_semihard_mine_x3_z": (_i, [_p, _i64, _p, _i64, _p, _i, _i, _p, _i64, _i64, _p, _p, _p, _sz, _p, _p]),
    "cdml_semihard_mine_h2": (_i, [_p, _i64, _p, _i64, _p, _i, _i, _p, _i64, _i64, _f, _p, _p, _p, _sz, _p, _p]),
    "cdml_triplet_hinge_indexed": (_i, [_p, _i64, _p, _i, _i, _f, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "cdml_x3_slab_steps": (_i, [_i]),
    "cdml_triplet_hinge_indexed_tail": (_i, [_p, _i64, _p, _i, _i, _f, _p, _p, _p, _p, _p, _p, _i64, _p, _i64, _f, _p, _i64,
                                             _p, _i64, _i64, _p]),
    "cdml_pair_dist": (_i, [_p, _i64, _i, _p, _i, _i, _p, _p, _p, _p]),
    "cdml_cowatch_workspace": (_sz, [_i64]),
    "cdml_cowatch_graph": (_i, [_p, _i64, _p, _p, _p, _p, _p, _sz, _p]),
    "cdml_cowatch_select": (_i, [_p, _i64, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "cdml_knn_list_capacity": (_i, []),
    "cdml_row_sqnorm": (_i, [_p, _i64, _i, _i, _p, _p]),
    "cdml_knn_merge": (_i, [_p, _i64, _i, _i, _i, _i, _p, _p, _i, _p, _p, _i, _p]),
    "cdml_knn_filter_x3": (_i, [_p, _i64, _i64, _p, _i64, _i64, _i, _i, _i, _p, _p, _p, _i, _i, _p, _p, _i, _p]),
    "cdml_knn_filter_h2": (_i, [_p, _i64, _i64, _p, _i64, _i64, _i, _i, _i, _f, _p, _p, _p, _i, _i, _p, _p, _i, _p]),
    "cdml_knn_merge_list": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "cdml_gemm_bf16_workspace": (_sz, [_i, _i, _i]),
    "cdml_gemm_bf16_epilogue_supported": (_i, [_i, _i, _i, _i, _i64, _i64, _i64, _i64]),
    "cdml_gemm_bf16_nt": (_i, [_i, _p, _i64, _p, _i64, _i, _i, _i, _p, _i64, _p, _p, _i64, _f, _p, _sz, _p]),
    "cdml_gemm_bf16_tn_supported": (_i, [_i, _i, _i, _i64, _i64]),
    "cdml_gemm_bf16_tn_workspace": (_sz, [_i, _i, _i]),
    "cdml_gemm_bf16_tn2_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "cdml_gemm_bf16_tn2": (_i, [_p, _i64, _p, _i64, _i, _i, _p, _i64, _p, _p, _i64, _p, _i64, _i, _i, _p, _i64, _p, _i,
                                _p, _sz, _p]),
    "cdml_gemm_bf16_tn": (_i, [_p, _i64, _p, _i64, _i, _i, _i, _p, _i64, _p, _p, _sz, _p]),
    "cdml_sample_gather_x3": (_i, [_i, _p, _i64, _u64, _u64, _p, _i, _i64, _i64, _p, _i64, _i64, _i,
                                   _p, _p, _p, _i64, _i, _i64, _i64, _p, _p]),
    "cdml_sample_gather_h2": (_i, [_i, _p, _i64, _u64, _u64, _p, _i, _i64, _i64, _p, _i64, _i64, _i,
                                   _p, _p, _p, _i64, _i, _i64, _i64, _p, _p]),
    "cdml_gather_rows_x3": (_i, [_p, _i64, _i64, _p, _i, _i, _i, _p, _i64, _p, _p]),
    "cdml_sample_gather_x3k": (_i, [_i, _p, _i64, _u64, _u64, _p, _i, _i64, _i64, _p, _i64, _i64, _i,
                                    _p, _p, _p, _i64, _i, _i64, _i64, _p, _p, _i64, _p]),
    "cdml_split_f32_bf16x3": (_i, [_p, _i64, _i, _i, _p, _i64, _i64, _i, _p]),
    "cdml_gemm_bf16x3_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "cdml_gemm_bf16x3_nt": (_i, [_i, _p, _i64, _i64, _p, _i64, _i64, _i, _i, _i, _i, _p, _i64, _i64, _p, _p, _i64, _f,
                                 _p, _p, _sz, _p]),
    "cdml_gemm_bf16x3_tn": (_i, [_p, _i64, _i64, _p, _i64, _i64, _i, _i, _i, _i, _p, _i64, _p, _f, _p, _p, _sz, _p]),
    "cdml_split_f32_f16x2": (_i, [_p, _i64, _i, _i, _p, _i64, _i64, _i, _f, _p]),
    "cdml_gemm_f16x2_workspace": (_sz, [_i, _i, _i, _i]),
    "cdml_gemm_f16x2_nt": (_i, [_i, _p, _i64, _i64, _p, _i64, _i64, _i, _i, _i, _p, _i64, _i64, _p, _p, _i64, _f, _f, _f,
                                _p, _sz, _p]),
    "cdml_gemm_f16x2_tn": (_i, [_p, _i64, _i64, _p, _i64, _i64, _i, _i, _i, _p, _i64, _f, _p, _f, _p, _sz, _p]),
    "cdml_interleave8_bf16x3": (_i, [_p, _i64, _i64, _i, _i, _p, _p]),
    "cdml_gemm_bf16x3_tnk": (_i, [_p, _i, _i, _p, _i, _i, _i, _i, _i, _p, _i64, _p, _p, _sz, _p]),
    "cdml_transpose_to_bf16": (_i, [_i, _p, _i64, _i, _i, _p, _i64, _p]),
    "cdml_cast_f32_bf16": (_i, [_p, _i64, _i, _i, _p, _i64, _p]),
    "cdml_colsum_workspace_floats": (_sz, [_i, _i]),
    "cdml_colsum": (_i, [_i, _p, _i64, _i, _i, _p, _p, _p]),
    "cdml_fill_uniform_table_f16": (_i, [_p, _i64, _i64, _i, _i64, _u64, _p]),
    "cdml_gather_rows_f16": (_i, [_p, _i64, _i64, _i64, _p, _i, _i, _p, _i64, _p, _p]),
    "cdml_ew_combine": (_i, [_i, _p, _i64, _p, _i64, _i, _i, _p, _i64, _p]),
    "cdml_ew_fusion_bwd": (_i, [_i, _p, _i64, _p, _i64, _p, _i64, _i, _i, _f, _p, _i64, _p, _i64, _p]),
    "cdml_lrelu_bwd": (_i, [_p, _i64, _p, _i64, _i, _i, _f, _p, _i64, _p]),
    "cdml_adam_step": (_i, [_p, _p, _p, _p, _i64, _f, _p, _f, _f, _f, _i64, _p, _i, _p, _p]),
    "cdml_adam_matrix_bf16": (_i, [_p, _p, _p, _p, _i, _i, _f, _p, _f, _f, _f, _i64, _p, _p, _i64, _p, _i64,
                                   _p, _p, _p, _p, _i, _i, _p, _p]),
    "cdml_adam_matrix_planes": (_i, [_p, _p, _p, _p, _i, _i, _f, _p, _f, _f, _f, _i64, _p, _p, _i64, _i64, _p, _i64, _i64,
                                     _p, _p, _p, _p, _i, _i, _p, _p]),
    "cdml_adam_matrix_h2": (_i, [_p, _p, _p, _p, _i, _i, _f, _p, _f, _f, _f, _i64, _p, _p, _i64, _i64, _p, _i64, _i64, _f,
                                 _p, _p, _p, _p, _i, _i, _p, _p]),
    "cdml_table_adam_rows": (_i, [_p, _i64, _i64, _i64, _i, _p, _i, _p, _i64, _p, _p, _p, _p, _f, _f, _p, _f, _f, _f,
                                  _i64, _p, _p]),
    "cdml_grad_prepare": (_i, [_p, _p, _i64, _f, _f, _p, _p, _p]),
    "cdml_momentum_step": (_i, [_p, _p, _p, _i64, _f, _p, _f, _i, _p]),
    "cdml_lars_scratch_floats": (_sz, []),
    "cdml_lars_step": (_i, [_p, _p, _p, _i64, _f, _p, _f, _f, _f, _f, _p, _p]),
    "cdml_lars_multi_scratch_floats": (_sz, []),
    "cdml_lars_multi": (_i, [_p, _p, _p, _p, _p, _i, _f, _p, _f, _f, _f, _f, _p, _p, _p, _p, _p]),
    "cdml_lars_multi_norms": (_i, [_p, _p, _p, _p, _i, _p, _p]),
    "cdml_lars_matrix": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p, _f, _f, _f, _f, _p, _p, _p, _i64, _i64, _p,
                              _i64, _i64, _i, _p, _p, _p]),
    "cdml_momentum_matrix": (_i, [_p, _p, _p, _i, _i, _f, _p, _f, _i, _p, _i64, _i64, _p, _i64, _i64, _i, _p, _p, _p, _i,
                                  _p, _p, _p]),
    "cdml_lars_matrix_h2": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p, _f, _f, _f, _f, _p, _p, _p, _i64, _i64, _p,
                                 _i64, _i64, _f, _p, _p, _p]),
    "cdml_momentum_matrix_h2": (_i, [_p, _p, _p, _i, _i, _f, _p, _f, _i, _p, _i64, _i64, _p, _i64, _i64, _f, _p, _p, _p, _i,
                                     _p, _p, _p]),
}

_lib = None


def load_library():
    """Load libcdml_hip.so (built by __graft_entry__.build()); raise if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise CdmlError(-2, f"{path} not found: build it with "
                            f"`python -c 'import __graft_entry__ as g; g.build()'` "
                            f"(there is no CPU fallback)")
    # torch first: the process must end up with ONE HIP runtime, the one torch ships and allocates
    # device memory with.  Loaded before torch, this library pulls in the system libamdhip64 and
    # its launches then fail with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    # the library must be the one THIS tree builds (a variant library named by CDML_LIB_PATH is built from a scratch
    # copy of the sources on purpose and is exempt; CDML_ALLOW_STALE_LIB=1 for a deliberate mismatch)
    want = source_id()
    if want is not None and not os.environ.get("CDML_LIB_PATH") and os.environ.get("CDML_ALLOW_STALE_LIB") != "1":
        have = lib.cdml_build_id().decode()
        if have != "CDML_BUILD_ID=" + want:
            raise CdmlError(-2, f"{path} was built from other sources ({have}; this tree is {want}): rebuild it with "
                                f"`python -c 'import __graft_entry__ as g; g.build()'`")
    _lib = lib
    return lib


def call(name, *args):
    """Invoke a status-returning entry point; raise CdmlError on failure."""
    lib = load_library()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise CdmlError(rc, lib.cdml_last_error().decode("utf-8", "replace"))
    return rc
