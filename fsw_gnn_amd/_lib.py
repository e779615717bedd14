"""ctypes binding of libfsw_hip.so (include/fsw_hip.h).

Mirrors how the reference binds its native library: ctypes.CDLL on a .so that sits next to the Python
sources, raw tensor.data_ptr() device pointers, caller-owned buffers (reference fsw_embedding.py:94-99,
195-206, 2952-2977).  There is NO fallback: if the library is missing or a call fails this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FSW_HIP_LIBRARY: another build of the same library (kernel tuning experiments, tools/exp_variants.sh)
LIB_PATH = os.environ.get("FSW_HIP_LIBRARY") or os.path.join(_HERE, "libfsw_hip.so")

FSW_ABI_VERSION = 5
REG_MAX_DEG = 32
LDS_MAX_DEG = 2048
MID_SIZES = (40, 48, 64, 80, 96, 128, 160, 192, 256)   # FSW_MID_SIZES: padded register-path networks above REG_MAX_DEG
NUM_LDS_BINS = 3                                          # FSW_NUM_LDS_BINS: degrees <= 512, 1024, 2048
NUM_HUB_BINS = 4                                          # FSW_NUM_HUB_BINS: degrees <= 4096, 8192, 16384, 32768
HUB_MAX_DEG = 32768
BIN_MID0 = REG_MAX_DEG + 1      # first bin above the register path (include/fsw_hip.h: FSW_BIN_MID0)
NUM_BINS = REG_MAX_DEG + 1 + len(MID_SIZES) + NUM_LDS_BINS + NUM_HUB_BINS + 1
NUM_STATS = 8
STAT_FLAGS, STAT_MAX_DEGREE, STAT_NUM_ZERO, STAT_NUM_REG, STAT_NUM_LDS, STAT_NUM_GLOBAL, STAT_NNZ, STAT_USER = 0, 1, 2, 3, 4, 5, 6, 7
BIN_BLOCK_ROWS, MAX_ROW_CHUNKS = 2048, 256
FLAG_INDEX_RANGE, FLAG_W_NONFINITE, FLAG_W_NEGATIVE, FLAG_X_NONFINITE = 1, 2, 4, 8

c_i64, c_i32, c_f32, c_vp, c_sz = ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t


class EmbedArgs(ctypes.Structure):
    """struct fsw_embed_args of include/fsw_hip.h (field order and types must match)."""
    _fields_ = [
        ("rowptr", c_vp), ("col", c_vp), ("w", c_vp), ("perm", c_vp), ("bin_start", c_vp), ("num_rows", c_i64),
        ("Xp", c_vp), ("ldp", c_i64), ("freqs", c_vp), ("S", c_i32), ("tau", c_f32),
        ("unit_table", c_vp), ("ldt", c_i64),
        ("out", c_vp), ("ldo", c_i64), ("bias", c_vp), ("out_scale", c_f32), ("has_mass", c_i32), ("mass_fn", c_i32),
        ("mass_scale", c_f32),
        ("num_reg_rows", c_i64), ("num_lds_rows", c_i64), ("num_global_rows", c_i64), ("num_zero_rows", c_i64),
        ("max_degree", c_i64),
        ("scratch", c_vp), ("scratch_bytes", c_sz),
        ("efeat", c_vp), ("Ve", c_vp), ("ldve", c_i64), ("d_edge", c_i32), ("reserved", c_i32),
        ("bin_start_host", c_vp),
    ]


class GenericArgs(ctypes.Structure):
    """struct fsw_generic_args of include/fsw_hip.h (field order and types must match)."""
    _fields_ = [
        ("value_dtype", c_i32), ("S", c_i32), ("rowptr", c_vp), ("col", c_vp), ("w", c_vp), ("num_rows", c_i64),
        ("max_degree", c_i64), ("Xp", c_vp), ("ldp", c_i64), ("Ke", c_vp), ("ldke", c_i64), ("freqs", c_vp),
        ("tau", ctypes.c_double), ("out", c_vp), ("ldo", c_i64), ("bias", c_vp), ("out_scale", ctypes.c_double),
        ("has_mass", c_i32), ("mass_fn", c_i32), ("mass_scale", ctypes.c_double), ("g", c_vp), ("ldg", c_i64),
        ("gkey", c_vp), ("ldk", c_i64), ("gfreq", c_vp), ("gw", c_vp), ("scratch", c_vp), ("scratch_bytes", c_sz),
    ]


_SIGNATURES = {
    "fsw_abi_version": (ctypes.c_int, []),
    "fsw_arch": (ctypes.c_char_p, []),
    "fsw_last_error": (ctypes.c_char_p, []),
    "fsw_graph_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "fsw_graph_build_coalesced": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                 c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "fsw_graph_build": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "fsw_graph_build_two_level": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "fsw_project_f32": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64, c_vp, ctypes.c_int, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "fsw_gemm_tn_workspace_bytes": (c_sz, [ctypes.c_int, ctypes.c_int]),
    "fsw_gemm_tn_f32": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, ctypes.c_int, ctypes.c_int, c_vp, c_i64, c_f32, c_vp, c_sz, c_vp]),
    "fsw_unit_table_rows": (c_sz, [ctypes.c_int]),
    "fsw_unit_coeff_table": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_i64, c_vp]),
    "fsw_embed_scratch_bytes": (c_sz, [c_i64]),
    "fsw_embed_f32": (ctypes.c_int, [ctypes.POINTER(EmbedArgs), c_vp]),
    "fsw_conv_fused_lds_bytes": (c_sz, [ctypes.c_int, ctypes.c_int]),
    "fsw_packed_linear_floats": (c_sz, [ctypes.c_int, ctypes.c_int]),
    "fsw_pack_linear_f32": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_vp, ctypes.c_int, c_vp, c_i64, c_vp]),
    "fsw_project_linear_f32": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64, c_vp, ctypes.c_int, c_i64, c_vp, c_i64,
                                              c_vp, ctypes.c_int, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "fsw_conv_fused_f32": (ctypes.c_int, [ctypes.POINTER(EmbedArgs), c_vp, c_i64, c_vp, ctypes.c_int, c_vp, c_i64, ctypes.c_int,
                                          ctypes.c_int, c_f32, c_vp, c_i64, c_vp]),
    "fsw_add_bias_act_f32": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, ctypes.c_int, ctypes.c_int, c_f32, c_vp]),
    "fsw_unit_dcoeff_table": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_i64, c_vp]),
    "fsw_embed_backward_f32": (ctypes.c_int, [ctypes.POINTER(EmbedArgs), c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "fsw_embed_backward_keys_f32": (ctypes.c_int, [ctypes.POINTER(EmbedArgs), c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "fsw_graph_transpose": (ctypes.c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "fsw_segment_sum_rows_f32": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, ctypes.c_int, c_vp, c_i64, c_vp]),
    "fsw_embed_generic_scratch_bytes": (c_sz, [c_i64, c_i64]),
    "fsw_embed_generic": (ctypes.c_int, [ctypes.POINTER(GenericArgs), c_vp]),
    "fsw_project_f64": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64, c_vp, ctypes.c_int, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "fsw_segcumsum_workspace_bytes": (c_sz, [c_i64]),
    "fsw_segcumsum": (ctypes.c_int, [ctypes.c_int, c_vp, c_vp, c_vp, ctypes.c_int, c_i64, ctypes.c_int, c_vp, c_sz, c_vp]),
    # legacy ABI, exact reference signatures (reference fsw_embedding.py:2952-2977)
    "segcumsum_wrapper": (None, [c_i64, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, ctypes.c_bool, c_i64, c_i64, c_sz]),
    "add_block_sums_wrapper": (None, [c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64]),
    "get_max_threads_per_block": (ctypes.c_int, [ctypes.c_int]),
    "launch_segcumsum_kernel_float": (None, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, ctypes.c_bool, c_i64, c_i64, c_i64]),
    "launch_segcumsum_kernel_double": (None, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, ctypes.c_bool, c_i64, c_i64, c_i64]),
    "launch_add_block_sums_kernel_float": (None, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64]),
    "launch_add_block_sums_kernel_double": (None, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib():
    """The loaded library; raises RuntimeError when it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                "fsw_gnn_amd: native library %s not found. Build it with `python -c \"import __graft_entry__ as g; "
                "g.build()\"` or `make -C fsw_gnn_amd/csrc` (needs hipcc, --offload-arch=gfx950). There is no "
                "CPU or pure-PyTorch fallback." % LIB_PATH)
        try:
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise RuntimeError("fsw_gnn_amd: cannot load %s: %s" % (LIB_PATH, e))
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError here = a symbol of include/fsw_hip.h is missing
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.fsw_abi_version() != FSW_ABI_VERSION:
            raise RuntimeError("fsw_gnn_amd: ABI version mismatch between %s and the Python binding" % LIB_PATH)
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (status %d): %s" % (what, rc, lib().fsw_last_error().decode()))


def ptr(t):
    """Device pointer of a torch tensor (or NULL for None)."""
    return None if t is None else c_vp(t.data_ptr())
