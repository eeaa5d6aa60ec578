"""ctypes binding of libctgcn_hip.so (C ABI: include/ctgcn_hip.h).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
Build it with `python -m ctgcn_amd.build` (hipcc --offload-arch=gfx950) or __graft_entry__.build().
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CTGCN_HIP_LIB") or os.path.join(_HERE, "csrc", "libctgcn_hip.so")   # env override: kernel A/B builds

F_SELF_LOOP, F_RELU, F_NESTED = 1, 2, 4
ACT_NONE, ACT_SELU = 0, 1
OP_KCORE = 1
OP_INGEST = 2
MAX_SLOTS = 255
ABI_VERSION = 28

_c = ctypes
_vp, _i64, _i32, _u32, _int, _sz = _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_uint32, _c.c_int, _c.c_size_t

class AggSplitGroup(_c.Structure):
    """ctgcn_agg_split_group_t (include/ctgcn_hip.h)"""
    _fields_ = [("row_ptr", _vp), ("col_idx", _vp), ("val", _vp), ("slot", _vp), ("X", _vp), ("ldx", _i64), ("K", _i32), ("flags", _u32),
                ("row_order", _vp), ("tile_mask", _vp), ("workspace", _vp), ("workspace_bytes", _sz),
                ("planes1", _vp), ("planes2", _vp), ("scales", _vp), ("tile_base", _vp)]


class GruLayerGroup(_c.Structure):
    """ctgcn_gru_layer_group_t (include/ctgcn_hip.h)"""
    _fields_ = [("planes", _vp), ("w_ih", _vp), ("w_hh", _vp), ("bias_gi", _vp), ("b_hn", _vp), ("ln_weight", _vp), ("ln_bias", _vp),
                ("ln_eps", _c.c_float), ("steps", _i32), ("out", _vp), ("ld_out", _i64), ("row_order", _vp), ("tile_mask", _vp), ("work", _i64)]


class GruSeqGroup(_c.Structure):
    """ctgcn_gru_seq_group_t (include/ctgcn_hip.h)"""
    _fields_ = [("gi", _vp), ("w_hh", _vp), ("b_hn", _vp), ("ln_weight", _vp), ("ln_bias", _vp), ("ln_eps", _c.c_float), ("steps", _i32),
                ("out", _vp), ("ld_out", _i64), ("row_order", _vp), ("tile_mask", _vp), ("tile_base", _vp), ("work", _i64)]


# name -> (restype, argtypes); must list every symbol include/ctgcn_hip.h declares
SIGNATURES = {
    "ctgcn_abi_version": (_int, []),
    "ctgcn_last_error": (_c.c_char_p, []),
    "ctgcn_device_info": (_int, [_c.c_char_p, _sz, _c.POINTER(_int)]),
    "ctgcn_transpose_bias_f32": (_int, [_i64, _i32, _vp, _i64, _vp, _vp, _i64, _vp]),
    "ctgcn_spmm_csr_f32": (_int, [_i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _int, _vp]),
    "ctgcn_hub_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "ctgcn_hub_split_entries": (_i32, []),
    "ctgcn_core_aggregate_f32": (_int, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _u32, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "ctgcn_core_aggregate_bwd_prep_f32": (_int, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _u32, _vp]),
    "ctgcn_core_aggregate_bwd_f32": (_int, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _u32, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "ctgcn_edges_to_csr": (_int, [_i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _c.POINTER(_i64), _vp, _sz, _vp]),
    "ctgcn_kcore_i32": (_int, [_i64, _vp, _vp, _vp, _vp, _sz, _i32, _c.POINTER(_i32), _vp]),
    "ctgcn_edge_levels_i32": (_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "ctgcn_slot_reorder": (_int, [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "ctgcn_gru_seq_f32": (_int, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _c.c_float, _int, _vp, _i64, _vp, _int, _int, _vp, _vp, _vp, _vp]),
    "ctgcn_layernorm_bwd_f32": (_int, [_i64, _i32, _i32, _vp, _vp, _i64, _vp, _c.c_float, _vp, _vp, _i32, _vp, _vp]),
    "ctgcn_group_table_bytes": (_sz, [_i32]),
    "ctgcn_table_uploads": (ctypes.c_uint64, [_int]),
    "ctgcn_core_aggregate_split_group_f32": (_int, [_i32, _i64, _i32, _vp, _vp, _sz, _vp, _vp]),
    "ctgcn_gru_layer_presplit_group_f32": (_int, [_i32, _i64, _i32, _vp, _vp, _sz, _vp, _vp]),
    "ctgcn_transpose_bias_group_f32": (_int, [_i32, _i64, _i32, _vp, _i64, _vp, _vp, _i64, _vp, _sz, _vp, _vp]),
    "ctgcn_gru_seq_group_f32": (_int, [_i32, _i64, _i32, _vp, _vp, _sz, _vp, _vp]),
    "ctgcn_linear_packed_chain_f32": (_int, [_i64, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _sz, _vp]),
    "ctgcn_linear_packed_group_f32": (_int, [_i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _sz, _vp, _vp]),
    "ctgcn_gru_bwd_blocks": (_i32, [_i64]),
    "ctgcn_gru_layer_presplit_save_f32": (_int, [_i64, _i32, _i32, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ctgcn_gru_bwd_rec_f32": (_int, [_i64, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "ctgcn_gru_bwd_in_f32": (_int, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp]),
    "ctgcn_lstm_seq_f32": (_int, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _c.c_float, _int, _vp, _vp, _vp]),
    "ctgcn_lstm_seq_bwd_f32": (_int, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "ctgcn_gru_layer_f32": (_int, [_i64, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_float, _int, _vp, _i64, _vp, _vp, _i64, _vp]),
    "ctgcn_linear_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "ctgcn_split_planes_bytes": (_sz, [_i64, _i32]),
    "ctgcn_split_rows_f32": (_int, [_i64, _i32, _vp, _i64, _vp, _sz, _vp]),
    "ctgcn_pack_weight_bytes": (_sz, [_i32, _i32]),
    "ctgcn_pack_weight_f32": (_int, [_i32, _i32, _vp, _i64, _vp, _sz, _vp]),
    "ctgcn_linear_packed_f32": (_int, [_i64, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _i64, _vp]),
    "ctgcn_linear_f32": (_int, [_i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i32, _vp, _i64, _vp, _sz, _vp]),
    "ctgcn_core_aggregate_split_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32, _i32]),
    "ctgcn_core_aggregate_split_f32": (_int, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _u32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _i32, _vp, _sz, _vp, _sz, _vp]),
    "ctgcn_gru_layer_presplit_f32": (_int, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_float, _vp, _i64, _vp, _vp, _vp]),
    "ctgcn_linear_presplit_f32": (_int, [_i64, _i32, _i32, _vp, _i64, _vp, _vp, _i64, _vp, _sz, _vp]),
    "ctgcn_gru_seq_bwd_f32": (_int, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _int, _vp]),
    "ctgcn_gru_input_proj_f32": (_int, [_i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _int, _i32, _vp]),
    "ctgcn_gru_input_grad_f32": (_int, [_i64, _i32, _i32, _vp, _vp, _vp, _i64, _vp]),
    "ctgcn_gru_weight_grad_f32": (_int, [_i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _int, _vp, _i32, _int, _vp]),
    "ctgcn_gru_row_granule": (_i64, []),
    "ctgcn_compute_units": (_i32, []),
    "ctgcn_row_cumsum_f32": (_int, [_i64, _vp, _vp, _vp, _vp]),
    "ctgcn_random_walk_pairs": (_int, [_i64, _vp, _vp, _vp, _i32, _i32, _i32, _c.c_uint64, _int, _vp, _vp, _vp, _vp]),
    "ctgcn_neg_sampling_indices": (_int, [_i64, _vp, _vp, _vp, _i32, _i64, _vp, _c.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ctgcn_write_embedding_tsv": (_int, [_c.c_char_p, _i64, _i32, _vp, _i64, _c.c_char_p, _vp, _c.c_char, _i32]),
    "ctgcn_workspace_bytes": (_sz, [_int, _i64, _i64, _i32, _i32]),
}

_lib = None


class CtgcnHipError(RuntimeError):
    pass


def load():
    """Load the library (once). Raises CtgcnHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CtgcnHipError(
            "libctgcn_hip.so not found at %s — the HIP extension is required (no CPU fallback). "
            "Build it with `python -m ctgcn_amd.build`." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.ctgcn_abi_version() != ABI_VERSION:
        raise CtgcnHipError("libctgcn_hip.so ABI %d != binding ABI %d; rebuild" % (lib.ctgcn_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ctgcn_last_error()
        raise CtgcnHipError("%s failed (code %d): %s" % (what, rc, msg.decode("utf-8", "replace") if msg else ""))


def ptr(t):
    """Device (or host) address of a torch tensor, None -> NULL."""
    return None if t is None else t.data_ptr()
