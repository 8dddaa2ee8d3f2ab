"""ctypes binding of libgnndelete_hip.so (include/gnndelete_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# GNNDELETE_HIP_LIB points at another build of the same ABI (A/B measurements); there is still no fallback
LIB_PATH = os.environ.get('GNNDELETE_HIP_LIB') or os.path.join(_HERE, 'lib', 'libgnndelete_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'gnndelete_hip.h')

_i32, _i64, _f32, _f64, _p = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol the header declares
ABI_VERSION = 9          # GD_ABI_VERSION of include/gnndelete_hip.h this binding was written against

PROTOTYPES = {
    'gd_abi_version': (ctypes.c_int, []),
    'gd_last_error_string': (ctypes.c_char_p, []),
    'gd_build_source_hash': (ctypes.c_char_p, []),
    'gd_matrix_split': (ctypes.c_int, []),
    'gd_set_matrix_split': (ctypes.c_int, [ctypes.c_int]),
    'gd_csr_from_coo_workspace': (_i64, [_i32, _i64]),
    'gd_csr_from_coo': (ctypes.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _p, _i64, _p]),
    'gd_gcn_norm_f32': (ctypes.c_int, [_p, _p, _i32, _p, _p]),
    'gd_spmm_csr_f32': (ctypes.c_int, [_p, _p, _p, _p, _i64, _p, _i64, _p, _f32, _i32, _i32, _p]),
    'gd_spmm_csr_balanced_f32': (ctypes.c_int, [_p, _i32, _p, _i32, _p, _p, _p, _i64, _p, _i64, _p, _f32, _p, _p, _i32,
                                                _i32, _i32, _p, _p]),
    'gd_spmm_csr_onepass_f32': (ctypes.c_int, [_p, _i32, _p, _p, _p, _i64, _p, _i64, _p, _f32, _p, _i32, _i32, _i32, _p, _p]),
    'gd_rows_gemm_wgrad_reduce_f32': (ctypes.c_int, [_p, _i32, _i32, _i32, _p, _i32, _p, _p, _p, _p, _f64, _f64, _f64, _f64, _p]),
    'gd_step_tail_f32': (ctypes.c_int, [_p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _f64, _f64, _f64, _f64,
                                        _p, _i32, _p, _i32, _p, _i32, _p, _p, _p, _p]),
    'gd_step_tail_parts_f32': (ctypes.c_int, [_p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _f64, _f64, _f64, _f64,
                                        _p, _i32, _p, _i32, _p, _i32, _p, _p, _p, _p]),
    'gd_rgcn_conv_f32': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _p, _i32, _i32, _p, _i64, _i32, _i32, _p]),
    'gd_rowpair_loss_f32': (ctypes.c_int, [_i32, _p, _i64, _p, _p, _i64, _p, _i32, _i32, _p, _p, _i64, _p]),
    'gd_random_walk': (ctypes.c_int, [_p, _p, _i32, _p, _i32, _i32, ctypes.c_uint64, _p, _p]),
    'gd_rgcn_tile_kl': (ctypes.c_int32, [_i32, _i32, _i32, _i32]),
    'gd_rgcn_pack_weight_f32': (ctypes.c_int, [_p, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    'gd_rgcn_wave_covers': (ctypes.c_int32, [_i32, _i32, _i32]),
    'gd_rgcn_wave_conv_f32': (ctypes.c_int, [_p, _i32, _i32, _p, _i32, _p, _p, _p, _p, _i64, _i32, _p, _i32, _p, _i64, _i32, _i32, _i32, _p]),
    'gd_rgcn_tile_conv_f32': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _i32, _p, _i64, _i32, _p, _i32, _i32, _p, _i64, _i32,
                                             _i32, _p, _p, _i32, _p, _p]),
    'gd_rgcn_mean_f32': (ctypes.c_int, [_p, _p, _p, _i64, _p, _i64, _i32, _i32, _i32, _p]),
    'gd_gat_aggregate_f32': (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _f32, _i32, _i32, _p]),
    'gd_gat_aggregate_bwd_f32': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _i64, _p, _i64,
                                                _p, _p, _p, _f32, _i32, _i32, _p]),
    'gd_gat_balanced_scratch': (_i64, [_i32, _i32]),
    'gd_gat_aggregate_balanced_f32': (ctypes.c_int, [_p, _i32, _p, _i32, _i32, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _p,
                                                     _p, _f32, _i32, _i32, _i32, _p]),
    'gd_gat_edge_grads_balanced_f32': (ctypes.c_int, [_p, _i32, _p, _i32, _p, _p, _p, _p, _p, _p, _i64, _p, _i64, _p,
                                                      _p, _p, _p, _f32, _i32, _i32, _p, _p]),
    'gd_row_dots_f32': (ctypes.c_int, [_p, _i64, _i32, _i32, _p, _p, _p, _p, _p]),
    'gd_gat_transpose_edges_f32': (ctypes.c_int, [_p, _p, _p, _i32, _p, _p, _p]),
    'gd_rank1_add2_f32': (ctypes.c_int, [_p, _i64, _i32, _i32, _p, _p, _p, _p, _p]),
    'gd_segment_sum_f32': (ctypes.c_int, [_p, _p, _p, _i32, _p, _p]),
    'gd_rows_gemm_f32': (ctypes.c_int, [_p, _i64, _p, _i32, _p, _i32, _i32, _i32, _p, _i32, _p, _i64, _p, _p]),
    'gd_rows_gemm_wgrad_workspace': (_i64, [_i32, _i32, _i32]),
    'gd_rows_gemm_ws_covers': (ctypes.c_int, [_i32, _i32, _i32]),
    'gd_rows_gemm_accumulate_f32': (ctypes.c_int, [_p, _i64, _p, _i32, _p, _i32, _i32, _i32, _p, _i64, _p]),
    'gd_rows_gemm_select_f32': (ctypes.c_int, [_p, _p, _p, _i64, _p, _i32, _p, _i32, _i32, _i32, _p, _i32, _p, _i64, _p]),
    'gd_rows_gemm_dots_f32': (ctypes.c_int, [_p, _p, _p, _i64, _p, _i32, _i32, _i32, _p, _i32, _p, _i64, _p, _i32, _p, _p, _p, _p,
                                             _p]),
    'gd_rows_gemm_signs_f32': (ctypes.c_int, [_p, _i64, _p, _i32, _p, _i32, _i32, _i32, _p, _i32, _p, _i64, _p, _p, _p]),
    'gd_rows_gemm_gated_f32': (ctypes.c_int, [_p, _i64, _p, _i32, _p, _i32, _i32, _i32, _p, _p, _i64, _p]),
    'gd_gate_rows_f32': (ctypes.c_int, [_p, _i64, _p, _i32, _p, _i32, _p, _i64, _p]),
    'gd_rows_gemm_gated_rank1_f32': (ctypes.c_int, [_p, _i64, _p, _i32, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _i64, _p]),
    'gd_rows_gemm_wgrad_f32': (ctypes.c_int, [_p, _i64, _p, _p, _i64, _p, _p, _p, _i32, _i32, _i32, _p, _i32, _p, _p]),
    'gd_rowpair_mse_workspace': (_i64, [_i32]),
    'gd_rowpair_mse_f32': (ctypes.c_int, [_p, _i64, _p, _i64, _i32, _p, _p, _i32, _p, _p, _p, _p, _i64, _i32,
                                          _p, _p, _p]),
    'gd_rowtarget_mse_workspace': (_i64, [_i32]),
    'gd_rowtarget_mse_f32': (ctypes.c_int, [_p, _i64, _p, _i32, _p, _p, _p, _p, _i32, _p, _i64, _p, _p, _p]),
    'gd_rowtarget_mse_pair_covers': (ctypes.c_int32, [_i32, _i32]),
    'gd_rowtarget_mse_pair_f32': (ctypes.c_int, [_p, _i64, _p, _i32, _p, _p, _p, _p, _i32, _p, _i64, _p, _p, _i64, _p, _i32, _p, _p, _p, _p, _i32, _p, _i64, _p, _p]),
    'gd_gemm_f32_workspace': (_i64, [_i32, _i32, _i32]),
    'gd_gemm_f32': (ctypes.c_int, [_p, _i64, _p, _i32, _p, _i32, _i32, _p, _p, _i64, _p, _p]),
    'gd_edge_dot_f32': (ctypes.c_int, [_p, _i64, _i32, _p, _p, _p, _i64, _p, _i64, _p, _p]),
    'gd_edge_dot_bwd_f32': (ctypes.c_int, [_p, _i64, _i32, _p, _p, _p, _i64, _p, _p, _i64, _p, _i64, _p]),
    'gd_rows_gemm_wgrad_adam_f32': (ctypes.c_int, [_p, _i64, _p, _p, _i64, _p, _p, _p, _i32, _i32, _i32, _p, _i32, _p,
                                                   _p, _p, _p, _p, _f64, _f64, _f64, _f64, _p]),
    'gd_rows_gemm_wgrad_blocks': (_i32, [_i32]),
    'gd_rows_gemm_wgrad_loss_f32': (ctypes.c_int, [_p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _i32,
                                                   _p, _p, _p, _p, _p, _p, _f64, _f64, _f64, _f64, _p]),
    'gd_del_loss_bwd_blocks': (_i32, [_i32]),
    'gd_del_loss_bwd_f32': (ctypes.c_int, [_p, _i64, _p, _i32, _p, _i32, _p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p]),
    'gd_del_loss_bwd_wgrad_f32': (ctypes.c_int, [_p, _i64, _p, _i32, _p, _i32, _p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _p]),
    'gd_del_loss_bwd_wgrad_parts': (ctypes.c_int32, [_i32, _i32]),
    'gd_del_loss_bwd_wgrad_parts_f32': (ctypes.c_int, [_p, _i64, _p, _i32, _p, _i32, _p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _i32, _p]),
    'gd_del1_loss_wgrad_covers': (ctypes.c_int32, [_i32, _i32]),
    'gd_del1_loss_wgrad_parts': (ctypes.c_int32, [_i32]),
    'gd_del1_chain_loss_wgrad_f32': (ctypes.c_int, [_p, _i64, _p, _i32, _p, _i32, _p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _i32, _p]),
    'gd_del1_loss_wgrad_f32': (ctypes.c_int, [_p, _i64, _p, _i32, _p, _i32, _p, _i64, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _i32, _p]),
    'gd_rowtarget_mse_blocks': (_i32, [_i32]),
    'gd_loss_finalize_f32': (ctypes.c_int, [_p, _i32, _p, _i32, _p, _p, _i32, _p, _p, _p]),
    'gd_pairs_sigmoid_mse_workspace': (_i64, [_i32, _i32]),
    'gd_pairs_sigmoid_mse_f32': (ctypes.c_int, [_p, _i64, _p, _i32, _i32, _p, _i64, _f32, _p, _p, _p, _p]),
    'gd_adam_at_f32': (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _f64, _f64, _f64, _f64, _p]),
    'gd_adam_f32': (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _f64, _f64, _f64, _f64, _p]),
    'gd_segment_softmax_f32': (ctypes.c_int, [_p, _p, _i32, _p, _p]),
    'gd_segment_softmax_bwd_f32': (ctypes.c_int, [_p, _p, _p, _i32, _p, _p]),
    'gd_rowpair_dot_f32': (ctypes.c_int, [_p, _i64, _p, _p, _i64, _p, _i64, _i32, _p, _p]),
    'gd_spmm_csr_onepass_aux_f32': (ctypes.c_int, [_p, _i32, _p, _p, _p, _p, _i64, _p, _i64, _p, _i32, _i32, _i32, _p, _p]),
    'gd_typed_wgrad_f32': (ctypes.c_int, [_p, _i32, _p, _p, _p, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _p]),
    'gd_typed_edge_dot_f32': (ctypes.c_int, [_p, _p, _p, _i64, _p, _i64, _p, _i64, _p, _i32, _i32, _i32, _p, _p]),
    'gd_comm_unique_id': (ctypes.c_int, [_p]),
    'gd_comm_init': (ctypes.c_int, [_p, _i32, _i32, ctypes.POINTER(ctypes.c_void_p)]),
    'gd_comm_destroy': (ctypes.c_int, [_p]),
    'gd_allreduce_f32': (ctypes.c_int, [_p, _i64, _p, _p]),
    'gd_exchange_rows_f32': (ctypes.c_int, [_p, _p, _p, _p, _i32, _i32, _p, _p]),
}

_lib = None


class GnnDeleteHipError(RuntimeError):
    pass


def declared_symbols():
    """Every function name include/gnndelete_hip.h declares."""
    with open(HEADER_PATH) as f:
        text = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    return sorted(set(re.findall(r'\b(gd_[a-z0-9_]+)\s*\(', text)))


def _sources_hash():
    from .source_hash import source_files, source_hash
    return source_hash() if all(os.path.exists(f) for f in source_files()) and os.path.isdir(os.path.join(_HERE, 'csrc')) else None


def build_stamp():
    """(stamp of the loaded library, hash of the sources next to it or None)."""
    return lib().gd_build_source_hash().decode(), _sources_hash()


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GnnDeleteHipError(
                f'{LIB_PATH} is missing: the HIP extension is not built. Run '
                '`python -c "import __graft_entry__ as g; g.build()"` (or `make -C gnndelete_amd/csrc`). '
                'There is no CPU fallback.')
        # torch first: its bundled HIP runtime must be the one in the process before this library (linked
        # against libamdhip64) is mapped, or the two end up on different runtime instances
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.gd_abi_version() != ABI_VERSION:
            raise GnnDeleteHipError(f'libgnndelete_hip.so ABI version {handle.gd_abi_version()}, this package binds version {ABI_VERSION}: rebuild (make -C gnndelete_amd/csrc)')
        # the library must have been built from the sources next to it: a stale .so next to newer sources
        # would run kernels nobody is looking at.  GNNDELETE_HIP_LIB (an A/B build of the same ABI) and GD_ALLOW_STALE_LIB=1
        # switch the check off; an installed copy without its sources has nothing to compare with.
        built, here = handle.gd_build_source_hash().decode(), _sources_hash()
        if here is not None and built != here and not os.environ.get('GNNDELETE_HIP_LIB') and os.environ.get('GD_ALLOW_STALE_LIB') != '1':
            raise GnnDeleteHipError(f'{LIB_PATH} was built from other sources (stamp {built}, sources here {here}): run '
                                    '`make -C gnndelete_amd/csrc` (or set GD_ALLOW_STALE_LIB=1 to load it anyway)')
        _lib = handle
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().gd_last_error_string().decode(errors='replace')
        raise GnnDeleteHipError(f'{what} failed with code {rc}: {msg}')


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr(device=None):
    import torch
    return torch.cuda.current_stream(device).cuda_stream
