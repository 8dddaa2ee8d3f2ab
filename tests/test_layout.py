"""Repository contract checks (CPU): the C-ABI library exports every symbol the header declares,
ctypes prototypes cover them all, and the product never touches the oracle."""
import ast
import ctypes
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gnndelete_amd import _lib
    declared = _lib.declared_symbols()
    assert len(declared) >= 15
    assert set(declared) == set(_lib.PROTOTYPES), set(declared) ^ set(_lib.PROTOTYPES)
    handle = ctypes.CDLL(_lib.LIB_PATH)          # loads without a GPU
    for name in declared:
        assert hasattr(handle, name), name
    assert _lib.lib().gd_abi_version() == _lib.ABI_VERSION == 9
    assert _lib.lib().gd_last_error_string() is not None
    # pure host helpers can be called without a GPU
    assert _lib.lib().gd_rows_gemm_wgrad_workspace(1000, 128, 128) >= 128 * 128
    assert _lib.lib().gd_rowpair_mse_workspace(10) >= 2 and _lib.lib().gd_rowtarget_mse_workspace(10) >= 2


def test_argument_errors_are_reported_not_thrown():
    from gnndelete_amd import _lib
    rc = _lib.lib().gd_spmm_csr_f32(None, None, None, None, 0, None, 0, None, 0.0, 4, 8, None)
    assert rc == 1 and b'null' in _lib.lib().gd_last_error_string()
    with pytest.raises(_lib.GnnDeleteHipError):
        _lib.check(rc, 'gd_spmm_csr_f32')


def _imports(path):
    with open(path) as f:
        tree = ast.parse(f.read())
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                yield a.name
        elif isinstance(node, ast.ImportFrom) and node.module:
            yield ('.' * node.level) + node.module


def test_product_code_never_imports_the_oracle():
    offenders = []
    product = [os.path.join(ROOT, f) for f in ('delete_gnn.py', 'train_gnn.py', 'prepare_dataset.py')]
    for base, _, files in os.walk(os.path.join(ROOT, 'gnndelete_amd')):
        product += [os.path.join(base, f) for f in files if f.endswith('.py')]
    for path in product:
        for mod in _imports(path):
            if mod.lstrip('.').split('.')[0] == 'oracle':
                offenders.append((path, mod))
        with open(path) as f:
            assert '/root/reference' not in f.read(), path
    assert not offenders, offenders


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from gnndelete_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.GnnDeleteHipError, match='no CPU fallback'):
        _lib.lib()


def test_cpu_tensors_are_rejected_by_the_ops():
    import torch
    from gnndelete_amd import _lib, ops
    with pytest.raises(_lib.GnnDeleteHipError):
        ops.rows_gemm(torch.zeros(4, 32), None, torch.zeros(32, 32))


def test_csr_from_coo_rejects_sizes_outside_the_int32_range():
    """Argument validation of gd_csr_from_coo_workspace needs no GPU: edge counts the int32 CSR cannot index."""
    from gnndelete_amd import _lib
    L = _lib.lib()
    assert L.gd_csr_from_coo_workspace(10, 2 ** 31) == -1
    assert L.gd_csr_from_coo_workspace(-1, 5) == -1
    assert L.gd_csr_from_coo_workspace(10, 0) == 256
