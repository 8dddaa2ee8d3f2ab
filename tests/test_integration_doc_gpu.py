"""INTEGRATION.md shows the ctypes + autograd.Function stub a maintainer of the reference would add around the C ABI
(the Del operator, framework/models/deletion.py:17-29).  This test EXECUTES that code block as printed, so the document
cannot drift from the library: forward and both gradients against the reference's DeletionLayer arithmetic in float64."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_integration_md_stub_runs_and_matches_the_del_operator():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    stub = next(b for b in blocks if 'class DelRows' in b)
    stub = stub.replace("ctypes.CDLL('libgnndelete_hip.so')",
                        f"ctypes.CDLL({os.path.join(ROOT, 'gnndelete_amd', 'lib', 'libgnndelete_hip.so')!r})")
    ns = {}
    exec(compile(stub, 'INTEGRATION.md', 'exec'), ns)
    g = torch.Generator().manual_seed(0)
    n, d = 500, 64
    x = torch.randn(n, d, generator=g)
    w = torch.randn(d, d, generator=g) * 0.1
    mask = torch.rand(n, generator=g) < 0.4
    up = torch.randn(n, d, generator=g)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = xd.clone()
    ref[mask] = ref[mask] @ wd                               # DeletionLayer.forward (deletion.py:24-25)
    ref.backward(up.double())
    xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    idx = mask.nonzero().flatten().int().cuda()
    z = ns['DelRows'].apply(xg, wg, idx)
    z.backward(up.cuda())
    rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
    assert rel(z.detach(), ref.detach()) < 1e-5
    assert rel(xg.grad, xd.grad) < 1e-5 and rel(wg.grad, wd.grad) < 1e-5
