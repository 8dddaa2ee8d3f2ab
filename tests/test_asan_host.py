"""Host-side AddressSanitizer build of the C-ABI library (ASAN host build; GPU ASAN / xnack+ is not
available on the pool): `make -C gnndelete_amd/csrc asan-check` compiles every source with -fsanitize=address on the HOST
side (hipcc ignores it for the gfx950 code objects) and runs tests/test_layout.py - symbol table, argument validation, error
strings, workspace arithmetic - against that library under LD_PRELOAD of the ASAN runtime.  No GPU needed."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get('GNNDELETE_HIP_LIB') is not None, reason='already running against an alternative library')
@pytest.mark.skipif(shutil.which('make') is None or not os.path.exists('/opt/rocm/bin/hipcc'), reason='needs make + hipcc')
def test_layout_suite_passes_under_host_asan():
    r = subprocess.run(['make', '-C', os.path.join(ROOT, 'gnndelete_amd', 'csrc'), '-j8', 'asan-check'], capture_output=True, text=True,
                       timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert 'passed' in r.stdout and 'ERROR: AddressSanitizer' not in tail, tail
