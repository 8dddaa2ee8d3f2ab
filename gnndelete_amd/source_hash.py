"""sha256 over the kernel sources and the C header (sorted by path): ONE definition of "the sources a library was built from",
used by the Makefile (stamped into libgnndelete_hip.so as gd_build_source_hash()), by gnndelete_amd/_lib.py (refuses a library
whose stamp differs from the sources next to it) and by bench.py (stamps a stage profile with the kernels it measured).
No imports beyond the standard library: the Makefile runs this file as a script."""
import glob
import hashlib
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)


def source_files():
    csrc = os.path.join(_PKG, 'csrc')
    return sorted(glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.h')) + glob.glob(os.path.join(csrc, '*.cpp'))
                  + [os.path.join(ROOT, 'include', 'gnndelete_hip.h')])


def source_hash():
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == '__main__':
    print(source_hash())
