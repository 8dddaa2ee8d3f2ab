#!/bin/bash
# build the library for the current sources, then hand the command to gpurun (a stale .so is refused by _lib.py on the box)
set -e
cd "$(dirname "$0")/.."
make -C gnndelete_amd/csrc -j6 2>&1 | grep -E "error|Error|warning: unused" || true
python -c "from gnndelete_amd import _lib; _lib.lib(); b, h = _lib.build_stamp(); assert b == h, (b, h); print('lib stamp', b)"
exec /usr/local/graft/bin/gpurun "$@"
