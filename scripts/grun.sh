#!/bin/bash
# build + load check, then gpurun: never send a tree whose library does not build
cd /root/repo
python - <<'PY' || { echo "BUILD FAILED: not calling gpurun"; exit 1; }
from chemprop_amd import _lib
_lib.build(force=False) if not _lib._stale() else _lib.build(force=True)
l = _lib.load()
assert l.dmpnn_version() == _lib.ABI_VERSION
print("lib ok")
PY
T=${TIMEOUT:-1200}
/usr/local/graft/bin/gpurun --timeout $T -- "$@"
