#!/bin/bash
# build libonet_hip.so; non-zero exit when hipcc fails (so that "tools/b.sh && gpurun ..." never ships a stale library)
cd "$(dirname "$0")/.." && python - <<'PY'
import sys
from onet_amd import build
try:
    build.build(verbose=False)
except Exception as e:
    print("BUILD FAILED:", e); sys.exit(1)
PY
