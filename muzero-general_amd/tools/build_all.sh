#!/bin/bash
# Rebuild the gfx950 library and the CPU test double (run from anywhere).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT"
set -o pipefail
python muzero-general_amd/build.py "$@" 2>&1 | { grep -v "^/opt/rocm" || true; }
python -c "
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'muzero-general_amd')
import hostcheck; hostcheck.build(force=True)"
echo "build ok"
