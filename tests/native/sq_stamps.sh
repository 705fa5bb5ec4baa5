#!/bin/bash
# per-phase time of one Adam step of sq_fit_kernel: builds the library with -DSQ_STAMPS into tests/native/_abl/libstamps.so (the stamped
# build overwrites out_points of object 0 with tick sums: diagnostics only) and runs perf_sq_stamps.py
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 - <<'PY'
import __graft_entry__ as g
g.build_library(lib="tests/native/_abl/libstamps.so", extra_flags=("-DSQ_STAMPS",))
PY
python3 tests/native/perf_sq_stamps.py 2>&1 | grep -v amdgpu
