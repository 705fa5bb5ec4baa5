#!/bin/bash
# Round 5: the same-box A/B runs behind profiles/r5_experiments.txt #6-#10, reproducible from this tree (one gpurun call each):
#   bash tests/native/r5_experiments.sh breg | wm8 | sq | batch
# Every variant library / harness binary is built here, from the sources as they are, with the build-time switch named; outputs go to
# gpurun_out/.  (#1-#5 used code that is gone again -- cg.stagger, cg_persist.hpp, the first SQ variants -- and #10's switch CG_TAIL_STAGGER too.)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
lib() {      # lib <name> <flags...>: odam_amd/libodam_amd_<name>.so built with the extra compile flags
  local name=$1; shift
  python3 - "$name" "$@" <<'PY'
import sys, __graft_entry__ as g
g.build_library(lib="odam_amd/libodam_amd_%s.so" % sys.argv[1], extra_flags=tuple(sys.argv[2:]))
PY
}
harness() {  # harness <name> <flags...>: tests/native/ring_stamps0_<name>, the torch-free layer harness
  local name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DCG_STAMP=0 "$@" -I odam_amd/csrc -o tests/native/ring_stamps0_$name tests/native/ring_stamps.hip
}
ab() {       # ab <tag> <macro>: harness and forward with <macro>=0 against the tree's default, alternating
  local tag=$1 macro=$2
  lib ${tag}_off -D$macro=0; harness ${tag}_off -D$macro=0; harness ${tag}_on
  (for rep in 1 2; do echo "== $macro=0"; ./tests/native/ring_stamps0_${tag}_off x16; echo "== this tree"; ./tests/native/ring_stamps0_${tag}_on x16; done) > $O/r5_ring_$tag.txt 2>&1
  python3 tests/native/ab_libs.py odam_amd/libodam_amd_${tag}_off.so odam_amd/libodam_amd.so resnet50 fp32 32 3 > $O/r5_ab_$tag.txt 2>&1
  cat $O/r5_ring_$tag.txt $O/r5_ab_$tag.txt
}
case "$1" in
  breg) ab breg CG_X16_BREG ;;        # 6: filter fragments of a column group in registers
  wm8)  ab wm8 CG_X16_WM8 ;;          # 7: 256-column tiles as eight waves of 32 x 256
  sq)                                 # 8: view split 16 against 32 (the scan itself has no switch: compare with a library of an older commit)
    (for cfg in sq.split=16 sq.split=32; do
       ODAM_CONFIG=$cfg python3 tests/native/perf_sq_fit30.py 2>&1 | grep -v amdgpu
       ODAM_CONFIG=$cfg python3 tests/native/perf_sq_fit30.py 8 2>&1 | grep -v amdgpu
     done; python3 tests/native/perf_sq.py 2>&1 | grep -v amdgpu) > $O/r5_sq8_ab.txt 2>&1
    (for cfg in sq.split=16 sq.split=32; do echo "== this kernel, $cfg"; ODAM_CONFIG=$cfg bash tests/native/sq_stamps.sh; done) > $O/r5_sq8_stamps.txt 2>&1
    cat $O/r5_sq8_ab.txt $O/r5_sq8_stamps.txt ;;
  batch)                              # 9: per-layer tables at 4 / 8 / 16 / 32 frames per forward, kernel choice pinned
    cd /tmp && export TMPDIR=/tmp ODAM_CONFIG=cg.pin=1
    for B in 4 8 16 32; do
      rm -rf $O/r5_b$B
      rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_b$B -o t -- python3 $R/tests/native/prof_detr1.py $B fp32 resnet50 > $O/r5_b$B.log 2>&1
      python3 $R/tests/native/analyze_trace.py $O/r5_b$B/t_kernel_trace.csv $B 2 resnet50 > $O/r5_per_layer_pinned_B$B.txt 2>&1
      rm -rf $O/r5_b$B
      echo "== B = $B"; tail -11 $O/r5_per_layer_pinned_B$B.txt
    done ;;
  *) echo "usage: $0 breg | wm8 | sq | batch" ;;
esac
