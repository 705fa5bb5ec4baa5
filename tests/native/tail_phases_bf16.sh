#!/bin/bash
# What the bf16 bottleneck-on-the-tile launches spend where: timing builds of cg_fused_bf16.hip (CG_TAILHACK, cg_tail_bf16.hpp; WRONG
# results) linked with the tree's other objects, each timed against the real library on one box by tests/native/ab_libs.py.
#   here (CPU):   bash tests/native/tail_phases_bf16.sh build      -> build/ab/libodam_th<N>.so
#   on the box:   bash tests/native/tail_phases_bf16.sh run [batch]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
if [ "$1" = "build" ]; then
  python3 -c "import __graft_entry__ as g; g.build()"
  mkdir -p build/ab
  for h in 1 2 3 4 8 9 11; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DCG_TAILHACK=$h -c odam_amd/csrc/cg_fused_bf16.hip -o build/ab/cg_fused_bf16_th$h.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libodam_th$h.so $(ls build/obj/*.o | grep -v cg_fused_bf16) build/ab/cg_fused_bf16_th$h.o -ldl -pthread ) &
  done
  wait
  rm -f build/ab/*.o
  ls -la build/ab
else
  B=${2:-32}
  for h in 1 2 3 4 8 9 11; do
    python3 tests/native/ab_libs.py odam_amd/libodam_amd.so build/ab/libodam_th$h.so resnet101 bf16 $B 3 2>&1 | grep "ms per forward" | sed "s/^/hack $h: /"
  done
fi
