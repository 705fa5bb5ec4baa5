#!/bin/bash
# on the box: per-kernel average durations of the fused bf16 launches for the real library and every timing build (see tail_phases_bf16.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tail_phases; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for h in real 1 2 3 4 8 9 11; do
  if [ $h = real ]; then export ODAM_AMD_LIB=$R/odam_amd/libodam_amd.so; else export ODAM_AMD_LIB=$R/build/ab/libodam_th$h.so; fi
  rm -rf $O/t_$h
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$h -o t -- python3 $R/tests/native/prof_detr1.py 32 bf16 resnet101 > $O/t_$h.log 2>&1
  python3 - $O/t_$h/t_kernel_stats.csv $h <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for r in rows:
    m = re.search(r"conv_gemm_big_kernel<0, (\d+), 4, 512, ([567])>", r["Name"])
    if m: out.append("P=%s fuse=%s: %7.1f us x %s" % (m.group(1), m.group(2), float(r["AverageNs"]) / 1e3, r["Calls"]))
print("hack %-4s " % sys.argv[2] + " | ".join(sorted(out)))
PY
  rm -rf $O/t_$h
done
