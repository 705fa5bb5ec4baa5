#!/bin/bash
# Where the waves of every contraction kernel of an fp32 R50 forward spend their cycles (rocprofv3 --pmc, two passes of SQ counters):
#   parked (SQ_WAIT_ANY: s_waitcnt / barriers), issue-stalled (SQ_WAIT_INST_ANY, of which SQ_WAIT_INST_LDS), issuing (SQ_ACTIVE_INST_*)
#   bash tests/native/pmc_conv_cycles.sh  -> gpurun_out/r5_conv_wave_cycles.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_cyc1 $O/pmc_cyc2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc_cyc1 -o t -- python3 $R/tests/native/prof_detr1.py 32 fp32 resnet50 > $O/pmc_cyc1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmc_cyc2 -o t -- python3 $R/tests/native/prof_detr1.py 32 fp32 resnet50 > $O/pmc_cyc2.log 2>&1
python3 - "$O" <<'PY' > $O/r5_conv_wave_cycles.txt
import csv, glob, sys, collections
O = sys.argv[1]
def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    return acc
a, b = load("pmc_cyc1"), load("pmc_cyc2")
print("kernel | share of wave cycles: parked (waitcnt / barrier)  issue-stalled (of which on LDS)  issuing  [VALU  LDS  VMEM in flight] | LDS conflict share | VALU instructions per wave-cycle")
for k in sorted(a, key=lambda k: -a[k].get("SQ_WAVE_CYCLES", 0)):
    A, B = a[k], b.get(k, {})
    w = A.get("SQ_WAVE_CYCLES", 0)
    if w <= 0 or "conv_gemm" not in k and "attention" not in k: continue
    g = lambda d, n: d.get(n, 0.0)
    print("%-70s | %.2f  %.2f (%.2f)  %.2f  [%.2f %.2f %.2f] | %.3f | %.3f" % (
        k[-70:], g(A, "SQ_WAIT_ANY") / w, g(A, "SQ_WAIT_INST_ANY") / w, g(A, "SQ_WAIT_INST_LDS") / w, g(A, "SQ_ACTIVE_INST_ANY") / w,
        g(A, "SQ_ACTIVE_INST_VALU") / w, g(A, "SQ_ACTIVE_INST_LDS") / w, g(A, "SQ_ACTIVE_INST_VMEM") / w,
        g(B, "SQ_LDS_BANK_CONFLICT") / max(g(B, "SQ_LDS_IDX_ACTIVE"), 1),
        g(B, "SQ_INSTS_VALU") / max(g(B, "SQ_WAVE_CYCLES"), 1)))
PY
cat $O/r5_conv_wave_cycles.txt
