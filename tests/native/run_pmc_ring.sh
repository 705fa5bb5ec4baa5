#!/bin/bash
# where the ring kernel's wave-cycles go (per layer shape of perf_conv_shape.py): two SQ counter passes per dtype
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for dt in bf16 fp32; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_ring_${dt}_a -o t -- python3 $R/tests/native/perf_conv_shape.py $dt > $R/gpurun_out/pmc_ring_${dt}_a.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/pmc_ring_${dt}_b -o t -- python3 $R/tests/native/perf_conv_shape.py $dt > $R/gpurun_out/pmc_ring_${dt}_b.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for dt in ("bf16", "fp32"):
    rows = collections.OrderedDict()
    for ab in "ab":
        f = glob.glob(f"{R}/gpurun_out/pmc_ring_{dt}_{ab}/**/*counter_collection.csv", recursive=True)
        if not f: continue
        for r in csv.DictReader(open(f[0])):
            if "conv_gemm" not in r["Kernel_Name"]: continue
            key = (r["Dispatch_Id"], r["Kernel_Name"][:60], r["Grid_Size"])
            rows.setdefault((ab, r["Dispatch_Id"]), {"k": r["Kernel_Name"].split("<")[1].split(">")[0] if "<" in r["Kernel_Name"] else r["Kernel_Name"], "grid": r["Grid_Size"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    # last dispatch of each distinct (kernel, grid) in each pass
    out = {}
    for (ab, d), v in rows.items():
        out.setdefault((v["k"], v["grid"]), {}).update(v)
    print("==", dt)
    for (k, g), v in out.items():
        wc = v.get("SQ_WAVE_CYCLES", 0)
        if wc <= 0: continue
        f = lambda n: v.get(n, 0) / wc
        print(f"{k:28s} grid {g:>9s}: wait_any {f('SQ_WAIT_ANY'):.2f} wait_inst {f('SQ_WAIT_INST_ANY'):.2f} active {f('SQ_ACTIVE_INST_ANY'):.2f} valu {f('SQ_ACTIVE_INST_VALU'):.2f} lds {f('SQ_ACTIVE_INST_LDS'):.2f} sca {f('SQ_ACTIVE_INST_SCA'):.2f} | mfma_busy/busy_cycles {v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(v.get('SQ_BUSY_CYCLES',1),1):.2f} | lds conflict/active {v.get('SQ_LDS_BANK_CONFLICT',0)/max(v.get('SQ_LDS_IDX_ACTIVE',1),1):.2f} insts valu {v.get('SQ_INSTS_VALU',0):.3g} salu {v.get('SQ_INSTS_SALU',0):.3g} lds {v.get('SQ_INSTS_LDS',0):.3g} vmem {v.get('SQ_INSTS_VMEM',0):.3g}")
PY
