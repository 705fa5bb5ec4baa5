"""MFMA pipe utilisation per kernel from one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass:
busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs -> util = busy / (active / 8 * 1024)"""
import csv, re, sys
from collections import defaultdict
busy, act, n = defaultdict(float), defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\([^()]*\)$", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))[:100]
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES": busy[k] += float(r["Counter_Value"]); n[k] += 1
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE": act[k] += float(r["Counter_Value"])
print("kernel,dispatches,sum_SQ_VALU_MFMA_BUSY_CYCLES,sum_GRBM_GUI_ACTIVE,mfma_util")
for k in sorted(busy, key=lambda k: -busy[k]):
    if busy[k] > 0 and act[k] > 0:
        print(f'"{k}",{n[k]},{busy[k]:.0f},{act[k]:.0f},{busy[k] / (act[k] / 8 * 1024):.3f}')
