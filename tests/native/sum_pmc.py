"""sum one rocprofv3 PMC counter per kernel: python3 sum_pmc.py <counter_collection.csv> <COUNTER> > summary.csv"""
import csv, re, sys
from collections import defaultdict
path, name = sys.argv[1], sys.argv[2]
tot, n = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] != name:
        continue
    k = re.sub(r"\([^()]*\)$", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))[:100]
    tot[k] += float(r["Counter_Value"]); n[k] += 1
print(f"kernel,dispatches,sum_{name}")
for k in sorted(tot, key=lambda k: -tot[k]):
    print(f'"{k}",{n[k]},{tot[k]:.1f}')
