"""HBM-side bytes of every contraction launch of one forward, from the two PMC passes of run_pmc.sh:
   python3 pmc_per_dispatch.py <dir with pmc_r50f32_FETCH_SIZE/ and pmc_r50f32_WRITE_SIZE/> [prefix]
FETCH_SIZE is doubled (gfx950 reports half the bytes of wide streaming reads); MB per launch, second forward of the run."""
import csv, glob, re, sys


def load(d, prefix, name):
    p = glob.glob(f"{d}/pmc_{prefix}_{name}/**/*counter_collection.csv", recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] == name:
            out[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]), r["Grid_Size"])
    return out


d = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else "r50f32"
F, W = load(d, prefix, "FETCH_SIZE"), load(d, prefix, "WRITE_SIZE")
conv = [i for i in sorted(F) if "conv_gemm" in F[i][0]]
second = conv[len(conv) // 2:]
tot = 0.0
for i in second:
    name = re.sub(r"void odam_cg::|\(odam_cg::ConvGemmArgs\)", "", F[i][0])[:46]
    f, w = 2 * F[i][1] / 1024, W.get(i, ("", float("nan"), ""))[1] / 1024
    tot += f + w
    print(f"{name:48s} grid {F[i][2]:>9s}  fetch {f:8.1f} MB  write {w:8.1f} MB")
print(f"contractions of one forward: {tot:.0f} MB")
