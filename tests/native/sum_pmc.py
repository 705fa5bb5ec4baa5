"""rocprofv3 PMC csv -> per-kernel sums.
  python3 sum_pmc.py <counter_collection.csv> <COUNTER>       one counter, one line per kernel (csv on stdout)
  python3 sum_pmc.py --summary <dir with pmc_*/ from run_pmc.sh>   profiles/pmc_summary.json on stdout (what bench.py reads)
Conventions (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reads
half of the bytes of wide streaming reads, so HBM-side bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024."""
import csv, glob, json, os, re, sys
from collections import defaultdict


def sums(path, names):
    tot = {n: defaultdict(float) for n in names}
    cnt = defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] not in tot:
            continue
        k = re.sub(r"\([^()]*\)$", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))[:100]
        tot[r["Counter_Name"]][k] += float(r["Counter_Value"])
        if r["Counter_Name"] == names[0]:
            cnt[k] += 1
    return tot, cnt


def _csv(d, name):
    f = glob.glob(os.path.join(d, "pmc_" + name, "**", "*counter_collection.csv"), recursive=True)
    return f[0] if f else None


def summary(d):
    out = {"generated_by": "tests/native/run_pmc.sh + sum_pmc.py --summary", "units": "bytes; FETCH_SIZE doubled (gfx950)"}

    def traffic(prefix, match, per):
        fp, wp = _csv(d, prefix + "_FETCH_SIZE"), _csv(d, prefix + "_WRITE_SIZE")
        if not fp or not wp:
            return None
        f, n = sums(fp, ["FETCH_SIZE"]); w, _ = sums(wp, ["WRITE_SIZE"])
        fk = sum(v for k, v in f["FETCH_SIZE"].items() if match in k); wk = sum(v for k, v in w["WRITE_SIZE"].items() if match in k)
        disp = sum(v for k, v in n.items() if match in k)
        return {"fetch_kb": fk, "write_kb": wk, "dispatches": disp, "bytes_total": (2 * fk + wk) * 1024, "per": per}
    t = traffic("r50f32", "conv_gemm", 64)
    if t:
        out["conv_gemm_fp32"] = {"bytes_per_frame": t["bytes_total"] / 64, "fetch_kb": t["fetch_kb"], "write_kb": t["write_kb"],
                                 "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over 2 forwards of 32 frames (prof_detr1.py 32 fp32 resnet50)"}
    t = traffic("r101bf16", "conv_gemm", 64)
    if t:
        out["conv_gemm_bf16_r101"] = {"bytes_per_frame": t["bytes_total"] / 64, "fetch_kb": t["fetch_kb"], "write_kb": t["write_kb"],
                                      "source": "the same passes on prof_detr1.py 32 bf16 resnet101"}
    t = traffic("sq", "sq_fit", 2)
    sqd = {}
    if t:
        sqd.update({"bytes_per_launch": t["bytes_total"] / max(t["dispatches"], 1), "fetch_kb": t["fetch_kb"], "write_kb": t["write_kb"],
                    "launches": t["dispatches"]})
    vp = _csv(d, "sq_valu")
    if vp:
        names = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE"]
        v, _ = sums(vp, names)
        g = lambda n: sum(x for k, x in v[n].items() if "sq_fit" in k)
        if g("SQ_WAVE_CYCLES") > 0:
            sqd.update({"valu_util": g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"), "wait_frac": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
                        "lds_frac": g("SQ_ACTIVE_INST_LDS") / g("SQ_WAVE_CYCLES"), "valu_insts": g("SQ_INSTS_VALU"),
                        "source": "rocprofv3 --pmc SQ_* on prof_sq.py (2 launches of 64 objects x 256 views x 200 steps): valu_util = "
                                  "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of resident-wave time with a VALU instruction in flight), "
                                  "wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked at s_waitcnt / barriers)"})
    if sqd:
        out["sq_fit"] = sqd
    for name, key in (("r50f32_mfma", "mfma_util_fp32"), ("r101bf16_mfma", "mfma_util_bf16_r101")):
        mp = _csv(d, name)
        if mp:
            v, n = sums(mp, ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])
            res = {}
            for k, b in v["SQ_VALU_MFMA_BUSY_CYCLES"].items():
                a = v["GRBM_GUI_ACTIVE"].get(k, 0)
                if b > 0 and a > 0:
                    res[k] = round(b / (a / 8 * 1024), 4)      # busy summed over 1024 SIMDs, GUI_ACTIVE over 8 XCDs
            out[key] = res
    return out


if __name__ == "__main__":
    if sys.argv[1] == "--summary":
        print(json.dumps(summary(sys.argv[2]), indent=1))
    else:
        path, name = sys.argv[1], sys.argv[2]
        tot, n = sums(path, [name])
        print(f"kernel,dispatches,sum_{name}")
        for k in sorted(tot[name], key=lambda k: -tot[name][k]):
            print(f'"{k}",{n[k]},{tot[name][k]:.1f}')
