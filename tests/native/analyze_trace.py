"""per-layer TFLOP/s of conv_gemm launches from a rocprofv3 kernel trace of tests/native/prof_detr.py
   python3 analyze_trace.py <kernel_trace.csv> B [nrep] [resnet101] [--pmc <dir with pmc_<prefix>_FETCH_SIZE/ and _WRITE_SIZE/> <prefix> <bytes per element>]
With --pmc (round 6): two more columns per launch -- the COMPULSORY bytes of the layer(s) it covers (every input, residual, filter and output
element once, at the element size given) and the HBM-side bytes the counters saw for that dispatch (2 x FETCH_SIZE + WRITE_SIZE, the gfx950
correction of MI355X_MICROARCH.md; the passes are separate runs of the same launch sequence, matched by position) -- and their ratio."""
import csv, glob, sys
path, B = sys.argv[1], int(sys.argv[2])
PMC = None
if "--pmc" in sys.argv:
    k_ = sys.argv.index("--pmc")
    PMC = (sys.argv[k_ + 1], sys.argv[k_ + 2], int(sys.argv[k_ + 3]))
    del sys.argv[k_:k_ + 4]
ESZ = PMC[2] if PMC else 2
nrep = int(sys.argv[3]) if len(sys.argv) > 3 else 3
BLOCKS = (3, 4, 23, 3) if (len(sys.argv) > 4 and sys.argv[4] == "resnet101") else (3, 4, 6, 3)
rows = [r for r in csv.DictReader(open(path)) if "conv_gemm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // nrep
convs = rows[(nrep - 1) * n:]
def co(n, k, s, p): return (n + 2 * p - k) // s + 1
H, W = 800, 1066
import re
def fuse_kind(row):          # 5th template argument of the ring kernel: 1 / 3 = 3x3 + expand fused, 2 / 4 = ... + the next block's reduce
    m_ = re.search(r"conv_gemm_big_kernel<\d+, \d+, \d+, \d+, (\d+)>", row["Kernel_Name"])
    return int(m_.group(1)) if m_ else 0
seq = []                    # (name, flops) in launch order; built while walking the trace: fused launches cover several layers
pos = 0
def take(name, fl, el=0.0):      # el: compulsory ELEMENTS of the launch (inputs + residual + filters + outputs, each once)
    global pos
    seq.append((name, fl, el * ESZ)); pos += 1
H1, W1 = co(H, 7, 2, 3), co(W, 7, 2, 3)
h, w = co(H1, 3, 2, 1), co(W1, 3, 2, 1); inpl = 64
# (the stem reads the framed NHWC4 image and, with the pool on its tile, writes the POOLED tensor)
take("stem7x7", 2.0 * B * H1 * W1 * 64 * 147, B * H * W * 4 + B * h * w * 64 + 64 * 147)
have_c1 = False
for l, nb in enumerate(BLOCKS):
    pl = 64 * 2 ** l
    for i in range(nb):
        s = 2 if (i == 0 and l > 0) else 1
        ho, wo = co(h, 3, s, 1), co(w, 3, s, 1)
        f_c1, f_c2, f_c3 = 2.0 * B * h * w * pl * inpl, 2.0 * B * ho * wo * pl * pl * 9, 2.0 * B * ho * wo * pl * 4 * pl
        px_i, px_o = B * h * w, B * ho * wo
        e_c2 = px_i * pl + 9 * pl * pl                      # the 3x3's input + filters (its output stays on chip when fused)
        e_c3 = 4 * pl * pl + 2 * px_o * 4 * pl              # expand filters + residual in + block output
        if not have_c1: take(f"l{l+1}.{i}.c1", f_c1, px_i * inpl + px_i * pl + pl * inpl)
        have_c1 = False
        if i == 0: take(f"l{l+1}.{i}.ds", 2.0 * B * ho * wo * pl * 4 * inpl, px_o * inpl + px_o * 4 * pl + 4 * pl * inpl)
        if i == 0: e_c3 -= 0                               # (block 0's residual is the downsample branch's output: read all the same)
        fk = fuse_kind(convs[pos]) if pos < len(convs) else 0
        if fk >= 6:           # bf16 bottleneck on the tile with the chained reduce: PN = 32 << (fk - 5) output channels
            pn = 32 << (fk - 5)
            take(f"l{l+1}.{i}.c2+c3+c1'", f_c2 + f_c3 + 2.0 * B * ho * wo * pn * pl * 4, e_c2 + e_c3 + px_o * pn + pn * 4 * pl); have_c1 = True
        elif fk in (2, 4):    # 4: the chained reduce is the next STAGE's (twice the planes)
            pn = pl if fk == 2 else 2 * pl
            take(f"l{l+1}.{i}.c2+c3+c1'", f_c2 + f_c3 + 2.0 * B * ho * wo * pn * pl * 4, e_c2 + e_c3 + px_o * pn + pn * 4 * pl); have_c1 = True
        elif fk:
            take(f"l{l+1}.{i}.c2+c3", f_c2 + f_c3, e_c2 + e_c3)
        else:
            take(f"l{l+1}.{i}.c2", f_c2, e_c2 + px_o * pl); take(f"l{l+1}.{i}.c3", f_c3, px_o * pl + e_c3)
        inpl = pl * 4; h, w = ho, wo
def add(name, m, nn, k): seq.append((name, 2.0 * m * nn * k, (m * k + m * nn + nn * k) * ESZ))
M = B * h * w; Mq = B * 100
add("input_proj", M, 256, 2048)
for i in range(6):
    for t_ in [(f"enc{i}.qk", M, 512, 256), (f"enc{i}.v", M, 256, 256), (f"enc{i}.out", M, 256, 256), (f"enc{i}.l1", M, 2048, 256), (f"enc{i}.l2", M, 256, 2048)]: add(*t_)
add("cross_k", M, 1536, 256); add("cross_v", M, 1536, 256)
for i in range(6):
    for t_ in [(f"dec{i}.qk", Mq, 512, 256), (f"dec{i}.v", Mq, 256, 256), (f"dec{i}.out", Mq, 256, 256), (f"dec{i}.cq", Mq, 256, 256), (f"dec{i}.cout", Mq, 256, 256), (f"dec{i}.l1", Mq, 2048, 256), (f"dec{i}.l2", Mq, 256, 2048)]: add(*t_)
add("cls", Mq, 19, 256)
for nm, no in (("bbox", 4), ("off", 2), ("ang", 30), ("size", 3), ("depth", 1)):
    add(nm + "0", Mq, 256, 256); add(nm + "1", Mq, 256, 256); add(nm + "2", Mq, no, 256)
assert len(convs) == len(seq), (len(convs), len(seq))
pmc_mb = None
if PMC:
    def load(name):
        p_ = glob.glob(f"{PMC[0]}/pmc_{PMC[1]}_{name}/**/*counter_collection.csv", recursive=True)[0]
        out = {}
        for r_ in csv.DictReader(open(p_)):
            if r_["Counter_Name"] == name and "conv_gemm" in r_["Kernel_Name"]:
                out[int(r_["Dispatch_Id"])] = float(r_["Counter_Value"])
        ids = sorted(out)
        ids = ids[len(ids) - len(seq):]                      # the last forward of that run
        return [out[i_] for i_ in ids]
    F_, W_ = load("FETCH_SIZE"), load("WRITE_SIZE")
    assert len(F_) == len(seq) == len(W_), (len(F_), len(W_), len(seq))
    pmc_mb = [(2 * f_ + w_) / 1024 for f_, w_ in zip(F_, W_)]
agg = {}
bagg = {}
for li, (r, (name, fl, by)) in enumerate(zip(convs, seq)):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cfg = r["Kernel_Name"].split("<")[1].split(">")[0]
    grp = name.split(".")[0] if name[0] == "l" and name[1].isdigit() else ("enc" if name.startswith("enc") else "dec" if name.startswith("dec") else name if name in ("stem7x7", "input_proj") else "cross" if name.startswith("cross") else "heads")
    a = agg.setdefault(grp, [0, 0]); a[0] += us; a[1] += fl
    extra = ""
    if pmc_mb is not None:
        b_ = bagg.setdefault(grp, [0.0, 0.0]); b_[0] += by / 1e6; b_[1] += pmc_mb[li] * 1.048576
        extra = f"  compulsory {by/1e6:7.1f} MB  PMC {pmc_mb[li]*1.048576:7.1f} MB  x{pmc_mb[li]*1.048576/(by/1e6):4.2f}  {(pmc_mb[li]*1.048576)/us:5.2f} TB/s"
    if name.startswith(("stem", "l1.0", "l1.1", "l2.0", "l2.1", "l3.0", "l3.1", "l4.0", "l4.1", "input", "enc0", "cross", "dec0")):
        print(f"{name:16s} {fl/1e9:8.2f} GF cfg={cfg:16s} wgs={int(r['Grid_Size_X'])//256:6d} {us:8.1f} us  {fl/us/1e6:6.1f} TF{extra}")
print()
tu = tf = 0
tb = [0.0, 0.0]
for g, (us, fl) in agg.items():
    extra = ""
    if pmc_mb is not None:
        c_, p_ = bagg[g]; tb[0] += c_; tb[1] += p_
        extra = f"  compulsory {c_:8.0f} MB  PMC {p_:8.0f} MB  x{p_/c_:4.2f}  {p_/us:5.2f} TB/s"
    print(f"{g:12s} {us/1e3:7.2f} ms {fl/1e9:8.1f} GF  {fl/us/1e6:6.1f} TF{extra}"); tu += us; tf += fl
extra = f"  compulsory {tb[0]:8.0f} MB  PMC {tb[1]:8.0f} MB  x{tb[1]/tb[0]:4.2f}  {tb[1]/tu:5.2f} TB/s" if pmc_mb is not None else ""
print(f"{'total':12s} {tu/1e3:7.2f} ms {tf/1e9:8.1f} GF  {tf/tu/1e6:6.1f} TF{extra}")
