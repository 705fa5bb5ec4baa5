"""per-layer TFLOP/s of conv_gemm launches from a rocprofv3 kernel trace of tests/native/prof_detr.py"""
import csv, sys
path, B = sys.argv[1], int(sys.argv[2])
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
def take(name, fl):
    global pos
    seq.append((name, fl)); pos += 1
H1, W1 = co(H, 7, 2, 3), co(W, 7, 2, 3); take("stem7x7", 2.0 * B * H1 * W1 * 64 * 147)
h, w = co(H1, 3, 2, 1), co(W1, 3, 2, 1); inpl = 64
have_c1 = False
for l, nb in enumerate(BLOCKS):
    pl = 64 * 2 ** l
    for i in range(nb):
        s = 2 if (i == 0 and l > 0) else 1
        ho, wo = co(h, 3, s, 1), co(w, 3, s, 1)
        f_c1, f_c2, f_c3 = 2.0 * B * h * w * pl * inpl, 2.0 * B * ho * wo * pl * pl * 9, 2.0 * B * ho * wo * pl * 4 * pl
        if not have_c1: take(f"l{l+1}.{i}.c1", f_c1)
        have_c1 = False
        if i == 0: take(f"l{l+1}.{i}.ds", 2.0 * B * ho * wo * pl * 4 * inpl)
        fk = fuse_kind(convs[pos]) if pos < len(convs) else 0
        if fk >= 6:           # bf16 bottleneck on the tile with the chained reduce: PN = 32 << (fk - 5) output channels
            take(f"l{l+1}.{i}.c2+c3+c1'", f_c2 + f_c3 + 2.0 * B * ho * wo * (32 << (fk - 5)) * pl * 4); have_c1 = True
        elif fk in (2, 4):    # 4: the chained reduce is the next STAGE's (twice the planes)
            take(f"l{l+1}.{i}.c2+c3+c1'", f_c2 + f_c3 + 2.0 * B * ho * wo * (pl if fk == 2 else 2 * pl) * pl * 4); have_c1 = True
        elif fk:
            take(f"l{l+1}.{i}.c2+c3", f_c2 + f_c3)
        else:
            take(f"l{l+1}.{i}.c2", f_c2); take(f"l{l+1}.{i}.c3", f_c3)
        inpl = pl * 4; h, w = ho, wo
def add(name, m, nn, k): seq.append((name, 2.0 * m * nn * k))
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
agg = {}
for r, (name, fl) in zip(convs, seq):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cfg = r["Kernel_Name"].split("<")[1].split(">")[0]
    grp = name.split(".")[0] if name[0] == "l" and name[1].isdigit() else ("enc" if name.startswith("enc") else "dec" if name.startswith("dec") else name if name in ("stem7x7", "input_proj") else "cross" if name.startswith("cross") else "heads")
    a = agg.setdefault(grp, [0, 0]); a[0] += us; a[1] += fl
    if name.startswith(("stem", "l1.0", "l1.1", "l2.0", "l2.1", "l3.0", "l3.1", "l4.0", "l4.1", "input", "enc0", "cross", "dec0")):
        print(f"{name:16s} {fl/1e9:8.2f} GF cfg={cfg:16s} wgs={int(r['Grid_Size_X'])//256:6d} {us:8.1f} us  {fl/us/1e6:6.1f} TF")
print()
tu = tf = 0
for g, (us, fl) in agg.items():
    print(f"{g:12s} {us/1e3:7.2f} ms {fl/1e9:8.1f} GF  {fl/us/1e6:6.1f} TF"); tu += us; tf += fl
print(f"{'total':12s} {tu/1e3:7.2f} ms {tf/1e9:8.1f} GF  {tf/tu/1e6:6.1f} TF")
