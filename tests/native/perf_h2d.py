"""PCIe-inclusive detection rate: frames start in pinned host memory (float32 [3,800,1066], what the reference's
transform hands to the detector) and are uploaded batch by batch under the kernels; compared with resident frames."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import detector, weights, synth
N, B = 384, 32
det = detector.Detector(max_batch=B, n_streams=2); det.load_state_dict(weights.make_state_dict())
host = torch.empty(N, 3, 800, 1066, dtype=torch.float32).pin_memory()
host.normal_()
dev = host.to("cuda:0")
K = synth.K_SCANNET
def run(fr):
    out = []
    for c0, rows in det.detect_resident_chunks(fr, (synth.IMG_W, synth.IMG_H), K, chunk=4 * B):
        out.append(rows.copy())
    return np.concatenate(out)
a = run(dev); b = run(host)
print("host-upload path equals resident path:", np.array_equal(a, b), flush=True)
raw = torch.randint(0, 256, (N, 480, 640, 3), dtype=torch.uint8)
raw_pin, raw_dev = raw.pin_memory(), raw.to("cuda:0")
print("raw uint8: pinned-upload path equals resident path:", np.array_equal(run(raw_dev), run(raw_pin)), flush=True)
for name, fr in (("resident", dev), ("pinned host + upload", host), ("raw uint8 resident + device transform", raw_dev),
                 ("raw uint8 pinned + upload + device transform", raw_pin)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2): run(fr)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    print(f"{name:44s}: {N/dt:7.1f} frames/s  ({fr[0].numel()*fr.element_size()*N/dt/1e9:.2f} GB/s of input)", flush=True)
