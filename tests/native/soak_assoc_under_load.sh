#!/bin/bash
# association (persistent matching kernel with device-wide barriers) while another process keeps every CU busy with the
# detector's one-workgroup-per-CU kernels: no barrier time-out (NaN assignment), same results
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 - > /dev/null 2>&1 <<'PY' &
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from odam_amd import detector, weights
det = detector.Detector(max_batch=32, n_streams=1); det.load_state_dict(weights.make_state_dict())
img = torch.randn(32, 3, 800, 1066, device="cuda:0")
t0 = time.time()
while time.time() - t0 < 75: det(img)
torch.cuda.synchronize()
PY
BG=$!
sleep 25
for i in 1 2 3; do
  timeout 300 python3 -m pytest tests/test_assoc_gpu.py tests/test_e2e.py -m gpu -q -x 2>&1 | tail -1
done
timeout 100 python3 tests/native/perf_assoc.py 2>&1 | grep "case 3"
wait $BG 2>/dev/null
echo done
