"""The reference against ITSELF across ATen's CPU dispatch levels: the un-nudged fits of the 46 survey problems
(make_golden_sq_survey.py) re-run with ATEN_CPU_CAPABILITY = default / avx2 / avx512 -- the vector width torch's CPU kernels
(exp, log, pow, sin, cos, the reductions of the loss) are dispatched to.  Same reference source, same inputs, same machine: only
the instruction set of the elementwise kernels differs, i.e. last-bit differences of the kind any two correct fp32
implementations have.  Stored: final parameters per level -> tests/golden/sq_survey_dispatch.npz.  tests/test_sq_survey.py
prints the reference-vs-reference deviation next to ours.
Run: python tests/golden/make_golden_sq_dispatch.py       (a few minutes on 7 processes)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

LEVELS = ("default", "avx2", "avx512")


def _worker(level, out_path):
    """runs inside a process started with ATEN_CPU_CAPABILITY=level (read by torch at import)"""
    from multiprocessing import Pool
    import numpy as np
    from make_golden_sq_survey import CASES, _one
    import torch
    got = torch.backends.cpu.get_cpu_capability()
    jobs = sorted([(ci, F, seed, -1) for ci, (F, seed) in enumerate(CASES)], key=lambda j: -j[1])
    with Pool(7) as pool:
        res = pool.map(_one, jobs, chunksize=1)
    final = np.zeros((len(CASES), 9), np.float32)
    for ci, _, p, _ in res:
        final[ci] = p
    np.savez(out_path, final=final, capability=str(got))


def main():
    import numpy as np
    if len(sys.argv) == 4 and sys.argv[1] == "--worker":
        return _worker(sys.argv[2], sys.argv[3])
    data = {}
    for lv in LEVELS:
        tmp = os.path.join("/tmp", f"sq_dispatch_{lv}.npz")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--worker", lv, tmp], env={**os.environ, "ATEN_CPU_CAPABILITY": lv})
        z = np.load(tmp)
        data[f"final_{lv}"] = z["final"]
        data[f"capability_{lv}"] = z["capability"]
        print(lv, "->", z["capability"])
    ref = np.load(os.path.join(HERE, "sq_survey.npz"))["final"]
    rel = lambda a, b: np.max(np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b.astype(np.float64)), 1e-3), axis=-1)
    for lv in LEVELS:
        d = rel(data[f"final_{lv}"], ref)
        print(f"{lv}: identical to sq_survey.npz on {int((d == 0).sum())} of {len(d)}; within 1e-4 on {int((d <= 1e-4).sum())}; max {d.max():.2e}")
    np.savez_compressed(os.path.join(HERE, "sq_survey_dispatch.npz"), **data)


if __name__ == "__main__":
    main()
