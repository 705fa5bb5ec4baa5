"""Final-parameter parity of the super-quadric fit as a measurement (see sq_survey_lib.py): the CPU oracle here, the
gfx950 kernel under -m gpu (it must also equal the oracle bit for bit, so both see the same numbers)."""
import os

import numpy as np
import pytest

import sq_survey_lib as S
from conftest import GOLDEN


def _oracle_finals(z, oracle):
    return np.stack([oracle.fit(z[f"c{i}_p0"], z[f"c{i}_P"], z[f"c{i}_tgt"], z[f"c{i}_mask"], int(z[f"c{i}_cls"]))[0]
                     for i in range(int(z["n_cases"]))])


def test_oracle_vs_reference_spread(oracle, golden):
    z = golden("sq_survey.npz")
    rows = S.survey_rows(z, _oracle_finals(z, oracle), oracle, dispatch=golden("sq_survey_dispatch.npz"))
    S.check(rows)
    # the reference's own answer depends on the instruction set ATen dispatches its CPU kernels to at least as much as ours
    # differs from it: the stored (AVX-512) run and the scalar-kernel run agree to 1e-4 on no more than about half of the problems
    assert sum(r["d"]["avx512"] == 0 for r in rows) == len(rows)
    assert sum(r["d"]["default"] <= 1e-4 for r in rows) <= 0.6 * len(rows)
    # what the first divergence is made of (tests/golden/make_sq_divergence.py: the reference's own torch operations substituted
    # into the oracle part by part): the oracle's deviation there is the one this test measures, the projection is already exact,
    # transcendentals (MKL VML: not restatable) and autograd's summation order each change the gradient's last bits in every problem
    div = golden("sq_divergence.npz")
    assert np.allclose(div["final_dev"][:, 0], [r["e"] for r in rows], rtol=1e-6, atol=0)      # H0 of that measurement = this oracle
    cause = [str(c) for c in div["cause"]]
    assert all("B" not in c for c in cause) and all("A" in c and "C" in c for c in cause)
    assert (div["first_param_step"][:, 0] >= 1).all() and (div["first_param_step"][:, 0] <= 10).all()
    assert (div["final_dev"][:, 1] <= 1e-4).sum() < 0.5 * len(rows)      # the reference's own transcendentals alone do not restore 1e-4
    # the committed table is this very measurement
    path = os.path.join(GOLDEN, "sq_survey_table.txt")
    if os.environ.get("ODAM_WRITE_SURVEY_TABLE"):
        open(path, "w").write(S.table(rows, div))
    assert open(path).read() == S.table(rows, div)


@pytest.mark.gpu
def test_gpu_vs_reference_spread(oracle, golden):
    from odam_amd import sq
    z = golden("sq_survey.npz")
    n = int(z["n_cases"])
    f = sq.SqFitter("cuda:0", 200)
    vc = [len(z[f"c{i}_tgt"]) for i in range(n)]
    out = f.fit(np.stack([z[f"c{i}_p0"] for i in range(n)]), [int(z[f"c{i}_cls"]) for i in range(n)], vc,
                np.concatenate([z[f"c{i}_P"] for i in range(n)]), np.concatenate([z[f"c{i}_tgt"] for i in range(n)]),
                np.concatenate([z[f"c{i}_mask"] for i in range(n)]))["params"].cpu().numpy()
    f.close()
    assert np.array_equal(out.view(np.uint32), _oracle_finals(z, oracle).view(np.uint32))
    S.check(S.survey_rows(z, out, oracle))
