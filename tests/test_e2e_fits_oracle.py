"""CPU: the two fit passes of the whole-chain fixture on the reference's own tracks, object by object (tests/e2e_lib.py),
with the oracle as the fitter.  tests/test_e2e.py asserts the same table through the GPU kernel (bit-identical to the oracle)."""
import numpy as np

import e2e_lib
from test_multi_view_host import OracleFitter


def test_fits_on_the_reference_tracks_object_by_object(golden, oracle, capsys):
    from odam_amd import multi_view
    z = golden("e2e.npz")
    seq, K, T_wcs, P_cws = e2e_lib.sequence_geometry()

    def fit_tracks(tracks):
        return multi_view.optim_process(tracks, seq["img_names"], T_wcs, P_cws, e2e_lib.SEQ["h"], e2e_lib.SEQ["w"], K,
                                        "super_quadric", True, 200, 10, fitter=OracleFitter(oracle), return_params=True)
    rows = []
    for w in (1, 2):
        rows += e2e_lib.check_fits_on_reference_tracks(z, w, fit_tracks)
    with capsys.disabled():
        print("\n" + e2e_lib.table(rows))
    assert sum(r[3] <= 1e-4 for r in rows) == 25 and len(rows) == 33
