"""The stages of the Frobenius-pruned additive FFT inside sot_qc_encode against the executable model
(flute_amd/csrc/fafft_model.py). The expected words were produced by that model (tests/golden/make_fafft_stages.py)
for one random row at m = 21; the model itself is checked against the oracle's GF(2)[x] product in
tests/test_fafft_model.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "fafft_stages_m21.npz")


def test_transform_stages_match_model():
    import flute_amd as fa
    g = np.load(GOLD)
    eng = fa.Engine(1 << 19)  # P just above 2^19 ... capacity 2^21 bits
    try:
        assert eng.qc_log2_capacity() == 21
        for stage, key in enumerate(("conv", "r8", "r12", "r16", "r21")):
            got = eng.qc_transform_row(g["input_bits"], stage)
            want = g[key]
            if stage == 0:
                got, want = got[: 1 << 15], want[: 1 << 15]
            bad = np.flatnonzero(got != want)
            assert bad.size == 0, "stage %d (%s): %d words differ, first at %d" % (stage, key, bad.size, bad[0])
    finally:
        eng.close()
