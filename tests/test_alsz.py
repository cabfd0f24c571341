"""ALSZ13 OT extension: the oracle restatement satisfies the reference's `ot_ext` test
(crates/zappot/src/ot_ext/mod.rs:280-307) and its generalisation to random choices. CPU only."""
import numpy as np
import pytest

from oracle import alsz_oracle as az


@pytest.mark.parametrize("count,zero_choices", [(1000, True), (1000, False), (8, False), (136, False)])
def test_oracle_ot_ext_property(count, zero_choices):
    rng = np.random.default_rng(42)
    pairs, chosen, bc = az.fake_base(rng)
    bits = np.zeros(count, np.uint8) if zero_choices else rng.integers(0, 2, count, dtype=np.uint8)
    u, recv = az.receive_random(pairs, np.packbits(bits, bitorder="little").tobytes(), count)
    sent = az.send_random(chosen, bc.tobytes(), u, count)
    assert np.array_equal(recv, sent[np.arange(count), bits])  # r == s[choice]
    assert not np.any(np.all(recv == sent[np.arange(count), 1 - bits], axis=1))
    assert len({recv[i].tobytes() for i in range(count)}) == count
