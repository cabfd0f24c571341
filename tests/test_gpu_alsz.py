"""GPU parity of the ALSZ13 OT extension (sot_alsz_send / sot_alsz_receive) against oracle/alsz_oracle.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("count", [8, 136, 1000, 65536, 100008])
def test_alsz_matches_oracle(count):
    import flute_amd
    from flute_amd import ot_ext
    from oracle import alsz_oracle as az
    if flute_amd.device_count() < 1:
        pytest.fail("no gfx950 device")
    rng = np.random.default_rng(count)
    pairs, chosen, bc = az.fake_base(rng)
    bits = rng.integers(0, 2, count, dtype=np.uint8)
    cb = np.packbits(bits, bitorder="little").tobytes()
    u, recv = ot_ext.Receiver(pairs).receive_random(cb, count)
    ou, orecv = az.receive_random(pairs, cb, count)
    assert np.array_equal(u, ou), "u rows differ from the oracle"
    assert np.array_equal(recv, orecv), "receiver messages differ from the oracle"
    sent = ot_ext.Sender(chosen, bc.tobytes()).send_random(u, count)
    assert np.array_equal(sent, az.send_random(chosen, bc.tobytes(), ou, count)), "sender messages differ"
    assert np.array_equal(recv, sent[np.arange(count), bits])  # ot_ext/mod.rs:303-305, any choices


def test_alsz_full_size_property():
    """2^24 extensions (the criterion bench size, zappot/benches): recv == sent[choice], size-independent check"""
    from flute_amd import ot_ext
    from oracle import alsz_oracle as az
    count = 1 << 24
    rng = np.random.default_rng(3)
    pairs, chosen, bc = az.fake_base(rng)
    bits = rng.integers(0, 2, count, dtype=np.uint8)
    u, recv = ot_ext.Receiver(pairs).receive_random(np.packbits(bits, bitorder="little").tobytes(), count)
    sent = ot_ext.Sender(chosen, bc.tobytes()).send_random(u, count)
    assert np.array_equal(recv, sent[np.arange(count), bits])
    assert not np.array_equal(recv[:1000], sent[:1000, 0]) or bits[:1000].sum() == 0
    with pytest.raises(Exception):
        ot_ext.Receiver(pairs).receive_random(b"\0" * 2, 12)
