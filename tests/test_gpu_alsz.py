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


def test_alsz_context_reused_across_calls_matches_oracle():
    """device-resident contexts (sot_alsz_ctx_*): three calls in a row on the same pair of contexts, sizes going down and
    up again (stale buffer contents must not leak into a smaller call), each equal to the oracle"""
    from flute_amd import ot_ext
    from oracle import alsz_oracle as az
    rng = np.random.default_rng(11)
    pairs, chosen, bc = az.fake_base(rng)
    R = ot_ext.Receiver(pairs, max_count=70000)
    S = ot_ext.Sender(chosen, bc.tobytes(), max_count=70000)
    for count in (65536, 1000, 8, 69992):
        bits = rng.integers(0, 2, count, dtype=np.uint8)
        cb = np.packbits(bits, bitorder="little").tobytes()
        u, recv = R.receive_random(cb, count)
        ou, orecv = az.receive_random(pairs, cb, count)
        assert np.array_equal(u, ou) and np.array_equal(recv, orecv), count
        sent = S.send_random(u, count)
        assert np.array_equal(sent, az.send_random(chosen, bc.tobytes(), ou, count)), count
    with pytest.raises(Exception):
        R.receive_random(b"\0" * 10000, 70008)  # beyond max_count


def test_alsz_device_resident_handoff():
    """receiver results left in HBM, the sender reads the u rows device-to-device (in-process two-party run), and the
    downloaded device buffers equal the host-buffer path. NO sync between the two calls on purpose: the contexts run on
    two non-blocking streams, and sot_alsz_ctx_send orders itself behind the receiver context whose u_dev it is handed
    (hipStreamWaitEvent on that context's `done` event, csrc/alsz.hip)"""
    import torch
    from flute_amd import ot_ext
    from oracle import alsz_oracle as az
    count = 100008
    rng = np.random.default_rng(12)
    pairs, chosen, bc = az.fake_base(rng)
    bits = rng.integers(0, 2, count, dtype=np.uint8)
    cbn = np.packbits(bits, bitorder="little")
    R = ot_ext.Receiver(pairs, max_count=count)
    S = ot_ext.Sender(chosen, bc.tobytes(), max_count=count)
    ch_dev = torch.from_numpy(cbn).cuda()
    torch.cuda.synchronize()
    assert R.receive_random(None, count, to_host=False, choices_dev=ch_dev.data_ptr()) == (None, None)
    assert S.send_random(None, count, to_host=False, u_dev=R.ctx.u_dev) is None
    R.ctx.sync()
    S.ctx.sync()
    u_ref, recv_ref = ot_ext.Receiver(pairs).receive_random(cbn.tobytes(), count)
    sent_ref = ot_ext.Sender(chosen, bc.tobytes()).send_random(u_ref, count)

    def dl(ptr, nbytes):
        import ctypes as C
        buf = np.empty(nbytes, dtype=np.uint8)
        hip = C.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(nbytes), 2) == 0
        return buf
    assert np.array_equal(dl(R.ctx.msgs_dev, count * 16).view(np.uint64).reshape(count, 2), recv_ref)
    assert np.array_equal(dl(S.ctx.msgs_dev, count * 32).view(np.uint64).reshape(count, 2, 2), sent_ref)
    assert np.array_equal(recv_ref, sent_ref[np.arange(count), bits])
