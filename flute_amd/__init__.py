"""flute_amd -- MI355X-native silent-OT correlation engine for FLUTE (host-side mirror).

The compute path is libflute_sot.so (hand-written HIP for gfx950, C ABI in include/flute_sot.h).
This package mirrors the reference's interface for the path (zappot silent_ot Sender/Receiver,
gmw SilentMtProvider) on top of that ABI. There is no CPU fallback: without the library or a
gfx950 device every compute call raises.
"""
from .lib import SotError, load_library, device_count  # noqa: F401
from .silent_ot import (QuasiCyclicConf, Engine, ShardedEngine, Sender, Receiver, MultType,  # noqa: F401
                        sample_base_choice_bits, fake_base)
from .mul_triple import MulTriples, SilentMtProvider  # noqa: F401
from .lut_eval import LutTemplate, online_delta_shares, online_delta_shares_dev  # noqa: F401
from .aes_lut import Aes128Flute  # noqa: F401
from .lut_setup import (SetupCircuit, build_setup_circuit, replicate_setup_circuit, AndCircuit,  # noqa: F401
                        aes128_lut_gate_parents, aes128_setup_circuit, gmw_and_layers, AES128_SBOXES,
                        AES128_LUT_SETUP_MTS, AES128_BRISTOL_AND_GATES)
from . import wire  # noqa: F401
