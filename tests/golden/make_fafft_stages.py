"""Generates tests/golden/fafft_stages_m21.npz with the executable model (about 8 minutes of pure Python)."""
import os, sys, random, numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "flute_amd", "csrc"))
import fafft_model as fm
random.seed(11)
m = 21
g = fm.Geom(m)
b = random.getrandbits(1 << (m - 1))
Z = fm.int_to_words(fm.convert(b, m - 1, False), g.nw)
out = {"input_bits": fm.int_to_words(b, g.nw), "conv": Z.copy()}
for r in range(1, m + 1):
    if r <= m - 5:
        fm.level_word(Z, g, r, False)
    else:
        fm.level_intra(Z, g, m - r, False)
    if r in (8, 12, 16, m):
        out["r%d" % r] = Z.copy()
    print("level", r, flush=True)
out["input_bits"] = out["input_bits"][: 1 << 15]
np.savez_compressed(os.path.join(HERE, "fafft_stages_m21.npz"), **out)
print("saved")
