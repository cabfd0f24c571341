#!/usr/bin/env python3
"""Extracts the eight truth tables of crates/flute_benchmark/to_eval/sbox8.lut (a benchmark INPUT file of the reference:
one grouped LUT, 8 inputs addr[0..7], 8 outputs dout[0..7], each `8 11111111 0x<64 hex digits> dout[k]`) into
tests/golden/sbox8_lut.json. Run in the build container only (the GPU box has no /root/reference)."""
import json
import re
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/crates/flute_benchmark/to_eval/sbox8.lut"
txt = open(src).read()
line = [l for l in txt.splitlines() if l.startswith("LUT ")][0]
tabs = re.findall(r"8 11111111 0x([0-9a-f]{64}) (dout\[\d\])", line)
assert len(tabs) == 8
out = {"source": "crates/flute_benchmark/to_eval/sbox8.lut (#INPUTS 8: addr[0..7], #OUTPUTS 8: dout[0..7], 1 grouped LUT)",
       "inputs": ["addr[%d]" % i for i in range(8)],
       "luts": [{"out_wire": w, "wire_mask": "11111111", "table_hex": h} for h, w in tabs]}
json.dump(out, open(__file__.replace("make_sbox8_lut.py", "sbox8_lut.json"), "w"), indent=1)
print("ok", [w for _, w in tabs])
