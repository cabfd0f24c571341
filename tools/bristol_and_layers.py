#!/usr/bin/env python3
"""AND-layer profile of a Bristol-format circuit: how many AND gates sit at each multiplicative depth (the layers a GMW
executor evaluates one network round at a time; XOR / INV are free and take the depth of their deepest input, executor.rs
/ circuit/base_circuit.rs layer iteration). Run in the build container on the reference's file; only the list of counts
is committed (tests/golden/aes128_bristol_and_layers.json) -- numbers, not the netlist.

  python3 tools/bristol_and_layers.py /root/reference/crates/flute_benchmark/Baseline/bristol/aes_128.bristol \\
      > tests/golden/aes128_bristol_and_layers.json"""
import json
import sys


def profile(path):
    with open(path) as f:
        toks = [ln.split() for ln in f if ln.strip()]
    n_gates, n_wires = int(toks[0][0]), int(toks[0][1])
    depth = [0] * n_wires
    layers = {}
    kinds = {}
    for t in toks[2:]:
        nin, nout = int(t[0]), int(t[1])
        ins = [int(x) for x in t[2:2 + nin]]
        out = int(t[2 + nin])
        op = t[-1]
        kinds[op] = kinds.get(op, 0) + 1
        d = max(depth[i] for i in ins)
        if op == "AND":
            layers[d] = layers.get(d, 0) + 1
            d += 1
        depth[out] = d
    counts = [layers.get(i, 0) for i in range(max(layers) + 1)]
    return {"source": "crates/flute_benchmark/Baseline/bristol/aes_128.bristol", "gates": n_gates, "wires": n_wires,
            "gate_kinds": kinds, "and_gates": sum(counts), "and_depth": len(counts), "and_gates_per_layer": counts}


if __name__ == "__main__":
    json.dump(profile(sys.argv[1]), sys.stdout, indent=1)
    print()
