#!/usr/bin/env python3
"""Writes tests/golden/synth_b.json: sha256 of a few chunks of the synthetic content "B" (Kafka v2 record batches, tsxform/synth.py), so
that the generator - whose chunks bench.py's value_B leg and the device parity test are built on - cannot drift unnoticed (numpy version,
a refactoring).  usage: python tests/golden/make_synth_b.py"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tsxform import synth  # noqa: E402

CASES = [(41, 2, 0, 60000), (41, 2, 1, 320000), (1000, 0, 0, 4 << 20), (7, 3, 5, 131072 + 29989)]


def main():
    out = []
    for seed, seg, chunk, size in CASES:
        c = synth.gen_chunk("B", seed, seg, chunk, size)
        out.append({"seed": seed, "segment": seg, "chunk": chunk, "size": size, "sha256": hashlib.sha256(c.tobytes()).hexdigest(),
                    "record_batches": len(synth.record_batches_of(c))})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_b.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path, len(out))


if __name__ == "__main__":
    main()
