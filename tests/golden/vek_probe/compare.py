#!/usr/bin/env python3
"""Which rounding does vek 0.17.2 use?  usage: python3 compare.py probe.txt   (probe.txt = output of `cargo run --release`)

For each of the six operations prints the variants of expected.txt that reproduce ALL 64 probe lines, next to the variant
include/rusterix_vek.hpp implements ("header").  If the header's variant is not among the matching ones, flip it there
(RXR_VEK_FUSED_MATVEC for the matrix products; `normalized` / `lerp` / `dot` by editing the functions -- `dot3` of the device
code, rusterix_amd/csrc/rxr_kernels.hip, follows the header's `dot`), rebuild, re-run the
tests and regenerate tests/golden/ -- every consumer (oracle, host mirror, device kernels) includes that one header."""
import collections
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
exp = collections.defaultdict(dict)  # (case, op) -> variant -> words
for line in open(os.path.join(here, "expected.txt")):
    case, op, variant, *words = line.split()
    exp[(int(case), op)][variant] = words
probe = {}
for line in open(sys.argv[1]):
    case, op, *words = line.split()
    probe[(int(case), op)] = words
for op in ("matvec", "matmat", "normalized", "lerp", "dot", "magnitude"):
    variants = [v for v in exp[(0, op)] if v != "header"]
    matching = [v for v in variants if all(exp[(c, op)][v] == probe[(c, op)] for c in range(64))]
    header = [v for v in variants if all(exp[(c, op)][v] == exp[(c, op)]["header"] for c in range(64))]
    distinct = len({tuple(tuple(exp[(c, op)][v]) for c in range(64)) for v in variants})
    print(f"{op:11s} vek matches: {matching or 'NONE of ' + str(variants)};  rusterix_vek.hpp implements: {header};  "
          f"{'OK' if set(header) & set(matching) else 'MISMATCH -- fix include/rusterix_vek.hpp'}  ({distinct} distinguishable variants)")
