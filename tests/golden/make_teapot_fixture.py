#!/usr/bin/env python3
"""Writes tests/golden/teapot_mesh.npz: the flattened arrays of the reference's examples/teapot.obj (the Utah teapot, public
domain; mesh DATA only -- 1202 positions, 2256 triangles; the OBJ has no `vt`, so uv = (x, y) as src/wavefront.rs:92-95 assigns).

/root/reference does not exist on the GPU box; with this fixture configuration C2 (BASELINE.json configs[1], examples/obj.rs)
runs there on the real geometry instead of a stand-in of equal counts.  Run it where the reference checkout is mounted:

    python tests/golden/make_teapot_fixture.py [/root/reference/examples/teapot.obj]

The file is parsed by the ORACLE's OBJ reader (oracle/rusterix_oracle.cpp, restating src/wavefront.rs:34-102), i.e. the arrays are
what `Batch3D::from_obj` produces; tests/test_host_and_abi.py checks that the product's reader gives the same bytes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.oracle_api import load_oracle  # noqa: E402


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/examples/teapot.obj"
    orc = load_oracle()
    v, idx, uv = orc.Batch3D.from_obj(open(src).read()).geometry()[:3]
    assert v.shape == (1202, 4) and idx.shape == (2256, 3), (v.shape, idx.shape)
    assert np.array_equal(uv, v[:, :2])
    out = os.path.join(ROOT, "tests", "golden", "teapot_mesh.npz")
    np.savez_compressed(out, positions=np.ascontiguousarray(v[:, :3], np.float32), indices=idx.astype(np.uint16))
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
