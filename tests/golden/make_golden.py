"""Generates the fixtures in tests/golden/ from the oracle (oracle/fftmesh_oracle.c).

The reference (Unity C#) holds no test vectors and cannot be executed in the build container, so
these are NOT outputs of the reference: they are regression pins of the restatement, small enough to
read, produced once by this script:   python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle as O  # noqa: E402
import workloads  # noqa: E402


def dump(name, p, seed, t):
    h0, h0c = O.generate_spectrum(p, seed)
    v, n, c = O.eval_literal_f32(p, h0, h0c, t)
    vd, nd, cd = O.eval_f64(p, h0, h0c, t)
    params = np.array([p.N, p.unit_width, p.length, p.wind_x, p.wind_y, p.amplitude, p.choppiness, p.gravity], np.float64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), params=params, seed=seed, t=np.float32(t), h0=h0, h0c=h0c,
                        vertices=v, normals=n, colors=c, vertices_f64=vd, normals_f64=nd, colors_f64=cd)


if __name__ == "__main__":
    dump("fftmesh_n16_t1p5", workloads.fftmesh_params(16, choppiness=1.0), 42, 1.5)
    dump("fftmesh_shipped_n12_t2", workloads.shipped_fftmesh_scene(), 7, 2.0)
    print("ok")
