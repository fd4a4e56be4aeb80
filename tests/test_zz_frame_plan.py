"""GPU tier, run last: the frame-at-a-time plan of the 1024^2 FFT path (FFTMesh.Update drives ONE step per call,
S/FFTMesh.cs:60-73) against the batched plan, bit for bit.

A single-step enqueue at 1024^2 launches pass 1 with one FIELD per workgroup (grid 257 x 3 instead of 257 workgroups doing
three fields in turn) and pass 2 with one wave per row (VT = 1) instead of two rows per fat wave: more, shorter workgroups
where a step cannot fill the device.  Neither changes the arithmetic of a row or a column, so every output of a step must
be the same bit pattern whichever plan produced it (csrc/mistral_water.hip, MW_LATENCY_PLAN)."""
import numpy as np
import pytest

import workloads

pytestmark = pytest.mark.gpu


def test_single_step_plan_equals_batched_plan_bit_for_bit_1024(mw):
    import torch
    p = workloads.fftmesh_params(1024)
    NN = 1024 * 1024
    times = [0.25 + 0.37 * k for k in range(5)]
    with mw.Ocean(resolution=p.N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude,
                  choppiness=p.choppiness, gravity=p.gravity, seed=3) as o:
        dv = torch.empty((5, NN, 3), dtype=torch.float32, device="cuda")
        dn = torch.empty((5, NN, 3), dtype=torch.float32, device="cuda")
        dw = torch.empty((5, NN), dtype=torch.float32, device="cuda")
        o.evaluate_device(times, dv.data_ptr(), dn.data_ptr(), dw.data_ptr())      # batched plan (5 steps per enqueue)
        o.synchronize()
        bv, bn, bw = dv.cpu().numpy(), dn.cpu().numpy(), dw.cpu().numpy()
        for k in (0, 2, 4):
            v, n, c = o.evaluate(times[k])                                          # frame plan (one step per call)
            assert (bv[k] == v).all(), k
            assert (bn[k] == n).all(), k
            assert (bw[k] == c[:, 0]).all(), k
