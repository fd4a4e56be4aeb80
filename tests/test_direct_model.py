"""CPU tier: the algebra of csrc/direct_kernels.h -- the separable direct sum of S/FFTMesh.cs:199-217 as four REAL float32 matrix
products on zero-padded operands -- restated in numpy float32 with the kernels' operand layouts, against the f64 oracle.

This pins, without a GPU: the four table layouts ([Er ; -Ei], [Ei ; Er], [Er^T | -Ei^T], [Ei^T | Er^T]), which real component of
which field each product delivers, the zero padding to multiples of 64, and that float32 accumulation over K = 2N terms stays
inside the tolerances the GPU tests state (2e-5 of the field scale; 2e-4 on the Inspector-default grid, where the reference's own
float32 phases are off by ~1e-4).  The kernel's thread / LDS / MFMA index maps are exercised on the GPU (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

import workloads

f32 = np.float32


def direct_model(oracle, p, h0, h0c, t):
    N = p.N
    Np = (N + 63) // 64 * 64
    j = np.arange(N)
    k = (f32(2) * f32(3.1415926536) * (j.astype(f32) - f32(N) / f32(2)) / f32(p.length)).astype(f32)        # wave_k, strict f32
    pos = ((j - N // 2).astype(f32) * f32(p.unit_width) + (f32(p.unit_width) / f32(2) if N % 2 == 0 else f32(0))).astype(f32)
    ph = np.outer(k.astype(np.float64), pos.astype(np.float64))                                             # k_direct_tables: f64 phase
    Er, Ei = np.cos(ph).astype(f32), np.sin(ph).astype(f32)
    B1re, B1im = np.zeros((2 * Np, Np), f32), np.zeros((2 * Np, Np), f32)
    A2re, A2im = np.zeros((Np, 2 * Np), f32), np.zeros((Np, 2 * Np), f32)
    B1re[:N, :N], B1re[Np:Np + N, :N] = Er, -Ei
    B1im[:N, :N], B1im[Np:Np + N, :N] = Ei, Er
    A2re[:N, :N], A2re[:N, Np:Np + N] = Er.T, -Ei.T
    A2im[:N, :N], A2im[:N, Np:Np + N] = Ei.T, Er.T
    F = oracle.htilde_fields_f64(p, h0, h0c, t)                # the spectrum kernel's output (f64 here, f32 on the device)
    sp = np.zeros((5, N, N), np.complex128)
    for f in range(5):
        A1 = np.zeros((Np, 2 * Np), f32)
        A1[:N, :N], A1[:N, Np:Np + N] = F[f].real.astype(f32), F[f].imag.astype(f32)
        T = np.concatenate([A1 @ B1re, A1 @ B1im], 0)          # step 1: [Tr ; Ti]
        out = (A2re if f == 0 else A2im) @ T                    # step 2: H = Re, the other four = Im
        assert out.dtype == f32 and not out[N:].any() and not out[:, N:].any()      # the padding stays zero
        sp[f] = out[:N, :N] if f == 0 else 1j * out[:N, :N].astype(np.float64)
    return oracle.assemble_f64(p, sp)


@pytest.mark.parametrize("N,u,L,rel", [(12, 1.0, 12.39, 2e-5), (33, 0.9, 33.0, 2e-5), (65, 0.5, 40.0, 2e-5), (200, 1.0, 212.5, 2e-5),
                                       (50, 1.0, 1.0, 2e-4), (1000, 1.0, 1000.0, 2e-5)])
def test_gemm_form_of_the_direct_sum(oracle, N, u, L, rel):
    if L == 1.0:      # the Inspector defaults, S/FFTMesh.cs:13-19
        p = oracle.Params(N=N, unit_width=u, length=L, wind_x=1.0, wind_y=1.0, amplitude=1.0, choppiness=1.0)
    else:
        p = oracle.Params(N=N, unit_width=u, length=L, wind_x=14.45, wind_y=12.0, amplitude=1.5e-8 * (1024.0 / N) ** 2 * (L / N) ** 2, choppiness=0.46)
    h0, h0c = oracle.generate_spectrum(p, 4)
    rest = oracle.rest_mesh(p)[0]
    for t in (0.5, 16.0):
        v, n, c, _ = direct_model(oracle, p, h0, h0c, t)
        vd, nd, cd, hds = oracle.eval_matmul_f64(p, h0, h0c, t, return_hds=True)
        workloads.assert_parity(v.astype(f32), n.astype(f32), c.astype(f32), vd, nd, cd, rest, rel=rel, tag=f"GEMM model N={N} t={t}", hds=hds)
