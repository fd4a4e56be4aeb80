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


def dump_pond(name):
    """W/MistralWaterLib.cginc:154-180 in its three modes on a jittered 12x12 lattice (oracle/pond_oracle.c)."""
    M = workloads.POND_MATERIAL
    common = dict(amplitude=M["_Amplitude"], frequency=M["_Frequency"], speed=M["_Speed"], steepness=M["_Steepness"],
                  wspeed=M["_WSpeed"], dir_ab=M["_WDirectionAB"], dir_cd=M["_WDirectionCD"])
    pos = workloads.pond_lattice(12, y=0.25, seed=5)
    out = {"pos": pos, "t": np.float32(3.25)}
    for tag, mode, smoothing, amp in (("wave", 0, 0.35, M["_Amplitude"]), ("gerstner", 1, 1.0, M["_Amplitude"]),
                                      ("level_one", 2, 1.0, 0.1)):
        v, n = O.pond_displace_f64(O.pond_params(mode, smoothing=smoothing, **{**common, "amplitude": amp}), pos, 3.25)
        out[tag + "_vertices"], out[tag + "_normals"] = v, n
        out[tag + "_mode_smoothing_amplitude"] = np.array([mode, smoothing, amp], np.float64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def dump_renderer(name):
    """Two OceanRenderer frames at resolution 8 (64^2 textures) and the material's vertex stage on the 8x8 mesh."""
    rp = O.RendererParams(resolution=8, length=434.48 * 64 / 1024.0, wind_x=14.45, wind_y=12.0, amplitude=0.41,
                          choppiness=0.46, gravity=9.81, mult=1.5)
    init4 = O.renderer_initial_spectrum(rp, 5)
    ph = np.zeros((rp.M, rp.M), np.float32)
    for dt in (0.016, 0.3):
        H, D, Nn, W = O.renderer_textures_f64(rp, init4, ph, dt)
    v, n, c = O.renderer_mesh_vertex_stage_f64(rp, 0.75, H[..., 0], D[..., [0, 2]], Nn[..., :3], W[..., 0])
    f32 = lambda a: np.ascontiguousarray(a, np.float32)  # noqa: E731
    np.savez_compressed(os.path.join(HERE, name + ".npz"), init4=init4, dts=np.array([0.016, 0.3], np.float32), unit_width=0.75,
                        params=np.array([rp.resolution, rp.length, rp.wind_x, rp.wind_y, rp.amplitude, rp.choppiness,
                                         rp.gravity, rp.mult], np.float64),
                        height_rgba=f32(H), disp_rgba=f32(D), normal_rgba=f32(Nn), white_rgba=f32(W), phase=ph,
                        mesh_vertices=v, mesh_normals=n, mesh_colors=c)


def dump_config1_sample(name):
    """BASELINE configs[0] / SURVEY 8d config 1 (the reference's own CPU-runnable case): 256 x 256 grid, unit_width 1, length 256,
    wind (5, 3), amplitude 0.01, choppiness 1, t = 1.0, seed 1 -- the literal float32 O(N^4) FFTMesh.Displacement loop
    (S/FFTMesh.cs:192-220) on 192 seeded vertices (the full grid is ~4 minutes on one core), and the f64 values beside it.  The
    spectrum is NOT stored (1 MB): it is regenerated from the seed by the documented counter RNG."""
    p = O.Params(N=256, unit_width=1.0, length=256.0, wind_x=5.0, wind_y=3.0, amplitude=0.01, choppiness=1.0, gravity=9.81)
    h0, h0c = O.generate_spectrum(p, 1)
    idx = np.sort(np.random.default_rng(256).choice(256 * 256, 192, replace=False)).astype(np.int32)
    hd, nor = O.displacement_subset_f32(p, h0, h0c, 1.0, idx)       # (d.x, h, d.z) and the unit normal per sampled vertex
    vf, nf, cf, hds = O.eval_fft_f64(p, h0, h0c, 1.0, return_hds=True)
    rest = O.rest_mesh(p)[0]
    params = np.array([p.N, p.unit_width, p.length, p.wind_x, p.wind_y, p.amplitude, p.choppiness, p.gravity], np.float64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), params=params, seed=1, t=np.float32(1.0), vertex_idx=idx,
                        literal_hd=hd, literal_normals=nor, f64_height=vf[idx, 1], f64_disp_x=hds[idx, 0], f64_disp_z=hds[idx, 1],
                        f64_normals=nf[idx], f64_white=cf[idx, 0], rest=rest[idx].astype(np.float32),
                        h0_checksum=np.array([np.float64(h0.astype(np.float64).sum()), np.float64(np.abs(h0).astype(np.float64).sum())]))


if __name__ == "__main__":
    dump_pond("pond_modes_t3p25")
    dump_renderer("renderer_res8_frame2")
    dump_config1_sample("fftmesh_config1_256_literal_sample")
    dump("fftmesh_n16_t1p5", workloads.fftmesh_params(16, choppiness=1.0), 42, 1.5)
    dump("fftmesh_shipped_n12_t2", workloads.shipped_fftmesh_scene(), 7, 2.0)
    print("ok")
