"""Per-texel error bounds of the OceanRenderer normal and whitecap passes (F/OceanNormal.shader:39-56,
F/WhiteCap.shader:33-45), from f64 textures.

The normal is normalize(c), c = right x top + top x left + left x bottom + bottom x right of four edge vectors built from
differences of neighbouring texels (with the shader's `center = D.rgb` quirk: center.y is Im hx, not the height, so the
y components of all four vectors carry a common offset of the size of the swell).  In exact arithmetic that offset cancels
(a closed fan: c = sum of P_i x P_(i+1), independent of the centre), in float32 it does not: |c| stays ~ 4 ts^2 while the
products inside it grow with the swell, and 1/|c| amplifies both the error of the transformed textures and the rounding of
the pass itself.  A fixed tolerance either fails there or gives the whole image a free pass; here every texel gets its own
bound from its own condition, component by component:

    c_k = sum over the 8 products a_p * b_q that enter component k of the four cross products
    stage:       |dc_k| <= K * 2^-24 * sum |a_p * b_q| + 2 d32 * sum (|a_p| + |b_q|)   (float32 rounding of the pass on FIXED
                                                                          textures; d32 = 2 ulp of the texel magnitude: the
                                                                          edge vectors are float32 differences of texels)
    end to end:  |dc_k| <= 2 delta * sum (|a_p| + |b_q|)  +  the above   (each vector component carries <= 2 delta of
                                                                          texture error: its own texel's and the centre's)
    |n - n_ref| <= |dc| / |c|
    |w - w_ref| <= 1.5 * ( dJ + 0.3 sqrt(2) |n - n_ref| )                 smoothstep'(t) <= 1.5;  dJ from the +-8-texel
                                                                          central differences (delta / 8 per derivative)
"""
import numpy as np


def _shift(a, dx, dy):
    """a[py + dy, px + dx] with clamp addressing (Unity's RenderTexture default, as the oracle)."""
    M = a.shape[0]
    iy = np.clip(np.arange(M) + dy, 0, M - 1)
    ix = np.clip(np.arange(M) + dx, 0, M - 1)
    return a[np.ix_(iy, ix)]


def normal_white_bounds(length_normal, Dr, Dg, Db, H, delta, K=16.0, safety=2.0):
    """Dr, Dg, Db: displacementTexture.r/.g/.b; H: heightTexture.r (f64, [M, M] indexed [py, px]).  delta: absolute error
    of those textures (0 for the stage bound).  Returns per-texel bounds (bound_n, bound_w)."""
    M = H.shape[0]
    ts = length_normal / M

    def edge(dx, dy, ox, oz):
        return (ox + _shift(Dr, dx, dy) - Dr, _shift(H, dx, dy) - Dg, oz + _shift(Db, dx, dy) - Db)
    r, l = edge(+1, 0, ts, 0.0), edge(-1, 0, -ts, 0.0)
    t, b = edge(0, -1, 0.0, -ts), edge(0, +1, 0.0, ts)
    c = [0.0, 0.0, 0.0]
    S = [0.0, 0.0, 0.0]      # sum |a_p b_q| per component
    T = [0.0, 0.0, 0.0]      # sum |a_p| + |b_q| per component
    for a_, b_ in ((r, t), (t, l), (l, b), (b, r)):
        for k, (p1, q1, p2, q2) in enumerate(((1, 2, 2, 1), (2, 0, 0, 2), (0, 1, 1, 0))):
            c[k] = c[k] + a_[p1] * b_[q1] - a_[p2] * b_[q2]
            S[k] = S[k] + np.abs(a_[p1] * b_[q1]) + np.abs(a_[p2] * b_[q2])
            T[k] = T[k] + np.abs(a_[p1]) + np.abs(b_[q1]) + np.abs(a_[p2]) + np.abs(b_[q2])
    cn = np.maximum(np.sqrt(c[0] ** 2 + c[1] ** 2 + c[2] ** 2), 1e-300)
    # the edge vectors themselves are float32 differences of texels of magnitude m: (ts + D_i) - D_0 carries up to ~2 ulp(m)
    m = np.maximum(np.maximum(np.abs(Dr), np.abs(Dg)), np.maximum(np.abs(Db), np.abs(H))) + ts
    for v in (r, l, t, b):
        for comp in v:
            m = np.maximum(m, np.abs(comp))
    d32 = 2.0 * 2.0 ** -24 * m
    dc = [K * 2.0 ** -24 * S[k] + 2.0 * (delta + d32) * T[k] for k in range(3)]
    bound_n = safety * np.sqrt(dc[0] ** 2 + dc[1] ** 2 + dc[2] ** 2) / cn + 2.0 ** -22
    bound_n = np.minimum(bound_n, 2.0)                                   # two unit vectors are never further apart
    ax = -0.5 * (_shift(Dr, -8, 0) - _shift(Dr, +8, 0)) / 8.0            # dDdx.x   (F/WhiteCap.shader:37)
    ay = -0.5 * (_shift(Db, -8, 0) - _shift(Db, +8, 0)) / 8.0            # dDdx.y
    bx = -0.5 * (_shift(Dr, 0, -8) - _shift(Dr, 0, +8)) / 8.0            # dDdy.x   (:36)
    by = -0.5 * (_shift(Db, 0, -8) - _shift(Db, 0, +8)) / 8.0            # dDdy.y
    scale = np.abs(1 + ax) + np.abs(1 + by) + np.abs(ay) + np.abs(bx)
    dJ = (delta / 8.0 + K * 2.0 ** -24 * (np.abs(Dr) + np.abs(Db) + 1.0) / 8.0) * scale + K * 2.0 ** -24 * scale ** 2
    bound_w = 1.5 * (safety * dJ + 0.3 * np.sqrt(2.0) * bound_n) + 2.0 ** -21
    return bound_n, bound_w


def _check(n, w, N_ref, W_ref, bn, bw, tag):
    en = np.abs(np.asarray(n, np.float64) - N_ref).max(-1)
    ew = np.abs(np.asarray(w, np.float64) - W_ref)
    bad_n, bad_w = en > bn, ew > bw
    assert not bad_n.any(), f"{tag} normals: {int(bad_n.sum())} texels above their bound, worst ratio {float((en / bn).max()):.2f}"
    assert not bad_w.any(), f"{tag} whitecap: {int(bad_w.sum())} texels above their bound, worst ratio {float((ew / bw).max()):.2f}"
    return float((en / bn).max()), float((ew / bw).max()), float(np.median(bn)), float(np.median(bw))


def assert_normal_white(n, w, N_ref, W_ref, length_normal, Dr, Dg, Db, H, got=None, rel=3e-6, tag=""):
    """END TO END: device normal [M,M,3] / whitecap [M,M] vs the oracle's (computed from the oracle's own textures), each texel
    against its own bound.  `got` = the device's (Dr, Dg, Db, H): delta is then the error actually MEASURED on the transformed
    textures (itself held to `rel` of their maximum); without it delta is the tolerance `rel` itself."""
    scale = max(float(np.abs(Dr).max()), float(np.abs(Db).max()), float(np.abs(H).max()), float(np.abs(Dg).max()))
    delta = rel * scale
    if got is not None:
        measured = max(float(np.abs(np.asarray(g_, np.float64) - r_).max()) for g_, r_ in zip(got, (Dr, Dg, Db, H)))
        assert measured <= delta, f"{tag} textures: {measured:.3e} > {delta:.3e}"
        delta = max(measured, 2.0 ** -24 * scale)
    bn, bw = normal_white_bounds(length_normal, Dr, Dg, Db, H, delta)
    return _check(n, w, N_ref, W_ref, bn, bw, tag)


def assert_normal_white_stage(oracle, rp_normal, Ht, Dt, Nt, Wt, tag=""):
    """THE STAGE ALONE: the device's normal / whitecap textures vs the oracle's f64 evaluation of the two shaders on the
    DEVICE's own height / displacement textures (RGBA targets, [M,M,4]) -- no input error, only the float32 rounding of the
    pass, bounded per texel by K ulp of the products that enter it."""
    import ctypes as C
    M = Ht.shape[0]
    dtex = np.ascontiguousarray(Dt, np.float64)
    hre = np.ascontiguousarray(Ht[..., 0], np.float64)
    N_ref, W_ref = np.empty((M, M, 3), np.float64), np.empty((M, M), np.float64)
    oracle.lib().orr_normal_white_f64(C.byref(rp_normal.c()), dtex.ctypes.data_as(C.c_void_p), hre.ctypes.data_as(C.c_void_p),
                                      N_ref.ctypes.data_as(C.c_void_p), W_ref.ctypes.data_as(C.c_void_p))
    bn, bw = normal_white_bounds(rp_normal.length, dtex[..., 0], dtex[..., 1], dtex[..., 2], hre, 0.0)
    return _check(Nt[..., :3], Wt[..., 0], N_ref, W_ref, bn, bw, tag + " (stage)")
