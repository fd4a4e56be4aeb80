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


def assert_normal_white(n, w, N_ref, W_ref, length_normal, Dr, Dg, Db, H, got=None, rel=3e-6, tag="", return_bounds=False):
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
    r = _check(n, w, N_ref, W_ref, bn, bw, tag)
    return (r, bn, bw, delta) if return_bounds else r


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


# ---- the ocean material's vertex stage (W/TestOcean.shader:61-79, or_mesh_vertex) ----------------------------------------------
# A mesh vertex samples the four textures bilinearly (tex2Dlod, clamp) and normalises the sampled normal.  Its error budget is
# therefore that of the texels it touches: a weighted mean of <= 4 texel errors, amplified by 1/|s| in the normalisation
# (s = the interpolated, not yet normalised normal: short where neighbouring texel normals disagree), plus the float32
# rounding of the stage itself.  No quantiles: every vertex is held to the bound of its own taps.
def _mesh_axis(M, res):
    """Tap indices and weight of the res mesh lines on an M-texel axis, as the kernel forms them (uv in f32, S/OceanRenderer.cs:184)."""
    u = np.arange(res, dtype=np.float32) / np.float32(res - 1)
    x = (u * np.float32(M) - np.float32(0.5)).astype(np.float64)
    fl = np.floor(x)
    i0 = fl.astype(np.int64)
    return np.clip(i0, 0, M - 1), np.clip(i0 + 1, 0, M - 1), x - fl


def _mesh_sample(tex, M, res, reduce=None):
    """tex [M, M(, C)] indexed [py, px] sampled at the res x res mesh vertices, vertex (i, j) at index i*res + j (u from i -> px,
    v from j -> py).  reduce=None: bilinear in f64; reduce=np.maximum: the largest of the four taps (error bounds)."""
    x0, x1, wx = _mesh_axis(M, res)
    y0, y1, wy = _mesh_axis(M, res)
    X0, Y0 = np.meshgrid(x0, y0, indexing="ij")
    X1, Y1 = np.meshgrid(x1, y1, indexing="ij")
    a00, a10, a01, a11 = tex[Y0, X0], tex[Y0, X1], tex[Y1, X0], tex[Y1, X1]
    if reduce is not None:
        return reduce(reduce(a00, a10), reduce(a01, a11)).reshape((res * res,) + tex.shape[2:])
    WX, WY = np.meshgrid(wx, wy, indexing="ij")
    if tex.ndim == 3:
        WX, WY = WX[..., None], WY[..., None]
    a0, a1 = a00 + (a10 - a00) * WX, a01 + (a11 - a01) * WX
    return (a0 + (a1 - a0) * WY).reshape((res * res,) + tex.shape[2:])


def _dilate(a):
    """3 x 3 maximum: a tap index that differs by one between the f32 kernel and this f64 restatement stays covered."""
    out = a.copy()
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            out = np.maximum(out, _shift(a, dx, dy))
    return out


ULP = 2.0 ** -24


def _tap_magnitude(tex_mag, M, res):
    """Largest |height| / |displacement| texel among each vertex's taps: the bilinear taps are float32 sums of those."""
    return _mesh_sample(_dilate(np.asarray(tex_mag, np.float64)), M, res, reduce=np.maximum)[:, None]


def assert_mesh_stage_alone(mesh_ref, got, N_tex, res, tex_mag, tag=""):
    """THE STAGE ALONE: device mesh (v, n, c) vs the oracle's f64 vertex stage evaluated on the DEVICE's own textures
    (mesh_ref): float32 rounding of the bilinear taps, of the /8 and of the normalisation only."""
    (V, Nn, Cc), (v, n, c) = mesh_ref, got
    M = N_tex.shape[0]
    s = np.maximum(np.linalg.norm(_mesh_sample(np.asarray(N_tex, np.float64), M, res), axis=-1), 1e-30)
    bv = 8 * ULP * (np.abs(V) + _tap_magnitude(tex_mag, M, res) / 8.0 + 1e-3)   # rest + lerp(texels) / 8: ulps of the TAPS, not of the sum
    assert (np.abs(v - V) <= bv).all(), f"{tag} mesh vertices (stage): worst ratio {float((np.abs(v - V) / bv).max()):.2f}"
    bn = np.minimum(16 * ULP / s + 4 * ULP, 2.0)
    en = np.abs(n - Nn).max(-1)
    assert (en <= bn).all(), f"{tag} mesh normals (stage): {int((en > bn).sum())} vertices above their bound, worst ratio {float((en / bn).max()):.2f}"
    bc = 8 * ULP * np.maximum(1.0, np.abs(Cc))
    assert (np.abs(c - Cc) <= bc).all(), f"{tag} mesh colours (stage): worst ratio {float((np.abs(c - Cc) / bc).max()):.2f}"
    return float((en / bn).max()), float(np.median(bn))


def assert_mesh_end_to_end(mesh_ref, got, N_ref_tex, bn_tex, bw_tex, delta, res, tex_mag, tag=""):
    """END TO END: device mesh vs the oracle's mesh from the oracle's own textures.  bn_tex / bw_tex: the per-texel bounds of
    the normal / whitecap textures (normal_white_bounds), delta: the absolute error of the height / displacement textures."""
    (V, Nn, Cc), (v, n, c) = mesh_ref, got
    M = N_ref_tex.shape[0]
    s = np.maximum(np.linalg.norm(_mesh_sample(np.asarray(N_ref_tex, np.float64), M, res), axis=-1), 1e-30)
    bv = delta / 8.0 + 8 * ULP * (np.abs(V) + _tap_magnitude(tex_mag, M, res) / 8.0 + 1e-3)
    assert (np.abs(v - V) <= bv).all(), f"{tag} mesh vertices: worst ratio {float((np.abs(v - V) / bv).max()):.2f}"
    tn = _mesh_sample(_dilate(bn_tex), M, res, reduce=np.maximum)
    bn = np.minimum(2.0 * np.sqrt(3.0) * tn / s + 16 * ULP / s + 4 * ULP, 2.0)
    en = np.abs(n - Nn).max(-1)
    assert (en <= bn).all(), f"{tag} mesh normals: {int((en > bn).sum())} vertices above their bound, worst ratio {float((en / bn).max()):.2f}"
    bc = _mesh_sample(_dilate(bw_tex), M, res, reduce=np.maximum) + 8 * ULP
    ec = np.abs(c - Cc)
    assert (ec <= bc).all(), f"{tag} mesh colours: {int((ec > bc).sum())} vertices above their bound, worst ratio {float((ec / bc).max()):.2f}"
    return float((en / bn).max()), float((ec / bc).max()), float(np.median(bn)), float(np.median(bc))
