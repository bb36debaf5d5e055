"""Seeded synthetic scan-to-map scenes (BASELINE.md section 3 / SURVEY.md section 8d).

Scene "room+clutter": the 6 planes of a 40 x 30 x 8 m box room, 12 random boxes and 8 vertical
cylinders.  Target = M surface samples (+ sigma noise, roughly one point per voxel), analytic normals
oriented towards a sensor at the origin (mirrors CloudRegistration.cpp:37), GICP covariances
R diag(1,1,1e-3) R^T.  Source = N independent samples within `radius` of the sensor, moved by the
INVERSE of T_true (rpy (0.5, -0.7, 2.0) deg, t (0.15, -0.10, 0.05) m); initial guess = identity.
Pure numpy; used by tests and bench.py (there is no network for real datasets).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

ROOM = (40.0, 30.0, 8.0)
SENSOR_HEIGHT = 1.5


def rpy_to_R(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def true_transform():
    T = np.eye(4)
    T[:3, :3] = rpy_to_R(math.radians(0.5), math.radians(-0.7), math.radians(2.0))
    T[:3, 3] = (0.15, -0.10, 0.05)
    return T


@dataclass
class _Rect:  # axis-aligned rectangle: centre c, half extents along two axes (u, v), normal axis k with sign
    c: np.ndarray
    u: int
    v: int
    hu: float
    hv: float
    k: int
    sgn: float

    @property
    def area(self):
        return 4.0 * self.hu * self.hv


@dataclass
class _Cyl:  # vertical cylinder lateral surface
    cx: float
    cy: float
    r: float
    z0: float
    z1: float

    @property
    def area(self):
        return 2 * math.pi * self.r * (self.z1 - self.z0)


def _box_faces(lo, hi, inward=False):
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    c = 0.5 * (lo + hi)
    h = 0.5 * (hi - lo)
    faces = []
    for k in range(3):
        u, v = [a for a in range(3) if a != k]
        for sgn in (-1.0, 1.0):
            cc = c.copy()
            cc[k] += sgn * h[k]
            faces.append(_Rect(cc, u, v, h[u], h[v], k, -sgn if inward else sgn))
    return faces


def make_primitives(rng):
    X, Y, Z = ROOM
    z0 = -SENSOR_HEIGHT
    prims = _box_faces((-X / 2, -Y / 2, z0), (X / 2, Y / 2, z0 + Z), inward=True)
    for _ in range(12):
        sx, sy, sz = rng.uniform(0.8, 3.5), rng.uniform(0.8, 3.5), rng.uniform(0.6, 3.0)
        while True:
            cx, cy = rng.uniform(-X / 2 + 3, X / 2 - 3), rng.uniform(-Y / 2 + 3, Y / 2 - 3)
            if math.hypot(cx, cy) > 4.0:
                break
        faces = _box_faces((cx - sx / 2, cy - sy / 2, z0), (cx + sx / 2, cy + sy / 2, z0 + sz))
        prims += [f for f in faces if not (f.k == 2 and f.sgn < 0)]  # no bottom face
    for _ in range(8):
        r, hgt = rng.uniform(0.15, 0.6), rng.uniform(2.0, Z)
        while True:
            cx, cy = rng.uniform(-X / 2 + 2, X / 2 - 2), rng.uniform(-Y / 2 + 2, Y / 2 - 2)
            if math.hypot(cx, cy) > 3.0:
                break
        prims.append(_Cyl(cx, cy, r, z0, z0 + hgt))
    return prims


class _PrimTable:
    """Vectorised view of the primitives: per-primitive parameter arrays, so that a batch of (primitive, u, v)
    triples turns into points and analytic normals without a Python loop over primitives."""

    def __init__(self, prims):
        n = len(prims)
        self.is_cyl = np.zeros(n, bool)
        self.c = np.zeros((n, 3))          # rect: centre; cyl: (cx, cy, z0)
        self.ax = np.zeros((n, 3), np.int64)   # rect: (u axis, v axis, normal axis)
        self.h = np.zeros((n, 2))          # rect: half extents (hu, hv); cyl: (r, z1 - z0)
        self.sgn = np.zeros(n)
        self.area = np.array([p.area for p in prims])
        # extent of the (u, v) parameter rectangle in metres: rect 2hu x 2hv, cylinder circumference x height
        self.len_u = np.zeros(n)
        self.len_v = np.zeros(n)
        for i, p in enumerate(prims):
            if isinstance(p, _Rect):
                self.c[i] = p.c
                self.ax[i] = (p.u, p.v, p.k)
                self.h[i] = (p.hu, p.hv)
                self.sgn[i] = p.sgn
                self.len_u[i], self.len_v[i] = 2 * p.hu, 2 * p.hv
            else:
                self.is_cyl[i] = True
                self.c[i] = (p.cx, p.cy, p.z0)
                self.h[i] = (p.r, p.z1 - p.z0)
                self.len_u[i], self.len_v[i] = 2 * math.pi * p.r, p.z1 - p.z0

    def points(self, prim, fu, fv):
        """prim: primitive index per point; fu, fv in [0, 1): position in the primitive's parameter rectangle."""
        n = prim.shape[0]
        P = np.empty((n, 3))
        Nn = np.zeros((n, 3))
        cyl = self.is_cyl[prim]
        r = ~cyl
        if r.any():
            pr = prim[r]
            Pr = self.c[pr].copy()
            rows = np.arange(pr.shape[0])
            Pr[rows, self.ax[pr, 0]] += (2.0 * fu[r] - 1.0) * self.h[pr, 0]
            Pr[rows, self.ax[pr, 1]] += (2.0 * fv[r] - 1.0) * self.h[pr, 1]
            P[r] = Pr
            Nr = np.zeros((pr.shape[0], 3))
            Nr[rows, self.ax[pr, 2]] = self.sgn[pr]
            Nn[r] = Nr
        if cyl.any():
            pc = prim[cyl]
            th = 2.0 * math.pi * fu[cyl]
            ct, st = np.cos(th), np.sin(th)
            P[cyl, 0] = self.c[pc, 0] + self.h[pc, 0] * ct
            P[cyl, 1] = self.c[pc, 1] + self.h[pc, 0] * st
            P[cyl, 2] = self.c[pc, 2] + fv[cyl] * self.h[pc, 1]
            Nn[cyl, 0] = ct
            Nn[cyl, 1] = st
        return P, Nn


_CHUNK = 1 << 20


def _rng(seed, stream, k):
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence([int(seed), int(stream), int(k)])))


def _run_chunks(fn, n_chunks):
    """Chunks are independent (one PCG64 stream each), so the result does not depend on the thread count."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(n_chunks, int(os.environ.get("O3D_SYNTH_THREADS", min(16, os.cpu_count() or 1)))))
    if workers == 1 or n_chunks == 1:
        return [fn(k) for k in range(n_chunks)]
    with ThreadPoolExecutor(workers) as ex:
        return list(ex.map(fn, range(n_chunks)))


def _sample_stratified(tab, n, seed, noise):
    """`n` points, one per cell of a regular (u, v) lattice over every primitive (jittered inside the cell): the
    surface density of a voxel-deduplicated map (param_velodyne_puck16.lua:50) without oversampling and sorting.
    The lattice edge is chosen so that there are ~8 % more cells than points; a random subset of exactly n is kept."""
    total_area = float(tab.area.sum())
    v = math.sqrt(total_area / (n * 1.08))
    while True:
        nu = np.maximum(np.ceil(tab.len_u / v), 1).astype(np.int64)
        nv = np.maximum(np.ceil(tab.len_v / v), 1).astype(np.int64)
        cum = np.concatenate([[0], np.cumsum(nu * nv)])
        if cum[-1] >= n:
            break
        v *= 0.98
    K = int(cum[-1])
    sel = _rng(seed, 11, 0).permutation(K)[:n]          # which cells carry a point, in random order
    n_chunks = (n + _CHUNK - 1) // _CHUNK

    def chunk(k):
        idx = sel[k * _CHUNK:(k + 1) * _CHUNK]
        prim = np.searchsorted(cum, idx, side="right") - 1
        local = idx - cum[prim]
        iu, iv = local // nv[prim], local % nv[prim]
        g = _rng(seed, 12, k)
        fu = (iu + g.random(idx.shape[0])) / nu[prim]
        fv = (iv + g.random(idx.shape[0])) / nv[prim]
        P, Nn = tab.points(prim, fu, fv)
        P += g.normal(0.0, noise, P.shape)
        return P.astype(np.float32), _orient_to_sensor(P, Nn).astype(np.float32)

    parts = _run_chunks(chunk, n_chunks)
    return np.concatenate([p for p, _ in parts]), np.concatenate([q for _, q in parts])


def _sample_uniform(tab, n, seed, noise, radius=None, stream=21):
    """`n` points uniformly by area (optionally only within `radius` of the origin, by rejection)."""
    cumprob = np.cumsum(tab.area / tab.area.sum())
    pts, nrms, have, k = [], [], 0, 0
    while have < n:
        want = min(_CHUNK, int((n - have) * (1.15 if radius is None else 2.2)) + 64)
        g = _rng(seed, stream, k)
        k += 1
        prim = np.minimum(np.searchsorted(cumprob, g.random(want), side="right"), cumprob.shape[0] - 1)
        P, Nn = tab.points(prim, g.random(want), g.random(want))
        if radius is not None:
            keep = np.einsum("ij,ij->i", P, P) <= radius * radius
            P, Nn = P[keep], Nn[keep]
        P = P + g.normal(0.0, noise, P.shape)
        pts.append(P)
        nrms.append(_orient_to_sensor(P, Nn))
        have += P.shape[0]
    return np.concatenate(pts)[:n], np.concatenate(nrms)[:n]


def _orient_to_sensor(P, Nn):
    flip = np.einsum("ij,ij->i", Nn, -P) < 0
    Nn = Nn.copy()
    Nn[flip] *= -1
    return Nn


def covs_from_normals(Nn, eps=1e-3):
    """R diag(1,1,eps) R^T with the third axis = normal  ==  I - (1-eps) n n^T; 6 unique entries."""
    n = Nn / np.linalg.norm(Nn, axis=1, keepdims=True)
    k = 1.0 - eps
    return np.stack([1 - k * n[:, 0] ** 2, -k * n[:, 0] * n[:, 1], -k * n[:, 0] * n[:, 2], 1 - k * n[:, 1] ** 2,
                     -k * n[:, 1] * n[:, 2], 1 - k * n[:, 2] ** 2], axis=1).astype(np.float32)


@dataclass
class Scene:
    tgt_xyz: np.ndarray   # (M,3) float32
    tgt_nrm: np.ndarray   # (M,3) float32
    src_xyz: np.ndarray   # (N,3) float32
    src_nrm: np.ndarray   # (N,3) float32
    T_true: np.ndarray    # (4,4) float64: reading -> reference

    @property
    def tgt_cov(self):
        return covs_from_normals(self.tgt_nrm)

    @property
    def src_cov(self):
        return covs_from_normals(self.src_nrm)


def make_scene(n_src: int, n_tgt: int, seed: int = 1234, noise: float = 0.01, radius: float = 25.0,
               dedup: bool = True) -> Scene:
    """Deterministic in (n_src, n_tgt, seed, noise, radius, dedup); independent of the thread count."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tab = _PrimTable(make_primitives(rng))
    if dedup and n_tgt >= 1000:
        T, Tn = _sample_stratified(tab, n_tgt, seed, noise)    # roughly one point per voxel, like a voxelised map
    else:
        T, Tn = _sample_uniform(tab, n_tgt, seed, noise, stream=22)
    S, Sn = _sample_uniform(tab, n_src, seed + 1, noise, radius=radius)
    Tt = true_transform()
    Ti = np.linalg.inv(Tt)
    S = S @ Ti[:3, :3].T + Ti[:3, 3]
    Sn = Sn @ Ti[:3, :3].T
    return Scene(np.asarray(T, np.float32), np.asarray(Tn, np.float32), S.astype(np.float32), Sn.astype(np.float32), Tt)


def pose_error(T, T_ref):
    """(translation error [m], rotation geodesic [rad]) between two 4x4 transforms."""
    T, T_ref = np.asarray(T, np.float64), np.asarray(T_ref, np.float64)
    dt = float(np.linalg.norm(T[:3, 3] - T_ref[:3, 3]))
    R = T[:3, :3] @ T_ref[:3, :3].T
    c = max(-1.0, min(1.0, (np.trace(R) - 1.0) / 2.0))
    # small-angle robust: use the skew part
    s = 0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return dt, float(math.atan2(s, c))


def make_corridor(n_src, n_tgt, seed=0, length=40.0, width=3.0, height=2.5, n_end=0, noise=0.005):
    """Degenerate scene for the localizability tests (R8x): a straight corridor along x (floor, ceiling, two walls)
    -- translation along x is unobservable -- plus `n_end` target points on an end wall (weak information along x).
    Returns (tgt_xyz, tgt_nrm, src_xyz, src_nrm) float32; the reading is sampled from the same surfaces."""
    rng = np.random.default_rng(seed)

    def sample(n, with_end):
        face = rng.integers(0, 4, size=n)
        x = rng.uniform(-length / 2, length / 2, size=n)
        u = rng.uniform(-0.5, 0.5, size=n)
        p = np.zeros((n, 3))
        nr = np.zeros((n, 3))
        p[:, 0] = x
        fl, ce, wl, wr = face == 0, face == 1, face == 2, face == 3
        p[fl, 1], p[fl, 2], nr[fl, 2] = u[fl] * width, 0.0, 1.0
        p[ce, 1], p[ce, 2], nr[ce, 2] = u[ce] * width, height, -1.0
        p[wl, 1], p[wl, 2], nr[wl, 1] = -width / 2, (u[wl] + 0.5) * height, 1.0
        p[wr, 1], p[wr, 2], nr[wr, 1] = width / 2, (u[wr] + 0.5) * height, -1.0
        if with_end:
            e = np.zeros((with_end, 3))
            e[:, 0] = length / 2
            e[:, 1] = rng.uniform(-0.5, 0.5, size=with_end) * width
            e[:, 2] = rng.uniform(0, 1, size=with_end) * height
            en = np.zeros((with_end, 3))
            en[:, 0] = -1.0
            p, nr = np.concatenate([p, e]), np.concatenate([nr, en])
        p = p + rng.normal(scale=noise, size=p.shape)
        return p.astype(np.float32), nr.astype(np.float32)

    tgt, tn = sample(n_tgt, n_end)
    src, sn = sample(n_src, n_end)
    return tgt, tn, src, sn
