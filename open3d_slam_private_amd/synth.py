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


def _sample(prims, n, rng, radius=None):
    """n points uniformly by area (optionally only within `radius` of the origin, by rejection)."""
    areas = np.array([p.area for p in prims])
    prob = areas / areas.sum()
    pts_l, nrm_l = [], []
    have = 0
    while have < n:
        want = int((n - have) * (1.3 if radius is None else 3.0)) + 64
        counts = rng.multinomial(want, prob)
        P = np.empty((want, 3))
        Nn = np.zeros((want, 3))
        o = 0
        for prim, cnt in zip(prims, counts):
            if cnt == 0:
                continue
            sl = slice(o, o + cnt)
            if isinstance(prim, _Rect):
                P[sl] = prim.c
                P[sl, prim.u] += rng.uniform(-prim.hu, prim.hu, cnt)
                P[sl, prim.v] += rng.uniform(-prim.hv, prim.hv, cnt)
                Nn[sl, prim.k] = prim.sgn
            else:
                th = rng.uniform(0, 2 * math.pi, cnt)
                P[sl, 0] = prim.cx + prim.r * np.cos(th)
                P[sl, 1] = prim.cy + prim.r * np.sin(th)
                P[sl, 2] = rng.uniform(prim.z0, prim.z1, cnt)
                Nn[sl, 0] = np.cos(th)
                Nn[sl, 1] = np.sin(th)
            o += cnt
        perm = rng.permutation(want)
        P, Nn = P[perm], Nn[perm]
        if radius is not None:
            keep = np.linalg.norm(P, axis=1) <= radius
            P, Nn = P[keep], Nn[keep]
        pts_l.append(P)
        nrm_l.append(Nn)
        have += P.shape[0]
    P = np.concatenate(pts_l)[:n]
    Nn = np.concatenate(nrm_l)[:n]
    return P, Nn


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
    rng = np.random.Generator(np.random.PCG64(seed))
    prims = make_primitives(rng)
    total_area = float(sum(p.area for p in prims))
    if dedup and n_tgt >= 1000:
        # roughly one point per voxel, like a voxelised map (param_velodyne_puck16.lua:50)
        v = math.sqrt(total_area / (n_tgt * 1.15))
        raw, rawn = _sample(prims, int(3.0 * n_tgt), rng)
        key = np.floor(raw / v).astype(np.int64)
        h = (key[:, 0] * 73856093) ^ (key[:, 1] * 19349663) ^ (key[:, 2] * 83492791)
        _, first = np.unique(h, return_index=True)
        first.sort()
        if first.shape[0] >= n_tgt:
            sel = first[rng.permutation(first.shape[0])[:n_tgt]]
        else:
            rest = np.setdiff1d(np.arange(raw.shape[0]), first, assume_unique=False)
            sel = np.concatenate([first, rest[: n_tgt - first.shape[0]]])
        T, Tn = raw[sel], rawn[sel]
    else:
        T, Tn = _sample(prims, n_tgt, rng)
    T = T + rng.normal(0.0, noise, T.shape)
    Tn = _orient_to_sensor(T, Tn)
    rng2 = np.random.Generator(np.random.PCG64(seed + 1))
    S, Sn = _sample(prims, n_src, rng2, radius=radius)
    S = S + rng2.normal(0.0, noise, S.shape)
    Sn = _orient_to_sensor(S, Sn)
    Tt = true_transform()
    Ti = np.linalg.inv(Tt)
    S = S @ Ti[:3, :3].T + Ti[:3, 3]
    Sn = Sn @ Ti[:3, :3].T
    return Scene(T.astype(np.float32), Tn.astype(np.float32), S.astype(np.float32), Sn.astype(np.float32), Tt)


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
