"""ctypes front-end of the CPU oracle (oracle/icp_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libicp_oracle.so")

F_TRIM, F_NORMAL, F_MAXDIST = 1, 2, 4


class Filters(C.Structure):
    _fields_ = [("flags", C.c_int32), ("trim_ratio", C.c_float), ("cos_max_angle", C.c_float),
                ("outlier_max_d2", C.c_float)]


class Params(C.Structure):
    _fields_ = [("max_dist", C.c_float), ("filt", Filters), ("max_iter", C.c_int32),
                ("min_diff_rot", C.c_float), ("min_diff_trans", C.c_float), ("smooth_len", C.c_int32),
                ("fixed_iters", C.c_int32), ("n_threads", C.c_int32),
                ("use_xicp", C.c_int32), ("xicp_enough", C.c_float), ("xicp_insufficient", C.c_float),
                ("xicp_cos_min", C.c_float), ("xicp_cos_strong", C.c_float)]


class Result(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("max_iter_reached", C.c_int32),
                ("status", C.c_int32), ("n_kept_last", C.c_int64), ("err_last", C.c_double),
                ("A_last", C.c_float * 36), ("b_last", C.c_float * 6), ("T_iter", C.c_float * 16),
                ("T_refMean_readMean", C.c_float * 16), ("loop_seconds", C.c_double),
                ("localizable", C.c_int32 * 6), ("n_constraints", C.c_int32), ("pad_", C.c_int32),
                ("xicp_combined", C.c_double * 6), ("xicp_high", C.c_double * 6)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds).  Returns the .so path.  O3D_ORACLE_LIB points the tests at another build of
    the same source (the ASan / UBSan build of tools/oracle_sanitizers.sh)."""
    if os.environ.get("O3D_ORACLE_LIB"):
        return os.environ["O3D_ORACLE_LIB"]
    src = os.path.join(_HERE, "icp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "_build/libicp_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_kd_build.restype = C.c_void_p
        _lib.orc_kd_build.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        _lib.orc_kd_free.argtypes = [C.c_void_p]
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def _f32(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if cols is not None:
        assert a.ndim == 2 and a.shape[1] == cols, a.shape
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def max_threads() -> int:
    return lib().orc_max_threads()


def centroid(xyz) -> np.ndarray:
    xyz = _f32(xyz)
    out = np.zeros(3, np.float32)
    lib().orc_centroid(_p(xyz), C.c_int64(xyz.shape[1]), C.c_int64(xyz.shape[0]), _p(out))
    return out


class KdTree:
    def __init__(self, xyz):
        self.xyz = _f32(xyz)
        self.h = lib().orc_kd_build(_p(self.xyz), C.c_int64(self.xyz.shape[1]), C.c_int64(self.xyz.shape[0]))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_kd_free(C.c_void_p(self.h))
            self.h = None

    def knn(self, src_xyz, T, max_dist=math.inf, n_threads=1):
        src = _f32(src_xyz)
        T = _f32(T).reshape(16)
        n = src.shape[0]
        ids = np.empty(n, np.int32)
        d2 = np.empty(n, np.float32)
        lib().orc_knn(C.c_void_p(self.h), _p(src), C.c_int64(src.shape[1]), C.c_int64(n), _p(T),
                      C.c_float(max_dist), _p(ids), _p(d2), C.c_int(n_threads))
        return ids, d2


def knn_brute(tgt_xyz, src_xyz, T, max_dist=math.inf):
    tgt, src = _f32(tgt_xyz), _f32(src_xyz)
    T = _f32(T).reshape(16)
    n = src.shape[0]
    ids = np.empty(n, np.int32)
    d2 = np.empty(n, np.float32)
    lib().orc_knn_brute(_p(tgt), C.c_int64(tgt.shape[1]), C.c_int64(tgt.shape[0]), _p(src),
                        C.c_int64(src.shape[1]), C.c_int64(n), _p(T), C.c_float(max_dist), _p(ids), _p(d2))
    return ids, d2


def make_filters(trim_ratio=None, max_normal_angle=None, outlier_max_dist=None) -> Filters:
    f = Filters(0, 1.0, -1.0, math.inf)
    if trim_ratio is not None:
        f.flags |= F_TRIM
        f.trim_ratio = trim_ratio
    if max_normal_angle is not None:
        f.flags |= F_NORMAL
        f.cos_max_angle = float(np.cos(np.float32(max_normal_angle)))  # cosf in T=float
    if outlier_max_dist is not None:
        f.flags |= F_MAXDIST
        f.outlier_max_d2 = float(np.float32(outlier_max_dist) ** 2)
    return f


def trim_limit(d2, ratio, n_threads=1):
    """n_threads <= 1: nth_element-class selection as the reference (Matches.cpp:83); > 1: parallel exact radix select."""
    d2 = _f32(d2)
    lim = C.c_float()
    nf = C.c_int64()
    rc = lib().orc_trim_limit_mt(_p(d2), C.c_int64(d2.shape[0]), C.c_float(ratio), C.byref(lim), C.byref(nf),
                                 C.c_int(n_threads))
    if rc != 0:
        raise RuntimeError("ConvergenceError: no matches available for computing distance quantiles")
    return lim.value, nf.value


def weights(filters: Filters, src_nrm, tgt_nrm, T, ids, d2, n_threads=1):
    n = ids.shape[0]
    w = np.empty(n, np.float32)
    sn = _f32(src_nrm) if src_nrm is not None else None
    tn = _f32(tgt_nrm) if tgt_nrm is not None else None
    T = _f32(T).reshape(16)
    lim = C.c_float()
    rc = lib().orc_weights_mt(C.byref(filters), _p(sn), C.c_int64(sn.shape[1] if sn is not None else 3), _p(tn),
                              C.c_int64(tn.shape[1] if tn is not None else 3), _p(T), _p(ids), _p(_f32(d2)),
                              C.c_int64(n), _p(w), C.byref(lim), C.c_int(n_threads))
    if rc != 0:
        raise RuntimeError("ConvergenceError: no matches available for computing distance quantiles")
    return w, lim.value


def p2pl_normal_eq(src_xyz, tgt_xyz, tgt_nrm, T, ids, d2, w, n_threads=1):
    src, tgt, tn = _f32(src_xyz), _f32(tgt_xyz), _f32(tgt_nrm)
    T = _f32(T).reshape(16)
    A = np.zeros((6, 6), np.float32)
    b = np.zeros(6, np.float32)
    err = C.c_double()
    kept = C.c_int64()
    lib().orc_p2pl_normal_eq(_p(src), C.c_int64(src.shape[1]), _p(tgt), C.c_int64(tgt.shape[1]), _p(tn),
                             C.c_int64(tn.shape[1]), _p(T), _p(ids), _p(_f32(d2)), _p(_f32(w)),
                             C.c_int64(src.shape[0]), _p(A), _p(b), C.byref(err), C.byref(kept), C.c_int(n_threads))
    return A, b, err.value, kept.value


def solve6(A, b):
    A, b = _f32(A).reshape(36), _f32(b)
    x = np.zeros(6, np.float32)
    lib().orc_solve6.restype = C.c_int
    rank = lib().orc_solve6(_p(A), _p(b), _p(x))
    return x, rank


def x_to_T(x):
    T = np.zeros(16, np.float32)
    lib().orc_x_to_T(_p(_f32(x)), _p(T))
    return T.reshape(4, 4)


def icp_p2pl(tgt_xyz, tgt_nrm, src_xyz, src_nrm=None, T_init=None, *, max_dist=math.inf, trim_ratio=None,
             max_normal_angle=None, outlier_max_dist=None, max_iter=40, min_diff_rot=0.001, min_diff_trans=0.001,
             smooth_len=3, fixed_iters=0, n_threads=1, xicp=None):
    """Full reference-chain registration (R1-R10).  Returns (T 4x4 float32, Result).
    xicp: None, or (enough, insufficient, min_angle_deg, strong_angle_deg) = R8x OptimizedEqualityConstraints."""
    tgt, tn, src = _f32(tgt_xyz), _f32(tgt_nrm), _f32(src_xyz)
    sn = _f32(src_nrm) if src_nrm is not None else None
    T0 = _f32(np.eye(4) if T_init is None else T_init).reshape(16)
    P = Params()
    P.max_dist = max_dist
    P.filt = make_filters(trim_ratio, max_normal_angle, outlier_max_dist)
    P.max_iter, P.min_diff_rot, P.min_diff_trans, P.smooth_len = max_iter, min_diff_rot, min_diff_trans, smooth_len
    P.fixed_iters, P.n_threads = fixed_iters, n_threads
    if xicp is not None:
        P.use_xicp = 1
        P.xicp_enough, P.xicp_insufficient = float(xicp[0]), float(xicp[1])
        P.xicp_cos_min = float(np.cos(np.float32(xicp[2]) * np.float32(np.pi) / np.float32(180.0), dtype=np.float32))
        P.xicp_cos_strong = float(np.cos(np.float32(xicp[3]) * np.float32(np.pi) / np.float32(180.0), dtype=np.float32))
    if (P.filt.flags & F_NORMAL) and sn is None:
        raise ValueError("InvalidField: SurfaceNormalOutlierFilter needs 'normals' on the reading")
    T = np.zeros(16, np.float32)
    res = Result()
    lib().orc_icp_p2pl(_p(tgt), C.c_int64(tgt.shape[1]), _p(tn), C.c_int64(tn.shape[1]), C.c_int64(tgt.shape[0]),
                       _p(src), C.c_int64(src.shape[1]), _p(sn), C.c_int64(sn.shape[1] if sn is not None else 3),
                       C.c_int64(src.shape[0]), _p(T0), C.byref(P), _p(T), C.byref(res))
    return T.reshape(4, 4), res


def gicp_normal_eq(src_xyz, src_cov, tgt_xyz, tgt_cov, T, ids):
    src, tgt = _f32(src_xyz), _f32(tgt_xyz)
    sc, tc = _f32(src_cov, 6), _f32(tgt_cov, 6)
    T = _f32(T).reshape(16)
    H = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    e = C.c_double()
    cnt = C.c_int64()
    lib().orc_gicp_normal_eq(_p(src), C.c_int64(src.shape[1]), _p(sc), _p(tgt), C.c_int64(tgt.shape[1]), _p(tc),
                             _p(T), _p(np.ascontiguousarray(ids, np.int32)), C.c_int64(src.shape[0]), _p(H), _p(b),
                             C.byref(e), C.byref(cnt))
    return H, b, e.value, cnt.value


def icp_gicp(tgt_xyz, tgt_cov, src_xyz, src_cov, T_init=None, *, max_dist=math.inf, max_iter=20, fixed_iters=0,
             rot_eps=0.1 * math.pi / 180.0, trans_eps=1e-3, n_threads=1, stop_rule=0, rel_fitness=1e-6, rel_rmse=1e-6):
    """stop_rule 1: Open3D ICPConvergenceCriteria semantics (CloudRegistration.cpp:16-21,45-52), see orc_icp_gicp2."""
    tgt, src = _f32(tgt_xyz), _f32(src_xyz)
    tc, sc = _f32(tgt_cov, 6), _f32(src_cov, 6)
    T0 = _f32(np.eye(4) if T_init is None else T_init).reshape(16)
    T = np.zeros(16, np.float32)
    res = Result()
    lib().orc_icp_gicp2(_p(tgt), C.c_int64(tgt.shape[1]), _p(tc), C.c_int64(tgt.shape[0]), _p(src),
                        C.c_int64(src.shape[1]), _p(sc), C.c_int64(src.shape[0]), _p(T0), C.c_float(max_dist),
                        C.c_int(max_iter), C.c_int(fixed_iters), C.c_double(rot_eps), C.c_double(trans_eps),
                        C.c_int(stop_rule), C.c_double(float(np.float32(rel_fitness))), C.c_double(float(np.float32(rel_rmse))),
                        C.c_int(n_threads), _p(T), C.byref(res))
    return T.reshape(4, 4), res


def knn_k(tree: "KdTree", q_xyz, k, max_dist=math.inf, n_threads=1):
    q = _f32(q_xyz)
    n = q.shape[0]
    ids = np.empty((n, k), np.int32)
    d2 = np.empty((n, k), np.float32)
    lib().orc_knn_k(C.c_void_p(tree.h), _p(q), C.c_int64(q.shape[1]), C.c_int64(n), C.c_int(k), C.c_float(max_dist),
                    _p(ids), _p(d2), C.c_int(n_threads))
    return ids, d2


def surface_normals(xyz, k, max_dist=math.inf, viewpoint=None, regularise=False, n_threads=1, extras=False):
    """(normals n x 3, eigenvalues ascending n x 3, covariances n x 6, neighbour ids n x k); with extras=True also
    (eigenvectors n x 9, densities n, mean distances n)."""
    x = _f32(xyz)
    n = x.shape[0]
    nrm = np.empty((n, 3), np.float32)
    ev = np.empty((n, 3), np.float32)
    cov = np.empty((n, 6), np.float32)
    ids = np.empty((n, k), np.int32)
    evec = np.empty((n, 9), np.float32) if extras else None
    dens = np.empty(n, np.float32) if extras else None
    md = np.empty(n, np.float32) if extras else None
    vp = _f32(viewpoint) if viewpoint is not None else None
    lib().orc_surface_normals(_p(x), C.c_int64(x.shape[1]), C.c_int64(n), C.c_int(k), C.c_float(max_dist), _p(vp),
                              C.c_int(1 if regularise else 0), _p(nrm), _p(ev), _p(cov), _p(ids), C.c_int(n_threads),
                              _p(evec), _p(dens), _p(md))
    if extras:
        return nrm, ev, cov, ids, evec, dens, md
    return nrm, ev, cov, ids


def solve6_xicp(A, b, flags):
    """Equality-constrained 6x6 solve of R8x (null-space form).  Returns (x float32[6], rank of the reduced system)."""
    A_, b_ = _f32(A).reshape(36), _f32(b).reshape(6)
    f = np.ascontiguousarray(flags, np.int32)
    x = np.zeros(6, np.float32)
    lib().orc_solve6_xicp.restype = C.c_int
    r = lib().orc_solve6_xicp(_p(A_), _p(b_), _p(f), _p(x))
    return x, int(r)


def xicp_eigvecs(A):
    A_ = _f32(A).reshape(36)
    Vr, Vt = np.zeros(9), np.zeros(9)
    lib().orc_xicp_eigvecs(_p(A_), _p(Vr), _p(Vt))
    return Vr.reshape(3, 3), Vt.reshape(3, 3)


def crop_mask(xyz, crop_type, center=(0.0, 0.0, 0.0), radius_min=0.0, radius_max=0.0, min_z=0.0, max_z=0.0):
    """CroppingVolume::isWithinVolume (open3d_slam/src/croppers.cpp:118-170) in float64, operation by operation:
    1 MaxRadius |p-t| <= r; 2 MinRadius |p-t| >= r; 3 MinMaxRadius; 4 Cylinder z in [minZ, maxZ] and |(p-t).xy| <= r."""
    p = np.asarray(xyz, np.float64)
    if crop_type == 0:
        return np.ones(p.shape[0], bool)
    d = p - np.asarray(center, np.float64)[None]
    if crop_type == 4:
        r = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])
        return (p[:, 2] >= min_z) & (p[:, 2] <= max_z) & (r <= radius_max)
    r = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
    if crop_type == 1:
        return r <= radius_max
    if crop_type == 2:
        return r >= radius_min
    return (r <= radius_max) & (r >= radius_min)


def voxelize_within_volume(xyz, voxel_size, inside_mask, normals=None, covs=None):
    """voxelizeWithinCroppingVolume (open3d_slam/src/helpers.cpp:117-192) restated with a plain dict: points outside the
    volume pass through in order; inside points are accumulated per voxel index floor(p * (1/voxel)) (VoxelHashMap.hpp:
    43-51) in index order in float64 (AccumulatedPoint, helpers.cpp:30-72).  Voxels are emitted in ascending (z, y, x)
    index (the reference's unordered_map order is unspecified).  Returns (xyz, normals, covs, n_outside)."""
    p = np.asarray(xyz, np.float64)
    nr = np.asarray(normals, np.float64) if normals is not None else None
    cv = np.asarray(covs, np.float64).reshape(-1, 9) if covs is not None else None
    if not voxel_size > 0:
        return p.copy(), nr, cv, p.shape[0]
    inv = 1.0 / float(voxel_size)
    out_idx = np.nonzero(~inside_mask)[0]
    acc = {}
    for i in np.nonzero(inside_mask)[0]:
        key = (int(np.floor(p[i, 2] * inv)), int(np.floor(p[i, 1] * inv)), int(np.floor(p[i, 0] * inv)))
        a = acc.get(key)
        if a is None:
            a = acc[key] = [np.zeros(3), np.zeros(3), np.zeros(9), 0]
        a[0] = a[0] + p[i]
        if nr is not None and not np.any(np.isnan(nr[i])):
            a[1] = a[1] + nr[i]
        if cv is not None:
            a[2] = a[2] + cv[i]
        a[3] += 1
    keys = sorted(acc)
    vp = np.array([acc[k][0] / float(acc[k][3]) for k in keys]).reshape(-1, 3)
    ox = np.concatenate([p[out_idx], vp])
    on = oc = None
    if nr is not None:
        vn = []
        for k in keys:
            a = acc[k][1] / float(acc[k][3])
            z2 = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]
            vn.append(a / np.sqrt(z2) if z2 > 0 else a)
        on = np.concatenate([nr[out_idx], np.array(vn).reshape(-1, 3)])
    if cv is not None:
        oc = np.concatenate([cv[out_idx], np.array([acc[k][2] / float(acc[k][3]) for k in keys]).reshape(-1, 9)])
    return ox, on, oc, out_idx.size


def carve_indices(map_xyz, scan_xyz, sensor, voxel_size=0.1, max_ray=20.0, truncation=0.1, min_dot=0.5, map_normals=None,
                  subset_mask=None):
    """getIdxsOfCarvedPoints (open3d_slam/src/helpers.cpp:238-283) restated with a dict of voxels; float64, operation by
    operation.  Returns the ascending indices of the map points a scan ray passes through."""
    mp = np.asarray(map_xyz, np.float64)
    sp = np.asarray(scan_xyz, np.float64)
    s = np.asarray(sensor, np.float64)
    nr = np.asarray(map_normals, np.float64) if map_normals is not None else None
    inv = 1.0 / float(voxel_size)
    vox = {}
    idxs = np.arange(mp.shape[0]) if subset_mask is None else np.nonzero(subset_mask)[0]
    for i in idxs:
        key = (int(np.floor(mp[i, 0] * inv)), int(np.floor(mp[i, 1] * inv)), int(np.floor(mp[i, 2] * inv)))
        vox.setdefault(key, []).append(int(i))
    remove = set()
    for p in sp:
        d = p - s
        length = np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
        if not (length > 0.0) or not np.isfinite(length):
            continue
        u = d / length
        reach = max(voxel_size, min(length - truncation, max_ray))
        dist = 0.0
        while dist < reach:
            cur = dist * u + s
            key = (int(np.floor(cur[0] * inv)), int(np.floor(cur[1] * inv)), int(np.floor(cur[2] * inv)))
            for j in vox.get(key, ()):
                ok = True
                if nr is not None:
                    n = nr[j]
                    z2 = (n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]
                    if z2 > 0:
                        n = n / np.sqrt(z2)
                    ok = abs((u[0] * n[0] + u[1] * n[1]) + u[2] * n[2]) > min_dot
                if ok:
                    remove.add(j)
            dist += voxel_size
    return np.array(sorted(remove), np.int32)


def information_matrix(tgt_xyz, src_xyz, T, max_dist):
    """GetInformationMatrixFromPointClouds as used in open3d_slam/src/constraint_builders.cpp:69-73 (Open3D 0.15.1
    arithmetic, not in the tree: restated, PARITY UNPINNED): exact NN of every transformed reading point within
    max_dist; G = [[0,z,-y,1,0,0],[-z,0,x,0,1,0],[y,-x,0,0,0,1]] at the matched reference point; sum G^T G in float64."""
    tree = KdTree(_f32(tgt_xyz))
    ids, d2 = tree.knn(_f32(src_xyz), _f32(T), max_dist=max_dist)
    q = _f32(tgt_xyz)[ids[ids >= 0]].astype(np.float64)
    info = np.zeros((6, 6))
    for x, y, z in q:
        G = np.array([[0, z, -y, 1, 0, 0], [-z, 0, x, 0, 1, 0], [y, -x, 0, 0, 0, 1]], np.float64)
        info += G.T @ G
    return info, q.shape[0]


def smooth_normals(normals, ids):
    """SurfaceNormalDataPointsFilter.smoothNormals (SurfaceNormal.cpp:259-283), sequential, in place on a copy."""
    nr = np.ascontiguousarray(normals, dtype=np.float32).copy()
    ii = np.ascontiguousarray(ids, dtype=np.int32)
    lib().orc_smooth_normals(_p(nr), _p(ii), C.c_int64(nr.shape[0]), C.c_int(ii.shape[1]))
    return nr
