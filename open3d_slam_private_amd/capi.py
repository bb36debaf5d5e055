"""ctypes binding of include/o3dslam_reg.h (the drop-in C ABI).  No torch types; numpy or raw
device pointers only.  Mirrors the header 1:1 -- see the header for the reference file:line each
entry point replaces."""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("O3D_REG_LIB") or os.path.join(_HERE, "lib", "libo3dslam_reg.so")   # O3D_REG_LIB: A/B builds (tools/)

STATUS_NAMES = {0: "OK", 1: "EMPTY_TARGET", 2: "EMPTY_SOURCE", 3: "NO_CORRESPONDENCES", 4: "BAD_TRANSFORM",
                5: "NOT_CONFIGURED", 6: "BAD_ARGUMENT", 7: "MISSING_FIELD", 8: "DEVICE_ERROR", 9: "UNSUPPORTED"}
COST_P2PL, COST_GICP = 0, 1


class RegError(RuntimeError):
    def __init__(self, status, msg=""):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {msg}")


_DEBUG_FIELDS = ("profile_loop", "match_variant", "debug_flags", "disable_halo", "lanes_per_point", "disable_fused")


class RegParams(C.Structure):
    """include/o3dslam_reg.h reg_params.  The experiment switches of include/o3dslam_reg_debug.h (profile_loop,
    match_variant, debug_flags, disable_halo, lanes_per_point, disable_fused) are NOT part of the C struct any more; for
    the tests' and tools' convenience they can still be set as plain Python attributes on a RegParams object --
    Registration() passes them on through reg_debug_configure."""
    _fields_ = [("struct_size", C.c_int32), ("cost", C.c_int32), ("knn", C.c_int32), ("max_dist", C.c_float),
                ("epsilon", C.c_float), ("use_trimmed", C.c_int32), ("trim_ratio", C.c_float),
                ("use_surface_normal", C.c_int32), ("max_normal_angle", C.c_float),
                ("use_max_dist_filter", C.c_int32), ("outlier_max_dist", C.c_float), ("max_iter", C.c_int32),
                ("min_diff_rot", C.c_float), ("min_diff_trans", C.c_float), ("smooth_len", C.c_int32),
                ("fixed_iters", C.c_int32), ("gicp_rot_eps", C.c_float), ("gicp_trans_eps", C.c_float),
                ("cell_size", C.c_float), ("device", C.c_int32), ("sort_source", C.c_int32), ("use_xicp", C.c_int32),
                ("xicp_enough", C.c_float), ("xicp_insufficient", C.c_float), ("xicp_min_angle_deg", C.c_float),
                ("xicp_strong_angle_deg", C.c_float), ("gicp_stop_rule", C.c_int32), ("gicp_rel_fitness", C.c_float),
                ("gicp_rel_rmse", C.c_float), ("reserved", C.c_int32)]
    profile_loop = match_variant = debug_flags = disable_halo = lanes_per_point = disable_fused = 0


class RegDebugParams(C.Structure):
    _fields_ = [("struct_size", C.c_int32)] + [(k, C.c_int32) for k in _DEBUG_FIELDS] + [("reserved", C.c_int32)]


class RegResult(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("max_iter_reached", C.c_int32),
                ("rank_last", C.c_int32), ("n_inliers", C.c_int64), ("n_matched", C.c_int64), ("error", C.c_double),
                ("fitness", C.c_double), ("inlier_rmse", C.c_double), ("H_last", C.c_float * 36),
                ("b_last", C.c_float * 6), ("target_build_ms", C.c_float), ("loop_ms", C.c_float),
                ("T_iter_last", C.c_float * 16), ("n_band_stalls", C.c_int32), ("n_constraints", C.c_int32),
                ("prof_ms", C.c_float * 4), ("prof_launches", C.c_int32 * 4), ("localizable", C.c_int32 * 6),
                ("xicp_combined", C.c_double * 6), ("xicp_high", C.c_double * 6), ("source_prep_ms", C.c_float),
                ("rotation_corrected", C.c_int32), ("T_iter_prev", C.c_float * 16), ("n_tail_launches", C.c_int32),
                ("n_tail_iterations", C.c_int32)]


class NormalsOut(C.Structure):
    _fields_ = [(name, C.c_void_p) for name in ("normals", "eigvals", "eigvecs", "covs", "densities", "mean_dists", "ids")]


class RegCrop(C.Structure):
    _fields_ = [("type", C.c_int32), ("reserved", C.c_int32), ("center", C.c_double * 3), ("radius_min", C.c_double),
                ("radius_max", C.c_double), ("min_z", C.c_double), ("max_z", C.c_double)]


CROP_NONE, CROP_MAX_RADIUS, CROP_MIN_RADIUS, CROP_MIN_MAX_RADIUS, CROP_CYLINDER = 0, 1, 2, 3, 4


class DistStatus(C.Structure):
    _fields_ = [("sequences_done", C.c_int64), ("sequences_enqueued", C.c_int64), ("iterations", C.c_int32),
                ("done", C.c_int32), ("stall", C.c_int32), ("stream_idle", C.c_int32), ("limit_last", C.c_float),
                ("limit_prev", C.c_float)]


class DistAction(C.Structure):
    _fields_ = [("kind", C.c_int32), ("count", C.c_int32), ("seq", C.c_int64)]


class DistReply(C.Structure):
    _fields_ = [("available", C.c_int32), ("iterations", C.c_int32), ("done", C.c_int32), ("stall", C.c_int32),
                ("limit_last", C.c_float), ("limit_prev", C.c_float)]


STEER_RECORD, STEER_GENERIC, STEER_FUSED, STEER_DRAIN, STEER_DONE = 0, 1, 2, 3, 4
DT_I32, DT_I64, DT_F64 = 0, 1, 2
DIST_ID_BYTES = 128
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class Collectives(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("all_reduce_sum", ALL_REDUCE_FN), ("all_gather", ALL_GATHER_FN)]


class TargetInfo(C.Structure):
    _fields_ = [("n_points", C.c_int64), ("n_bricks", C.c_int64), ("n_cells_occupied", C.c_int64),
                ("table_bytes", C.c_int64), ("cell_size", C.c_float), ("origin", C.c_float * 3),
                ("centroid", C.c_float * 3), ("dims", C.c_int32 * 3)]


EXPORTS = ["reg_default_params", "reg_shipped_params", "reg_create", "reg_destroy", "reg_last_error",
           "reg_set_stream", "reg_set_target", "reg_set_source", "reg_register", "reg_compute", "reg_prepare",
           "reg_linearize", "reg_get_correspondences", "reg_match_local", "reg_trim_histogram", "reg_reduce_local",
           "reg_solve_update", "reg_host_solve6", "reg_host_x_to_T", "reg_host_centroid", "reg_get_target_info", "reg_profile_kernels", "reg_source_centroid_sums", "reg_prepare_centroid", "reg_compose",
           "reg_dist_begin", "reg_dist_buffers", "reg_dist_phase", "reg_dist_finish",
           "reg_dist_fused_buffers", "reg_dist_poll", "reg_estimate_normals", "reg_smooth_normals", "reg_host_solve6_xicp",
           "reg_set_target_f64", "reg_get_target_source_indices", "reg_voxelize_within_volume", "reg_carve_indices", "reg_dist_xicp_buffers", "reg_dist_gather_buffers",
           "reg_dist_record", "reg_dist_centroid_sums", "reg_dist_prepare",
           "reg_information_matrix", "reg_set_source_f64", "reg_debug_configure",
           "reg_dist_get_unique_id", "reg_dist_init", "reg_dist_init_custom", "reg_dist_register", "reg_dist_shutdown",
           "reg_dist_info", "reg_dist_steer_create", "reg_dist_steer_destroy", "reg_dist_steer_step",
           "reg_dist_steer_counts", "reg_host_tail_plan"]


def lib_path() -> str:
    return _SO


def build_library(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", src_dir]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return _SO


_lib = None


def load_library():
    """Load the HIP extension; fails loudly when it has not been built (no fallback exists).

    Note: PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  A process that uses BOTH torch (device buffers,
    torch.distributed) and this library must `import torch` first, so that one HIP runtime serves both."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError(f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the registration path.")
    lib = C.CDLL(_SO)
    vp, i64, f32p = C.c_void_p, C.c_int64, C.c_void_p
    lib.reg_create.argtypes = [C.POINTER(RegParams), C.POINTER(vp)]
    lib.reg_destroy.argtypes = [vp]
    lib.reg_last_error.argtypes = [vp]
    lib.reg_last_error.restype = C.c_char_p
    lib.reg_set_stream.argtypes = [vp, vp]
    lib.reg_debug_configure.argtypes = [vp, C.POINTER(RegDebugParams)]
    lib.reg_dist_get_unique_id.argtypes = [C.c_char_p]
    lib.reg_dist_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    lib.reg_dist_init_custom.argtypes = [vp, C.POINTER(Collectives), C.c_int, C.c_int]
    lib.reg_dist_register.argtypes = [vp, f32p, f32p, C.POINTER(RegResult)]
    lib.reg_dist_shutdown.argtypes = [vp]
    lib.reg_dist_info.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.reg_dist_steer_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
    lib.reg_dist_steer_create.restype = vp
    lib.reg_dist_steer_destroy.argtypes = [vp]
    lib.reg_dist_steer_destroy.restype = None
    lib.reg_dist_steer_step.argtypes = [vp, C.POINTER(DistReply)]
    lib.reg_dist_steer_step.restype = DistAction
    lib.reg_dist_steer_counts.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.reg_dist_steer_counts.restype = None
    lib.reg_set_source_f64.argtypes = [vp, vp, vp, vp, i64, C.c_int]
    lib.reg_set_target.argtypes = [vp, f32p, i64, f32p, i64, f32p, i64, C.c_int]
    lib.reg_set_source.argtypes = [vp, f32p, i64, f32p, i64, f32p, i64, C.c_int]
    lib.reg_register.argtypes = [vp, f32p, f32p, C.POINTER(RegResult)]
    lib.reg_compute.argtypes = [vp, f32p, i64, f32p, i64, f32p, i64, C.c_int, f32p, f32p, C.POINTER(RegResult)]
    lib.reg_prepare.argtypes = [vp, f32p]
    lib.reg_linearize.argtypes = [vp, f32p, f32p, f32p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.reg_get_correspondences.argtypes = [vp, vp, vp, vp]
    lib.reg_match_local.argtypes = [vp, f32p]
    lib.reg_trim_histogram.argtypes = [vp, C.c_int, C.c_uint32, vp]
    lib.reg_reduce_local.argtypes = [vp, f32p, C.c_float, vp]
    lib.reg_solve_update.argtypes = [C.POINTER(RegParams), vp, f32p, f32p, C.POINTER(C.c_int32)]
    lib.reg_host_solve6.argtypes = [f32p, f32p, f32p]
    lib.reg_host_solve6.restype = C.c_int
    lib.reg_host_x_to_T.argtypes = [f32p, f32p]
    lib.reg_host_solve6_xicp.argtypes = [f32p, f32p, vp, f32p]
    lib.reg_host_solve6_xicp.restype = C.c_int
    lib.reg_set_target_f64.argtypes = [vp, vp, vp, vp, i64, C.c_int, C.POINTER(RegCrop), C.POINTER(C.c_int64)]
    lib.reg_get_target_source_indices.argtypes = [vp, vp]
    lib.reg_carve_indices.argtypes = [vp, vp, vp, i64, vp, i64, C.c_int, C.POINTER(C.c_double), C.POINTER(RegCrop),
                                      C.c_double, C.c_double, C.c_double, C.c_double, vp, C.POINTER(C.c_int64)]
    lib.reg_voxelize_within_volume.argtypes = [vp, vp, vp, vp, i64, C.c_int, C.POINTER(RegCrop), C.c_double, vp, vp, vp,
                                               C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.reg_host_centroid.argtypes = [f32p, i64, i64, f32p]
    lib.reg_get_target_info.argtypes = [vp, C.POINTER(TargetInfo)]
    lib.reg_profile_kernels.argtypes = [vp, f32p, C.c_int, f32p]
    lib.reg_source_centroid_sums.argtypes = [vp, vp]
    lib.reg_prepare_centroid.argtypes = [vp, f32p, f32p]
    lib.reg_compose.argtypes = [vp, f32p, f32p]
    lib.reg_dist_begin.argtypes = [vp, f32p]
    lib.reg_dist_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.reg_dist_phase.argtypes = [vp, C.c_int]
    lib.reg_dist_finish.argtypes = [vp, f32p, C.POINTER(RegResult)]
    lib.reg_dist_fused_buffers.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int64)]
    lib.reg_dist_poll.argtypes = [vp, C.POINTER(DistStatus)]
    lib.reg_dist_record.argtypes = [vp, i64, C.POINTER(DistStatus)]
    lib.reg_dist_centroid_sums.argtypes = [vp, C.POINTER(vp)]
    lib.reg_dist_prepare.argtypes = [vp, f32p, i64]
    lib.reg_information_matrix.argtypes = [vp, f32p, C.c_float, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.reg_dist_xicp_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.reg_dist_gather_buffers.argtypes = [vp, C.c_int, i64, C.POINTER(vp), C.POINTER(vp)]
    lib.reg_smooth_normals.argtypes = [vp, vp, vp, i64, C.c_int, C.c_int, vp]
    lib.reg_estimate_normals.argtypes = [vp, vp, i64, i64, C.c_int, C.c_int, C.c_float, vp, C.c_int,
                                         C.POINTER(NormalsOut), C.POINTER(C.c_int64)]
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError if the library does not export what the header declares
    _lib = lib
    return lib


def default_params() -> RegParams:
    p = RegParams()
    load_library().reg_default_params(C.byref(p))
    return p


def shipped_params() -> RegParams:
    p = RegParams()
    load_library().reg_shipped_params(C.byref(p))
    return p


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _T_in(T):
    """numpy 4x4 (math layout) -> column-major float[16] as the ABI wants (== Eigen::Matrix4f::data())."""
    T = np.asarray(T, dtype=np.float32).reshape(4, 4)
    return np.ascontiguousarray(T.T).reshape(16)


def _T_out(buf):
    return np.array(buf, dtype=np.float32).reshape(4, 4).T.copy()


class Registration:
    """One registration context (== one reg_handle)."""

    def __init__(self, params: RegParams | None = None, **overrides):
        self._lib = load_library()
        p = params if params is not None else default_params()
        for k, v in overrides.items():
            if not hasattr(p, k):
                raise TypeError(f"unknown parameter {k}")
            setattr(p, k, v)
        p.struct_size = C.sizeof(RegParams)
        self.params = p
        self._h = C.c_void_p()
        st = self._lib.reg_create(C.byref(p), C.byref(self._h))
        if st != 0:
            msg = self._lib.reg_last_error(self._h).decode() if self._h else "reg_create rejected the parameters"
            if self._h:
                self._lib.reg_destroy(self._h)
                self._h = C.c_void_p()
            raise RegError(st, msg)
        dbg = RegDebugParams()
        dbg.struct_size = C.sizeof(RegDebugParams)
        for k in _DEBUG_FIELDS:
            setattr(dbg, k, int(getattr(p, k, 0)))
        if any(getattr(dbg, k) for k in _DEBUG_FIELDS):
            self._check(self._lib.reg_debug_configure(self._h, C.byref(dbg)))
        self._keep = []
        self.n_source = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.reg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != 0:
            raise RegError(st, self._lib.reg_last_error(self._h).decode())

    def set_stream(self, hip_stream: int):
        self._check(self._lib.reg_set_stream(self._h, C.c_void_p(hip_stream)))

    # ---- host (numpy) entry points ------------------------------------------------------------
    def set_target(self, xyz, normals=None, covs=None):
        xyz = _f32(xyz)
        nrm = _f32(normals) if normals is not None else None
        cov = _f32(covs) if covs is not None else None
        m = xyz.shape[0] if xyz.ndim == 2 else 0
        self._check(self._lib.reg_set_target(self._h, _ptr(xyz), xyz.shape[1] if xyz.ndim == 2 else 3, _ptr(nrm),
                                             nrm.shape[1] if nrm is not None else 3, _ptr(cov), m, 0))

    def set_target_f64(self, xyz, normals=None, covs=None, crop=None):
        """Target-side preparation on the device (croppers.cpp:76-170 + open3d_conversions.cpp:57-118): fp64 AoS cloud,
        optional cropping volume `crop` = dict(type=, center=, radius_min=, radius_max=, min_z=, max_z=).
        Returns the number of points kept."""
        x = np.ascontiguousarray(xyz, np.float64)
        nr = np.ascontiguousarray(normals, np.float64) if normals is not None else None
        cv = np.ascontiguousarray(covs, np.float64).reshape(-1, 9) if covs is not None else None
        c = self._crop_struct(crop)
        kept = C.c_int64(0)
        st = self._lib.reg_set_target_f64(self._h, _ptr(x), _ptr(nr), _ptr(cv), x.shape[0] if x.ndim == 2 else 0, 0,
                                          C.byref(c) if c is not None else None, C.byref(kept))
        self.n_target_kept = int(kept.value)
        self._check(st)
        return self.n_target_kept

    def set_target_f64_device(self, xyz_ptr, m, nrm_ptr=None, cov_ptr=None, crop=None):
        """As set_target_f64 with the fp64 cloud already resident in HBM (m x 3 doubles; normals m x 3; covs m x 9)."""
        c = self._crop_struct(crop)
        kept = C.c_int64(0)
        st = self._lib.reg_set_target_f64(self._h, C.c_void_p(xyz_ptr), C.c_void_p(nrm_ptr) if nrm_ptr else None,
                                          C.c_void_p(cov_ptr) if cov_ptr else None, m, 1,
                                          C.byref(c) if c is not None else None, C.byref(kept))
        self.n_target_kept = int(kept.value)
        self._check(st)
        return self.n_target_kept

    @staticmethod
    def _crop_struct(crop):
        if crop is None:
            return None
        c = RegCrop()
        c.type = int(crop.get("type", CROP_NONE))
        for k in range(3):
            c.center[k] = float(crop.get("center", (0, 0, 0))[k])
        c.radius_min = float(crop.get("radius_min", 0.0))
        c.radius_max = float(crop.get("radius_max", 0.0))
        c.min_z, c.max_z = float(crop.get("min_z", 0.0)), float(crop.get("max_z", 0.0))
        return c

    def voxelize_within_volume(self, xyz, voxel_size, volume=None, normals=None, covs=None):
        """voxelizeWithinCroppingVolume (helpers.cpp:117-192) on the device.  Returns (xyz, normals, covs, n_outside):
        the first n_outside rows are the untouched points outside `volume`, the rest one averaged point per voxel."""
        x = np.ascontiguousarray(xyz, np.float64)
        m = x.shape[0]
        nr = np.ascontiguousarray(normals, np.float64) if normals is not None else None
        cv = np.ascontiguousarray(covs, np.float64).reshape(-1, 9) if covs is not None else None
        ox = np.empty((m, 3), np.float64)
        on = np.empty((m, 3), np.float64) if nr is not None else None
        oc = np.empty((m, 9), np.float64) if cv is not None else None
        c = self._crop_struct(volume)
        n_out, n_outside = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.reg_voxelize_within_volume(self._h, _ptr(x), _ptr(nr), _ptr(cv), m, 0,
                                                         C.byref(c) if c is not None else None, float(voxel_size), _ptr(ox),
                                                         _ptr(on), _ptr(oc), C.byref(n_out), C.byref(n_outside)))
        k = int(n_out.value)
        return ox[:k], (on[:k] if on is not None else None), (oc[:k] if oc is not None else None), int(n_outside.value)

    def carve_indices(self, map_xyz, scan_xyz, sensor, voxel_size=0.1, max_ray=20.0, truncation=0.1, min_dot=0.5,
                      map_normals=None, subset=None):
        """Space carving (helpers.cpp:238-283): ascending indices of the map points to remove."""
        mp = np.ascontiguousarray(map_xyz, np.float64)
        sp = np.ascontiguousarray(scan_xyz, np.float64)
        nr = np.ascontiguousarray(map_normals, np.float64) if map_normals is not None else None
        out = np.empty(max(mp.shape[0], 1), np.int32)
        sen = (C.c_double * 3)(*[float(v) for v in sensor])
        c = self._crop_struct(subset)
        n = C.c_int64(0)
        self._check(self._lib.reg_carve_indices(self._h, _ptr(mp), _ptr(nr), mp.shape[0], _ptr(sp), sp.shape[0], 0, sen,
                                                C.byref(c) if c is not None else None, float(voxel_size), float(max_ray),
                                                float(truncation), float(min_dot), _ptr(out), C.byref(n)))
        return out[:int(n.value)].copy()

    def target_source_indices(self):
        idx = np.empty(self.n_target_kept, np.int32)
        self._check(self._lib.reg_get_target_source_indices(self._h, _ptr(idx)))
        return idx

    def set_source(self, xyz, normals=None, covs=None):
        xyz = _f32(xyz)
        nrm = _f32(normals) if normals is not None else None
        cov = _f32(covs) if covs is not None else None
        n = xyz.shape[0] if xyz.ndim == 2 else 0
        self._check(self._lib.reg_set_source(self._h, _ptr(xyz), xyz.shape[1] if xyz.ndim == 2 else 3, _ptr(nrm),
                                             nrm.shape[1] if nrm is not None else 3, _ptr(cov), n, 0))
        self.n_source = n

    def set_source_f64(self, xyz, normals=None, covs=None):
        """R11, reading side: Open3D's fp64 arrays (points_ / normals_ n x 3, covariances_ n x 3 x 3) cast on the device
        as open3dToPointmatcher does (open3d_conversions.cpp:57-118)."""
        x = np.ascontiguousarray(xyz, np.float64)
        nr = np.ascontiguousarray(normals, np.float64) if normals is not None else None
        cv = np.ascontiguousarray(covs, np.float64).reshape(-1, 9) if covs is not None else None
        n = x.shape[0] if x.ndim == 2 else 0
        self._check(self._lib.reg_set_source_f64(self._h, _ptr(x), _ptr(nr), _ptr(cv), n, 0))
        self.n_source = n

    def set_source_f64_device(self, xyz_ptr, n, nrm_ptr=None, cov_ptr=None):
        self._check(self._lib.reg_set_source_f64(self._h, C.c_void_p(xyz_ptr), C.c_void_p(nrm_ptr) if nrm_ptr else None,
                                                 C.c_void_p(cov_ptr) if cov_ptr else None, n, 1))
        self.n_source = n

    def estimate_normals(self, xyz, k=10, max_dist=np.inf, viewpoint=None, regularise=False, want_eigvals=False,
                         want_covs=False, want_ids=False, want_eigvecs=False, want_densities=False,
                         want_mean_dists=False):
        """Exact k-NN + PCA normals (SurfaceNormal.cpp:152-252 / CloudRegistration.cpp:25-43).  Returns a dict with
        `normals` (n,3) and, on request, `eigvals` (n,3), `eigvecs` (n,9), `covs` (n,6), `densities` (n,),
        `mean_dists` (n,), `ids` (n,k), plus `n_rescanned`."""
        xyz = _f32(xyz)
        n = xyz.shape[0] if xyz.ndim == 2 else 0
        out = {"normals": np.zeros((n, 3), np.float32)}
        for key, want, shape, dt in (("eigvals", want_eigvals, (n, 3), np.float32), ("eigvecs", want_eigvecs, (n, 9), np.float32),
                                     ("covs", want_covs, (n, 6), np.float32), ("densities", want_densities, (n,), np.float32),
                                     ("mean_dists", want_mean_dists, (n,), np.float32), ("ids", want_ids, (n, k), np.int32)):
            if want:
                out[key] = np.zeros(shape, dt)
        o = NormalsOut()
        for key in ("normals", "eigvals", "eigvecs", "covs", "densities", "mean_dists", "ids"):
            setattr(o, key, out[key].ctypes.data if key in out else None)
        vp_ = _f32(viewpoint) if viewpoint is not None else None
        resc = C.c_int64(0)
        self._check(self._lib.reg_estimate_normals(
            self._h, _ptr(xyz), xyz.shape[1] if xyz.ndim == 2 else 3, n, 0, int(k), float(max_dist), _ptr(vp_),
            1 if regularise else 0, C.byref(o), C.byref(resc)))
        out["n_rescanned"] = int(resc.value)
        return out

    def smooth_normals(self, normals, ids):
        """SurfaceNormalDataPointsFilter's smoothNormals (SurfaceNormal.cpp:259-283) on the device, sequential semantics.
        Returns (smoothed normals (n,3), sweeps launched)."""
        nr = np.ascontiguousarray(normals, dtype=np.float32).copy()
        ii = np.ascontiguousarray(ids, dtype=np.int32)
        passes = C.c_int32(0)
        self._check(self._lib.reg_smooth_normals(self._h, _ptr(nr), _ptr(ii), nr.shape[0], ii.shape[1], 0, C.byref(passes)))
        return nr, int(passes.value)

    def estimate_normals_device(self, xyz_ptr, xyz_stride, n, normals_ptr, k=10, max_dist=np.inf, viewpoint=None,
                                regularise=False, eigvals_ptr=None, covs_ptr=None, ids_ptr=None, eigvecs_ptr=None,
                                densities_ptr=None, mean_dists_ptr=None):
        vp_ = _f32(viewpoint) if viewpoint is not None else None
        o = NormalsOut()
        o.normals, o.eigvals, o.eigvecs, o.covs = normals_ptr, eigvals_ptr, eigvecs_ptr, covs_ptr
        o.densities, o.mean_dists, o.ids = densities_ptr, mean_dists_ptr, ids_ptr
        resc = C.c_int64(0)
        self._check(self._lib.reg_estimate_normals(
            self._h, C.c_void_p(xyz_ptr), xyz_stride, n, 1, int(k), float(max_dist), _ptr(vp_), 1 if regularise else 0,
            C.byref(o), C.byref(resc)))
        return int(resc.value)

    # ---- device-pointer entry points (inputs already resident in HBM) ---------------------------
    def set_target_device(self, xyz_ptr, xyz_stride, m, nrm_ptr=None, nrm_stride=3, cov_ptr=None):
        self._check(self._lib.reg_set_target(self._h, C.c_void_p(xyz_ptr), xyz_stride,
                                             C.c_void_p(nrm_ptr) if nrm_ptr else None, nrm_stride,
                                             C.c_void_p(cov_ptr) if cov_ptr else None, m, 1))

    def set_source_device(self, xyz_ptr, xyz_stride, n, nrm_ptr=None, nrm_stride=3, cov_ptr=None):
        self._check(self._lib.reg_set_source(self._h, C.c_void_p(xyz_ptr), xyz_stride,
                                             C.c_void_p(nrm_ptr) if nrm_ptr else None, nrm_stride,
                                             C.c_void_p(cov_ptr) if cov_ptr else None, n, 1))
        self.n_source = n

    def register(self, T_init=None):
        Ti = _T_in(np.eye(4) if T_init is None else T_init)
        To = np.zeros(16, np.float32)
        res = RegResult()
        st = self._lib.reg_register(self._h, _ptr(Ti), _ptr(To), C.byref(res))
        self.last_result = res
        self._check(st)
        return _T_out(To), res

    def information_matrix(self, T, max_dist):
        """GetInformationMatrixFromPointClouds analogue (constraint_builders.cpp:69-73): (6x6 float64, n_pairs)."""
        Ti = _T_in(T)
        info = (C.c_double * 36)()
        n = C.c_int64(0)
        self._check(self._lib.reg_information_matrix(self._h, _ptr(Ti), float(max_dist), info, C.byref(n)))
        return np.array(info[:], np.float64).reshape(6, 6), int(n.value)

    def prepare(self, T_init=None):
        Ti = _T_in(np.eye(4) if T_init is None else T_init)
        self._check(self._lib.reg_prepare(self._h, _ptr(Ti)))

    def linearize(self, T_iter=None):
        Ti = _T_in(np.eye(4) if T_iter is None else T_iter)
        H = np.zeros(36, np.float32)
        b = np.zeros(6, np.float32)
        err = C.c_double()
        cnt = C.c_int64()
        self._check(self._lib.reg_linearize(self._h, _ptr(Ti), _ptr(H), _ptr(b), C.byref(err), C.byref(cnt)))
        return H.reshape(6, 6), b, err.value, cnt.value

    def correspondences(self, want_w=True):
        n = self.n_source
        ids = np.empty(n, np.int32)
        d2 = np.empty(n, np.float32)
        w = np.empty(n, np.float32) if want_w else None
        self._check(self._lib.reg_get_correspondences(self._h, _ptr(ids), _ptr(d2), _ptr(w)))
        return ids, d2, w

    def target_info(self) -> TargetInfo:
        info = TargetInfo()
        self._check(self._lib.reg_get_target_info(self._h, C.byref(info)))
        return info

    def profile_kernels(self, T_iter=None, reps=20):
        ms = np.zeros(3, np.float32)
        Ti = _T_in(np.eye(4) if T_iter is None else T_iter)
        self._check(self._lib.reg_profile_kernels(self._h, _ptr(Ti), reps, _ptr(ms)))
        return {"match_ms": float(ms[0]), "select_ms": float(ms[1]), "linearize_ms": float(ms[2])}

    # ---- distributed halves ---------------------------------------------------------------------
    def source_centroid_sums(self):
        s = np.zeros(3, np.int64)
        self._check(self._lib.reg_source_centroid_sums(self._h, _ptr(s)))
        return s

    def prepare_centroid(self, T_init, c_read):
        c = np.ascontiguousarray(c_read, np.float32)
        self._check(self._lib.reg_prepare_centroid(self._h, _ptr(_T_in(T_init)), _ptr(c)))

    def compose(self, T_iter):
        To = np.zeros(16, np.float32)
        self._check(self._lib.reg_compose(self._h, _ptr(_T_in(T_iter)), _ptr(To)))
        return _T_out(To)

    # ---- stream-ordered distributed path ----------------------------------------------------------
    def dist_begin(self, T_start=None):
        self._check(self._lib.reg_dist_begin(self._h, _ptr(_T_in(T_start)) if T_start is not None else None))

    def dist_gather_buffers(self, n_ranks, n_max):
        """(d2_local_ptr, d2_all_ptr) of the select-by-gather iteration: n_max and n_ranks * n_max float32."""
        lp, ap = C.c_void_p(), C.c_void_p()
        self._check(self._lib.reg_dist_gather_buffers(self._h, int(n_ranks), int(n_max), C.byref(lp), C.byref(ap)))
        return lp.value, ap.value

    def dist_xicp_buffers(self):
        """(center_ptr, sums_ptr): 4 and 12 float64 to all-reduce after phases 7 and 8 of the first iteration."""
        cp, sp = C.c_void_p(), C.c_void_p()
        self._check(self._lib.reg_dist_xicp_buffers(self._h, C.byref(cp), C.byref(sp)))
        return cp.value, sp.value

    def dist_buffers(self):
        """(hist_ptr, sums_ptr): device addresses of the 3x2048 int32 histograms and the 32 float64 sums."""
        hp, sp = C.c_void_p(), C.c_void_p()
        self._check(self._lib.reg_dist_buffers(self._h, C.byref(hp), C.byref(sp)))
        return hp.value, sp.value

    def dist_fused_buffers(self, n_ranks, rank):
        """(contrib_ptr, gathered_ptr, contrib_bytes) of the fused multi-GPU iteration."""
        cp, gp, nb = C.c_void_p(), C.c_void_p(), C.c_int64()
        self._check(self._lib.reg_dist_fused_buffers(self._h, n_ranks, rank, C.byref(cp), C.byref(gp), C.byref(nb)))
        return cp.value, gp.value, nb.value

    def dist_poll(self):
        st = DistStatus()
        self._check(self._lib.reg_dist_poll(self._h, C.byref(st)))
        return st

    def dist_centroid_sums(self):
        """Enqueues this slice's integer centroid sums; returns the device address of the 3 int64 to all-reduce."""
        p = C.c_void_p()
        self._check(self._lib.reg_dist_centroid_sums(self._h, C.byref(p)))
        return p.value

    def dist_prepare(self, T_init, n_global):
        Ti = _T_in(np.eye(4) if T_init is None else T_init)
        self._check(self._lib.reg_dist_prepare(self._h, _ptr(Ti), int(n_global)))

    def dist_record(self, seq_rel):
        """Report of ONE specific sequence (1-based since dist_begin); .sequences_done == seq_rel when available."""
        st = DistStatus()
        self._check(self._lib.reg_dist_record(self._h, int(seq_rel), C.byref(st)))
        return st

    def dist_phase(self, phase):
        self._check(self._lib.reg_dist_phase(self._h, phase))

    def dist_finish(self):
        To = np.zeros(16, np.float32)
        res = RegResult()
        st = self._lib.reg_dist_finish(self._h, _ptr(To), C.byref(res))
        self.last_result = res
        self._check(st)
        return _T_out(To), res

    # ---- multi-GPU registration behind the C ABI (RCCL, or a custom transport) --------------------------
    def dist_init(self, unique_id: bytes, rank: int, n_ranks: int):
        """ncclCommInitRank on the handle's device; `unique_id` = the 128 bytes of dist_unique_id() of rank 0."""
        buf = C.create_string_buffer(bytes(unique_id), DIST_ID_BYTES)
        self._check(self._lib.reg_dist_init(self._h, buf, int(rank), int(n_ranks)))

    def dist_init_custom(self, all_reduce_sum, all_gather, rank: int, n_ranks: int):
        """Transport given as Python callables (buf_ptr, count, dtype, stream) / (send_ptr, recv_ptr, bytes_per_rank,
        stream) -> 0 on success; both operate on DEVICE memory."""
        self._cb = (ALL_REDUCE_FN(lambda ctx, buf, n, dt, st: int(all_reduce_sum(buf, n, dt, st))),
                    ALL_GATHER_FN(lambda ctx, snd, rcv, nb, st: int(all_gather(snd, rcv, nb, st))))
        c = Collectives(None, self._cb[0], self._cb[1])
        self._check(self._lib.reg_dist_init_custom(self._h, C.byref(c), int(rank), int(n_ranks)))

    def dist_register(self, T_init=None):
        """Collective: == ICP::compute for the reading that is partitioned over the ranks of the group."""
        Ti = _T_in(np.eye(4) if T_init is None else T_init)
        To = np.zeros(16, np.float32)
        res = RegResult()
        st = self._lib.reg_dist_register(self._h, _ptr(Ti), _ptr(To), C.byref(res))
        self.last_result = res
        self._check(st)
        return _T_out(To), res

    def dist_info(self):
        n, g, f, s = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._lib.reg_dist_info(self._h, C.byref(n), C.byref(g), C.byref(f), C.byref(s)))
        return {"n_global": n.value, "n_generic": g.value, "n_fused": f.value, "n_stalls": s.value}

    def dist_shutdown(self):
        self._check(self._lib.reg_dist_shutdown(self._h))

    def match_local(self, T_iter):
        self._check(self._lib.reg_match_local(self._h, _ptr(_T_in(T_iter))))

    def trim_histogram(self, level, prefix=0):
        hist = np.zeros(2048, np.uint32)
        self._check(self._lib.reg_trim_histogram(self._h, level, prefix, _ptr(hist)))
        return hist

    def reduce_local(self, T_iter, trim_limit=math.inf):
        sums = np.zeros(32, np.float64)
        self._check(self._lib.reg_reduce_local(self._h, _ptr(_T_in(T_iter)), trim_limit, _ptr(sums)))
        return sums


def dist_unique_id() -> bytes:
    """ncclGetUniqueId (rank 0); hand the 128 bytes to the other ranks by any means."""
    buf = C.create_string_buffer(DIST_ID_BYTES)
    st = load_library().reg_dist_get_unique_id(buf)
    if st != 0:
        raise RegError(st, "reg_dist_get_unique_id (is librccl loadable?)")
    return buf.raw


class Steer:
    """The steering state machine of reg_dist_register (pure host code in the library: no device needed)."""

    def __init__(self, trimming, fixed_iters, max_iter, settle_tol=0.05, can_fuse=True):
        self._lib = load_library()
        self._s = C.c_void_p(self._lib.reg_dist_steer_create(int(bool(trimming)), int(fixed_iters), int(max_iter),
                                                             float(settle_tol), int(bool(can_fuse))))

    def step(self, reply: "DistReply | None" = None) -> DistAction:
        return self._lib.reg_dist_steer_step(self._s, C.byref(reply) if reply is not None else None)

    def counts(self):
        g, f, s = C.c_int32(), C.c_int32(), C.c_int32()
        self._lib.reg_dist_steer_counts(self._s, C.byref(g), C.byref(f), C.byref(s))
        return g.value, f.value, s.value

    def __del__(self):
        try:
            if self._s:
                self._lib.reg_dist_steer_destroy(self._s)
                self._s = None
        except Exception:
            pass


def solve_update(params: RegParams, sums, T_iter):
    """R8 + T_iter update on the host (identical on every rank)."""
    lib = load_library()
    s = np.ascontiguousarray(sums, np.float64)
    To = np.zeros(16, np.float32)
    rank = C.c_int32()
    st = lib.reg_solve_update(C.byref(params), _ptr(s), _ptr(_T_in(T_iter)), _ptr(To), C.byref(rank))
    if st != 0:
        raise RegError(st, "reg_solve_update")
    return _T_out(To), rank.value


def host_solve6(A, b):
    x = np.zeros(6, np.float32)
    rank = load_library().reg_host_solve6(_ptr(_f32(A).reshape(36)), _ptr(_f32(b)), _ptr(x))
    return x, rank


def host_solve6_xicp(A, b, flags):
    x = np.zeros(6, np.float32)
    f = np.ascontiguousarray(flags, np.int32)
    rank = load_library().reg_host_solve6_xicp(_ptr(_f32(A).reshape(36)), _ptr(_f32(b)), _ptr(f), _ptr(x))
    return x, rank


def host_x_to_T(x):
    T = np.zeros(16, np.float32)
    load_library().reg_host_x_to_T(_ptr(_f32(x)), _ptr(T))
    return _T_out(T)


def host_centroid(xyz):
    xyz = _f32(xyz)
    out = np.zeros(3, np.float32)
    load_library().reg_host_centroid(_ptr(xyz), xyz.shape[1], xyz.shape[0], _ptr(out))
    return out


def host_tail_plan(n, cus=256, tile=0):
    """Launch plan of the persistent tail kernel for an n-point reading: (usable, workgroups, workgroups per XCD class, reading
    points per XCD class) -- host-only (reg_host_tail_plan).  tile: octets per XCD tile (0: contiguous eighths)."""
    lib = load_library()
    lib.reg_host_tail_plan.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    lib.reg_host_tail_plan.restype = None
    out = (C.c_int32 * 4)()
    lib.reg_host_tail_plan(int(n), int(cus), int(tile), out)
    return bool(out[0]), int(out[1]), int(out[2]), int(out[3])
