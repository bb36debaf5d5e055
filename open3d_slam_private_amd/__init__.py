"""MI355X-native scan-to-map registration path for open3d_slam (hot path only).

The package is a thin host-side mirror of the reference's registration interfaces over a
C-ABI shared library (include/o3dslam_reg.h) whose compute is hand-written HIP for gfx950.
There is NO CPU fallback: importing works anywhere (the build check runs on CPU-only hosts),
but every compute entry point raises if the HIP library or a GPU is missing.
"""
from .capi import (RegError, RegParams, RegResult, Registration, TargetInfo, default_params, lib_path, load_library,
                   shipped_params)
from .icp import ICP, DataPoints, RegistrationIcpGeneralized, RegistrationResult, SurfaceNormalDataPointsFilter

__all__ = ["RegError", "RegParams", "RegResult", "Registration", "TargetInfo", "default_params", "shipped_params",
           "lib_path", "load_library", "ICP", "DataPoints", "RegistrationIcpGeneralized", "RegistrationResult",
           "SurfaceNormalDataPointsFilter"]
