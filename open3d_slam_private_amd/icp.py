"""Host-side mirror of the reference's registration interfaces for the hot path.

  ICP                          <-> PointMatcher<float>::ICP   (libpointmatcher/pointmatcher/PointMatcher.h:1023-1060,
                                   ICP.cpp:793-898; configured like ICPChainBase::setDefault / loadFromYaml,
                                   ICP.cpp:100-210), as used by o3d_slam::Mapper (Mapper.cpp:343,372-373)
  RegistrationIcpGeneralized   <-> o3d_slam::RegistrationIcpGeneralized::registerClouds
                                   (open3d_slam/src/CloudRegistration.cpp:16-21, CloudRegistration.hpp:19-73)

Same method names, argument meaning and error behaviour (exceptions named after the reference's);
all compute goes through the C ABI (capi.Registration) to the HIP kernels -- nothing is computed here.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from . import capi
from .capi import RegError, RegParams


class ConvergenceError(RuntimeError):
    """PointMatcher<T>::ConvergenceError (ErrorMinimizer.cpp:75-77, Matches.cpp:76-80)."""


class InvalidField(RuntimeError):
    """DataPoints::InvalidField (DataPoints.cpp:1112)."""


class InvalidModuleType(RuntimeError):
    """PointMatcherSupport::InvalidModuleType (ICP.cpp:203-209)."""


class InvalidParameter(RuntimeError):
    """Parametrizable::InvalidParameter (Registrar.h:103-109)."""


@dataclass
class DataPoints:
    """The fields of PointMatcher<float>::DataPoints the path touches (PointMatcher.h:222-403):
    `features` N x 4 ({x,y,z,1} per point == the column-major 4 x N Eigen matrix in memory) or N x 3,
    descriptor `normals` N x 3."""
    features: np.ndarray
    normals: np.ndarray | None = None
    covariances: np.ndarray | None = None

    def getNbPoints(self) -> int:
        return 0 if self.features is None else int(np.asarray(self.features).shape[0])


def _translate(e: RegError):
    if e.status == 3:
        return ConvergenceError(str(e))
    if e.status == 7:
        return InvalidField(str(e))
    if e.status == 6:
        return InvalidParameter(str(e))
    return RuntimeError(str(e))


_KNOWN_TOP = {"readingDataPointsFilters", "referenceDataPointsFilters", "readingStepDataPointsFilters", "matcher",
              "outlierFilters", "errorMinimizer", "transformationCheckers", "inspector", "logger",
              "degeneracyDebug", "printingDegeneracy", "ceresDegeneracyAnalysis", "degeneracyAwareness"}


class ICP:
    """Drop-in for the `icp_` member of o3d_slam::Mapper: initReference() once per map refresh,
    compute() per scan."""

    def __init__(self):
        self.params: RegParams | None = None
        self._reg: capi.Registration | None = None
        self.matcherIsInitialized = False
        self.maxNumIterationsReached = False
        self.last_result = None

    # -- configuration --------------------------------------------------------------------------
    def setDefault(self):
        """ICPChainBase::setDefault (ICP.cpp:100-113)."""
        self.params = capi.default_params()
        self._reg = None
        self.matcherIsInitialized = False

    def loadFromYaml(self, stream_or_text):
        """The hot-path subset of ICPChainBase::loadFromYaml (ICP.cpp:116-210)."""
        import yaml
        text = stream_or_text.read() if hasattr(stream_or_text, "read") else stream_or_text
        doc = yaml.safe_load(text) or {}
        for key in doc:
            if key not in _KNOWN_TOP:
                raise InvalidModuleType(f"Module type {key} does not exist")
        p = capi.default_params()
        p.use_trimmed = 0
        p.max_iter = 1 << 30
        p.smooth_len = 0
        for name in ("readingDataPointsFilters", "referenceDataPointsFilters", "readingStepDataPointsFilters"):
            if doc.get(name):
                raise NotImplementedError(f"{name}: data-point filters are outside the accelerated path "
                                          "(the shipped icp.yaml leaves these chains empty)")
        m = doc.get("matcher")
        if m:
            (mname, margs), = (m.items() if isinstance(m, dict) else [(m, {})])
            if mname != "KDTreeMatcher":
                raise NotImplementedError(f"matcher {mname}")
            margs = margs or {}
            p.knn = int(margs.get("knn", 1))
            p.max_dist = float(margs.get("maxDist", math.inf))
            p.epsilon = float(margs.get("epsilon", 0.0))
        for f in doc.get("outlierFilters") or []:
            (fname, fargs), = (f.items() if isinstance(f, dict) else [(f, {})])
            fargs = fargs or {}
            if fname == "TrimmedDistOutlierFilter":
                p.use_trimmed, p.trim_ratio = 1, float(fargs.get("ratio", 0.85))
            elif fname == "SurfaceNormalOutlierFilter":
                p.use_surface_normal, p.max_normal_angle = 1, float(fargs.get("maxAngle", 1.57))
            elif fname == "MaxDistOutlierFilter":
                p.use_max_dist_filter, p.outlier_max_dist = 1, float(fargs.get("maxDist", 1.0))
            elif fname == "NullOutlierFilter":
                pass
            else:
                raise NotImplementedError(f"outlier filter {fname}")
        em = doc.get("errorMinimizer", "PointToPlaneErrorMinimizer")
        emname = next(iter(em)) if isinstance(em, dict) else em
        if emname != "PointToPlaneErrorMinimizer":
            raise NotImplementedError(f"errorMinimizer {emname}")
        for c in doc.get("transformationCheckers") or []:
            (cname, cargs), = (c.items() if isinstance(c, dict) else [(c, {})])
            cargs = cargs or {}
            if cname == "CounterTransformationChecker":
                p.max_iter = int(cargs.get("maxIterationCount", 40))
            elif cname == "DifferentialTransformationChecker":
                p.min_diff_rot = float(cargs.get("minDiffRotErr", 0.001))
                p.min_diff_trans = float(cargs.get("minDiffTransErr", 0.001))
                p.smooth_len = int(cargs.get("smoothLength", 3))
            else:
                raise NotImplementedError(f"transformation checker {cname}")
        da = doc.get("degeneracyAwareness")
        if da:
            # ICPChainBase::loadAdditionalYAMLContent (ICP.cpp:575-790): only the shipped method is accelerated
            (dname, dargs), = (da.items() if isinstance(da, dict) else [(da, {})])
            dargs = dargs or {}
            if dname == "OptimizedEqualityConstraints":
                need = ("enoughInformationThreshold", "insufficientInformationThreshold",
                        "point2NormalMinimalAlignmentAngleThreshold", "point2NormalStrongAlignmentAngleThreshold")
                if any(k not in dargs for k in need):
                    raise InvalidParameter("OptimizedEqualityConstraints needs " + ", ".join(need))   # ICP.cpp:632-672
                p.use_xicp = 1
                p.xicp_enough = float(dargs[need[0]])
                p.xicp_insufficient = float(dargs[need[1]])
                p.xicp_min_angle_deg = float(dargs[need[2]])
                p.xicp_strong_angle_deg = float(dargs[need[3]])
            elif dname in ("None", "none", "kNone"):
                pass
            else:
                raise NotImplementedError(f"degeneracyAwareness method {dname}")
        if p.knn != 1:
            raise InvalidParameter("knn must be 1 on the accelerated path")
        self.params = p
        self._reg = None
        self.matcherIsInitialized = False

    # -- the two calls Mapper makes ----------------------------------------------------------------
    def _ensure(self):
        if self.params is None:
            raise RuntimeError("You must setup a matcher before running ICP")  # ICP.cpp:819-824
        if self._reg is None:
            self._reg = capi.Registration(self.params)

    def hasMap(self) -> bool:
        return self.matcherIsInitialized

    def initReference(self, referenceIn: DataPoints) -> bool:
        """ICP::initReference (ICP.cpp:847-898).  Returns False on an empty reference."""
        self._ensure()
        if referenceIn.getNbPoints() == 0:
            print("The reference point cloud is empty. (libpointmatcher)")
            self.matcherIsInitialized = False
            return False
        try:
            self._reg.set_target(referenceIn.features, referenceIn.normals, referenceIn.covariances)
        except RegError as e:
            raise _translate(e) from None
        self.matcherIsInitialized = True
        return True

    def compute(self, readingIn: DataPoints, referenceIn: DataPoints | None = None, T_refIn_readIn=None,
                initializeMatcherWithInputReference: bool = True) -> np.ndarray:
        """ICP::compute (ICP.cpp:813-844)."""
        self._ensure()
        if initializeMatcherWithInputReference or not self.matcherIsInitialized:
            if referenceIn is None or not self.initReference(referenceIn):
                return np.eye(4, dtype=np.float32)
        T = np.eye(4, dtype=np.float32) if T_refIn_readIn is None else np.asarray(T_refIn_readIn, np.float32)
        if T.shape != (4, 4):
            raise RuntimeError("The initial transformation matrix must be squared.")  # ICP.cpp:910-918
        if readingIn.getNbPoints() == 0:
            raise RuntimeError("The reading point cloud is empty.")  # ICP.cpp:958-960
        try:
            self._reg.set_source(readingIn.features, readingIn.normals, readingIn.covariances)
            T_out, res = self._reg.register(T)
        except RegError as e:
            raise _translate(e) from None
        self.last_result = res
        self.maxNumIterationsReached = bool(res.max_iter_reached)
        return T_out

    def __call__(self, readingIn, referenceIn, T_refIn_readIn=None):
        return self.compute(readingIn, referenceIn, T_refIn_readIn, True)


@dataclass
class SurfaceNormalDataPointsFilter:
    """SurfaceNormalDataPointsFilter (DataPointsFilters/SurfaceNormal.h:68-78, SurfaceNormal.cpp:152-252) on the
    device: parameters by the reference's names; `epsilon` must be 0 (the search is exact).  `filter` returns a
    DataPoints with the `normals` descriptor added (keepNormals) and keeps `eigValues` / `matchedIds` on the filter."""

    def __init__(self, knn=5, maxDist=float("inf"), epsilon=0.0, keepNormals=True, keepDensities=False,
                 keepEigenValues=False, keepEigenVectors=False, keepMatchedIds=False, keepMeanDist=False, viewpoint=None,
                 smoothNormals=False):
        if knn < 3:
            raise InvalidParameter("knn: minimum 3 (SurfaceNormal.h:68)")
        if knn > 32:
            raise InvalidParameter("knn: this build supports at most 32 neighbours")
        if epsilon != 0.0:
            raise InvalidParameter("epsilon: only the exact search (0) is implemented")
        if not (maxDist > 0):
            raise InvalidParameter("maxDist: must be > 0")
        self.knn, self.maxDist, self.keepNormals = int(knn), float(maxDist), keepNormals
        self.keepEigenValues, self.keepMatchedIds, self.viewpoint = keepEigenValues, keepMatchedIds, viewpoint
        self.keepDensities, self.keepEigenVectors, self.keepMeanDist = keepDensities, keepEigenVectors, keepMeanDist
        self.eigValues = self.matchedIds = self.densities = self.eigVectors = self.meanDists = None
        self.smoothNormals = bool(smoothNormals)   # SurfaceNormal.cpp:259-283 (in place, index order)
        self._reg = None

    def filter(self, cloud: DataPoints) -> DataPoints:
        if self._reg is None:
            self._reg = capi.Registration(capi.default_params())
        try:
            out = self._reg.estimate_normals(cloud.features, k=self.knn, max_dist=self.maxDist, viewpoint=self.viewpoint,
                                             want_eigvals=self.keepEigenValues, want_ids=self.keepMatchedIds or self.smoothNormals,
                                             want_densities=self.keepDensities, want_eigvecs=self.keepEigenVectors,
                                             want_mean_dists=self.keepMeanDist)
            if self.smoothNormals:
                out["normals"], _ = self._reg.smooth_normals(out["normals"], out["ids"])
        except RegError as e:
            raise _translate(e) from None
        self.eigValues, self.matchedIds = out.get("eigvals"), (out.get("ids") if self.keepMatchedIds else None)
        self.densities, self.eigVectors, self.meanDists = out.get("densities"), out.get("eigvecs"), out.get("mean_dists")
        return DataPoints(cloud.features, out["normals"] if self.keepNormals else cloud.normals, cloud.covariances)


class RegistrationResult:
    """open3d::pipelines::registration::RegistrationResult fields consumed by the reference
    (Odometry.cpp:56,77, PlaceRecognition.cpp:118)."""
    transformation_: np.ndarray = field(default_factory=lambda: np.eye(4))
    fitness_: float = 0.0
    inlier_rmse_: float = 0.0
    correspondence_set_: np.ndarray | None = None


class RegistrationIcpGeneralized:
    """o3d_slam::RegistrationIcpGeneralized (CloudRegistration.hpp:59-68, CloudRegistration.cpp:16-21)."""

    def __init__(self, maxCorrespondenceDistance_=1.0, max_iteration_=30):
        self.maxCorrespondenceDistance_ = maxCorrespondenceDistance_
        self.max_iteration_ = max_iteration_
        self.relative_fitness_ = 1e-6   # open3d::pipelines::registration::ICPConvergenceCriteria defaults
        self.relative_rmse_ = 1e-6
        self._reg = None

    def registerClouds(self, source: DataPoints, target: DataPoints, init=None) -> RegistrationResult:
        p = capi.default_params()
        p.cost = capi.COST_GICP
        p.use_trimmed = 0
        p.max_dist = self.maxCorrespondenceDistance_
        p.max_iter = self.max_iteration_
        # icpConvergenceCriteria_: only max_iteration_ is configured (CloudRegistration.cpp:45-52), so Open3D's defaults
        # relative_fitness_ = relative_rmse_ = 1e-6 end the loop
        p.gicp_stop_rule = 1
        p.gicp_rel_fitness = self.relative_fitness_
        p.gicp_rel_rmse = self.relative_rmse_
        reg = capi.Registration(p)
        try:
            reg.set_target(target.features, None, target.covariances)
            reg.set_source(source.features, None, source.covariances)
            T, res = reg.register(np.eye(4) if init is None else init)
            ids, _, _ = reg.correspondences(want_w=False)
        except RegError as e:
            raise _translate(e) from None
        finally:
            pass
        sel = np.nonzero(ids >= 0)[0]
        out = RegistrationResult(T.astype(np.float64), float(res.fitness), float(res.inlier_rmse),
                                 np.stack([sel, ids[sel]], axis=1))
        reg.close()
        return out
