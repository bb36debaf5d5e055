// o3dslam_icp.hpp -- header-only C++ host-side mirror of the reference's registration object for the hot
// path, over the C ABI (o3dslam_reg.h).  Same method names, argument meaning and error behaviour as
//   PointMatcher<float>::ICP          libpointmatcher/pointmatcher/PointMatcher.h:1023-1060, ICP.cpp:793-898
// as used by o3d_slam::Mapper (open3d_slam/src/Mapper.cpp:343,372-373).  No Eigen dependency: clouds are
// passed as views on the caller's memory, which for a PointMatcher<float>::DataPoints is
//   DataPointsView{ dp.features.data(), dp.features.rows() /*4*/, dp.getNbPoints(),
//                   dp.getDescriptorViewByName("normals").data(), 3 }      (column-major Eigen == AoS per point)
// and transforms as column-major float[16] == Eigen::Matrix4f::data().
#pragma once
#include <array>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "o3dslam_reg.h"

namespace o3dreg {

// exception names follow the reference (PointMatcher.h:130-160, DataPoints.h)
struct ConvergenceError : std::runtime_error { using std::runtime_error::runtime_error; };
struct InvalidField : std::runtime_error { using std::runtime_error::runtime_error; };
struct InvalidParameter : std::runtime_error { using std::runtime_error::runtime_error; };
struct DeviceError : std::runtime_error { using std::runtime_error::runtime_error; };

struct DataPointsView {
    const float* features = nullptr;   // {x,y,z,1} per point (stride 4) or packed xyz (stride 3)
    int64_t feature_stride = 4;
    int64_t n = 0;
    const float* normals = nullptr;    // descriptor "normals", 3 per point
    int64_t normal_stride = 3;
    const float* covariances = nullptr;  // GICP only, 6 per point
    bool on_device = false;            // pointers are HIP device pointers (already resident in HBM)
    int64_t getNbPoints() const { return n; }
};

using TransformationParameters = std::array<float, 16>;  // column-major 4x4

inline TransformationParameters identity4() {
    TransformationParameters T{};
    T[0] = T[5] = T[10] = T[15] = 1.f;
    return T;
}

class ICP {
public:
    ICP() { reg_default_params(&params_); }
    ~ICP() { if (h_) reg_destroy(h_); }
    ICP(const ICP&) = delete;
    ICP& operator=(const ICP&) = delete;

    // ICPChainBase::setDefault (ICP.cpp:100-113)
    void setDefault() { reg_default_params(&params_); reset(); }
    // the chain of open3d_slam_ros/param/icp.yaml (what Mapper loads through loadFromYaml)
    void setShippedChain() { reg_shipped_params(&params_); reset(); }
    // direct access to the string-free parameter block (call before the first initReference)
    reg_params& parameters() { reset(); return params_; }

    bool hasMap() const { return matcherIsInitialized_; }
    bool getMaxNumIterationsReached() const { return last_.max_iter_reached != 0; }
    const reg_result& lastResult() const { return last_; }

    // ICP::initReference (ICP.cpp:847-898): false on an empty reference.
    bool initReference(const DataPointsView& referenceIn) {
        ensure();
        if (referenceIn.getNbPoints() == 0) { matcherIsInitialized_ = false; return false; }
        check(reg_set_target(h_, referenceIn.features, referenceIn.feature_stride, referenceIn.normals,
                             referenceIn.normal_stride, referenceIn.covariances, referenceIn.n,
                             referenceIn.on_device ? 1 : 0));
        matcherIsInitialized_ = true;
        return true;
    }

    // ICP::compute (ICP.cpp:813-844)
    TransformationParameters compute(const DataPointsView& readingIn, const DataPointsView& referenceIn,
                                     const TransformationParameters& T_refIn_readIn,
                                     bool initializeMatcherWithInputReference = true) {
        ensure();
        if (initializeMatcherWithInputReference || !matcherIsInitialized_)
            if (!initReference(referenceIn)) return identity4();
        if (readingIn.getNbPoints() == 0) throw std::runtime_error("The reading point cloud is empty.");
        TransformationParameters out = T_refIn_readIn;
        check(reg_compute(h_, readingIn.features, readingIn.feature_stride, readingIn.normals,
                          readingIn.normal_stride, readingIn.covariances, readingIn.n, readingIn.on_device ? 1 : 0,
                          T_refIn_readIn.data(), out.data(), &last_));
        return out;
    }
    TransformationParameters operator()(const DataPointsView& readingIn, const DataPointsView& referenceIn) {
        return compute(readingIn, referenceIn, identity4(), true);
    }

    // The scan as Mapper::addRangeMeasurement holds it before open3dToPointmatcher (Mapper.cpp:288-289): Open3D's fp64
    // arrays (points_ / normals_ n x 3 doubles).  Cast on the device (open3d_conversions.cpp:57-118), then compute().
    TransformationParameters computeF64(const double* points, const double* normals, int64_t n, bool on_device,
                                        const TransformationParameters& T_refIn_readIn) {
        ensure();
        if (!matcherIsInitialized_) throw std::runtime_error("You must call initReference first");
        if (n == 0) throw std::runtime_error("The reading point cloud is empty.");
        check(reg_set_source_f64(h_, points, normals, nullptr, n, on_device ? 1 : 0));
        TransformationParameters out = T_refIn_readIn;
        check(reg_register(h_, T_refIn_readIn.data(), out.data(), &last_));
        return out;
    }

    // ---- one process per GPU, reading partitioned over the group (BASELINE config C4) --------------------------------
    // Rank 0 creates the 128-byte id (ICP::makeGroupId) and hands it to the other ranks; every rank joins with its rank.
    static std::array<char, REG_DIST_ID_BYTES> makeGroupId() {
        std::array<char, REG_DIST_ID_BYTES> id{};
        if (reg_dist_get_unique_id(id.data()) != REG_OK) throw DeviceError("ncclGetUniqueId failed (is librccl loadable?)");
        return id;
    }
    void joinGroup(const std::array<char, REG_DIST_ID_BYTES>& id, int rank, int n_ranks) {
        ensure();
        check(reg_dist_init(h_, id.data(), rank, n_ranks));
    }
    void leaveGroup() { if (h_) check(reg_dist_shutdown(h_)); }
    // ICP::compute for THIS RANK'S SLICE of the reading; collective: every rank calls it with the same T and returns the
    // same transform (lastResult() carries the global figures).  The reference is the one given to initReference on
    // every rank (replicated).
    TransformationParameters computePartitioned(const DataPointsView& readingSlice,
                                                const TransformationParameters& T_refIn_readIn) {
        ensure();
        if (!matcherIsInitialized_) throw std::runtime_error("You must call initReference first");
        if (readingSlice.getNbPoints() == 0) throw std::runtime_error("The reading point cloud is empty.");
        check(reg_set_source(h_, readingSlice.features, readingSlice.feature_stride, readingSlice.normals,
                             readingSlice.normal_stride, readingSlice.covariances, readingSlice.n,
                             readingSlice.on_device ? 1 : 0));
        TransformationParameters out = T_refIn_readIn;
        check(reg_dist_register(h_, T_refIn_readIn.data(), out.data(), &last_));
        return out;
    }

private:
    void reset() { if (h_) { reg_destroy(h_); h_ = nullptr; } matcherIsInitialized_ = false; }
    void ensure() {
        if (h_) return;
        params_.struct_size = (int32_t)sizeof(reg_params);
        reg_status s = reg_create(&params_, &h_);
        if (s != REG_OK) {
            std::string msg = h_ ? reg_last_error(h_) : "reg_create rejected the parameters";
            if (h_) { reg_destroy(h_); h_ = nullptr; }
            if (s == REG_DEVICE_ERROR) throw DeviceError(msg);
            throw InvalidParameter(msg);
        }
    }
    void check(reg_status s) {
        if (s == REG_OK) return;
        const std::string msg = reg_last_error(h_);
        switch (s) {
            case REG_NO_CORRESPONDENCES: throw ConvergenceError(msg);
            case REG_MISSING_FIELD: throw InvalidField(msg);
            case REG_BAD_ARGUMENT: throw InvalidParameter(msg);
            case REG_DEVICE_ERROR: throw DeviceError(msg);
            default: throw std::runtime_error(msg);
        }
    }
    reg_params params_;
    reg_handle* h_ = nullptr;
    reg_result last_{};
    bool matcherIsInitialized_ = false;
};

// SurfaceNormalDataPointsFilter (DataPointsFilters/SurfaceNormal.cpp:152-252) on the device: exact k-NN (the point
// itself included) + PCA.  Outputs are written to caller-owned arrays laid out like the `normals` (3 x N),
// `eigValues` (3 x N, ascending == sortEigen) and `matchedIds` (knn x N) descriptors.
class SurfaceNormalFilter {
public:
    unsigned knn = 5;                                             // SurfaceNormal.h:68
    float maxDist = std::numeric_limits<float>::infinity();      // SurfaceNormal.h:69
    bool smoothNormals = false;                                   // SurfaceNormal.h:76, SurfaceNormal.cpp:259-283
    bool orientTowardsViewpoint = false;                          // CloudRegistration.cpp:37 (camera location)
    std::array<float, 3> viewpoint{{0.f, 0.f, 0.f}};

    SurfaceNormalFilter() = default;
    ~SurfaceNormalFilter() { if (h_) reg_destroy(h_); }
    SurfaceNormalFilter(const SurfaceNormalFilter&) = delete;
    SurfaceNormalFilter& operator=(const SurfaceNormalFilter&) = delete;

    // optional outputs by the reference's descriptor names: eigValues 3xN (ascending), matchedIds knn x N, densities N,
    // meanDists N, eigVectors 9xN; covariances6 {xx xy xz yy yz zz} for the GICP operator
    void compute(const DataPointsView& cloud, float* normals, float* eigValues = nullptr, int32_t* matchedIds = nullptr,
                 float* covariances6 = nullptr, bool regularisedCovariances = false, float* densities = nullptr,
                 float* meanDists = nullptr, float* eigVectors = nullptr) {
        if (!h_) {
            reg_params p;
            reg_default_params(&p);
            reg_status s = reg_create(&p, &h_);
            if (s != REG_OK) {
                std::string msg = h_ ? reg_last_error(h_) : "reg_create failed";
                if (h_) { reg_destroy(h_); h_ = nullptr; }
                throw DeviceError(msg);
            }
        }
        if (cloud.getNbPoints() == 0) throw std::runtime_error("The point cloud is empty.");
        reg_normals_out out{};
        out.normals = normals;
        out.eigvals = eigValues;
        out.covs = covariances6;
        std::vector<int32_t> own_ids;   // smoothNormals needs the neighbour lists even when the caller does not ask for them
        if (smoothNormals && !matchedIds && !cloud.on_device) {
            own_ids.resize((size_t)cloud.n * knn);
            matchedIds = own_ids.data();
        }
        if (smoothNormals && !matchedIds)
            throw InvalidParameter("smoothNormals on device buffers needs a matchedIds buffer (n x knn int32)");
        out.ids = matchedIds;
        out.densities = densities;
        out.mean_dists = meanDists;
        out.eigvecs = eigVectors;
        const reg_status s = reg_estimate_normals(h_, cloud.features, cloud.feature_stride, cloud.n,
                                                  cloud.on_device ? 1 : 0, (int)knn, maxDist,
                                                  orientTowardsViewpoint ? viewpoint.data() : nullptr,
                                                  regularisedCovariances ? 1 : 0, &out, nullptr);
        if (s == REG_BAD_ARGUMENT) throw InvalidParameter(reg_last_error(h_));
        if (s == REG_DEVICE_ERROR) throw DeviceError(reg_last_error(h_));
        if (s != REG_OK) throw std::runtime_error(reg_last_error(h_));
        if (smoothNormals) {
            const reg_status t = reg_smooth_normals(h_, normals, matchedIds, cloud.n, (int)knn, cloud.on_device ? 1 : 0, nullptr);
            if (t == REG_DEVICE_ERROR) throw DeviceError(reg_last_error(h_));
            if (t != REG_OK) throw std::runtime_error(reg_last_error(h_));
        }
    }

private:
    reg_handle* h_ = nullptr;
};

}  // namespace o3dreg
