// map_optimization_s2m.hpp — C++ host-side mirror of the reference's mapOptimization members
// and methods that belong to the scan-to-map path, on top of the C ABI (include/liorf_s2m.h).
//
// The reference keeps this path's state as members of `class mapOptimization`
// (src/mapOptmization.cpp:87-233) and runs it from laserCloudInfoHandler() (:236-275, step :265).
// This class carries the same names with the same meaning so that the node's handler can keep
// its shape: fill laserCloudSurfFromMapDS / laserCloudSurfLastDS / transformTobeMapped / cloudInfo,
// call scan2MapOptimization(), read transformTobeMapped / isDegenerate /
// incrementalOdometryAffineBack.  All arithmetic runs in the HIP library; there is no CPU path here.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/liorf_s2m.h"

namespace liorf_amd {

// pcl::PointXYZI layout (reference `typedef pcl::PointXYZI PointType`, include/utility.h:61):
// x,y,z,pad at 0..15, intensity at 16, padded to 32 bytes.
struct alignas(16) PointXYZI {
    float x, y, z, pad0;
    float intensity, pad1, pad2, pad3;
};
static_assert(sizeof(PointXYZI) == 32, "PointXYZI must match pcl::PointXYZI");

// The fields of liorf::cloud_info this path reads (reference msg/cloud_info.msg:10-16).
struct CloudInfo {
    int64_t imuAvailable = 0;
    int64_t odomAvailable = 0;
    float imuRollInit = 0, imuPitchInit = 0, imuYawInit = 0;
    float initialGuessX = 0, initialGuessY = 0, initialGuessZ = 0;
    float initialGuessRoll = 0, initialGuessPitch = 0, initialGuessYaw = 0;
};

class MapOptimizationS2M {
public:
    // reference members, same names -----------------------------------------------------------
    std::vector<PointXYZI> laserCloudSurfFromMapDS;   // local surf map, voxel-filtered (:124)
    std::vector<PointXYZI> laserCloudSurfLastDS;      // current scan, voxel-filtered (:108)
    int   laserCloudSurfLastDSNum = 0;                // (:131)
    float transformTobeMapped[6] = { 0, 0, 0, 0, 0, 0 };   // roll,pitch,yaw,x,y,z (:134)
    bool  isDegenerate = false;                       // (:139)
    float incrementalOdometryAffineBack[12] = { 0 };  // row-major 3x4 (:157)
    CloudInfo cloudInfo;                              // (:99)
    bool  haveKeyPoses = false;                       // !cloudKeyPoses3D->points.empty() (:1297)
    // ParamServer values the path reads (include/utility.h:211-233)
    int   imuType = 0;
    float imuRPYWeight = 0.01f, z_tollerance = 3.4028235e38f, rotation_tollerance = 3.4028235e38f;
    // what the last call did
    s2m_result lastResult{};

    explicit MapOptimizationS2M(int device_id = 0, void* hip_stream = nullptr)
    {
        s2m_params p;
        s2m_default_params(&p);
        p.device_id = device_id;
        p.stream = hip_stream;
        p.imu_type = imuType; p.imu_rpy_weight = imuRPYWeight;
        p.z_tol = z_tollerance; p.rot_tol = rotation_tollerance;
        const int rc = s2m_create(&p, &h_);
        if (rc != S2M_OK)
            throw std::runtime_error("s2m_create failed (" + std::to_string(rc) + "): no gfx950 device; there is no CPU fallback");
    }
    ~MapOptimizationS2M() { if (h_) s2m_destroy(h_); }
    MapOptimizationS2M(const MapOptimizationS2M&) = delete;
    MapOptimizationS2M& operator=(const MapOptimizationS2M&) = delete;

    // kdtreeSurfFromMap->setInputCloud(laserCloudSurfFromMapDS) (:1302), hoisted so that a map
    // that did not change between scans is not rebuilt
    void setInputCloud()
    {
        check(s2m_set_map(h_, laserCloudSurfFromMapDS.data(), haveKeyPoses ? laserCloudSurfFromMapDS.size() : 0,
                          sizeof(PointXYZI)), "s2m_set_map");
    }

    // void scan2MapOptimization() (:1295-1321)
    void scan2MapOptimization()
    {
        laserCloudSurfLastDSNum = (int)laserCloudSurfLastDS.size();
        s2m_imu_init imu;
        imu.imuAvailable = cloudInfo.imuAvailable;
        imu.imuRollInit = cloudInfo.imuRollInit; imu.imuPitchInit = cloudInfo.imuPitchInit; imu.imuYawInit = cloudInfo.imuYawInit;
        check(s2m_optimize(h_, laserCloudSurfLastDS.data(), laserCloudSurfLastDS.size(), sizeof(PointXYZI),
                           transformTobeMapped, &imu, &lastResult), "s2m_optimize");
        if (lastResult.skipped == 2) {
            // ROS_WARN("Not enough features! Only %d planar features available.", ...) (:1319)
            return;
        }
        if (lastResult.skipped == 0) {
            isDegenerate = lastResult.is_degenerate != 0;
            std::memcpy(incrementalOdometryAffineBack, lastResult.affine, sizeof(incrementalOdometryAffineBack));
        }
    }

    s2m_handle handle() const { return h_; }

private:
    void check(int rc, const char* what)
    {
        if (rc != S2M_OK) throw std::runtime_error(std::string(what) + ": " + s2m_last_error(h_));
    }
    s2m_handle h_ = nullptr;
};

}  // namespace liorf_amd
