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
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
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

// PointTypePose (reference include/utility.h:66-83): key pose with the id in `intensity`.
struct PointTypePose {
    float x = 0, y = 0, z = 0, intensity = 0;
    float roll = 0, pitch = 0, yaw = 0;
    double time = 0;
};

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
    int   laserCloudSurfFromMapDSNum = 0;             // (:130)
    std::vector<PointXYZI> laserCloudSurfLast;        // current scan before the filter (:107)
    std::vector<std::vector<PointXYZI>> surfCloudKeyFrames;   // key-frame clouds (:93)
    std::vector<PointTypePose> cloudKeyPoses6D;       // (:96)
    float mappingSurfLeafSize = 0.2f;                 // include/utility.h:227 (the shipped yaml files set 0.4)
    float surroundingKeyframeMapLeafSize = 0.2f;      // include/utility.h:228
    float surroundingKeyframeSearchRadius = 50.0f;    // include/utility.h:240
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
        const int rc = s2m_create(&p, &h_);       // the ParamServer members below are pushed before every call (pushParams)
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

    // void downsampleCurrentScan() (:1061-1067): laserCloudSurfLast -> laserCloudSurfLastDS on the device;
    // the filtered scan stays resident for scan2MapOptimization() and is copied back for the key-frame store
    void downsampleCurrentScan()
    {
        laserCloudSurfLastDS.resize(laserCloudSurfLast.size());
        size_t n_out = 0;
        checkVoxel(s2m_downsample_scan(h_, laserCloudSurfLast.data(), laserCloudSurfLast.size(), sizeof(PointXYZI), 0,
                                       mappingSurfLeafSize, laserCloudSurfLastDS.data(), sizeof(PointXYZI),
                                       laserCloudSurfLastDS.size(), &n_out), "s2m_downsample_scan");
        laserCloudSurfLastDS.resize(n_out);
        laserCloudSurfLastDSNum = (int)n_out;
        scanResident_ = true;
    }

    // void extractCloud(cloudToExtract) (:1014-1039): `keyInds` are the key-frame ids the caller's radius
    // search and time filter chose (extractNearby, :973-1012). Frames farther than
    // surroundingKeyframeSearchRadius from the newest key pose are dropped (:1018), the rest are transformed
    // by their key pose, concatenated, voxel-filtered and indexed on the device; there is no
    // laserCloudMapContainer cache to maintain.
    void extractCloud(const std::vector<int>& keyInds)
    {
        std::vector<const void*> frames;
        std::vector<size_t> sizes;
        std::vector<float> poses;
        size_t total = 0;
        if (!cloudKeyPoses6D.empty()) {
            const PointTypePose& last = cloudKeyPoses6D.back();
            for (int k : keyInds) {
                const PointTypePose& p = cloudKeyPoses6D[(size_t)k];
                const float dx = p.x - last.x, dy = p.y - last.y, dz = p.z - last.z;
                if (std::sqrt(dx * dx + dy * dy + dz * dz) > surroundingKeyframeSearchRadius) continue;
                frames.push_back(surfCloudKeyFrames[(size_t)k].data());
                sizes.push_back(surfCloudKeyFrames[(size_t)k].size());
                const float v[6] = { p.x, p.y, p.z, p.roll, p.pitch, p.yaw };
                poses.insert(poses.end(), v, v + 6);
                total += sizes.back();
            }
        }
        laserCloudSurfFromMapDS.resize(total);
        size_t n_out = 0;
        checkVoxel(s2m_extract_cloud(h_, (int)frames.size(), frames.data(), sizes.data(), sizeof(PointXYZI), 0, poses.data(),
                                     surroundingKeyframeMapLeafSize, laserCloudSurfFromMapDS.data(), sizeof(PointXYZI),
                                     total, &n_out), "s2m_extract_cloud");
        laserCloudSurfFromMapDS.resize(n_out);
        laserCloudSurfFromMapDSNum = (int)n_out;
        haveKeyPoses = !cloudKeyPoses6D.empty();
    }

    // The reference reads imuType / imuRPYWeight / z_tollerance / rotation_tollerance (ParamServer members, set from
    // the yaml after construction) whenever transformUpdate() runs (:1325-1350): the current member values are
    // handed to the library before every registration.
    void pushParams()
    {
        s2m_params p;
        check(s2m_get_params(h_, &p), "s2m_get_params");
        if (p.imu_type == imuType && p.imu_rpy_weight == imuRPYWeight && p.z_tol == z_tollerance && p.rot_tol == rotation_tollerance) return;
        p.imu_type = imuType; p.imu_rpy_weight = imuRPYWeight;
        p.z_tol = z_tollerance; p.rot_tol = rotation_tollerance;
        check(s2m_set_params(h_, &p), "s2m_set_params");
    }

    // void scan2MapOptimization() (:1295-1321)
    void scan2MapOptimization()
    {
        pushParams();
        s2m_imu_init imu;
        imu.imuAvailable = cloudInfo.imuAvailable;
        imu.imuRollInit = cloudInfo.imuRollInit; imu.imuPitchInit = cloudInfo.imuPitchInit; imu.imuYawInit = cloudInfo.imuYawInit;
        if (scanResident_) {        // downsampleCurrentScan() left the scan on the device
            scanResident_ = false;
            check(s2m_optimize_resident(h_, transformTobeMapped, &imu, &lastResult), "s2m_optimize_resident");
        } else {
            laserCloudSurfLastDSNum = (int)laserCloudSurfLastDS.size();
            check(s2m_optimize(h_, laserCloudSurfLastDS.data(), laserCloudSurfLastDS.size(), sizeof(PointXYZI),
                               transformTobeMapped, &imu, &lastResult), "s2m_optimize");
        }
        if (lastResult.skipped == 2) {
            // ROS_WARN("Not enough features! Only %d planar features available.", ...) (:1319)
            return;
        }
        if (lastResult.skipped == 0) {
            isDegenerate = lastResult.is_degenerate != 0;
            std::memcpy(incrementalOdometryAffineBack, lastResult.affine, sizeof(incrementalOdometryAffineBack));
        }
    }

    // No counterpart in the reference (it registers one scan at a time under `mtx`, :252): n scans against the resident map at
    // once - s2m_optimize_batch, the scans' LM loops in lockstep inside one graph.  `scans[b]` = laserCloudSurfLastDS of scan b,
    // `poses[6*b..]` in: initial guess, out: transformTobeMapped of scan b; results as lastResult would hold them.  Bitwise
    // what n calls of scan2MapOptimization() give.
    std::vector<s2m_result> scan2MapOptimizationBatch(const std::vector<const std::vector<PointXYZI>*>& scans, std::vector<float>& poses,
                                                      const std::vector<CloudInfo>* infos = nullptr)
    {
        pushParams();
        const int n = (int)scans.size();
        if ((int)poses.size() != 6 * n) throw std::runtime_error("scan2MapOptimizationBatch: poses must hold 6 floats per scan");
        std::vector<const void*> ptrs((size_t)n);
        std::vector<size_t> sizes((size_t)n);
        for (int b = 0; b < n; b++) { ptrs[(size_t)b] = scans[(size_t)b]->data(); sizes[(size_t)b] = scans[(size_t)b]->size(); }
        std::vector<s2m_imu_init> imu((size_t)n);
        for (int b = 0; b < n; b++) {
            const CloudInfo& ci = infos ? (*infos)[(size_t)b] : cloudInfo;
            imu[(size_t)b].imuAvailable = ci.imuAvailable;
            imu[(size_t)b].imuRollInit = ci.imuRollInit; imu[(size_t)b].imuPitchInit = ci.imuPitchInit; imu[(size_t)b].imuYawInit = ci.imuYawInit;
        }
        std::vector<s2m_result> res((size_t)n);
        check(s2m_optimize_batch(h_, n, ptrs.data(), sizes.data(), sizeof(PointXYZI), poses.data(), imu.data(), res.data()), "s2m_optimize_batch");
        return res;
    }

    // A stream of scans through two slots (s2m_slot_*): prepareNextScan(slot, cloud) orders the NEXT scan on that slot's stream
    // while the loop launched with launchSlot() on the other slot runs; collectSlot() is scan2MapOptimization()'s second half
    // (synchronise, transformUpdate(), members updated).  The reference does the steps strictly one after the other (:257-265).
    void prepareNextScan(int slot, const std::vector<PointXYZI>& cloud)
    {
        check(s2m_slot_set_scan(h_, slot, cloud.data(), cloud.size(), sizeof(PointXYZI), 0), "s2m_slot_set_scan");
    }
    void launchSlot(int slot)
    {
        pushParams();
        check(s2m_slot_optimize_launch(h_, slot, transformTobeMapped), "s2m_slot_optimize_launch");
    }
    void collectSlot(int slot)
    {
        s2m_imu_init imu;
        imu.imuAvailable = cloudInfo.imuAvailable;
        imu.imuRollInit = cloudInfo.imuRollInit; imu.imuPitchInit = cloudInfo.imuPitchInit; imu.imuYawInit = cloudInfo.imuYawInit;
        check(s2m_slot_optimize_collect(h_, slot, transformTobeMapped, &imu, &lastResult), "s2m_slot_optimize_collect");
        if (lastResult.skipped == 0) {
            isDegenerate = lastResult.is_degenerate != 0;
            std::memcpy(incrementalOdometryAffineBack, lastResult.affine, sizeof(incrementalOdometryAffineBack));
        }
    }

    // The ICP block of performRSLoopClosure / performSCLoopClosure (:571-586, :663-678): settings, align(),
    // hasConverged(), getFitnessScore(), getFinalTransformation().  Returns false where the reference returns early.
    float historyKeyframeSearchRadius = 10.0f;        // include/utility.h:245
    float historyKeyframeFitnessScore = 0.3f;         // include/utility.h:248
    bool icpAlign(const std::vector<PointXYZI>& cureKeyframeCloud, const std::vector<PointXYZI>& prevKeyframeCloud,
                  float finalTransformation[16])
    {
        if (cureKeyframeCloud.size() < 300 || prevKeyframeCloud.size() < 1000) return false;       // :565-566
        s2m_icp_params p;
        s2m_icp_default_params(&p);
        p.max_correspondence_distance = historyKeyframeSearchRadius * 2;                            // :573
        s2m_icp_result r;
        check(s2m_icp_align(h_, cureKeyframeCloud.data(), cureKeyframeCloud.size(), prevKeyframeCloud.data(),
                            prevKeyframeCloud.size(), sizeof(PointXYZI), &p, &r), "s2m_icp_align");
        if (!r.converged || r.fitness_score > historyKeyframeFitnessScore) return false;            // :585-586
        std::memcpy(finalTransformation, r.T, sizeof(r.T));
        return true;
    }

    s2m_handle handle() const { return h_; }

private:
    void check(int rc, const char* what)
    {
        if (rc != S2M_OK) throw std::runtime_error(std::string(what) + ": " + s2m_last_error(h_));
    }
    // S2M_WARN_LEAF_TOO_SMALL is PCL's PCL_WARN case (output = input): not an error
    void checkVoxel(int rc, const char* what) { if (rc != S2M_WARN_LEAF_TOO_SMALL) check(rc, what); }
    s2m_handle h_ = nullptr;
    bool scanResident_ = false;
};

// SCManager (reference include/Scancontext.h:56-113) on top of the same handle: the descriptor store and the
// loop detector live on the device; method names and return values are the reference's.
class SCManagerS2M {
public:
    explicit SCManagerS2M(s2m_handle h) : h_(h) {}
    // void makeAndSaveScancontextAndKeys(pcl::PointCloud<SCPointType>& _scan_down) (Scancontext.cpp:236-250)
    void makeAndSaveScancontextAndKeys(const std::vector<PointXYZI>& scan_down)
    {
        check(s2m_sc_add_scan(h_, scan_down.data(), scan_down.size(), sizeof(PointXYZI)), "s2m_sc_add_scan");
    }
    // std::pair<int, float> detectLoopClosureID(void) (Scancontext.cpp:253-344): {loop_id or -1, yaw_diff_rad}
    std::pair<int, float> detectLoopClosureID()
    {
        int32_t id = -1; float yaw = 0.0f;
        check(s2m_sc_detect_loop(h_, &id, &yaw, &lastMatch), "s2m_sc_detect_loop");
        return { (int)id, yaw };
    }
    int size() const { return s2m_sc_size(h_); }
    s2m_sc_match lastMatch{};

private:
    void check(int rc, const char* what)
    {
        if (rc != S2M_OK) throw std::runtime_error(std::string(what) + ": " + s2m_last_error(h_));
    }
    s2m_handle h_;
};

}  // namespace liorf_amd
