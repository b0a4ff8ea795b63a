// scan2map_shim.cpp — the translation unit a maintainer adds to jimmyshe/liorf to route the scan-to-map path of the
// mapOptimization node through the MI355X library.  It is compiled only inside the reference's catkin workspace (it needs
// the reference's include/utility.h, i.e. ROS + PCL; neither exists in this repository's image, so this file is NOT part of
// the default build - `make -C liorf_amd/host` skips it).  Everything the three replaced member functions need is on the
// other side of the C ABI (include/liorf_s2m.h); nothing here touches the node's ROS topics, GTSAM graph or threads.
//
// Integration (see INTEGRATION.md for the CMake lines):
//   1. add this file to the `${PROJECT_NAME}_mapOptmization` target and link `liorf_s2m`;
//   2. in src/mapOptmization.cpp give `class mapOptimization` a member `liorf_amd::Scan2MapGpu gpu;`
//   3. replace the bodies of
//        void extractCloud(pcl::PointCloud<PointType>::Ptr cloudToExtract)   (:1012-1044)
//        void downsampleCurrentScan()                                          (:1059-1067)
//        void scan2MapOptimization()                                           (:1295-1321)
//      with one-line calls to the functions below (shown at the end of this file); surfOptimization(),
//      combineOptimizationCoeffs(), LMOptimization(), transformUpdate() and the kd-tree member are no longer used.
#include "utility.h"                 // the reference's: PointType, PointTypePose, ParamServer members, pcl, ROS logging

#include <cstring>
#include <stdexcept>
#include <vector>

#include "liorf_s2m.h"

namespace liorf_amd {

static_assert(sizeof(PointType) == 32, "pcl::PointXYZI is handed over as 32-byte records");

class Scan2MapGpu {
public:
    Scan2MapGpu()
    {
        s2m_params prm;
        s2m_default_params(&prm);
        if (s2m_create(&prm, &h_) != S2M_OK) {
            ROS_ERROR("liorf: no MI355X (gfx950) device for the scan-to-map path; there is no CPU fallback");
            ros::shutdown();
        }
    }
    ~Scan2MapGpu() { if (h_) s2m_destroy(h_); }
    Scan2MapGpu(const Scan2MapGpu&) = delete;
    Scan2MapGpu& operator=(const Scan2MapGpu&) = delete;

    // The ParamServer values transformUpdate() reads (:1325-1350) are pushed before every registration: they are plain
    // members of the node, set from the yaml after construction.
    void pushParams(int imuType, float imuRPYWeight, float z_tollerance, float rotation_tollerance)
    {
        s2m_params p;
        if (s2m_get_params(h_, &p) != S2M_OK) return;
        if (p.imu_type == imuType && p.imu_rpy_weight == imuRPYWeight && p.z_tol == z_tollerance && p.rot_tol == rotation_tollerance) return;
        p.imu_type = imuType; p.imu_rpy_weight = imuRPYWeight; p.z_tol = z_tollerance; p.rot_tol = rotation_tollerance;
        check(s2m_set_params(h_, &p), "s2m_set_params");
    }

    // extractCloud (:1012-1044): the key frames within surroundingKeyframeSearchRadius of the newest key pose (:1018) are
    // transformed by their key poses, concatenated, voxel-filtered (surroundingKeyframeMapLeafSize) and indexed on the
    // device.  laserCloudSurfFromMapDS is filled for the node's other consumers (publishing, map saving).
    void extractCloud(const pcl::PointCloud<PointType>::Ptr& cloudToExtract, const pcl::PointCloud<PointType>::Ptr& cloudKeyPoses3D,
                      const pcl::PointCloud<PointTypePose>::Ptr& cloudKeyPoses6D,
                      const std::vector<pcl::PointCloud<PointType>::Ptr>& surfCloudKeyFrames, float surroundingKeyframeSearchRadius,
                      float surroundingKeyframeMapLeafSize, pcl::PointCloud<PointType>::Ptr& laserCloudSurfFromMapDS,
                      int& laserCloudSurfFromMapDSNum)
    {
        std::vector<const void*> frames;
        std::vector<size_t> sizes;
        std::vector<float> poses;
        size_t total = 0;
        for (int i = 0; i < (int)cloudToExtract->size(); ++i) {
            const PointType& p = cloudToExtract->points[i];
            const PointType& last = cloudKeyPoses3D->back();
            const float dx = p.x - last.x, dy = p.y - last.y, dz = p.z - last.z;
            if (std::sqrt(dx * dx + dy * dy + dz * dz) > surroundingKeyframeSearchRadius) continue;       // pointDistance (:1018)
            const int thisKeyInd = (int)p.intensity;
            const PointTypePose& t = cloudKeyPoses6D->points[thisKeyInd];
            frames.push_back(surfCloudKeyFrames[thisKeyInd]->points.data());
            sizes.push_back(surfCloudKeyFrames[thisKeyInd]->size());
            const float v[6] = { t.x, t.y, t.z, t.roll, t.pitch, t.yaw };
            poses.insert(poses.end(), v, v + 6);
            total += sizes.back();
        }
        laserCloudSurfFromMapDS->points.resize(total);
        size_t n_out = 0;
        const int rc = s2m_extract_cloud(h_, (int)frames.size(), frames.data(), sizes.data(), sizeof(PointType), 0, poses.data(),
                                         surroundingKeyframeMapLeafSize, laserCloudSurfFromMapDS->points.data(), sizeof(PointType), total, &n_out);
        if (rc != S2M_OK && rc != S2M_WARN_LEAF_TOO_SMALL) check(rc, "s2m_extract_cloud");
        laserCloudSurfFromMapDS->points.resize(n_out);
        laserCloudSurfFromMapDS->width = (uint32_t)n_out; laserCloudSurfFromMapDS->height = 1;
        laserCloudSurfFromMapDSNum = (int)n_out;
        haveMap_ = !cloudKeyPoses3D->points.empty();
    }

    // downsampleCurrentScan (:1059-1067): VoxelGrid(mappingSurfLeafSize); the filtered scan stays on the device for the
    // registration and is copied back for the key-frame store.
    void downsampleCurrentScan(const pcl::PointCloud<PointType>::Ptr& laserCloudSurfLast, float mappingSurfLeafSize,
                               pcl::PointCloud<PointType>::Ptr& laserCloudSurfLastDS, int& laserCloudSurfLastDSNum)
    {
        laserCloudSurfLastDS->points.resize(laserCloudSurfLast->size());
        size_t n_out = 0;
        const int rc = s2m_downsample_scan(h_, laserCloudSurfLast->points.data(), laserCloudSurfLast->size(), sizeof(PointType), 0,
                                           mappingSurfLeafSize, laserCloudSurfLastDS->points.data(), sizeof(PointType),
                                           laserCloudSurfLastDS->points.size(), &n_out);
        if (rc != S2M_OK && rc != S2M_WARN_LEAF_TOO_SMALL) check(rc, "s2m_downsample_scan");
        laserCloudSurfLastDS->points.resize(n_out);
        laserCloudSurfLastDS->width = (uint32_t)n_out; laserCloudSurfLastDS->height = 1;
        laserCloudSurfLastDSNum = (int)n_out;
    }

    // scan2MapOptimization (:1295-1321) on the scan downsampleCurrentScan() left on the device.
    void scan2MapOptimization(bool keyPosesEmpty, int laserCloudSurfLastDSNum, const liorf::cloud_info& cloudInfo, float transformTobeMapped[6],
                              bool& isDegenerate, Eigen::Affine3f& incrementalOdometryAffineBack)
    {
        if (keyPosesEmpty || !haveMap_) return;                                                  // :1297 (no key pose yet: no map was extracted)
        s2m_imu_init imu;
        imu.imuAvailable = cloudInfo.imuAvailable;
        imu.imuRollInit = cloudInfo.imuRollInit; imu.imuPitchInit = cloudInfo.imuPitchInit; imu.imuYawInit = cloudInfo.imuYawInit;
        s2m_result res;
        const int rc = s2m_optimize_resident(h_, transformTobeMapped, &imu, &res);
        if (rc != S2M_OK) { ROS_ERROR("scan2map: %s", s2m_last_error(h_)); return; }
        if (res.skipped == 2) {
            ROS_WARN("Not enough features! Only %d planar features available.", laserCloudSurfLastDSNum);   // :1319
            return;
        }
        if (res.skipped != 0) return;
        isDegenerate = res.is_degenerate != 0;                                                   // -> pose.covariance[0] (:1724-1727)
        Eigen::Matrix4f m = Eigen::Matrix4f::Identity();
        for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) m(r, c) = res.affine[4 * r + c];
        incrementalOdometryAffineBack = Eigen::Affine3f(m);                                      // :1352
    }

private:
    void check(int rc, const char* what)
    {
        if (rc != S2M_OK) { ROS_ERROR("%s: %s", what, s2m_last_error(h_)); throw std::runtime_error(what); }
    }
    s2m_handle h_ = nullptr;
    bool haveMap_ = false;
};

}  // namespace liorf_amd

// ---- the three member functions of class mapOptimization, as they read after the change (src/mapOptmization.cpp) -------
//
//   void extractCloud(pcl::PointCloud<PointType>::Ptr cloudToExtract)
//   {
//       gpu.extractCloud(cloudToExtract, cloudKeyPoses3D, cloudKeyPoses6D, surfCloudKeyFrames, surroundingKeyframeSearchRadius,
//                        surroundingKeyframeMapLeafSize, laserCloudSurfFromMapDS, laserCloudSurfFromMapDSNum);
//   }
//   void downsampleCurrentScan()
//   {
//       gpu.downsampleCurrentScan(laserCloudSurfLast, mappingSurfLeafSize, laserCloudSurfLastDS, laserCloudSurfLastDSNum);
//   }
//   void scan2MapOptimization()
//   {
//       gpu.pushParams(imuType, imuRPYWeight, z_tollerance, rotation_tollerance);
//       gpu.scan2MapOptimization(cloudKeyPoses3D->points.empty(), laserCloudSurfLastDSNum, cloudInfo, transformTobeMapped,
//                                isDegenerate, incrementalOdometryAffineBack);
//   }
