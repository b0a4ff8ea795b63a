// scan2map_shim.cpp — SOURCE ONLY (needs ROS + PCL, absent from this image; not built here).
//
// What a maintainer adds to the reference's src/mapOptmization.cpp to route the scan-to-map
// path through the MI355X library: the body of scan2MapOptimization() (:1295-1321) and the
// setInputCloud() it contains (:1302) are replaced; surfOptimization(), combineOptimizationCoeffs()
// and LMOptimization() (:1074-1293) are no longer called; every other member, topic and factor
// stays as it is.  The cloud_info subscription (:177), the odometry publications (:173-174) and
// the degenerate flag in pose.covariance[0] (:1724-1727) are unchanged.
//
//   #include "liorf_s2m.h"
//
//   // new member, created once in the constructor (:164-200)
//   s2m_handle s2m = nullptr;
//   ...
//       s2m_params prm; s2m_default_params(&prm);
//       prm.imu_type = imuType; prm.imu_rpy_weight = imuRPYWeight;
//       prm.z_tol = z_tollerance; prm.rot_tol = rotation_tollerance;
//       if (s2m_create(&prm, &s2m) != S2M_OK) { ROS_ERROR("liorf: no MI355X for scan2map"); ros::shutdown(); }
//
//   void scan2MapOptimization()
//   {
//       if (cloudKeyPoses3D->points.empty())
//           return;
//       // pcl::PointXYZI is a 32-byte record with x,y,z first: handed over as is
//       s2m_set_map(s2m, laserCloudSurfFromMapDS->points.data(), laserCloudSurfFromMapDS->size(), sizeof(PointType));
//       s2m_imu_init imu{ cloudInfo.imuAvailable, cloudInfo.imuRollInit, cloudInfo.imuPitchInit, cloudInfo.imuYawInit };
//       s2m_result res;
//       int rc = s2m_optimize(s2m, laserCloudSurfLastDS->points.data(), laserCloudSurfLastDSNum, sizeof(PointType),
//                             transformTobeMapped, &imu, &res);
//       if (rc != S2M_OK) { ROS_ERROR("scan2map: %s", s2m_last_error(s2m)); return; }
//       if (res.skipped == 2) {
//           ROS_WARN("Not enough features! Only %d planar features available.", laserCloudSurfLastDSNum);
//           return;
//       }
//       isDegenerate = res.is_degenerate;
//       incrementalOdometryAffineBack = Eigen::Map<Eigen::Matrix<float, 3, 4, Eigen::RowMajor>>(res.affine) ... ; // 3x4 -> Affine3f
//   }
//
// CMakeLists.txt: target_link_libraries(${PROJECT_NAME}_mapOptmization ... liorf_s2m) and the
// include path of liorf_s2m.h.  Nothing else in the package changes.
