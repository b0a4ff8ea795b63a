// s2m_harness.cpp — ROS-free C++ harness of the drop-in boundary: reads a map and a scan as raw
// PointXYZI records (32-byte stride, the reference's wire layout) and an initial guess, runs
// scan2MapOptimization() through the host mirror, prints the result.  Fails loudly without a GPU.
//
//   s2m_harness map.bin scan.bin roll pitch yaw x y z [imuType imuRPYWeight z_tollerance rotation_tollerance imuAvailable imuRollInit imuPitchInit]
//       the optional tail sets the ParamServer members AFTER the node is constructed (as the reference's yaml
//       loading does) and the cloud_info IMU fields transformUpdate() reads (:1325-1350)
//   s2m_harness --chain frames.bin frames.txt raw_scan.bin map_leaf scan_leaf roll pitch yaw x y z
//       the handler's three steps in order: extractCloud() over the key frames listed in frames.txt (one line
//       "n_points x y z roll pitch yaw" per frame, clouds back to back in frames.bin), downsampleCurrentScan(),
//       scan2MapOptimization().
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "map_optimization_s2m.hpp"

static std::vector<liorf_amd::PointXYZI> read_cloud(const char* path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<liorf_amd::PointXYZI> pts((size_t)n / sizeof(liorf_amd::PointXYZI));
    f.read(reinterpret_cast<char*>(pts.data()), (std::streamsize)(pts.size() * sizeof(liorf_amd::PointXYZI)));
    return pts;
}

static void print_result(const liorf_amd::MapOptimizationS2M& node)
{
    const s2m_result& r = node.lastResult;
    std::printf("skipped %d iters %d converged %d degenerate %d n_sel %d\n", r.skipped, r.iters_run, r.converged, r.is_degenerate, r.n_sel_last);
    std::printf("transformTobeMapped %.9g %.9g %.9g %.9g %.9g %.9g\n", node.transformTobeMapped[0], node.transformTobeMapped[1],
                node.transformTobeMapped[2], node.transformTobeMapped[3], node.transformTobeMapped[4], node.transformTobeMapped[5]);
    std::printf("incrementalOdometryAffineBack");
    for (int k = 0; k < 12; k++) std::printf(" %.9g", node.incrementalOdometryAffineBack[k]);
    std::printf("\n");
}

static int run_chain(char** argv)
{
    liorf_amd::MapOptimizationS2M node;
    const std::vector<liorf_amd::PointXYZI> all = read_cloud(argv[2]);
    std::ifstream tab(argv[3]);
    if (!tab) throw std::runtime_error(std::string("cannot open ") + argv[3]);
    size_t n, at = 0;
    liorf_amd::PointTypePose p;
    std::vector<int> keyInds;
    while (tab >> n >> p.x >> p.y >> p.z >> p.roll >> p.pitch >> p.yaw) {
        if (at + n > all.size()) throw std::runtime_error("frames.txt asks for more points than frames.bin holds");
        p.intensity = (float)node.cloudKeyPoses6D.size();
        keyInds.push_back((int)node.cloudKeyPoses6D.size());
        node.cloudKeyPoses6D.push_back(p);
        node.surfCloudKeyFrames.emplace_back(all.begin() + (std::ptrdiff_t)at, all.begin() + (std::ptrdiff_t)(at + n));
        at += n;
    }
    node.laserCloudSurfLast = read_cloud(argv[4]);
    node.surroundingKeyframeMapLeafSize = (float)std::atof(argv[5]);
    node.mappingSurfLeafSize = (float)std::atof(argv[6]);
    for (int k = 0; k < 6; k++) node.transformTobeMapped[k] = (float)std::atof(argv[7 + k]);
    node.extractCloud(keyInds);
    node.downsampleCurrentScan();
    node.scan2MapOptimization();
    std::printf("laserCloudSurfFromMapDSNum %d laserCloudSurfLastDSNum %d\n", node.laserCloudSurfFromMapDSNum, node.laserCloudSurfLastDSNum);
    print_result(node);
    return 0;
}

// --many map.bin roll pitch yaw x y z scan0.bin scan1.bin ...: the same initial guess for every scan; the scans once as a batch
// (scan2MapOptimizationBatch), once as a stream through two slots (prepareNextScan / launchSlot / collectSlot), once one by one
// (scan2MapOptimization): prints "batch|stream|single <i> iters <n> pose ..." - the three must agree bit for bit.
static int run_many(int argc, char** argv)
{
    liorf_amd::MapOptimizationS2M node;
    node.laserCloudSurfFromMapDS = read_cloud(argv[2]);
    node.haveKeyPoses = !node.laserCloudSurfFromMapDS.empty();
    float guess[6];
    for (int k = 0; k < 6; k++) guess[k] = (float)std::atof(argv[3 + k]);
    std::vector<std::vector<liorf_amd::PointXYZI>> scans;
    for (int a = 9; a < argc; a++) scans.push_back(read_cloud(argv[a]));
    const int n = (int)scans.size();
    node.setInputCloud();
    auto show = [](const char* tag, int i, int iters, const float* p) {
        std::printf("%s %d iters %d pose %.9g %.9g %.9g %.9g %.9g %.9g\n", tag, i, iters, p[0], p[1], p[2], p[3], p[4], p[5]);
    };
    {
        std::vector<const std::vector<liorf_amd::PointXYZI>*> ptrs;
        std::vector<float> poses;
        for (int b = 0; b < n; b++) { ptrs.push_back(&scans[(size_t)b]); for (int k = 0; k < 6; k++) poses.push_back(guess[k]); }
        const std::vector<s2m_result> res = node.scan2MapOptimizationBatch(ptrs, poses);
        for (int b = 0; b < n; b++) show("batch", b, res[(size_t)b].iters_run, &poses[(size_t)6 * (size_t)b]);
    }
    node.prepareNextScan(0, scans[0]);
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 6; k++) node.transformTobeMapped[k] = guess[k];
        node.launchSlot(i & 1);
        if (i + 1 < n) node.prepareNextScan((i + 1) & 1, scans[(size_t)i + 1]);
        node.collectSlot(i & 1);
        show("stream", i, node.lastResult.iters_run, node.transformTobeMapped);
    }
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 6; k++) node.transformTobeMapped[k] = guess[k];
        node.laserCloudSurfLastDS = scans[(size_t)i];
        node.scan2MapOptimization();
        show("single", i, node.lastResult.iters_run, node.transformTobeMapped);
    }
    return 0;
}

int main(int argc, char** argv)
{
    try {
        if (argc == 2 && std::string(argv[1]) == "--version") { std::puts(s2m_version()); return 0; }
        if (argc == 13 && std::string(argv[1]) == "--chain") return run_chain(argv);
        if (argc >= 10 && std::string(argv[1]) == "--many") return run_many(argc, argv);
        if (argc != 9 && argc != 16) {
            std::fprintf(stderr, "usage: %s map.bin scan.bin roll pitch yaw x y z [imuType imuRPYWeight z_tol rot_tol imuAvailable imuRoll imuPitch]\n", argv[0]);
            return 2;
        }
        liorf_amd::MapOptimizationS2M node;                 // throws without a gfx950 device
        node.laserCloudSurfFromMapDS = read_cloud(argv[1]);
        node.laserCloudSurfLastDS = read_cloud(argv[2]);
        node.haveKeyPoses = !node.laserCloudSurfFromMapDS.empty();
        for (int k = 0; k < 6; k++) node.transformTobeMapped[k] = (float)std::atof(argv[3 + k]);
        if (argc == 16) {       // members set after construction, like ParamServer's yaml values
            node.imuType = std::atoi(argv[9]);
            node.imuRPYWeight = (float)std::atof(argv[10]);
            node.z_tollerance = (float)std::atof(argv[11]);
            node.rotation_tollerance = (float)std::atof(argv[12]);
            node.cloudInfo.imuAvailable = std::atoll(argv[13]);
            node.cloudInfo.imuRollInit = (float)std::atof(argv[14]);
            node.cloudInfo.imuPitchInit = (float)std::atof(argv[15]);
        }
        node.setInputCloud();
        node.scan2MapOptimization();
        print_result(node);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "s2m_harness: %s\n", e.what());
        return 1;
    }
}
