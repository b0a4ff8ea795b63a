// s2m_harness.cpp — ROS-free C++ harness of the drop-in boundary: reads a map and a scan as raw
// PointXYZI records (32-byte stride, the reference's wire layout) and an initial guess, runs
// scan2MapOptimization() through the host mirror, prints the result.  Fails loudly without a GPU.
//
//   s2m_harness map.bin scan.bin roll pitch yaw x y z
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

int main(int argc, char** argv)
{
    try {
        if (argc == 2 && std::string(argv[1]) == "--version") { std::puts(s2m_version()); return 0; }
        if (argc != 9) { std::fprintf(stderr, "usage: %s map.bin scan.bin roll pitch yaw x y z\n", argv[0]); return 2; }
        liorf_amd::MapOptimizationS2M node;                 // throws without a gfx950 device
        node.laserCloudSurfFromMapDS = read_cloud(argv[1]);
        node.laserCloudSurfLastDS = read_cloud(argv[2]);
        node.haveKeyPoses = !node.laserCloudSurfFromMapDS.empty();
        for (int k = 0; k < 6; k++) node.transformTobeMapped[k] = (float)std::atof(argv[3 + k]);
        node.setInputCloud();
        node.scan2MapOptimization();
        const s2m_result& r = node.lastResult;
        std::printf("skipped %d iters %d converged %d degenerate %d n_sel %d\n", r.skipped, r.iters_run, r.converged, r.is_degenerate, r.n_sel_last);
        std::printf("transformTobeMapped %.9g %.9g %.9g %.9g %.9g %.9g\n", node.transformTobeMapped[0], node.transformTobeMapped[1],
                    node.transformTobeMapped[2], node.transformTobeMapped[3], node.transformTobeMapped[4], node.transformTobeMapped[5]);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "s2m_harness: %s\n", e.what());
        return 1;
    }
}
