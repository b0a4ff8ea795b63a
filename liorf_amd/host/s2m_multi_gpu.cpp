// s2m_multi_gpu.cpp — the batch of independent scans across the GPUs of one node (BASELINE config 4), in C++:
// one host thread + one s2m handle + one HIP stream per device, the local surf map replicated on every device, scans
// assigned round-robin (scan i -> device i % n), no data-path collective; the only exchange is ONE RCCL all-gather of the
// 8-float result records ({roll, pitch, yaw, x, y, z, iterations, correspondences} per scan) over xGMI, after which every
// device - and the host - holds all poses.  The reference has no counterpart (it registers one scan at a time under a
// mutex, src/mapOptmization.cpp:252); this is the sharding SURVEY.md section 8(e) describes.
//
//   s2m_multi_gpu <n_gpus|0=all> map.bin manifest.txt [reps]
//     map.bin       local surf map, pcl::PointXYZI records (32-byte stride)
//     manifest.txt  one scan per line:  scan.bin roll pitch yaw x y z      (initial guess = transformTobeMapped)
//   prints one line per scan: "scan <i> dev <d> iters <n> n_sel <m> pose r p y x y z" (read back from the gathered table of
//   device 0), and the wall time per batch over `reps` repetitions.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "map_optimization_s2m.hpp"

namespace {

constexpr int kRecord = 8;

struct ScanJob { std::vector<liorf_amd::PointXYZI> pts; float pose[6]; };

std::vector<liorf_amd::PointXYZI> read_cloud(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + path);
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<liorf_amd::PointXYZI> pts((size_t)n / sizeof(liorf_amd::PointXYZI));
    f.read(reinterpret_cast<char*>(pts.data()), (std::streamsize)(pts.size() * sizeof(liorf_amd::PointXYZI)));
    return pts;
}

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define CHECK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) throw std::runtime_error(std::string(#x) + ": " + ncclGetErrorString(r_)); } while (0)

struct Rank {
    int dev = 0;
    hipStream_t stream = nullptr;
    liorf_amd::MapOptimizationS2M* node = nullptr;
    float* d_send = nullptr;      // [per_rank][8]
    float* d_recv = nullptr;      // [n_ranks * per_rank][8]
    std::vector<float> h_send;
    std::string error;
};

}  // namespace

int main(int argc, char** argv)
{
    try {
        if (argc < 4) { std::fprintf(stderr, "usage: %s <n_gpus|0=all> map.bin manifest.txt [reps]\n", argv[0]); return 2; }
        int n_dev = 0;
        CHECK_HIP(hipGetDeviceCount(&n_dev));
        if (n_dev <= 0) throw std::runtime_error("no HIP device: there is no CPU fallback");
        int n = std::atoi(argv[1]);
        if (n <= 0 || n > n_dev) n = n_dev;
        const int reps = argc > 4 ? std::max(1, std::atoi(argv[4])) : 1;
        const std::vector<liorf_amd::PointXYZI> map = read_cloud(argv[2]);
        std::vector<ScanJob> jobs;
        {
            std::ifstream mf(argv[3]);
            if (!mf) throw std::runtime_error(std::string("cannot open ") + argv[3]);
            std::string line;
            while (std::getline(mf, line)) {
                std::istringstream ss(line);
                std::string path;
                ScanJob j;
                if (!(ss >> path >> j.pose[0] >> j.pose[1] >> j.pose[2] >> j.pose[3] >> j.pose[4] >> j.pose[5])) continue;
                j.pts = read_cloud(path);
                jobs.push_back(std::move(j));
            }
        }
        const int n_scans = (int)jobs.size();
        if (n_scans == 0) throw std::runtime_error("empty manifest");
        const int per_rank = (n_scans + n - 1) / n;

        // ---- one handle + stream per device; RCCL communicators for the single-process group
        std::vector<Rank> ranks((size_t)n);
        std::vector<int> devs((size_t)n);
        for (int d = 0; d < n; d++) devs[(size_t)d] = d;
        std::vector<ncclComm_t> comms((size_t)n);
        CHECK_NCCL(ncclCommInitAll(comms.data(), n, devs.data()));
        for (int d = 0; d < n; d++) {
            Rank& r = ranks[(size_t)d];
            r.dev = d;
            CHECK_HIP(hipSetDevice(d));
            CHECK_HIP(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
            r.node = new liorf_amd::MapOptimizationS2M(d, r.stream);       // throws without a gfx950 device
            r.node->laserCloudSurfFromMapDS = map;                          // replicated local map
            r.node->haveKeyPoses = !map.empty();
            r.node->setInputCloud();
            CHECK_HIP(hipMalloc((void**)&r.d_send, sizeof(float) * kRecord * (size_t)per_rank));
            CHECK_HIP(hipMalloc((void**)&r.d_recv, sizeof(float) * kRecord * (size_t)per_rank * (size_t)n));
            r.h_send.assign((size_t)kRecord * (size_t)per_rank, NAN);
        }

        std::vector<float> table((size_t)kRecord * (size_t)per_rank * (size_t)n, NAN);
        double total_s = 0.0;
        for (int rep = 0; rep < reps; rep++) {
            const auto t0 = std::chrono::steady_clock::now();
            // ---- every device registers its shard (scan i -> device i % n), no exchange on the data path
            std::vector<std::thread> threads;
            for (int d = 0; d < n; d++) {
                threads.emplace_back([&, d]() {
                    Rank& r = ranks[(size_t)d];
                    try {
                        if (hipSetDevice(d) != hipSuccess) throw std::runtime_error("hipSetDevice");
                        std::fill(r.h_send.begin(), r.h_send.end(), NAN);
                        int k = 0;
                        for (int i = d; i < n_scans; i += n, k++) {
                            r.node->laserCloudSurfLastDS = jobs[(size_t)i].pts;
                            for (int q = 0; q < 6; q++) r.node->transformTobeMapped[q] = jobs[(size_t)i].pose[q];
                            r.node->scan2MapOptimization();
                            float* rec = &r.h_send[(size_t)kRecord * (size_t)k];
                            for (int q = 0; q < 6; q++) rec[q] = r.node->transformTobeMapped[q];
                            rec[6] = (float)r.node->lastResult.iters_run;
                            rec[7] = (float)r.node->lastResult.n_sel_last;
                        }
                        if (hipMemcpyAsync(r.d_send, r.h_send.data(), sizeof(float) * r.h_send.size(), hipMemcpyHostToDevice, r.stream) != hipSuccess)
                            throw std::runtime_error("record upload");
                    } catch (const std::exception& e) { r.error = e.what(); }
                });
            }
            for (std::thread& t : threads) t.join();
            for (const Rank& r : ranks) if (!r.error.empty()) throw std::runtime_error("device " + std::to_string(r.dev) + ": " + r.error);
            // ---- the one collective of the batch: all-gather of the records over RCCL (xGMI between the GPUs of a node)
            CHECK_NCCL(ncclGroupStart());
            for (int d = 0; d < n; d++)
                CHECK_NCCL(ncclAllGather(ranks[(size_t)d].d_send, ranks[(size_t)d].d_recv, (size_t)kRecord * (size_t)per_rank, ncclFloat,
                                         comms[(size_t)d], ranks[(size_t)d].stream));
            CHECK_NCCL(ncclGroupEnd());
            for (int d = 0; d < n; d++) { CHECK_HIP(hipSetDevice(d)); CHECK_HIP(hipStreamSynchronize(ranks[(size_t)d].stream)); }
            total_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        // every device holds the whole table; read device 0's copy and, as a check, the last device's
        CHECK_HIP(hipSetDevice(0));
        CHECK_HIP(hipMemcpy(table.data(), ranks[0].d_recv, sizeof(float) * table.size(), hipMemcpyDeviceToHost));
        std::vector<float> other(table.size());
        CHECK_HIP(hipSetDevice(n - 1));
        CHECK_HIP(hipMemcpy(other.data(), ranks[(size_t)n - 1].d_recv, sizeof(float) * other.size(), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < table.size(); k++)
            if (!(table[k] == other[k] || (table[k] != table[k] && other[k] != other[k]))) throw std::runtime_error("gathered tables differ between devices");
        for (int i = 0; i < n_scans; i++) {
            const int d = i % n, k = i / n;
            const float* rec = &table[(size_t)kRecord * ((size_t)d * (size_t)per_rank + (size_t)k)];
            std::printf("scan %d dev %d iters %d n_sel %d pose %.9g %.9g %.9g %.9g %.9g %.9g\n", i, d, (int)rec[6], (int)rec[7], rec[0], rec[1], rec[2],
                        rec[3], rec[4], rec[5]);
        }
        std::printf("gpus %d scans %d reps %d seconds_per_batch %.6f\n", n, n_scans, reps, total_s / reps);
        for (int d = 0; d < n; d++) {
            Rank& r = ranks[(size_t)d];
            (void)hipSetDevice(d);
            delete r.node;
            (void)hipFree(r.d_send); (void)hipFree(r.d_recv);
            (void)hipStreamDestroy(r.stream);
            (void)ncclCommDestroy(comms[(size_t)d]);
        }
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "s2m_multi_gpu: %s\n", e.what());
        return 1;
    }
}
