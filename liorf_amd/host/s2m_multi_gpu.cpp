// s2m_multi_gpu.cpp — the batch of independent scans across the GPUs of one node (BASELINE config 4), in C++.
//
// One persistent worker thread + one s2m handle + one HIP stream per device, created once.  The local surf map is replicated
// and indexed on every device and every device's shard of the scans (scan i -> device i % n) is uploaded ONCE, before the
// clock starts: what a batch costs is what north_star measures - the registrations, with their inputs resident in HBM.
// Per batch every worker registers its shard with ONE call (s2m_optimize_batch on device records: the shard's LM loops
// advance in lockstep inside one captured graph; a shard of one scan takes the single-scan entry points), writes its
// 8-float records ({roll, pitch, yaw, x, y, z, iterations, correspondences} per scan) and joins the ONE collective of the
// batch, an RCCL all-gather over xGMI issued from its own thread on its own stream - no data-path collective, no thread
// creation, no host copy of a point inside the timed region.  Afterwards every device, and the host, holds all poses.
// The reference has no counterpart (it registers one scan at a time under a mutex, src/mapOptmization.cpp:252); this is
// the sharding SURVEY.md section 8(e) describes.
//
//   s2m_multi_gpu <n_gpus|0=all> map.bin manifest.txt [reps] [early_exit]
//     early_exit    1 (default): the loop breaks on convergence as the reference does (:1313); 0: all 30 iterations (the bench's step)
//     map.bin       local surf map, pcl::PointXYZI records (32-byte stride)
//     manifest.txt  one scan per line:  scan.bin roll pitch yaw x y z      (initial guess = transformTobeMapped)
//   A worker that fails (a library or HIP error in its shard) reports it at a host-side agreement point in front of the
//   collective: every worker then skips the all-gather together, and the program prints the error and exits non-zero - no
//   rank is left waiting in a collective its peer never joins.  A worker that does not come back at all (a collective that
//   never completes) is given up after S2M_MULTI_GPU_TIMEOUT_S seconds (default 120): the communicators are aborted and the
//   process exits with code 3.
//   prints one line per scan: "scan <i> dev <d> iters <n> n_sel <m> pose r p y x y z" (read back from the gathered table of
//   device 0), and "gpus G scans S reps R seconds_per_batch T seconds_per_scan T/S" - T the median over `reps` timed batches
//   behind one untimed warm-up batch, setup excluded.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/liorf_s2m.h"

namespace {

constexpr int kRecord = 8;
constexpr size_t kStride = 32;          // pcl::PointXYZI

struct ScanJob { std::vector<unsigned char> bytes; size_t n = 0; float pose[6]; };

std::vector<unsigned char> read_file(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + path);
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<unsigned char> b((size_t)n - (size_t)n % kStride);
    f.read(reinterpret_cast<char*>(b.data()), (std::streamsize)b.size());
    return b;
}

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define CHECK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) throw std::runtime_error(std::string(#x) + ": " + ncclGetErrorString(r_)); } while (0)
#define CHECK_S2M(h, x) do { int r_ = (x); if (r_ != S2M_OK) throw std::runtime_error(std::string(#x) + " -> " + std::to_string(r_) + ": " + s2m_last_error(h)); } while (0)

// everything one device needs, owned by its worker thread
struct Rank {
    int dev = 0, n_ranks = 1, per_rank = 0;
    hipStream_t stream = nullptr;
    s2m_handle h = nullptr;
    ncclComm_t comm = nullptr;
    std::vector<int> mine;                       // indices of this device's scans
    std::vector<void*> d_scans;                  // their records, resident
    std::vector<size_t> sizes;
    std::vector<float> poses_in;                 // [n_mine][6]
    std::vector<float> poses;                    // in/out of a batch
    std::vector<s2m_result> results;
    float* h_send = nullptr;                     // pinned [per_rank][8]
    float* d_send = nullptr;                     // [per_rank][8]
    float* d_recv = nullptr;                     // [n_ranks * per_rank][8]
    std::string error;
    std::thread thread;
};

// the batches: the main thread raises `generation`, every worker runs one batch and reports
struct Gate {
    std::mutex m;
    std::condition_variable go, done, agree;
    long generation = 0;
    int finished = 0;
    bool quit = false;
    // the agreement point in front of the collective: every worker reports whether its shard went through; the last to arrive
    // releases the others, and they all see the same verdict
    int n = 1, arrived = 0, failed = 0;
    long phase = 0;
};

// test hook: S2M_MULTI_GPU_FAIL_RANK=<device> makes that device's worker fail in its second batch (the first timed one)
int fail_rank()
{
    const char* e = std::getenv("S2M_MULTI_GPU_FAIL_RANK");
    return e ? std::atoi(e) : -1;
}

// the registrations of this device's shard (may throw)
void register_shard(Rank& r, long batch_no)
{
    if (batch_no >= 2 && r.dev == fail_rank()) throw std::runtime_error("S2M_MULTI_GPU_FAIL_RANK: injected failure");
    const int n_mine = (int)r.mine.size();
    std::copy(r.poses_in.begin(), r.poses_in.end(), r.poses.begin());
    for (int k = 0; k < kRecord * r.per_rank; k++) r.h_send[k] = NAN;
    if (n_mine == 1) {
        // a shard of one scan: the single-scan entry points (the whole GPU for one loop)
        CHECK_S2M(r.h, s2m_set_scan_device(r.h, r.d_scans[0], r.sizes[0], kStride));
        CHECK_S2M(r.h, s2m_optimize_launch(r.h, r.poses.data()));
        CHECK_S2M(r.h, s2m_optimize_collect(r.h, r.poses.data(), nullptr, &r.results[0]));
    } else if (n_mine > 1) {
        // the shard's loops in lockstep inside one graph, scans already on the device
        CHECK_S2M(r.h, s2m_batch_set_scans(r.h, n_mine, r.d_scans.data(), r.sizes.data(), kStride, 1));
        CHECK_S2M(r.h, s2m_optimize_batch_launch(r.h, n_mine, r.poses.data()));
        CHECK_S2M(r.h, s2m_optimize_batch_collect(r.h, n_mine, r.poses.data(), nullptr, r.results.data()));
    }
    for (int k = 0; k < n_mine; k++) {
        float* rec = r.h_send + (size_t)kRecord * (size_t)k;
        for (int q = 0; q < 6; q++) rec[q] = r.poses[(size_t)6 * (size_t)k + (size_t)q];
        rec[6] = (float)r.results[(size_t)k].iters_run;
        rec[7] = (float)r.results[(size_t)k].n_sel_last;
    }
}

// the one collective of the batch (entered by every worker or by none)
void gather_records(Rank& r)
{
    CHECK_HIP(hipMemcpyAsync(r.d_send, r.h_send, sizeof(float) * kRecord * (size_t)r.per_rank, hipMemcpyHostToDevice, r.stream));
    // the one collective of the batch: all-gather of the records over RCCL (xGMI between the GPUs of a node), on this device's stream
    CHECK_NCCL(ncclAllGather(r.d_send, r.d_recv, (size_t)kRecord * (size_t)r.per_rank, ncclFloat, r.comm, r.stream));
    CHECK_HIP(hipStreamSynchronize(r.stream));
}

void worker(Rank* rp, Gate* g)
{
    Rank& r = *rp;
    long seen = 0;
    if (hipSetDevice(r.dev) != hipSuccess) r.error = "hipSetDevice";
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(g->m);
            g->go.wait(lk, [&] { return g->quit || g->generation != seen; });
            if (g->quit) return;
            seen = g->generation;
        }
        bool ok = r.error.empty();
        if (ok) {
            try { register_shard(r, seen); } catch (const std::exception& e) { r.error = e.what(); ok = false; }
        }
        bool all_ok = false;
        {   // agreement: nobody enters the all-gather unless everybody can
            std::unique_lock<std::mutex> lk(g->m);
            if (!ok) g->failed++;
            if (++g->arrived == g->n) { g->phase++; g->agree.notify_all(); }
            else { const long p = g->phase; g->agree.wait(lk, [&] { return g->phase != p; }); }
            all_ok = g->failed == 0;
        }
        if (all_ok) {
            try { gather_records(r); } catch (const std::exception& e) { r.error = e.what(); }
        } else if (r.error.empty())
            r.error = "skipped the all-gather: another device failed";
        {
            std::lock_guard<std::mutex> lk(g->m);
            g->finished++;
        }
        g->done.notify_one();
    }
}

}  // namespace

int main(int argc, char** argv)
{
    try {
        if (argc < 4) { std::fprintf(stderr, "usage: %s <n_gpus|0=all> map.bin manifest.txt [reps] [early_exit]\n", argv[0]); return 2; }
        int n_dev = 0;
        CHECK_HIP(hipGetDeviceCount(&n_dev));
        if (n_dev <= 0) throw std::runtime_error("no HIP device: there is no CPU fallback");
        int n = std::atoi(argv[1]);
        if (n <= 0 || n > n_dev) n = n_dev;
        const int reps = argc > 4 ? std::max(1, std::atoi(argv[4])) : 1;
        const int early_exit = argc > 5 ? std::atoi(argv[5]) : -1;
        const std::vector<unsigned char> map = read_file(argv[2]);
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
                j.bytes = read_file(path);
                j.n = j.bytes.size() / kStride;
                jobs.push_back(std::move(j));
            }
        }
        const int n_scans = (int)jobs.size();
        if (n_scans == 0) throw std::runtime_error("empty manifest");
        const int per_rank = (n_scans + n - 1) / n;
        if (per_rank > 64) throw std::runtime_error("more than 64 scans per device");

        // ---- setup, outside the clock: communicators, handles, the replicated map, the shards' scans on their devices, the workers
        std::vector<Rank> ranks((size_t)n);
        std::vector<int> devs((size_t)n);
        for (int d = 0; d < n; d++) devs[(size_t)d] = d;
        std::vector<ncclComm_t> comms((size_t)n);
        CHECK_NCCL(ncclCommInitAll(comms.data(), n, devs.data()));
        for (int d = 0; d < n; d++) {
            Rank& r = ranks[(size_t)d];
            r.dev = d; r.n_ranks = n; r.per_rank = per_rank; r.comm = comms[(size_t)d];
            CHECK_HIP(hipSetDevice(d));
            CHECK_HIP(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
            s2m_params p;
            s2m_default_params(&p);
            p.device_id = d;
            p.stream = r.stream;
            if (early_exit >= 0) p.early_exit = early_exit;
            const int rc = s2m_create(&p, &r.h);
            if (rc != S2M_OK) throw std::runtime_error("s2m_create failed (" + std::to_string(rc) + "): no gfx950 device; there is no CPU fallback");
            CHECK_S2M(r.h, s2m_set_map(r.h, map.data(), map.size() / kStride, kStride));        // replicated local map + its index
            for (int i = d; i < n_scans; i += n) {
                void* dp = nullptr;
                CHECK_HIP(hipMalloc(&dp, std::max<size_t>(jobs[(size_t)i].bytes.size(), kStride)));
                CHECK_HIP(hipMemcpy(dp, jobs[(size_t)i].bytes.data(), jobs[(size_t)i].bytes.size(), hipMemcpyHostToDevice));
                r.mine.push_back(i);
                r.d_scans.push_back(dp);
                r.sizes.push_back(jobs[(size_t)i].n);
                for (int q = 0; q < 6; q++) r.poses_in.push_back(jobs[(size_t)i].pose[q]);
            }
            r.poses.resize(r.poses_in.size());
            r.results.resize(r.mine.size() ? r.mine.size() : 1);
            CHECK_HIP(hipHostMalloc((void**)&r.h_send, sizeof(float) * kRecord * (size_t)per_rank));
            CHECK_HIP(hipMalloc((void**)&r.d_send, sizeof(float) * kRecord * (size_t)per_rank));
            CHECK_HIP(hipMalloc((void**)&r.d_recv, sizeof(float) * kRecord * (size_t)per_rank * (size_t)n));
        }
        Gate gate;
        gate.n = n;
        const double timeout_s = std::getenv("S2M_MULTI_GPU_TIMEOUT_S") ? std::atof(std::getenv("S2M_MULTI_GPU_TIMEOUT_S")) : 120.0;
        for (int d = 0; d < n; d++) ranks[(size_t)d].thread = std::thread(worker, &ranks[(size_t)d], &gate);
        // (whatever happens below, the workers are told to quit and joined before `ranks` goes away: a joinable std::thread
        // destroyed during stack unwinding would end the process in std::terminate instead of with the error message)
        auto stop_workers = [&]() {
            {
                std::lock_guard<std::mutex> lk(gate.m);
                gate.quit = true;
            }
            gate.go.notify_all();
            for (Rank& r : ranks) if (r.thread.joinable()) r.thread.join();
        };

        auto one_batch = [&]() {
            const auto t0 = std::chrono::steady_clock::now();
            {
                std::lock_guard<std::mutex> lk(gate.m);
                gate.finished = 0;
                gate.generation++;
            }
            gate.go.notify_all();
            {
                std::unique_lock<std::mutex> lk(gate.m);
                if (!gate.done.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return gate.finished == n; })) {
                    // a worker is stuck (in a collective that will never complete, or on a device that hangs): nothing to join
                    lk.unlock();
                    std::fprintf(stderr, "s2m_multi_gpu: %d of %d devices did not finish a batch within %.0f s - aborting the communicators\n",
                                 n - gate.finished, n, timeout_s);
                    for (ncclComm_t c : comms) (void)ncclCommAbort(c);
                    std::fflush(stderr);
                    std::_Exit(3);
                }
            }
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::string first;                           // the error of the device that failed, not of those that stood by
            for (const Rank& r : ranks)
                if (!r.error.empty() && r.error.rfind("skipped", 0) != 0 && first.empty()) first = "device " + std::to_string(r.dev) + ": " + r.error;
            for (const Rank& r : ranks)
                if (!r.error.empty() && first.empty()) first = "device " + std::to_string(r.dev) + ": " + r.error;
            if (!first.empty()) throw std::runtime_error(first);
            return s;
        };
        std::vector<double> times;
        try {
            {
                std::lock_guard<std::mutex> lk(gate.m);
                gate.arrived = 0; gate.failed = 0;
            }
            one_batch();                                     // warm-up: buffers sized, graphs captured, RCCL channels up
            for (int rep = 0; rep < reps; rep++) {
                {
                    std::lock_guard<std::mutex> lk(gate.m);
                    gate.arrived = 0; gate.failed = 0;
                }
                times.push_back(one_batch());
            }
        } catch (...) {
            stop_workers();
            throw;
        }
        std::sort(times.begin(), times.end());
        const double t_batch = times[times.size() / 2];

        // every device holds the whole table; read device 0's copy and, as a check, the last device's
        std::vector<float> table((size_t)kRecord * (size_t)per_rank * (size_t)n, NAN), other(table.size());
        CHECK_HIP(hipSetDevice(0));
        CHECK_HIP(hipMemcpy(table.data(), ranks[0].d_recv, sizeof(float) * table.size(), hipMemcpyDeviceToHost));
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
        std::printf("gpus %d scans %d reps %d seconds_per_batch %.6f seconds_per_scan %.6f\n", n, n_scans, reps, t_batch, t_batch / n_scans);

        stop_workers();
        for (int d = 0; d < n; d++) {
            Rank& r = ranks[(size_t)d];
            (void)hipSetDevice(d);
            (void)s2m_destroy(r.h);
            for (void* p : r.d_scans) (void)hipFree(p);
            (void)hipHostFree(r.h_send);
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
