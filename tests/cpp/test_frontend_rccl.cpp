// test_frontend_rccl.cpp -- a C++ caller drives RecFilter::realize_sharded with a REAL RCCL collective.
//
// The C ABI's stepping contract says "the caller all-gathers" (include/recfilter_amd.h); test_frontend.cpp honours it with
// device copies between thread barriers.  Here the collective is ncclAllGather on a communicator this process initialises
// itself (ncclCommInitRank), handed to the front-end exactly as include/recfilter.hpp:AllGather documents:
//     ncclAllGather(send, gathered, bytes, ncclChar, comm, (hipStream_t)stream)
// The pool's boxes have one GPU and RCCL refuses two ranks on one device, so the communicator has ONE rank and the plans
// are built with RF_PLAN_FORCE_EXCHANGE: one slab with the whole exchange structure (exit carries, the gather walk, entering
// carries, the correction inside the final pass), every call of an N-GPU rank.  Cases: a row-sharded image (merged
// exchange, collective on the filter's own stream) and a z-sharded volume (early exchange: the front-end hands the
// collective a side stream and runs rf_plan_interior beside it).  Results against RecFilter::realize() of the same filter
// and against raster loops.  Run as a fresh child process by tests/test_gpu_parity.py (it must own the GPU from its
// first HIP call); with RANK / WORLD_SIZE / RCCL_ID_FILE set it joins a larger communicator (an id file written by rank 0),
// which is how a node with several GPUs would run it.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

#include <rccl/rccl.h>

#include "recfilter.hpp"

static int failures = 0;

#define NCCL_OK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { \
    std::fprintf(stderr, "%s failed: %s\n", #call, ncclGetErrorString(r_)); std::exit(3); } } while (0)
#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    std::fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_)); std::exit(2); } } while (0)

static float *upload(const std::vector<float> &h) {
    float *d = nullptr;
    HIP_OK(hipMalloc(&d, h.size() * sizeof(float)));
    HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

static std::vector<float> random_image(size_t n, unsigned seed) {
    std::vector<float> v(n);
    unsigned long long s = 0x9E3779B97F4A7C15ull * (seed + 1);
    for (auto &x : v) {       // SplitMix64 -> [0,1)
        s += 0x9E3779B97F4A7C15ull;
        unsigned long long z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        x = (float)(z >> 40) / 16777216.0f;
    }
    return v;
}

// strict pointwise metric of SURVEY 8d
static double rel_err(const std::vector<float> &ref, const std::vector<float> &out) {
    double worst = ref.size() == out.size() ? 0.0 : 1e30;
    for (size_t i = 0; i < ref.size() && i < out.size(); i++)
        worst = std::fmax(worst, std::fabs((double)ref[i] - out[i]) / std::fmax(std::fabs((double)ref[i]), 1e-6));
    return worst;
}

static void report(const char *name, double err, double tol = 1e-4) {
    std::printf("%-52s max rel err %.3e %s\n", name, err, err < tol ? "ok" : "FAILED");
    if (!(err < tol)) failures++;
}

// scan loops in the style of /root/reference/tests/test_generic_xyz.cpp:45-110; clamped: the taps beyond the border read the
// (partially updated) border sample, lib/recfilter.cpp:330-336
static void loop_scan(std::vector<float> &ref, int w, int h, int c, int dim, bool causal, const std::vector<float> &W, bool clamped = false) {
    const int ext[3] = {w, h, c};
    const int n = ext[dim];
    const long stride[3] = {1, w, (long)w * h};
    const int o1 = (dim + 1) % 3, o2 = (dim + 2) % 3;
    for (int u = 0; u < ext[o1]; u++) for (int v = 0; v < ext[o2]; v++) {
        const long base = u * stride[o1] + v * stride[o2];
        for (int r = 0; r < n; r++) {
            const int i = causal ? r : n - 1 - r;
            float acc = 0.0f;
            for (size_t j = 1; j < W.size(); j++) {
                if (r > (int)j - 1) acc += W[j] * ref[base + (causal ? i - (long)j : i + (long)j) * stride[dim]];
                else if (clamped) acc += W[j] * ref[base + (causal ? 0 : n - 1) * stride[dim]];
            }
            ref[base + i * stride[dim]] = W[0] * ref[base + i * stride[dim]] + acc;
        }
    }
}

int main() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::fprintf(stderr, "no GPU\n"); return 2; }
    const int rank = std::getenv("RANK") ? std::atoi(std::getenv("RANK")) : 0;
    const int world = std::getenv("WORLD_SIZE") ? std::atoi(std::getenv("WORLD_SIZE")) : 1;
    HIP_OK(hipSetDevice(rank % ndev));

    // ---- the communicator: ncclGetUniqueId on rank 0, passed to the others through a file ---------------------------------
    ncclUniqueId id;
    const char *id_file = std::getenv("RCCL_ID_FILE");
    if (rank == 0) {
        NCCL_OK(ncclGetUniqueId(&id));
        if (world > 1) {
            if (!id_file) { std::fprintf(stderr, "WORLD_SIZE > 1 needs RCCL_ID_FILE\n"); return 2; }
            std::ofstream f(std::string(id_file) + ".tmp", std::ios::binary);
            f.write(reinterpret_cast<const char *>(&id), sizeof(id));
            f.close();
            std::rename((std::string(id_file) + ".tmp").c_str(), id_file);
        }
    } else {
        if (!id_file) { std::fprintf(stderr, "WORLD_SIZE > 1 needs RCCL_ID_FILE\n"); return 2; }
        for (int tries = 0;; tries++) {
            std::ifstream f(id_file, std::ios::binary);
            if (f.read(reinterpret_cast<char *>(&id), sizeof(id))) break;
            if (tries > 600) { std::fprintf(stderr, "no RCCL id after 60 s\n"); return 2; }
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    }
    ncclComm_t comm;
    NCCL_OK(ncclCommInitRank(&comm, world, id, rank));
    int comm_ranks = 0;
    NCCL_OK(ncclCommCount(comm, &comm_ranks));
    std::printf("RCCL communicator: %d rank(s), this is rank %d\n", comm_ranks, rank);

    int collectives = 0, on_side_stream = 0;
    hipStream_t st;
    HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    RecFilter::AllGather gather = [&](const void *send, void *gathered, size_t bytes, void *stream) {
        collectives++;
        if ((hipStream_t)stream != st) on_side_stream++;
        NCCL_OK(ncclAllGather(send, gathered, bytes, ncclChar, comm, (hipStream_t)stream));
    };
    const std::vector<float> W = {0.0975842401f, 1.5283848f, -0.625968993f};       // gaussian_weights(5, 2), SURVEY a-14

    {   // ---- a row-sharded image: every rank owns `rows` rows of a (world * rows) x width image ---------------------------
        const int width = 1024, rows = 384;
        std::vector<float> image = random_image((size_t)width * rows * world, 5);
        std::vector<float> slab(image.begin() + (size_t)rank * rows * width, image.begin() + (size_t)(rank + 1) * rows * width);
        float *d = upload(slab);
        RecFilterDim x("x", width), y("y", rows);
        RecFilter F;
        F.set_clamped_image_border();
        F(x, y) = RecFilterImage(d);
        F.add_filter(+x, W); F.add_filter(-x, W); F.add_filter(+y, W); F.add_filter(-y, W);
        F.split_all_dimensions(32);
        if (world > 1) F.shard(rank, world, std::vector<int64_t>((size_t)world, rows));
        else F.plan_options(RF_PLAN_FORCE_EXCHANGE | RF_PLAN_TILED_ONLY);
        F.set_stream((void *)st);
        const int before = collectives;
        std::vector<float> got = F.realize_sharded(gather).to_host<float>();
        std::vector<float> again = F.realize_sharded(gather).to_host<float>();      // the exchange buffers are reused
        if (collectives - before != 2) { failures++; std::printf("row shards: expected ONE ncclAllGather per realization, saw %d for two\n", collectives - before); }
        report("row shards, second realization == first", rel_err(got, again), 1e-12);
        {
            std::vector<float> ref = image;
            for (int dim = 0; dim < 2; dim++) { loop_scan(ref, width, rows * world, 1, dim, true, W, true); loop_scan(ref, width, rows * world, 1, dim, false, W, true); }
            std::vector<float> mine(ref.begin() + (size_t)rank * rows * width, ref.begin() + (size_t)(rank + 1) * rows * width);
            report("row shards over ncclAllGather vs raster loops", rel_err(mine, got));
        }
        if (world == 1) {
            // the same filter without the exchange structure
            RecFilter G;
            G.set_clamped_image_border();
            G(x, y) = RecFilterImage(d);
            G.add_filter(+x, W); G.add_filter(-x, W); G.add_filter(+y, W); G.add_filter(-y, W);
            G.split_all_dimensions(32);
            G.plan_options(RF_PLAN_TILED_ONLY);
            G.set_stream((void *)st);
            report("row shards over ncclAllGather vs realize()", rel_err(G.realize().to_host<float>(), got));
        }
        (void)hipFree(d);
    }
    {   // ---- a z-sharded volume: the early exchange, collective on a side stream beside rf_plan_interior ------------------
        const int nx = 256, ny = 64, nz = 64;
        const std::vector<float> Wz = {0.40f, 0.70f, -0.20f};
        std::vector<float> volume = random_image((size_t)nx * ny * nz * world, 9);
        std::vector<float> slab(volume.begin() + (size_t)rank * nx * ny * nz, volume.begin() + (size_t)(rank + 1) * nx * ny * nz);
        float *d = upload(slab);
        RecFilterDim x("x", nx), y("y", ny), z("z", nz);
        auto define = [&](RecFilter &F) {
            F(x, y, z) = RecFilterImage(d);
            F.add_filter(+x, Wz); F.add_filter(-x, Wz); F.add_filter(+y, Wz); F.add_filter(-y, Wz);
            F.add_filter(+z, Wz); F.add_filter(-z, Wz);
            F.split(x, 32, y, 16, z, 32);
            F.set_stream((void *)st);
        };
        RecFilter F;
        define(F);
        if (world > 1) F.shard(rank, world, std::vector<int64_t>((size_t)world, nz));
        else F.plan_options(RF_PLAN_FORCE_EXCHANGE | RF_PLAN_TILED_ONLY);
        const int before = collectives, side_before = on_side_stream;
        std::vector<float> got = F.realize_sharded(gather).to_host<float>();
        if (collectives - before != 1 || on_side_stream - side_before != 1) {
            failures++;
            std::printf("z slabs: expected ONE ncclAllGather, on a side stream; saw %d, %d on a side stream\n", collectives - before,
                        on_side_stream - side_before);
        }
        if (world == 1) {
            RecFilter G;
            define(G);
            G.plan_options(RF_PLAN_TILED_ONLY);
            report("z slabs over ncclAllGather vs realize()", rel_err(G.realize().to_host<float>(), got));
            std::vector<float> ref = slab;
            for (int dim = 0; dim < 3; dim++) { loop_scan(ref, nx, ny, nz, dim, true, Wz); loop_scan(ref, nx, ny, nz, dim, false, Wz); }
            report("z slabs over ncclAllGather vs raster loops", rel_err(ref, got));
        }
        if (world > 1) {
            std::vector<float> ref = volume;
            for (int dim = 0; dim < 3; dim++) { loop_scan(ref, nx, ny, nz * world, dim, true, Wz); loop_scan(ref, nx, ny, nz * world, dim, false, Wz); }
            std::vector<float> mine(ref.begin() + (size_t)rank * nx * ny * nz, ref.begin() + (size_t)(rank + 1) * nx * ny * nz);
            report("z slabs over ncclAllGather vs raster loops", rel_err(mine, got));
        }
        (void)hipFree(d);
    }
    HIP_OK(hipStreamSynchronize(st));
    NCCL_OK(ncclCommDestroy(comm));
    (void)hipStreamDestroy(st);
    std::printf("%s (%d ncclAllGather calls on %d rank(s))\n", failures ? "SOME TESTS FAILED" : "all rccl front-end tests passed", collectives,
                comm_ranks);
    return failures ? 1 : 0;
}
