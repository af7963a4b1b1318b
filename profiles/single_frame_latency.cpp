// Latency of the reference's call pattern: ONE 640x480 host frame in, primitives out (src/rgbd_slam.cpp:291-297), measured
// in C++ on the host clock, PCIe both ways included.  Three ways to make the call:
//   abi/pageable : cape_extract_host on an ordinary malloc'ed image (what a cv::Mat holds) + cape_host_results
//   abi/pinned   : the image lives in cape_host_alloc'ed memory: the streaming kernel reads it over PCIe, no staging copy
//   abi/pinned u16 : the same through cape_extract_u16_host on the raw uint16 image (half the bytes over the link)
//   overlay      : Primitive_Detection::find_primitives of the host overlay (pageable image, boundary polygons included)
// usage: latency_bench <frames.f32> <n_frames> <width> <height> <fx> <fy> <cx> <cy> <cylinders 0|1>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "outputs/logger.hpp"
#include "parameters.hpp"
#include "primitives_shim.hpp"

using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
static void report(const char* name, std::vector<double>& t)
{
    std::sort(t.begin(), t.end());
    std::printf("%-14s median %7.1f us   p10 %7.1f   p90 %7.1f   (%zu calls)\n", name, t[t.size() / 2], t[t.size() / 10], t[t.size() * 9 / 10], t.size());
}

int main(int argc, char** argv)
{
    if (argc < 10)
        return 2;
    const int nFrames = std::atoi(argv[2]), W = std::atoi(argv[3]), H = std::atoi(argv[4]);
    const double fx = std::atof(argv[5]), fy = std::atof(argv[6]), cx = std::atof(argv[7]), cy = std::atof(argv[8]);
    const bool cyl = std::atoi(argv[9]) != 0;
    const size_t px = static_cast<size_t>(W) * H;
    std::vector<float> frames(px * nFrames);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(frames.data(), sizeof(float), frames.size(), f) != frames.size())
        return 3;
    std::fclose(f);
    const int reps = 40;

    cape_config cfg {};
    cfg.width = W; cfg.height = H; cfg.fx = fx; cfg.fy = fy; cfg.cx = cx; cfg.cy = cy;
    cfg.flags = cyl ? CAPE_FLAG_CYLINDERS : 0u;
    cfg.max_batch = 1;
    cape_handle h = nullptr;
    if (cape_create(&cfg, &h) != CAPE_OK)
    {
        std::fprintf(stderr, "cape_create: %s\n", cape_last_error());
        return 4;
    }
    const cape_frame_record* rec = nullptr;
    const double* bnd = nullptr;
    long planes = 0;
    std::vector<double> t;
    // ---- abi / pageable
    for (int r = 0; r < reps + 2; ++r)
        for (int k = 0; k < nFrames; ++k)
        {
            const auto t0 = clk::now();
            if (cape_extract_host(h, frames.data() + px * k, 1, nullptr) != CAPE_OK || cape_host_results(h, &rec, nullptr, nullptr, &bnd) != CAPE_OK)
                return 5;
            planes += rec->header.n_planes;
            if (r >= 2)
                t.push_back(us(t0, clk::now()));
        }
    report("abi/pageable", t);
    // ---- abi / pinned
    float* pinned = nullptr;
    if (cape_host_alloc(h, px * nFrames * sizeof(float), reinterpret_cast<void**>(&pinned)) != CAPE_OK)
        return 6;
    std::memcpy(pinned, frames.data(), px * nFrames * sizeof(float));
    t.clear();
    cape_enable_timing(h, 1);
    for (int r = 0; r < reps + 2; ++r)
        for (int k = 0; k < nFrames; ++k)
        {
            if (r == 2 && k == 0)
                cape_reset_timings(h);
            const auto t0 = clk::now();
            if (cape_extract_host(h, pinned + px * k, 1, nullptr) != CAPE_OK || cape_host_results(h, &rec, nullptr, nullptr, &bnd) != CAPE_OK)
                return 7;
            planes += rec->header.n_planes;
            if (r >= 2)
                t.push_back(us(t0, clk::now()));
        }
    report("abi/pinned", t);
    cape_timings tm {};
    cape_get_timings(h, &tm);
    if (tm.calls)
        std::printf("               device time per call (HIP events, pinned run): moments %.1f us  plane %.1f us  grow %.1f us\n",
                    1e6 * tm.cell_moments_s / tm.calls, 1e6 * tm.cell_plane_s / tm.calls, 1e6 * tm.grow_s / tm.calls);
    cape_host_free(h, pinned);
    // ---- abi / pinned, raw uint16 (the sensor's own format, examples/main_TUM.cpp:242: half the bytes over the link); only when the
    // frames ARE whole millimetres below 65 536, so that the call sees the same depths as the rows above
    bool whole = true;
    for (size_t i = 0; i < frames.size() && whole; ++i)
        whole = frames[i] >= 0.0f && frames[i] < 65536.0f && frames[i] == static_cast<float>(static_cast<uint16_t>(frames[i]));
    uint16_t* pinned16 = nullptr;
    if (whole && cape_host_alloc(h, px * nFrames * sizeof(uint16_t), reinterpret_cast<void**>(&pinned16)) == CAPE_OK)
    {
        for (size_t i = 0; i < frames.size(); ++i)
            pinned16[i] = static_cast<uint16_t>(frames[i]);
        t.clear();
        long planes16 = 0, planes32 = 0;
        for (int r = 0; r < reps + 2; ++r)
            for (int k = 0; k < nFrames; ++k)
            {
                const auto t0 = clk::now();
                if (cape_extract_u16_host(h, pinned16 + px * k, 1.0f, 1, nullptr) != CAPE_OK || cape_host_results(h, &rec, nullptr, nullptr, &bnd) != CAPE_OK)
                    return 9;
                planes16 += rec->header.n_planes;
                if (r >= 2)
                    t.push_back(us(t0, clk::now()));
            }
        for (int k = 0; k < nFrames; ++k)
        {
            if (cape_extract_host(h, frames.data() + px * k, 1, nullptr) != CAPE_OK || cape_host_results(h, &rec, nullptr, nullptr, &bnd) != CAPE_OK)
                return 9;
            planes32 += rec->header.n_planes * (reps + 2);
        }
        report("abi/pinned u16", t);
        if (planes16 != planes32)
            std::printf("               !! the uint16 calls saw %ld planes, the float32 calls %ld\n", planes16, planes32);
        cape_host_free(h, pinned16);
    }
    cape_destroy(h);

    // ---- overlay (cylinder branch always on, like the reference)
    rgbd_slam::Parameters::set_camera_1(W, H, fx, fy, cx, cy);
    rgbd_slam::outputs::set_log_callback([](int, const std::string&) {});
    rgbd_slam::features::primitives::Primitive_Detection det(W, H);
    if (!det.is_ready())
        return 8;
    rgbd_slam::matrixf cloud;
    rgbd_slam::features::primitives::plane_container pc;
    rgbd_slam::features::primitives::cylinder_container cc;
    t.clear();
    for (int r = 0; r < reps + 2; ++r)
        for (int k = 0; k < nFrames; ++k)
        {
            const rgbd_slam::features::primitives::depth_image img(H, W, frames.data() + px * k);
            const auto t0 = clk::now();
            det.find_primitives(cloud, img, pc, cc);
            planes += static_cast<long>(pc.size());
            if (r >= 2)
                t.push_back(us(t0, clk::now()));
        }
    report("overlay", t);
    std::printf("               (%ld planes seen in total)\n", planes);

    // ---- overlay, batches of 64 host frames (the additions): containers per frame, polygons on the device or by the host class,
    //      float32 or raw uint16 input
    {
        const char* eB = std::getenv("CAPE_BENCH_BATCH");
        const char* eC = std::getenv("CAPE_BENCH_CHUNK");
        const int B = eB ? std::atoi(eB) : 64;
        if (eC)
            det.set_chunk_frames(std::atoi(eC));
        std::vector<float> batch(px * B);
        std::vector<uint16_t> raw(px * B);
        for (int k = 0; k < B; ++k)
            std::memcpy(batch.data() + px * k, frames.data() + px * (k % nFrames), px * sizeof(float));
        for (size_t i = 0; i < batch.size(); ++i)
        {
            const float v = batch[i] * 5.0f;
            raw[i] = v > 0.0f && v < 65535.0f ? static_cast<uint16_t>(v + 0.5f) : 0;
        }
        std::vector<rgbd_slam::features::primitives::plane_container> bp;
        std::vector<rgbd_slam::features::primitives::cylinder_container> bc;
        for (int mode = 0; mode < 9; ++mode)
        {
            // (more shards than devices: the handles' copies, kernels and host work overlap on the one device)
            det.set_shard_count(mode == 3 || mode == 7 ? 2 : mode == 4 || mode == 8 ? 4 : mode == 5 || mode == 6 ? 0 : 1);
            det.set_device_polygons(mode != 1);
            double best = 1e30;
            for (int r = 0; r < 6; ++r)
            {
                const auto t0 = clk::now();
                if (mode >= 2 && mode < 6)
                    det.find_primitives_batch(raw.data(), 0.2f, B, bp, bc);
                else
                    det.find_primitives_batch(batch.data(), B, bp, bc);
                const double dt = us(t0, clk::now());
                if (r >= 1 && dt < best)
                    best = dt;
            }
            size_t np = 0;
            for (const auto& c : bp)
                np += c.size();
            std::printf("overlay batch of %d  %-44s %8.1f us per frame (%.0f frames/s, %zu planes per batch)\n", B,
                        mode == 0 ? "float32 frames, polygons on the device" : mode == 1 ? "float32 frames, polygons by the host class" : mode == 2 ? "raw uint16 frames, polygons on the device" : mode == 3 ? "raw uint16 frames, two shards on the device" : mode == 4 ? "raw uint16 frames, four shards on the device" : mode == 5 ? "raw uint16 frames, default shards" : mode == 6 ? "float32 frames, default shards" : mode == 7 ? "float32 frames, two shards on the device" : "float32 frames, four shards on the device",
                        best / B, 1e6 * B / best, np);
        }
    }
    return 0;
}
