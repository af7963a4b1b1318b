// Latency of the reference's call pattern: ONE 640x480 host frame in, primitives out (src/rgbd_slam.cpp:291-297), measured
// in C++ on the host clock, PCIe both ways included.  Three ways to make the call:
//   abi/pageable : cape_extract_host on an ordinary malloc'ed image (what a cv::Mat holds) + cape_host_results
//   abi/pinned   : the image lives in cape_host_alloc'ed memory: the streaming kernel reads it over PCIe, no staging copy
//   overlay      : Primitive_Detection::find_primitives of the host overlay (pageable image, boundary polygons included)
// usage: latency_bench <frames.f32> <n_frames> <width> <height> <fx> <fy> <cx> <cy> <cylinders 0|1>
#include <algorithm>
#include <chrono>
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
    return 0;
}
