// Drives the host-side mirror of the reference interface the way src/rgbd_slam.cpp:48-57,109-112,291-297 does and
// prints the primitives as hex doubles; tests/test_gpu_host_shim.py compares them with the CPU oracle.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../rgb-d-slam_amd/host/primitives_shim.hpp"

using namespace rgbd_slam;
using namespace rgbd_slam::features::primitives;

int main(int argc, char** argv)
{
    if (argc < 8)
        return 2;
    const uint W = std::atoi(argv[2]), H = std::atoi(argv[3]);
    Parameters::set_camera_1(W, H, std::atof(argv[4]), std::atof(argv[5]), std::atof(argv[6]), std::atof(argv[7]));
    std::vector<float> depth(static_cast<size_t>(W) * H);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(depth.data(), sizeof(float), depth.size(), f) != depth.size())
        return 3;
    std::fclose(f);

    Depth_Map_Transformation depthOps(W, H, 20);
    Primitive_Detection detector(W, H);
    if (!detector.is_ready())
        return 4;
    const DepthImageView img {depth.data(), static_cast<int>(H), static_cast<int>(W), W};
    if (!depthOps.get_organized_cloud_array(img))
        return 5;
    plane_container planes;
    cylinder_container cylinders;
    detector.find_primitives(img, planes, cylinders);
    std::printf("planes %zu cylinders %zu\n", planes.size(), cylinders.size());
    for (const Plane& p : planes)
    {
        const auto n = p.get_normal();
        std::printf("P %a %a %a %a %zu %zu %.3f\n", n[0], n[1], n[2], p.get_d(), p.get_boundary_points().size(),
                    p.get_boundary_polygon().boundary_length(), p.get_boundary_polygon().get_area());
    }
    for (const Cylinder& c : cylinders)
        std::printf("C %a %a %a\n", c._normal[0], c._normal[1], c._normal[2]);
    // N2: MapPlane::find_matches selection with the frame's own planes standing in for projected map planes
    for (size_t i = 0; i < planes.size(); ++i)
    {
        const auto n = planes[i].get_normal();
        const int m = find_plane_match(planes, std::vector<bool>(planes.size(), false), {n[0], n[1], n[2], planes[i].get_d()},
                                       planes[i].get_boundary_polygon());
        std::printf("M %zu %d\n", i, m);
    }
    // N2 device part: the same frame twice -> every plane of "frame 1" overlaps its own copy in "frame 0"
    {
        std::vector<float> two(depth);
        two.insert(two.end(), depth.begin(), depth.end());
        std::vector<plane_container> bp;
        std::vector<cylinder_container> bc;
        detector.find_primitives_batch(two.data(), 2, bp, bc);
        std::vector<cape_frame_match> mm;
        if (!detector.match_consecutive(2, mm) || mm.size() != 2)
            return 7;
        std::printf("D %d %d", mm[1].n_prev, mm[1].n_cur);
        for (int j = 0; j < mm[1].n_prev; ++j)
            std::printf(" %d", mm[1].match[j]);
        std::printf("\n");
    }
    // rectify_depth with the default (identity) camera2 -> camera1 transform, then the rectified frame through the path
    std::vector<float> rect(depth.size());
    if (!depthOps.rectify_depth(img, rect.data()))
        return 6;
    size_t hits = 0;
    for (float v : rect)
        hits += v > 0;
    plane_container planes2;
    cylinder_container cylinders2;
    detector.find_primitives(DepthImageView {rect.data(), static_cast<int>(H), static_cast<int>(W), W}, planes2, cylinders2);
    std::printf("R %zu %zu\n", hits, planes2.size());
    detector.show_statistics(0.01, 1, true);
    return 0;
}
