// Dump harness for pinning the oracle against a REAL build of the reference (see README.md in this directory).
// This repo's own file; it is compiled against the reference tree where it lies and is never built in this image
// (Eigen / OpenCV / Boost are absent).  TEST INFRASTRUCTURE ONLY.
//
// Layout of <out.bin> (little endian): magic "CAPEREF1", int32 hCells, vCells, nPlanes, nCylinders,
//   int32 planeLabels[cells], int32 cylLabels[cells],
//   per cell: uint8 planar, uint32 pointCount, float64 normal[3], d, mse, score ; float32 tol
//   per plane: float64 normal[3], d, covariance[9] (row-major) ; per cylinder: float64 axis[3], radius
#define private public // harness only: reach _gridPlaneSegmentMap / _gridCylinderSegMap / _planeGrid / _cellDistanceTols
#define protected public
#include "features/primitives/depth_map_transformation.hpp"
#include "features/primitives/primitive_detection.hpp"
#undef private
#undef protected

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "parameters.hpp"

using namespace rgbd_slam;
using namespace rgbd_slam::features::primitives;

template <typename T> static void put(FILE* f, const T& v) { std::fwrite(&v, sizeof(T), 1, f); }

int main(int argc, char** argv)
{
    if (argc < 9)
    {
        std::fprintf(stderr, "usage: ref_dump depth.f32 width height fx fy cx cy out.bin\n");
        return 2;
    }
    const uint W = std::atoi(argv[2]), H = std::atoi(argv[3]);
    // intrinsics: the reference reads them from its YAML through Parameters::parse_file; the harness expects a config
    // file next to the depth frame named <depth.f32>.yaml written from examples/configuration_example.yaml with
    // fx fy cx cy substituted (the four command-line values are echoed for the record only)
    if (!Parameters::parse_file(std::string(argv[1]) + ".yaml"))
        return 3;
    cv::Mat_<float> depth(H, W);
    FILE* in = std::fopen(argv[1], "rb");
    if (!in || std::fread(depth.ptr<float>(0), sizeof(float), static_cast<size_t>(W) * H, in) != static_cast<size_t>(W) * H)
        return 4;
    std::fclose(in);

    Depth_Map_Transformation depthOps(W, H, parameters::detection::depthMapPatchSize_px);
    Primitive_Detection detector(W, H);
    matrixf cloud;
    if (!depthOps.get_organized_cloud_array(depth, cloud))
        return 5;
    plane_container planes;
    cylinder_container cylinders;
    detector.find_primitives(cloud, depth, planes, cylinders);

    FILE* f = std::fopen(argv[8], "wb");
    if (!f)
        return 6;
    std::fwrite("CAPEREF1", 1, 8, f);
    const int32_t hCells = detector._horizontalCellsCount, vCells = detector._verticalCellsCount;
    put(f, hCells);
    put(f, vCells);
    put(f, static_cast<int32_t>(planes.size()));
    put(f, static_cast<int32_t>(cylinders.size()));
    for (int r = 0; r < vCells; ++r)
        for (int c = 0; c < hCells; ++c)
            put(f, static_cast<int32_t>(detector._gridPlaneSegmentMap(r, c)));
    for (int r = 0; r < vCells; ++r)
        for (int c = 0; c < hCells; ++c)
            put(f, static_cast<int32_t>(detector._gridCylinderSegMap(r, c)));
    for (int i = 0; i < hCells * vCells; ++i)
    {
        const Plane_Segment& s = detector._planeGrid[i];
        put(f, static_cast<uint8_t>(s.is_planar()));
        put(f, static_cast<uint32_t>(s.get_point_count()));
        const vector3 n = s.get_normal();
        put(f, n.x());
        put(f, n.y());
        put(f, n.z());
        put(f, s.get_plane_d());
        put(f, s.get_MSE());
        put(f, s.get_score());
        put(f, static_cast<float>(detector._cellDistanceTols[i]));
    }
    for (const Plane& p : planes)
    {
        const vector3 n = p.get_normal();
        put(f, n.x());
        put(f, n.y());
        put(f, n.z());
        put(f, p.get_d());
        const matrix33 cov = p.get_point_cloud_covariance();
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                put(f, cov(r, c));
    }
    for (const Cylinder& c : cylinders)
    {
        put(f, c._normal.x());
        put(f, c._normal.y());
        put(f, c._normal.z());
        put(f, c._radius);
    }
    std::fclose(f);
    return 0;
}
