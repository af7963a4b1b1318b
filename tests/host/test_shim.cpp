// Drives the host-side replacement of the reference's `primitives` library the way src/rgbd_slam.cpp:48-57,109-112,
// 291-297,335 drives the original, and prints the primitives as hex doubles; tests/test_gpu_host_shim.py compares them
// with the CPU oracle.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "primitives_shim.hpp"

using namespace rgbd_slam;
using namespace rgbd_slam::features::primitives;

static void print_frame(const char* tag, const plane_container& planes, const cylinder_container& cylinders)
{
    std::printf("%splanes %zu cylinders %zu\n", tag, planes.size(), cylinders.size());
    for (const Plane& p : planes)
    {
        const vector3 n = p.get_normal();
        const matrix33 cov = p.get_point_cloud_covariance();
        std::printf("%sP %a %a %a %a %zu %.3f %a %a\n", tag, n.x(), n.y(), n.z(), p.get_d(), p.get_boundary_polygon().boundary_length(),
                    p.get_boundary_polygon().get_area(), cov(0, 0), cov(2, 1));
    }
    for (const Cylinder& c : cylinders)
        std::printf("%sC %a %a %a %d\n", tag, c._normal.x(), c._normal.y(), c._normal.z(), c._radius != c._radius ? 1 : 0);
}

int main(int argc, char** argv)
{
    if (argc < 8)
        return 2;
    const uint W = std::atoi(argv[2]), H = std::atoi(argv[3]);
    Parameters::set_camera_1(W, H, std::atof(argv[4]), std::atof(argv[5]), std::atof(argv[6]), std::atof(argv[7]));
    const int batchFrames = argc > 8 ? std::atoi(argv[8]) : 0; // extra frames appended to the file: sharded batch test
    const int shards = argc > 9 ? std::atoi(argv[9]) : 0;
    std::vector<float> depth(static_cast<size_t>(W) * H * (1 + batchFrames));
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(depth.data(), sizeof(float), depth.size(), f) != depth.size())
        return 3;
    std::fclose(f);

    // src/rgbd_slam.cpp:48-57
    auto depthOps = std::make_unique<Depth_Map_Transformation>(W, H, parameters::detection::depthMapPatchSize_px);
    auto detector = std::make_unique<Primitive_Detection>(W, H);
    if (!depthOps->is_ok() || !detector->is_ready())
        return 4;
    const depth_image img(static_cast<int>(H), static_cast<int>(W), depth.data());
    // :109-112
    matrixf cloudArrayOrganized;
    if (!depthOps->get_organized_cloud_array(img, cloudArrayOrganized))
        return 5;
    // :293-296
    plane_container planes;
    cylinder_container cylinders;
    detector->find_primitives(cloudArrayOrganized, img, planes, cylinders);
    print_frame("", planes, cylinders);

    // N2: MapPlane::find_matches selection with the frame's own planes standing in for projected map planes
    for (size_t i = 0; i < planes.size(); ++i)
    {
        const int m = find_plane_match(planes, std::vector<bool>(planes.size(), false), planes[i].get_parametrization(),
                                       planes[i].get_boundary_polygon());
        std::printf("M %zu %d\n", i, m);
    }
    // N2 device part: the same frame twice -> every plane of "frame 1" overlaps its own copy in "frame 0"
    {
        std::vector<float> two(depth.begin(), depth.begin() + static_cast<size_t>(W) * H);
        two.insert(two.end(), depth.begin(), depth.begin() + static_cast<size_t>(W) * H);
        std::vector<plane_container> bp;
        std::vector<cylinder_container> bc;
        detector->set_shard_count(1);
        detector->find_primitives_batch(two.data(), 2, bp, bc);
        std::vector<cape_frame_match> mm;
        if (!detector->match_consecutive(2, mm) || mm.size() != 2)
            return 7;
        std::printf("D %d %d", mm[1].n_prev, mm[1].n_cur);
        for (int j = 0; j < mm[1].n_prev; ++j)
            std::printf(" %d", mm[1].match[j]);
        std::printf("\n");
    }
    // sharded batch: every frame of the file, cut in contiguous blocks over `shards` handles (device = shard % devices)
    if (batchFrames > 0)
    {
        std::vector<plane_container> bp;
        std::vector<cylinder_container> bc;
        detector->set_shard_count(shards);
        detector->find_primitives_batch(depth.data(), 1 + batchFrames, bp, bc);
        std::printf("S %d %zu\n", detector->shard_count(), bp.size());
        for (size_t k = 0; k < bp.size(); ++k)
        {
            char tag[16];
            std::snprintf(tag, sizeof tag, "B%zu ", k);
            print_frame(tag, bp[k], bc[k]);
        }
        // the same batch with the boundary polygons built by the host class, plane by plane: identical lines expected
        // (the device kernel and the host class run the same statements)
        std::vector<plane_container> hp;
        std::vector<cylinder_container> hc;
        detector->set_device_polygons(false);
        detector->find_primitives_batch(depth.data(), 1 + batchFrames, hp, hc);
        detector->set_device_polygons(true);
        for (size_t k = 0; k < hp.size(); ++k)
        {
            char tag[16];
            std::snprintf(tag, sizeof tag, "H%zu ", k);
            print_frame(tag, hp[k], hc[k]);
            // and vertex for vertex
            bool same = hp[k].size() == bp[k].size();
            for (size_t i = 0; same && i < hp[k].size(); ++i)
            {
                // get_boundary_polygon() returns by value (like the reference's): keep the copies alive while their rings are read
                const auto pa = hp[k][i].get_boundary_polygon(), pb = bp[k][i].get_boundary_polygon();
                const auto& a = pa.boundary();
                const auto& b = pb.boundary();
                same = a.size() == b.size();
                for (size_t v = 0; same && v < a.size(); ++v)
                {
                    same = a[v][0] == b[v][0] && a[v][1] == b[v][1];
                    if (!same)
                    {
                        std::fprintf(stderr, "polygon vertex differs: frame %zu plane %zu vertex %zu of %zu: host %a %a device %a %a\n", k, i, v, a.size(),
                                     a[v][0], a[v][1], b[v][0], b[v][1]);
                        for (size_t w = 0; w < a.size(); ++w)
                            std::fprintf(stderr, "   %zu host %.3f %.3f device %.3f %.3f\n", w, a[w][0], a[w][1], b[w][0], b[w][1]);
                        const auto ca = pa.get_center(), cb = pb.get_center(), xa = pa.get_x_axis(), xb = pb.get_x_axis();
                        std::fprintf(stderr, "   centers %.3f %.3f %.3f | %.3f %.3f %.3f  x axes %.4f %.4f %.4f | %.4f %.4f %.4f\n", ca[0], ca[1], ca[2], cb[0], cb[1],
                                     cb[2], xa[0], xa[1], xa[2], xb[0], xb[1], xb[2]);
                    }
                }
                if (a.size() != b.size())
                    std::fprintf(stderr, "polygon size differs: frame %zu plane %zu: host %zu device %zu\n", k, i, a.size(), b.size());
            }
            std::printf("V%zu %d\n", k, same ? 1 : 0);
        }
    }
    // N4 through the overlay: the batch as raw 16-bit sensor images (depth quantised to 0.2 mm like a TUM PNG) == the same
    // quantised depths handed over as float32
    if (batchFrames > 0)
    {
        const size_t count = static_cast<size_t>(W) * H * (1 + batchFrames);
        std::vector<uint16_t> raw(count);
        std::vector<float> quantised(count);
        for (size_t i = 0; i < count; ++i)
        {
            const float v = depth[i] * 5.0f;
            raw[i] = v > 0.0f && v < 65535.0f ? static_cast<uint16_t>(v + 0.5f) : 0;
            quantised[i] = static_cast<float>(raw[i]) * 0.2f;
        }
        std::vector<plane_container> fp, up;
        std::vector<cylinder_container> fc, uc;
        detector->set_shard_count(shards);
        detector->find_primitives_batch(quantised.data(), 1 + batchFrames, fp, fc);
        detector->find_primitives_batch(raw.data(), 0.2f, 1 + batchFrames, up, uc);
        bool same = fp.size() == up.size();
        size_t planesSeen = 0;
        for (size_t k = 0; same && k < fp.size(); ++k)
        {
            same = fp[k].size() == up[k].size() && fc[k].size() == uc[k].size();
            for (size_t i = 0; same && i < fp[k].size(); ++i)
            {
                const vector3 a = fp[k][i].get_normal(), b = up[k][i].get_normal();
                same = a.x() == b.x() && a.y() == b.y() && a.z() == b.z() && fp[k][i].get_d() == up[k][i].get_d();
                ++planesSeen;
            }
        }
        std::printf("U %d %zu\n", same ? 1 : 0, planesSeen);
    }
    // N2 on the polygons: the device's find_matches between consecutive frames of a one-shard batch against find_plane_match
    // on the containers the batch returned (same planes, same polygons, same order)
    if (batchFrames > 0)
    {
        std::vector<plane_container> bp;
        std::vector<cylinder_container> bc;
        detector->set_shard_count(1);
        const int n = 1 + batchFrames;
        detector->find_primitives_batch(depth.data(), n, bp, bc);
        std::vector<cape_frame_match_exact> pm;
        if (!detector->match_consecutive_polygons(n, pm) || pm.size() != static_cast<size_t>(n))
            return 8;
        int previous = 0, mismatches = 0, matched = 0;
        for (int f = 1; f < n; ++f)
        {
            if (pm[f].flags & CAPE_MATCH_EXACT_OVERFLOW)
                continue;
            if (pm[f].n_prev != static_cast<int>(bp[f - 1].size()) || pm[f].n_cur != static_cast<int>(bp[f].size()))
            {
                ++mismatches;
                continue;
            }
            std::vector<bool> isMatched(bp[f].size(), false);
            for (size_t j = 0; j < bp[f - 1].size(); ++j)
            {
                const int m = find_plane_match(bp[f], isMatched, bp[f - 1][j].get_parametrization(), bp[f - 1][j].get_boundary_polygon());
                if (m >= 0)
                    isMatched[m] = true;
                ++previous;
                matched += m >= 0;
                mismatches += m != pm[f].match[j];
            }
        }
        std::printf("X %d %d %d %d\n", n, previous, matched, mismatches);
    }
    // N2 over a whole batch, whatever its sharding: set_batch_matching matches every chunk on the device while the batch runs and
    // stitches the chunk / shard boundaries with the host class.  Three shards (frame pairs across two shard boundaries are the
    // host's) must give the decisions of one shard (every pair on the device) -- with a camera pose in both.
    if (batchFrames > 0)
    {
        const int n = 1 + batchFrames;
        // a small rigid motion between consecutive frames: 1.5 degrees about y, (20, -10, 15) mm
        const double c = 0.99965732497555726, sn = 0.026176948307873153;
        std::vector<double> poses(static_cast<size_t>(n) * 16, 0.0);
        for (int f = 0; f < n; ++f)
        {
            double* T = poses.data() + static_cast<size_t>(f) * 16;
            T[0] = c, T[2] = sn, T[5] = 1.0, T[8] = -sn, T[10] = c, T[15] = 1.0;
            T[3] = 20.0, T[7] = -10.0, T[11] = 15.0;
        }
        std::vector<plane_container> p1, p3;
        std::vector<cylinder_container> c1, c3;
        detector->set_batch_matching(true, false, true, poses.data());
        detector->set_shard_count(1);
        detector->find_primitives_batch(depth.data(), n, p1, c1);
        const std::vector<cape_frame_match_exact> one = detector->batch_matches();
        detector->set_shard_count(3);
        detector->find_primitives_batch(depth.data(), n, p3, c3);
        const std::vector<cape_frame_match_exact>& three = detector->batch_matches();
        std::vector<cape_frame_match_exact> viaCall;
        const bool served = detector->match_consecutive_polygons(n, viaCall, false, true, poses.data()) && viaCall.size() == static_cast<size_t>(n);
        // the table answers only the question it was computed for -- other flags on a batch that is spread over three
        // shards are refused, not served with the wrong parameters
        std::vector<cape_frame_match_exact> other;
        const bool refused = !detector->match_consecutive_polygons(n, other, true, true, poses.data()) && other.empty();
        detector->set_batch_matching(false);
        int hostEntries = 0, hostEntriesOne = 0, mismatches = 0, matched = 0;
        if (one.size() != static_cast<size_t>(n) || three.size() != static_cast<size_t>(n) || !served)
            return 9;
        if (!refused)
            return 10;
        for (int f = 0; f < n; ++f)
        {
            hostEntries += (three[f].flags & CAPE_MATCH_EXACT_HOST) != 0;
            hostEntriesOne += (one[f].flags & CAPE_MATCH_EXACT_HOST) != 0;
            mismatches += one[f].n_prev != three[f].n_prev || one[f].n_cur != three[f].n_cur;
            for (int j = 0; j < one[f].n_prev && j < CAPE_MATCH_MAX_PLANES; ++j)
            {
                mismatches += one[f].match[j] != three[f].match[j] || viaCall[f].match[j] != three[f].match[j];
                matched += one[f].match[j] >= 0;
            }
        }
        std::printf("Y %d %d %d %d %d\n", n, hostEntriesOne, hostEntries, matched, mismatches);
    }
    // rectify_depth with the default (identity) camera2 -> camera1 transform, then the rectified frame through the path
    depth_image rect;
    if (!depthOps->rectify_depth(img, rect))
        return 6;
    size_t hits = 0;
    for (int r = 0; r < rect.rows; ++r)
        for (int c = 0; c < rect.cols; ++c)
            hits += rect(r, c) > 0;
    plane_container planes2;
    cylinder_container cylinders2;
    detector->find_primitives(cloudArrayOrganized, rect, planes2, cylinders2);
    std::printf("R %zu %zu\n", hits, planes2.size());
    detector->show_statistics(0.01, 1, true); // :335
    return 0;
}
