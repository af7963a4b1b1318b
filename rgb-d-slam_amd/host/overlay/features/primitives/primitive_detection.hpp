// Replacement for reference src/features/primitives/primitive_detection.hpp:27-241 inside the `primitives` library:
// same class name, constructor, find_primitives and show_statistics signatures (src/rgbd_slam.cpp:57, :293-296, :335
// compile against it unchanged).  Everything numeric happens behind the C ABI of libcape_hip.so (include/cape_hip.h);
// this class converts containers, builds the boundary polygons on the host (primitive_detection.cpp:622) and keeps the
// reference's error convention (noexcept, log and skip).
//
// Additions (the reference extracts one frame per call on one device):
//   find_primitives_batch : frames are independent, so a batch is cut in contiguous blocks over every visible GPU
//                           (one handle + one host thread per device) and the lists come back in frame
//                           order;
//   match_consecutive     : device pre-filter for MapPlane::find_matches;
//   match_consecutive_polygons : find_matches itself between consecutive frames, on the device polygons.
#ifndef RGBDSLAM_FEATURES_PRIMITIVES_PRIMITIVEDETECTION_HPP
#define RGBDSLAM_FEATURES_PRIMITIVES_PRIMITIVEDETECTION_HPP

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "cape_hip.h"
#include "depth_image.hpp"
#include "shape_primitives.hpp"
#include "types.hpp"

namespace rgbd_slam::features::primitives {

struct PolygonPool; // a few worker threads for the host-side boundary polygons of the one-frame call (primitive_detection.cpp)

class Primitive_Detection
{
  public:
    Primitive_Detection(const uint width, const uint height);
    ~Primitive_Detection();

    // primitive_detection.hpp:41-44.  depthMatrix (the organised cloud) is not read: see depth_map_transformation.hpp
    void find_primitives(const matrixf& depthMatrix,
                         const depth_image& depthImage,
                         plane_container& planeContainer,
                         cylinder_container& primitiveContainer) noexcept;

    // primitive_detection.hpp:46-49
    void show_statistics(const double meanFrameTreatmentDuration,
                         const uint frameCount,
                         const bool shouldDisplayDetails = false) const noexcept;

    // ---- additions -------------------------------------------------------------------------------------------
    // depth: n_frames contiguous row-major float32 images on the host.  Sharded over the visible devices (or over
    // set_shard_count() handles); planes[f] / cylinders[f] are frame f's containers whatever device produced them.
    void find_primitives_batch(const float* depth,
                               int n_frames,
                               std::vector<plane_container>& planes,
                               std::vector<cylinder_container>& cylinders) noexcept;
    // the same from the raw 16-bit sensor images (what the datasets' depth PNGs decode to; depth = raw * scale, e.g. 1/5 for
    // TUM, examples/main_TUM.cpp:221,242): half the bytes over PCIe, the conversion happens on the device
    void find_primitives_batch(const uint16_t* raw,
                               float scale,
                               int n_frames,
                               std::vector<plane_container>& planes,
                               std::vector<cylinder_container>& cylinders) noexcept;
    // 0 (default): one to four shards per visible device, by the size of the batch.  k > 0: k shards, shard i on device i % device_count -- more shards than
    // devices are legal (used by the tests to exercise the sharding on a one-GPU box).  Takes effect at the next batch.
    void set_shard_count(int shards) noexcept { _requestedShards = shards; }
    // frames a shard sends through its handle at a time (default 256).  Larger chunks amortise the fixed latencies of a pass
    // (the polygon kernels of 64 frames take 0.76 ms, of 512 frames 1.0 ms); takes effect when the shards are (re)created:
    // call it before the first batch.
    void set_chunk_frames(int frames) noexcept { _maxBatch = frames > 8 ? frames : 9; }
    // batches: boundary polygons on the device (default) or with the host class, plane by plane (A/B runs, tests)
    void set_device_polygons(bool on) noexcept { _devicePolygons = on; }
    // The five buckets show_statistics(.., shouldDisplayDetails = true) prints -- reset / init / grow / merge / refine,
    // primitive_detection.cpp:69-117 -- need HIP events around the kernels and three atomics per frame inside the grow kernel
    // (cape_enable_timing): a few microseconds on the one-frame call, hence opt-in (also: environment CAPE_DETAILED_STATISTICS=1).
    void set_detailed_statistics(bool on) noexcept;
    // The reference's utils::Random::_seed (src/utils/random.hpp:59-64): 0 under MAKE_DETERMINISTIC (the default here), std::time(0)
    // of the process otherwise; every frame's RANSAC restarts there (thread_local engine on a fresh thread per frame).
    void set_random_seed(uint32_t seed) noexcept;
    [[nodiscard]] int shard_count() const noexcept { return static_cast<int>(_shards.size()); }

    // candidate matches between consecutive frames still resident on the device after find_primitives_batch with ONE
    // shard (the last chunk of <= 256 frames, set_chunk_frames); see cape_match_consecutive.  Enforced: returns false (and logs why) when the
    // last batch was cut over several shards, or when n_frames exceeds the frames of its last chunk.  Plane indices count the segments flagged
    // is_output, i.e. the planes BEFORE the polygon validity test drops any.
    bool match_consecutive(int n_frames,
                           std::vector<cape_frame_match>& matches,
                           bool useAdvancedSearch = false,
                           bool allowIndexZero = false) noexcept;

    // the same between the boundary polygons of the batch, with the reference's own measure (intersection area over the
    // detected polygon's area, map_primitive.cpp:137): see cape_match_polygons.  Same residency rules; needs the device
    // polygons of the batch (set_device_polygons(true), the default).  Plane indices are those of the plane_containers
    // find_primitives_batch returned.
    // prevToCur (optional): n_frames x 16 doubles, row-major [R t; 0 0 0 1], entry f taking camera f-1's frame into camera f's --
    // the worldToCamera the reference hands to find_matches when the map is the previous frame (map_primitive.cpp:100-104);
    // nullptr = a static camera (see cape_match_polygons_pose)
    bool match_consecutive_polygons(int n_frames,
                                    std::vector<cape_frame_match_exact>& matches,
                                    bool useAdvancedSearch = false,
                                    bool allowIndexZero = false,
                                    const double* prevToCur = nullptr) noexcept;

    // Matches between ALL consecutive frames of the following batches, computed while a batch runs: every chunk is matched on
    // the device right behind its polygons (cape_match_polygons_pose); the frame pairs the device cannot serve -- across a chunk
    // or shard boundary (the two frames went through different passes, maybe on different devices) and the frames it flags
    // CAPE_MATCH_EXACT_OVERFLOW -- are matched by the host class (find_plane_match on the containers, entries marked
    // CAPE_MATCH_EXACT_HOST).  batch_matches() then holds one complete entry per frame of the last batch, whatever its sharding.
    // prevToCur: as for match_consecutive_polygons, n_frames x 16 doubles that must stay valid during find_primitives_batch.
    void set_batch_matching(bool on, bool useAdvancedSearch = false, bool allowIndexZero = false, const double* prevToCur = nullptr) noexcept
    {
        _matchInBatch = on;
        _matchAdvanced = useAdvancedSearch;
        _matchIndexZero = allowIndexZero;
        _matchPoses = prevToCur;
    }
    // An entry flagged CAPE_MATCH_EXACT_HOST | CAPE_MATCH_EXACT_OVERFLOW was filled by the host class from a table of 16 previous
    // planes: match[] is then PARTIAL (previous planes beyond the table keep -1).
    [[nodiscard]] const std::vector<cape_frame_match_exact>& batch_matches() const noexcept { return _batchMatches; }

    [[nodiscard]] bool is_ready() const noexcept { return _single.handle != nullptr; }

  private:
    struct Shard
    {
        cape_handle handle = nullptr;
        void* stream = nullptr; // batch shards: a non-blocking stream of the handle's device (cape_stream_create), so that the copies
                                // and kernels of two shards on one device overlap; the one-frame handle stays on the null stream
        int device = 0;
        int maxBatch = 0;
        // where the last chunk's records / boundary points are: the shard's own copies (batch handles, results in HBM) or
        // the library's pinned host memory (the one-frame handle: nothing is copied, see cape_host_results)
        const cape_frame_record* records = nullptr;
        const double* boundary = nullptr;
        std::vector<cape_frame_record> recordCopy;
        std::vector<double> boundaryCopy;
        // boundary polygons built on the device (batch shards: cape_build_polygons; empty for the one-frame handle, whose few
        // polygons are cheaper on the host than one more kernel on the latency path)
        std::vector<cape_polygon> polygonCopy;
        std::vector<double> vertexCopy;
        bool devicePolygons = false;
        std::string error;
    };
    bool make_shard(Shard& s, int device, int maxBatch) noexcept;
    bool ensure_shards(int wanted) noexcept;
    bool extract_chunk(Shard& shard, const float* depth, const uint16_t* raw, float scale, int m) const;
    void batch_impl(const float* depth, const uint16_t* raw, float scale, int n_frames, std::vector<plane_container>& planes,
                    std::vector<cylinder_container>& cylinders) noexcept;
    void run_shard(Shard& shard,
                   const float* depth,
                   const uint16_t* raw,
                   float scale,
                   int firstFrame,
                   int n,
                   std::vector<plane_container>& planes,
                   std::vector<cylinder_container>& cylinders,
                   bool& ok) const;
    void collect(const Shard& shard, int f, plane_container& planes, cylinder_container& cylinders) const;
    void host_match(int f, const std::vector<plane_container>& planes) const;

    uint _width, _height;
    int _cells = 0, _boundaryCapacity = 0;
    int _maxBatch = 256;
    int _requestedShards = 0;
    bool _devicePolygons = true;
    bool _matchInBatch = false, _matchAdvanced = false, _matchIndexZero = false;
    const double* _matchPoses = nullptr;
    bool _batchMatchAdvanced = false, _batchMatchIndexZero = false; // the settings _batchMatches was computed with
    const double* _batchMatchPoses = nullptr;
    mutable std::vector<cape_frame_match_exact> _batchMatches; // set_batch_matching: one entry per frame of the last batch
    mutable std::vector<char> _matchOnHost;                    // frames whose entry the host class owes (chunk / shard starts, flagged frames)
    int _lastBatchShards = 0;              // how the last find_primitives_batch was cut: match_consecutive needs 1 shard,
    int _lastBatchResident = 0;            // and the frames of its LAST chunk are the ones still on the device
    mutable Shard _single;                 // max_batch = 1: the reference's call pattern, results read in place
    mutable std::unique_ptr<PolygonPool> _polygonPool; // created at the first frame that shows three or more planes
    mutable bool _expectHostPolygons = false;          // the last one-frame call had three or more polygons for the host class
    mutable std::vector<Shard> _shards;    // batch shards (max_batch = set_chunk_frames each), created at the first find_primitives_batch
    mutable double _meanPrimitiveTreatmentDuration = 0.0; // seconds, accumulated like primitive_detection.cpp:164
    mutable double _hostRefineTime = 0.0;  // seconds the host spent in collect(): containers + the polygons it builds itself (part of _refineTime)
    mutable double _hostResetTime = 0.0;   // seconds spent clearing the output containers (the host's share of reset_data)
    bool _detailedStatistics = false;
    uint32_t _randomSeed = 0;

    // remove copy functions, like the reference (primitive_detection.hpp:228-230)
    Primitive_Detection(const Primitive_Detection&) = delete;
    Primitive_Detection& operator=(const Primitive_Detection&) = delete;
};

// ---- "next" row N2, host part ----------------------------------------------------------------------------------
// The selection loop of MapPlane::find_matches (src/map_management/map_features/map_primitive.cpp:91-161) for one map
// plane already projected into camera space: a detected plane is a candidate if is_distance_similar and
// is_normal_similar (shape_primitives.cpp:66-86); the candidate with the greatest polygon intersection area wins if
// inter / area(detected) >= 0.4 (0.2 with useAdvancedSearch).  Returns the selected index or -1; like the reference it
// can never return index 0 (`if (selectedIndex <= 0)`, map_primitive.cpp:146).
int find_plane_match(const plane_container& detectedPlanes,
                     const std::vector<bool>& isDetectedFeatureMatched,
                     const PlaneCameraCoordinates& projectedPlane,
                     const CameraPolygon& projectedPolygon,
                     bool useAdvancedSearch = false,
                     bool allowIndexZero = false) noexcept; // allowIndexZero: drop the quirk (what CAPE_MATCH_ALLOW_INDEX0 does on the device)

} // namespace rgbd_slam::features::primitives
#endif
