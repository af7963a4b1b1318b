/*
 * cape_hip.h -- C ABI of the MI355X-native CAPE plane/cylinder extractor (libcape_hip.so).
 *
 * This is the drop-in boundary for ONE path of BaptisteHudyma/RGB-D-SLAM: the `primitives` library
 * (reference CMakeLists.txt:117-123), i.e. Depth_Map_Transformation::get_organized_cloud_array +
 * Primitive_Detection::find_primitives.  The reference has no FFI layer; the functions below are what a
 * binding for that library would call.  Each entry point cites the reference interface it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the reference-side shim.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returns CAPE_OK (0) or a negative
 * cape_status and never throws, exits or logs to stdout (the reference's exit(-1)/terminate paths,
 * histogram.hpp:105-109, become error codes).  A handle is not thread-safe; distinct handles are.
 * All device work of a call is enqueued on the caller's HIP stream (`stream` is a hipStream_t passed as
 * void*; NULL = the null stream).
 *
 * Streams: ONE stream is in flight per handle -- the per-handle scratch and result buffers carry no per-buffer events.
 * A call made with another stream than the previous call's is ordered behind the handle's earlier work ON THE DEVICE
 * (hipStreamWaitEvent on a handle-owned event recorded behind every enqueue): it neither blocks the host nor touches
 * the previous stream, which the caller may destroy as soon as it has no further use for it.  The stream of the LAST
 * call must stay alive until that work has been waited for (cape_copy_results / cape_host_results / a stream
 * synchronisation of the caller's) -- the usual HIP rule for any enqueued work.
 */
#ifndef CAPE_HIP_H
#define CAPE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI number of this header: bumped whenever a struct below changes size or meaning (cape_abi_version() returns the library's;
 * a binding built against another number must not call it).  2 = round 6: cape_frame_header grew by next_record / segment_base
 * (frames of more than 64 plane segments continue in spill records), cape_config by spill_records, cape_timings holds the
 * reference's five buckets (12 fields), the frame status carries bit 7 and a count in bits 8..15. */
#define CAPE_ABI_VERSION 2

#define CAPE_CELL_SIZE 20          /* parameters::detection::depthMapPatchSize_px, src/parameters.hpp:79-80 */
#define CAPE_MAX_PLANES 64         /* plane segments ONE RECORD holds.  _planeSegments is an unbounded std::vector in the reference
                                      (primitive_detection.hpp:206): a frame with more continues in spill records, see
                                      cape_frame_header.next_record.  Frames with up to 32 segments run entirely in the fast
                                      kernels, up to 64 in a 64-segment instance, the others in the general instance */
#define CAPE_MAX_CYLINDERS 64      /* cylinder labels one record holds (cylinder2regionMap, likewise unbounded: same chain) */

typedef enum cape_status
{
    CAPE_OK = 0,
    CAPE_ERR_INVALID_ARGUMENT = -1,
    CAPE_ERR_NO_DEVICE = -2,       /* no HIP device / runtime error at create: the product has NO CPU fallback */
    CAPE_ERR_HIP = -3,             /* a HIP runtime call failed; cape_last_error() has the text */
    CAPE_ERR_CAPACITY = -4,        /* n_frames > max_batch */
    CAPE_ERR_UNSUPPORTED = -5
} cape_status;

enum
{
    CAPE_FLAG_CYLINDERS = 1u << 0, /* run cylinder RANSAC on low-score regions (primitive_detection.cpp:385-388);
                                      cleared = "plane-only" mode of BASELINE.json configs[0..1] */
    CAPE_FLAG_ASYNC_SECOND_PASS = 1u << 1, /* batches with cylinders: the second pass of stage B (the frames that reach a cylinder
                                      candidate; it lasts as long as its slowest frame) runs on a stream of the handle's own and
                                      cape_extract's caller stream does NOT wait for it.  Every later entry point on the handle
                                      (the next cape_extract, cape_copy_results, cape_pack_primitives, ...) orders itself behind
                                      it; a consumer of the raw cape_device_results pointers calls cape_sync_results first.
                                      Meant for TWO handles fed alternately: one's streaming kernels run under the other's tail */
};

/* per-frame status bits (cape_frame_header.status) */
enum
{
    CAPE_FRAME_PLANE_OVERFLOW = 1u << 0,    /* the handle's pool of spill records (cape_config.spill_records) ran out while this frame
                                               needed one more: the frame is truncated at the records it got */
    CAPE_FRAME_BOUNDARY_OVERFLOW = 1u << 1, /* boundary point capacity exceeded */
    CAPE_FRAME_CYL_OVERFLOW = 1u << 2,
    CAPE_FRAME_BIN_NEAR_EDGE = 1u << 3,     /* a cell's histogram angle fell within 1e-9 of a bin edge (libm tie risk) */
    CAPE_FRAME_INORDER_CELLS = 1u << 4,     /* >=1 cell took the in-order accumulation path (exactness guard) */
    CAPE_FRAME_RNG_EXHAUSTED = 1u << 5,     /* RANSAC asked for more draws than the precomputed mt19937 table */
    CAPE_FRAME_SEED_LIMIT = 1u << 6,        /* the seed loop hit its iteration guard (4 * cells + 1024; provably unreachable) */
    CAPE_FRAME_INVALID_SEED = 1u << 7       /* the seed loop ended on "Could not find a single plane segment: invalid seed"
                                               (log_warning, primitive_detection.cpp:299-304) */
};
/* bits 8..15 of the status: how many times the frame logged "Plane segment is not planar after merge" (primitive_detection.cpp:374
 * for a grown region, :497 for a cylinder sub-segment), saturating at 255 */
#define CAPE_FRAME_NOT_PLANAR_SHIFT 8
#define CAPE_FRAME_NOT_PLANAR_COUNT(status) (((status) >> CAPE_FRAME_NOT_PLANAR_SHIFT) & 0xFFu)

/*
 * Replaces: Depth_Map_Transformation(width,height,cellSize) + Primitive_Detection(width,height) constructors
 * (src/rgbd_slam.cpp:48-57) and the process-global camera intrinsics they read lazily
 * (Parameters::get_camera_1_intrinsics, src/parameters.hpp:144-149).
 */
typedef struct cape_config
{
    int32_t width;      /* multiple of 20; at most 256 cells wide (5120 px) and 65 535 cells in all -- grids of up to 128 x 64 cells */
    int32_t height;     /* multiple of 20     (2560 x 1280 px) run in the fast kernels, larger ones in the general instance      */
    double fx, fy, cx, cy;
    uint32_t flags;     /* CAPE_FLAG_* */
    int32_t device;     /* HIP device ordinal */
    int32_t max_batch;  /* frames per cape_extract call (sizes the per-frame scratch and result buffers) */
    int32_t boundary_capacity; /* boundary points per frame; 0 = 2 * cells */
    int32_t sub_batches; /* 0/1: the whole batch runs kernel after kernel on the caller's stream.  k > 1: the batch is
                            cut in k sub-batches that alternate between two internal streams (forked from / joined
                            into the caller's stream), so the latency-bound grow kernel of one sub-batch overlaps
                            the streaming cell kernel of the next */
    int32_t spill_records; /* records (each with its own boundary slab) in the handle's spill pool, shared by the frames of a batch
                            that hold more than 64 plane segments or cylinder labels; 0 = max(8, max_batch / 8).  A frame needs
                            ceil(n / 64) - 1 of them; at most cells / max(1, min(6, uint(0.0065 cells))) segments can exist */
} cape_config;

typedef struct cape_handle_s* cape_handle;

/* One plane segment of a frame = one element of Primitive_Detection::_planeSegments after merge_planes()
 * (primitive_detection.cpp:503-560) plus what Plane(planeSeg, polygon) derives from it
 * (shape_primitives.cpp:48-56).  Field names follow Plane_Segment (plane_segment.hpp:122-139). */
typedef struct cape_plane_segment
{
    double normal[3];       /* PlaneCoordinates::_normal of the segment */
    double d;
    double centroid[3];
    double mse;
    double score;
    double sums[9];         /* Sx Sy Sz Sxs Sys Szs Sxy Syz Szx */
    double out_normal[3];   /* Plane::_parametrization normal (one more normalisation) ; valid if is_output */
    double cov[9];          /* Plane_Segment::get_point_cloud_covariance(), row-major ; valid if is_output */
    uint32_t point_count;
    uint32_t merge_label;   /* planeMergeLabels[i]: index in the FRAME's segment list (may point into an earlier record of the chain) */
    uint32_t planar;
    uint32_t is_output;     /* root of its merge group, planar and >= 3 boundary points: becomes a `Plane` */
    uint32_t boundary_offset; /* first point in the boundary slab of the record that holds this segment */
    uint32_t boundary_count;
} cape_plane_segment;

typedef struct cape_cylinder
{
    double axis[3];         /* Cylinder::_normal */
    double radius;          /* NaN, as in the reference (shape_primitives.cpp:17-24 over a copy with 0 segments) */
    uint32_t kept;          /* survived add_cylinders_to_primitives (primitive_detection.cpp:705-734) */
    uint32_t region;        /* cylinder2regionMap[i].first */
} cape_cylinder;

typedef struct cape_frame_header
{
    int32_t n_plane_segments; /* _planeSegments.size() -- counted FROM THIS RECORD ON: the frame's first record holds the whole
                                 frame's count, the record itself the first min(64, n) of them, the rest follow next_record */
    int32_t n_planes;         /* number of segments with is_output (= planeContainer.size() before polygon tests), from this record on */
    int32_t n_cylinder_labels;/* cylinder2regionMap.size(), from this record on (the record holds the first min(64, n)) */
    int32_t n_cylinders;      /* cylinderContainer.size(), from this record on */
    int32_t n_boundary_points;/* points in THIS record's boundary slab */
    int32_t n_seeds;          /* iterations of the seed loop (debug) */
    uint32_t status;          /* CAPE_FRAME_* */
    int32_t n_planar_cells;
    int32_t next_record;      /* -1, or the index in the handle's record array (>= max_batch: a spill record) of the record that
                                 continues this frame: segments / cylinders 64.. of it, with its own boundary slab, polygons, vertices */
    int32_t segment_base;     /* index of segments[0] (and cylinders[0]) of this record in the frame's lists: 0, 64, 128, ... */
} cape_frame_header;

/* Fixed-capacity record (stays on the producing GPU / goes to its host; the multi-GPU gather ships the packed lists below).
 * The handle's record array holds max_batch records -- record f is frame f of the batch -- followed by the spill pool
 * (cape_config.spill_records).  A frame with more than 64 plane segments (a checkerboard of small facets) or cylinder labels is a
 * CHAIN of records linked by header.next_record; every record of the chain owns a boundary slab, a polygon row and a vertex slab
 * at its own index, so whatever consumes "record i" (polygons, the host conversion) works on spill records unchanged. */
typedef struct cape_frame_record
{
    cape_frame_header header;
    cape_plane_segment segments[CAPE_MAX_PLANES];
    cape_cylinder cylinders[CAPE_MAX_CYLINDERS];
} cape_frame_record;

/*
 * Multi-GPU exchange (BASELINE.json configs[3], [4]).  Frames shard by contiguous blocks, one GPU per
 * block, no collective inside a frame; once per batch the ranks all-gather what plane_container / cylinder_container
 * hold after find_primitives (shape_primitives.hpp:129-130), optionally with the two label grids.  The lists are ragged,
 * so each rank PACKS its shard on the device into one buffer of a fixed byte count (what ncclAllGather needs):
 *
 *   cape_packed_header | frames_capacity x cape_packed_frame | planes_capacity x cape_packed_plane |
 *   cylinders_capacity x cape_packed_cylinder | [frames_capacity x cells u8 plane labels | same, cylinder labels]
 *
 * planes_capacity = frames_capacity x planes_per_frame is a budget for the whole shard, not per frame: nothing is
 * truncated unless the shard's TOTAL exceeds it, which the header reports (planes_per_frame = CAPE_MAX_PLANES can never
 * overflow).  Sections start on 16-byte boundaries; cape_gather_layout has the offsets.
 */
#define CAPE_PACKED_MAGIC 0x43415045u /* "CAPE" */
enum
{
    CAPE_GATHER_LABELS = 1u << 0 /* also ship _gridPlaneSegmentMap / _gridCylinderSegMap, one byte per cell each */
};
enum
{
    CAPE_PACKED_PLANES_DROPPED = 1u << 0,   /* cape_packed_header.overflow */
    CAPE_PACKED_CYLINDERS_DROPPED = 1u << 1,
    CAPE_PACKED_LABELS_CLIPPED = 1u << 2    /* a frame of the shard holds more than 255 plane segments / cylinder labels: its label
                                               grids (one byte per cell on the wire) read 255 where the label is larger */
};
typedef struct cape_packed_header
{
    uint32_t magic;             /* CAPE_PACKED_MAGIC */
    int32_t n_frames;           /* frames of this shard */
    int32_t first_frame;        /* index of the shard's first frame in the whole batch (caller supplied) */
    int32_t n_planes_total;     /* planes found in the shard; only min(total, capacity) are listed */
    int32_t n_cylinders_total;
    int32_t planes_capacity;
    int32_t cylinders_capacity;
    uint32_t overflow;          /* CAPE_PACKED_*_DROPPED */
    uint32_t status_or;         /* OR of the frames' CAPE_FRAME_* flag bits (bits 0..7; the count in bits 8..15 of a frame's status is not folded) */
    int32_t cells;
    int32_t frames_capacity;
    uint32_t flags;             /* CAPE_GATHER_* */
} cape_packed_header;
typedef struct cape_packed_frame
{
    int32_t plane_offset;       /* first plane of the frame in the planes section */
    int32_t n_planes;           /* planeContainer.size() before the polygon validity test */
    int32_t cylinder_offset;
    int32_t n_cylinders;        /* cylinderContainer.size() */
    uint32_t status;            /* CAPE_FRAME_* */
    int32_t n_plane_segments;
} cape_packed_frame;
typedef struct cape_packed_plane /* SURVEY.md 8e record: what Plane(planeSeg, polygon) is built from, minus the polygon */
{
    double normal[3];           /* Plane::get_normal() */
    double d;                   /* Plane::get_d() */
    double centroid[3];
    double mse;
    double score;
    double sums[9];             /* Sx Sy Sz Sxs Sys Szs Sxy Syz Szx: get_point_cloud_covariance() follows from them */
    uint32_t point_count;
    uint32_t segment;           /* index of the segment in the producing frame's record */
} cape_packed_plane;
typedef struct cape_packed_cylinder
{
    double axis[3];
    double radius;              /* NaN, as in the reference */
} cape_packed_cylinder;

typedef struct cape_gather_config
{
    int32_t frames_capacity;     /* largest shard (frames per rank) this handle will pack; <= max_batch */
    int32_t planes_per_frame;    /* budget: planes_capacity = frames_capacity x planes_per_frame; 0 = 16; <= 4096 */
    int32_t cylinders_per_frame; /* 0 = 8 */
    uint32_t flags;              /* CAPE_GATHER_LABELS */
} cape_gather_config;
typedef struct cape_gather_layout
{
    uint64_t bytes_per_rank;
    uint64_t frames_offset, planes_offset, cylinders_offset, plane_labels_offset, cyl_labels_offset; /* labels: 0 if absent */
    int32_t frames_capacity, planes_capacity, cylinders_capacity, cells;
} cape_gather_layout;

/* Per-cell statistics (debug / parity access to Primitive_Detection::_planeGrid, _cellDistanceTols,
 * Histogram::_bins; primitive_detection.hpp:205-218).  One struct per cell, cell-row-major. */
typedef struct cape_cell_stats
{
    double sums[9];
    double normal[3];
    double d;
    double centroid[3];
    double mse;
    double score;
    float tol;
    uint32_t point_count;
    int32_t bin;            /* histogram bin right after init_histogram, -1 if not planar */
    uint32_t planar;
    uint32_t inorder;       /* 1 if the exactness guard sent this cell through the in-order path */
    uint32_t pad;
} cape_cell_stats;

typedef struct cape_timings
{
    /* mirrors the reference's stage buckets (primitive_detection.hpp:233-239), from HIP events, seconds,
     * accumulated over calls made with CAPE timing enabled */
    double cell_fit_s;      /* _initTime : back-projection + per-cell PCA (stage A = the two kernels below) */
    double cell_moments_s;  /*   A1 cape_cell_moments_kernel (streaming pass over the depth image) */
    double cell_plane_s;    /*   A2 cape_cell_plane_kernel (per-cell plane fit, tolerance, histogram bin) */
    double grow_s;          /* _growTime + _mergeTime + _refineTime : stage B kernel */
    double total_s;
    uint64_t frames;
    uint64_t calls;         /* number of cape_extract calls folded into the sums (= launches of each kernel) */
    /* The reference's five buckets, one to one (find_primitives, primitive_detection.cpp:126-160; show_statistics :69-117):
     *   reset_s       _resetTime : reset_data().  Nothing persists between frames here (row A14) and the hand-over counters are
     *                 cleared by a thread of stage A2: there is no reset pass to time, the bucket is 0 by construction
     *   init_s        _initTime  : init_planar_cell_fitting + init_histogram = kernels A1 + A2 (= cell_fit_s; the histogram's
     *                 bins are A2's, its counting is the first microseconds of the grow kernel)
     *   grow_phase_s  _growTime  : grow_planes_and_cylinders (seed loop, region growing, cylinder_fitting)
     *   merge_s       _mergeTime : merge_planes
     *   refine_s      _refineTime: add_planes_to_primitives + add_cylinders_to_primitives WITHOUT the boundary polygon (that is
     *                 cape_build_polygons or the host class; the overlay adds its own clock for it)
     * grow / merge / refine share the stage-B kernels: every frame's wave books the shader-clock ticks it spends in each of the
     * three (three atomics per frame, only while timing is on) and grow_s -- the kernels' HIP-event time -- is split in those
     * proportions: grow_phase_s + merge_s + refine_s == grow_s.  The split is an ESTIMATE weighted by wave occupancy, not three wall
     * times: a frame that is redone books its grow ticks in both passes, the workgroup finisher books one of its four waves, and only
     * the SUM of the three is measured (the HIP events). */
    double reset_s, init_s, grow_phase_s, merge_s, refine_s;
} cape_timings;

typedef struct cape_layout
{
    int32_t h_cells, v_cells, cells;
    int32_t boundary_capacity;
    uint64_t frame_record_bytes;  /* sizeof(cape_frame_record) */
    int32_t compute_units;        /* CUs of the handle's device */
    int32_t grow_frames_per_cu;   /* frames (one wavefront each) the grow kernel keeps in flight per CU (occupancy API) */
    uint32_t effective_flags;     /* the CAPE_FLAG_* bits that are ACTIVE on this handle: CAPE_FLAG_ASYNC_SECOND_PASS is
                                     cleared here when the handle cannot overlap (cylinders off, max_batch <= 8 or
                                     sub_batches > 1), so a caller can tell whether the mode it asked for is in force */
    uint32_t reserved;
} cape_layout;

/* Number of HIP devices this process sees (0 and CAPE_ERR_NO_DEVICE without a GPU): what a host-side batch API shards
 * over (one handle per device). */
int cape_device_count(int32_t* count_out);

/* Depth_Map_Transformation / Primitive_Detection constructors (src/rgbd_slam.cpp:48-57). */
int cape_create(const cape_config* cfg, cape_handle* out);
void cape_destroy(cape_handle h);
int cape_get_layout(cape_handle h, cape_layout* out);

/*
 * Replaces, for a batch of frames: get_organized_cloud_array (depth_map_transformation.hpp:36-39) followed by
 * find_primitives (primitive_detection.hpp:41-44).  `depth_dev` is a DEVICE pointer to n_frames row-major
 * float32 images in millimetres (0 = invalid), already resident in HBM, 16-byte aligned.  Asynchronous on `stream`.
 * Results stay on the device until fetched (cape_copy_results) or gathered (cape_device_results).
 */
int cape_extract(cape_handle h, const float* depth_dev, int32_t n_frames, void* stream);

/* Same path fed with the raw 16-bit sensor image (what the depth PNGs of the TUM / CAPE datasets hold): the device
 * converts exactly like the reference's host-side cv::Mat::convertTo(CV_32F, scale) (examples/main_TUM.cpp:221,242
 * with scale = 1/5 ; examples/main_CAPE.cpp:58-59 with scale = 1): z = float(raw) * scale, 0 = invalid.  Halves the
 * bytes read per frame ("next" row N4 of SURVEY.md 8f). */
int cape_extract_u16(cape_handle h, const uint16_t* depth_dev, float scale, int32_t n_frames, void* stream);

/* "Next" row N3: Depth_Map_Transformation::rectify_depth (depth_map_transformation.hpp:29, .cpp:23-87) for a batch --
 * registers the depth camera's image to the colour camera.  `cam2_to_cam1` is the row-major 4x4 matrix
 * Parameters::get_camera_2_to_camera_1_transformation() (host pointer, 16 doubles; the last row is ignored).  Both
 * images are device pointers, n_frames x H x W float32, not in place.  Collisions keep the last source pixel in
 * row-major order, which is what the reference's MAKE_DETERMINISTIC loop does.  The intrinsics are the handle's
 * (the reference also uses camera 1's for both directions, depth_map_transformation.cpp:156 / point_coordinates.cpp:90). */
int cape_rectify_depth(cape_handle h, const float* depth_dev, float* rectified_dev, int32_t n_frames,
                       const double* cam2_to_cam1, void* stream);

/* Same with host images (H2D copy, rectify, D2H copy, synchronous): the host boundary of the reference signature. */
int cape_rectify_depth_host(cape_handle h, const float* depth_host, float* rectified_host, int32_t n_frames,
                            const double* cam2_to_cam1);

/* "Next" row N2, device part: a cell-mask pre-filter for MapPlane::find_matches
 * (src/map_management/map_features/map_primitive.cpp:91-161) between CONSECUTIVE frames of the last cape_extract batch.
 * For frame f >= 1 the planes of frame f-1 play the map planes ("projected" with the identity pose) and the planes of
 * frame f the detected ones; plane i = i-th segment with is_output, its mask = the cells whose label belongs to its
 * merge group (the mask add_planes_to_primitives builds, primitive_detection.cpp:586-594).  Areas are counted in
 * cells instead of polygon mm^2; everything else follows the reference: |d_i - d_j| < 100 mm and |n_i . n_j| >
 * |cos 20 deg| (shape_primitives.cpp:70-86), inter > best so far and inter / area(detected) >= 0.4f (0.2 with
 * CAPE_MATCH_ADVANCED), previous planes visited in order with the is-matched flags updated between them
 * (feature_map.hpp:651-669), and the `selectedIndex <= 0` quirk that never returns detected plane 0
 * (map_primitive.cpp:146) unless CAPE_MATCH_ALLOW_INDEX0 is set.  Frame 0 of a batch has no predecessor (n_prev = 0).
 * A frame that continues in spill records (more than 64 plane segments) takes part with the planes of its FIRST record only: this
 * pre-filter has no reference counterpart and its tables hold 64 planes; the polygon matcher below flags such a frame instead. */
enum
{
    CAPE_MATCH_ADVANCED = 1u << 0,
    CAPE_MATCH_ALLOW_INDEX0 = 1u << 1
};
typedef struct cape_frame_match
{
    int32_t n_prev;                      /* planes of frame f-1 */
    int32_t n_cur;                       /* planes of frame f */
    int32_t match[CAPE_MAX_PLANES];      /* per previous plane j: matched plane of this frame, or -1 */
    uint16_t area_prev[CAPE_MAX_PLANES]; /* cells */
    uint16_t area_cur[CAPE_MAX_PLANES];
    uint16_t inter[CAPE_MAX_PLANES][CAPE_MAX_PLANES]; /* inter[j][i]: cells shared by previous plane j and plane i */
} cape_frame_match;
/* Asynchronous on `stream`; reads the device results of the last cape_extract (n_frames <= that batch). */
int cape_match_consecutive(cape_handle h, int32_t n_frames, uint32_t flags, void* stream);
/* Device pointer to / synchronous copy of the n_frames x cape_frame_match written by cape_match_consecutive. */
int cape_device_matches(cape_handle h, void** matches);
int cape_copy_matches(cape_handle h, int32_t n_frames, cape_frame_match* out);

/* "Next" row N1 on the device: the boundary polygon of every output plane of the last cape_extract batch -- what the
 * reference builds on the host right after the boundary candidates, `utils::Polygon(points, normal, center)`
 * (primitive_detection.cpp:622 -> src/utils/polygon.cpp:168-229: plane frame :74-115, projection :125-144, concave hull
 * :283-318 with the convex hull :268-281 as fallback, area :453-461, simplify :578-601), one wavefront per plane.
 * A polygon is a ring of 2-D vertices in the plane frame (x_axis, y_axis, center), clockwise, first vertex not repeated:
 * exactly the arguments of the reference's Polygon(ring, xAxis, yAxis, center) constructor (polygon.cpp:236-266), which is
 * how the overlay turns a record into a CameraPolygon without running a hull on the host.  The vertices are bit-identical
 * to this repo's host class (rgb-d-slam_amd/host/boundary_polygon.cpp; the reference's own vertices are not reproducible:
 * FLANN's randomized kd-trees over a nondeterministically ordered point list). */
enum
{
    CAPE_POLY_VALID = 1u << 0,           /* simple ring of >= 3 vertices: Primitive_Detection keeps the plane (:623-631) */
    CAPE_POLY_CONVEX_FALLBACK = 1u << 1, /* no concave hull on the k ladder: compute_convex_hull was used */
    CAPE_POLY_SIMPLIFIED = 1u << 2,      /* simplify() replaced the ring (area stayed above 75 %) */
    CAPE_POLY_OVERFLOW = 1u << 3,        /* more than 1024 boundary points: not built, left to the host class */
    CAPE_POLY_REJECTED = 1u << 4,        /* the host constructor would throw (fewer than 3 points / normal not unit) */
    CAPE_POLY_DISSOLVED = 1u << 5        /* the walk's hull crossed itself (the reference's Intersects misses crossings with axis-parallel
                                            edges) and was cut apart at its crossings like correct_boost_polygon.hpp does; the ring may
                                            hold vertices that are no boundary candidates (the crossing points) */
};
typedef struct cape_polygon
{
    double x_axis[3], y_axis[3]; /* get_plane_coordinate_system(segment normal) */
    double center[3];            /* Plane_Segment::get_center() = normal * (-d) (plane_coordinates.hpp:52), like primitive_detection.cpp:622 */
    double area;                 /* Polygon::_area after simplify */
    uint32_t vertex_offset;      /* first vertex in the frame's vertex array (= the segment's boundary_offset) */
    uint32_t vertex_count;
    uint32_t flags;              /* CAPE_POLY_* ; 0 for a segment that is not an output plane */
    uint32_t segment;            /* index of the segment in the frame record */
} cape_polygon;
/* Builds the polygons of frames [0, n_frames) of the last cape_extract; asynchronous on `stream`. */
int cape_build_polygons(cape_handle h, int32_t n_frames, void* stream);
/* Device pointers: polygons = n_frames x CAPE_MAX_PLANES cape_polygon (indexed by segment), vertices = n_frames x
 * boundary_capacity x 2 doubles.  Valid until the next cape_build_polygons / destroy.  CAPE_ERR_CAPACITY when no
 * cape_build_polygons has run since the last cape_extract (the arrays would describe the PREVIOUS batch); the copy calls
 * below likewise refuse more frames than the last cape_build_polygons / cape_match_polygons of the current batch covered. */
int cape_device_polygons(cape_handle h, cape_polygon** polygons, double** vertices);
/* Synchronous D2H copy of both arrays (either pointer may be NULL). */
int cape_copy_polygons(cape_handle h, int32_t n_frames, cape_polygon* polygons, double* vertices);

/* Debug / parity: the device polygon of an arbitrary point set (host pointers; n <= boundary_capacity points x 3 doubles,
 * unit normal, centre) -- one launch of the same kernel over a one-plane record.  `vertices_out` takes up to n x 2 doubles.
 * Synchronous; does not disturb the results of the last cape_extract. */
int cape_debug_polygon(cape_handle h, const double* points3, int32_t n, const double* normal, const double* center,
                       cape_polygon* polygon_out, double* vertices_out);

/* Row N2 with the reference's own area measure: the selection of MapPlane::find_matches (map_primitive.cpp:91-161) between
 * consecutive frames on the boundary polygons of cape_build_polygons -- `detectedPolygon.inter_area(projectedPolygon)` in mm^2
 * (map_primitive.cpp:137; the previous frame's polygon projected into the detected plane's frame, Polygon::project,
 * polygon.cpp:338-382) divided by the detected polygon's area -- instead of the shared cells cape_match_consecutive counts.
 * Plane indices count the planes Primitive_Detection KEEPS (output plane with a valid polygon of >= 3 vertices,
 * primitive_detection.cpp:623-631), i.e. the indices of the reference's plane_container.  Needs cape_build_polygons of the
 * same batch first.  The areas are bit-identical to this repo's host class (Polygon::inter_area). */
#define CAPE_MATCH_MAX_PLANES 16
enum
{
    CAPE_MATCH_EXACT_OVERFLOW = 1u << 0 /* more than 16 kept planes in one of the two frames, a frame that continues in spill records
                                           (cape_frame_header.next_record), an output plane of either frame
                                           whose polygon was left to the host class (CAPE_POLY_OVERFLOW: the host may keep it,
                                           so the kept-plane indices are not known here), or a polygon pair beyond the
                                           kernel's capacities (512 vertices per ring, 2 048 slab boundaries, 32 edges of a
                                           ring over one slab): no match is reported for the frame -- use the host class */
    ,
    CAPE_MATCH_EXACT_HOST = 1u << 1     /* never set by the library: the C++ overlay marks the entries its host class computed (frame
                                           pairs across a chunk / shard boundary, frames the device flagged); match[] is filled,
                                           seg_prev / seg_cur / inter_area are not */
};
typedef struct cape_frame_match_exact
{
    int32_t n_prev, n_cur;                    /* kept planes of frame f-1 / f */
    int32_t match[CAPE_MATCH_MAX_PLANES];     /* per previous plane j: matched plane of this frame, or -1 */
    int32_t seg_prev[CAPE_MATCH_MAX_PLANES];  /* segment index of previous plane j (-1 beyond n_prev) */
    int32_t seg_cur[CAPE_MATCH_MAX_PLANES];
    uint32_t flags;                           /* CAPE_MATCH_EXACT_* */
    uint32_t pad;
    double inter_area[CAPE_MATCH_MAX_PLANES][CAPE_MATCH_MAX_PLANES]; /* [j][i] mm^2 ; -1 where the distance / normal gates
                                                 failed (the reference does not intersect those), NaN: capacity exceeded */
} cape_frame_match_exact;
/* flags: CAPE_MATCH_ADVANCED, CAPE_MATCH_ALLOW_INDEX0 as for cape_match_consecutive.  Asynchronous on `stream`. */
int cape_match_polygons(cape_handle h, int32_t n_frames, uint32_t flags, void* stream);
/* The same with the camera motion between consecutive frames -- what the reference does before its gates: the map plane goes
 * through PlaneWorldCoordinates::to_camera_coordinates (map_primitive.cpp:100-101, plane_coordinates.cpp:20-24 with the plane
 * matrix of camera_transformation.cpp:53-71) and its polygon through WorldPolygon::to_camera_space (map_primitive.cpp:103,
 * polygon_coordinates.cpp:135-165) with `worldToCamera`.  Here the map is frame f-1: prev_to_cur = n_frames x 16 doubles in
 * HOST memory (read before the call returns), row-major 4x4 [R t; 0 0 0 1], entry f taking a point of camera f-1's frame into
 * camera f's; entry 0 is not read.  NULL = the identity = cape_match_polygons (a static camera). */
int cape_match_polygons_pose(cape_handle h, int32_t n_frames, const double* prev_to_cur, uint32_t flags, void* stream);
int cape_copy_polygon_matches(cape_handle h, int32_t n_frames, cape_frame_match_exact* out);

/* A stream of the handle's device for callers that do not link the HIP runtime themselves (the overlay): non-blocking, so the
 * work of several handles driven from several host threads overlaps instead of meeting on the legacy null stream.  Pass it as
 * the `stream` argument of the calls below; destroy it before the handle. */
int cape_stream_create(cape_handle h, void** stream_out);
int cape_stream_destroy(cape_handle h, void* stream);

/* Same, from host memory: H2D copy on `stream`, then cape_extract (host boundary of the reference's
 * cv::Mat_<float> argument).  The copy is part of the call; throughput numbers never use this entry. */
int cape_extract_host(cape_handle h, const float* depth_host, int32_t n_frames, void* stream);
/* The raw 16-bit sensor images from host memory (what a depth PNG decodes to): half the bytes over PCIe, the conversion of
 * cape_extract_u16 on the device.  88 k frames/s against 44 k for float32 input on a PCIe 5 x16 link. */
int cape_extract_u16_host(cape_handle h, const uint16_t* depth_host, float scale, int32_t n_frames, void* stream);

/* Device pointers to the results of the last cape_extract (valid until the next call / destroy):
 * records: n_frames x cape_frame_record ; plane_labels / cyl_labels: n_frames x cells int32
 * (_gridPlaneSegmentMap / _gridCylinderSegMap, primitive_detection.hpp:212-214) ; boundary: n_frames x
 * boundary_capacity x 3 doubles (compute_plane_segment_boundary, primitive_detection.cpp:650-703). */
int cape_device_results(cape_handle h, void** records, int32_t** plane_labels, int32_t** cyl_labels, double** boundary);
/* Sizes the packed buffer (two staging slots of bytes_per_rank on the device) and reports its layout.  May be called
 * again to change the capacities (synchronises). */
int cape_gather_configure(cape_handle h, const cape_gather_config* cfg, cape_gather_layout* layout_out);
/* Packs the results of the last cape_extract (frames [0, n_frames) of it) into the next staging slot, asynchronously on
 * `stream`; first_frame goes into the header.  *packed_dev (optional) receives the slot's device address. */
int cape_pack_primitives(cape_handle h, int32_t n_frames, int32_t first_frame, void** packed_dev, void* stream);
/* Synchronous copy of the slot filled by the last cape_pack_primitives / cape_gather_primitives (bytes_per_rank bytes). */
int cape_copy_packed(cape_handle h, void* packed_host);

/* RCCL communicator of the handle (librccl is dlopen'ed on first use).  Rank 0 makes the 128-byte id with
 * cape_comm_unique_id and hands it to the other ranks by any means (file, socket, MPI, torch.distributed store);
 * every rank then calls cape_comm_init, which is collective (ncclCommInitRank on the handle's device). */
#define CAPE_COMM_ID_BYTES 128
int cape_comm_unique_id(void* id_out);
int cape_comm_init(cape_handle h, const void* id, int32_t rank, int32_t world);
int cape_comm_destroy(cape_handle h);
/* What RCCL itself reports for the handle's communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), next to what
 * cape_comm_init was called with: a launcher prints it per rank so that a run on N GPUs can be seen to have had N ranks on N
 * devices.  has_comm = 0 (and -1 in the RCCL fields) when the handle has no communicator -- e.g. when the caller moves the packed
 * bytes with another transport (torch.distributed). */
typedef struct cape_comm_info_t
{
    int32_t has_comm;       /* 1: cape_comm_init succeeded on this handle */
    int32_t has_gather;     /* 1: the loaded librccl offers ncclGather (cape_gather_primitives_root) */
    int32_t nranks, rank;   /* ncclCommCount, ncclCommUserRank (-1: not available) */
    int32_t device;         /* ncclCommCuDevice: the HIP device RCCL bound the communicator to */
    int32_t init_nranks, init_rank; /* world / rank passed to cape_comm_init */
    int32_t handle_device;  /* cape_config.device */
} cape_comm_info_t;
int cape_comm_info(cape_handle h, cape_comm_info_t* out);
/* One batch's exchange: cape_pack_primitives on `stream`, then ONE ncclAllGather of bytes_per_rank per rank on the
 * handle's own communication stream (behind an event, so it runs under whatever the caller enqueues next on `stream`).
 * recv_dev: world x bytes_per_rank device bytes, rank r's shard at r x bytes_per_rank; it must stay untouched until the
 * gather has completed.  A staging slot is reused only after the gather that read it is done (two slots). */
int cape_gather_primitives(cape_handle h, int32_t n_frames, int32_t first_frame, void* recv_dev, void* stream);
/* "gather" in the narrow sense (BASELINE.json north_star): the same exchange with ONE receiver -- ncclGather (an RCCL
 * extension) of bytes_per_rank per rank to rank `root`; recv_dev is read on the root only (may be NULL elsewhere).  At
 * world 8 every other GPU receives nothing instead of 8 x bytes_per_rank.  CAPE_ERR_UNSUPPORTED if the loaded librccl has
 * no ncclGather. */
int cape_gather_primitives_root(cape_handle h, int32_t n_frames, int32_t first_frame, int32_t root, void* recv_dev, void* stream);
/* Totals of the last cape_extract batch (frames [0, n_frames)): what a caller sizes cape_gather_config.planes_per_frame
 * with -- e.g. ceil(1.25 x n_planes / n_frames) + 1 from the previous batch of the same stream -- instead of the default
 * budget.  Synchronous (waits for the batch).  Any output pointer may be NULL. */
int cape_count_primitives(cape_handle h, int32_t n_frames, int32_t* n_planes, int32_t* n_cylinders, int32_t* max_planes_per_frame);
/* Orders after the last cape_gather_primitives: with host_sync != 0 the call returns when the gather has landed,
 * otherwise `stream` is made to wait for it (hipStreamWaitEvent). */
int cape_gather_wait(cape_handle h, void* stream, int32_t host_sync);

/* Makes `stream` wait for whatever the handle still has in flight for the last cape_extract (the asynchronous second pass of
 * CAPE_FLAG_ASYNC_SECOND_PASS; a no-op otherwise): after it, work enqueued on `stream` may read cape_device_results. */
int cape_sync_results(cape_handle h, void* stream);

/* Synchronous D2H of the results of the last cape_extract.  Any pointer may be NULL. */
int cape_copy_results(cape_handle h, int32_t n_frames, cape_frame_record* records, int32_t* plane_labels,
                      int32_t* cyl_labels, double* boundary);

/* The spill pool of the last cape_extract (synchronous): *used = spill records handed out (a frame's chain is followed through
 * header.next_record, spill record k has index max_batch + k), *capacity = cape_config.spill_records as resolved at create,
 * *frames = frames of the batch that went through the general instance.  Any pointer may be NULL. */
int cape_spill_info(cape_handle h, int32_t* used, int32_t* capacity, int32_t* frames);
/* Synchronous D2H of spill records [first, first + count) -- record indices max_batch + first ... -- and their boundary slabs
 * (count x boundary_capacity x 3 doubles).  Either pointer may be NULL. */
int cape_copy_spill(cape_handle h, int32_t first, int32_t count, cape_frame_record* records, double* boundary);
/* The polygon rows (count x CAPE_MAX_PLANES) and vertex slabs (count x boundary_capacity x 2 doubles) of the same spill records,
 * after cape_build_polygons (which builds the polygons of every spill record in use together with the batch's). */
int cape_copy_spill_polygons(cape_handle h, int32_t first, int32_t count, cape_polygon* polygons, double* vertices);

/* Handles created with max_batch <= 8 (the reference's call pattern: one frame per call) keep records, label grids and
 * boundary points in pinned, device-mapped HOST memory: the kernels write them over PCIe directly and no device-to-host
 * copy exists on the latency path.  cape_host_results waits for the handle's stream and returns pointers to that memory
 * (valid until the next call / destroy); CAPE_ERR_UNSUPPORTED for larger handles, whose results live in HBM.
 * Such a handle also runs the ONE-FRAME CHAIN: stage A as one launch of strip workgroups whenever the frame is in
 * device memory (a device pointer, or the staged copy of a pageable host frame; a frame read in place from pinned memory keeps the
 * two streaming kernels), then ONE grow kernel whose last wave stores the completion number cape_host_results spins on -- two or
 * three launches per call instead of five.  Same results bit for bit; the debug knob CAPE_STAGE_A=bands (read and validated at
 * cape_create, like CAPE_RESUME / CAPE_SCHEDULE) keeps the batch kernels on such a handle, CAPE_STAGE_A=strips forces the strip
 * kernel for pinned input too. */
int cape_host_results(cape_handle h, const cape_frame_record** records, const int32_t** plane_labels,
                      const int32_t** cyl_labels, const double** boundary);

/* Pinned, device-mapped host memory for depth frames (hipHostMalloc / hipHostRegister on the handle's device).  When
 * cape_extract_host is given such a buffer, batches of up to 8 frames are read by the streaming kernel straight from host
 * memory (the image is read exactly once: the PCIe transfer is the kernel's input stream, no staging copy); larger
 * batches take one DMA.  A registered range must stay allocated until it is unregistered. */
int cape_host_alloc(cape_handle h, uint64_t bytes, void** out);
int cape_host_free(cape_handle h, void* p);
int cape_host_register(cape_handle h, void* p, uint64_t bytes);
int cape_host_unregister(cape_handle h, void* p);

/* Debug / parity: per-cell stats of one frame of the last batch (synchronous). */
int cape_copy_cell_stats(cape_handle h, int32_t frame, cape_cell_stats* cells_out);

/* Debug / parity: the seed cells of one frame of the last batch in the order grow_planes_and_cylinders tried them
 * (primitive_detection.cpp:277-307; header.n_seeds of them, at most `capacity` are copied, *n_out = header.n_seeds).
 * Synchronous. */
int cape_copy_seed_sequence(cape_handle h, int32_t frame, int32_t* seeds_out, int32_t capacity, int32_t* n_out);

/* show_statistics (primitive_detection.hpp:46-49): stage timings from HIP events recorded on the caller's stream
 * around each kernel of every cape_extract made while timing is enabled.  cape_get_timings synchronises the
 * pending events, folds them into the running sums and returns the sums; cape_reset_timings zeroes them. */
int cape_enable_timing(cape_handle h, int32_t enable);
int cape_get_timings(cape_handle h, cape_timings* out);
/* the same for a caller compiled against another revision of cape_timings: writes min(out_bytes, sizeof(cape_timings)) bytes */
int cape_get_timings_sized(cape_handle h, void* out, uint64_t out_bytes);
int cape_reset_timings(cape_handle h);

/* Debug / parity: evaluate device scalar math (f64 sqrt / div, ocml acos / atan2, the eigen-solver and plane fit)
 * on host operands so tests can compare gfx950 results with the CPU oracle bit for bit.  `a`,`b`,`out` are HOST
 * pointers; EIGEN3: a = n x 6 (m00 m10 m11 m20 m21 m22), out = n x 12 ; FIT_PLANE: a = n x 10 (9 sums, count),
 * out = n x 10 (normal[3], d, centroid[3], mse, score, planar). */
enum
{
    CAPE_DEBUG_SQRT = 0, CAPE_DEBUG_DIV = 1, CAPE_DEBUG_ACOS = 2, CAPE_DEBUG_ATAN2 = 3, CAPE_DEBUG_QUANT = 4,
    CAPE_DEBUG_SQRTF = 5, CAPE_DEBUG_EIGEN3 = 6, CAPE_DEBUG_FIT_PLANE = 7
};
int cape_debug_eval(int op, const double* a, const double* b, double* out, int n);
/* Debug: shader-clock ticks spent per phase of the grow kernel, n_frames x 32 (all zero unless the library was built
 * with -DCAPE_B_PROFILE).  Synchronises. */
int cape_debug_cycles(cape_handle h, int32_t n_frames, unsigned long long* out);
/* frames of the last cape_rectify_depth that its band kernel handed to the general kernels (tests; synchronises) */
int cape_debug_rectify_flagged(cape_handle h, int32_t* count);
/* the task queue of the last cape_build_polygons (tests; synchronises): slots reserved by spawned tasks and quit marks, tickets
 * taken, and the slots the call could use.  reserved > slots means the queue overflowed and waves walked rungs they could not
 * enqueue -- impossible in the shipped library (the queue holds every task a batch can spawn), forced by a test build. */
int cape_debug_polygon_queue(cape_handle h, uint32_t* reserved, uint32_t* tickets, uint32_t* slots);
/* the work lists of the last cape_match_polygons (profiling; synchronises): 32 words -- [0..3] pairs each capacity tier of the
 * intersection kernel was handed, [8 + 4 * tier + reason] pairs that left tier `tier` for a larger one because of reason 1 = ring
 * vertices, 2 = slab boundaries, 3 = edges over one slab */
int cape_debug_match_lists(cape_handle h, uint32_t* words32);

/* The seed of the reference's random engine (src/utils/random.hpp:59-64): 0 under MAKE_DETERMINISTIC -- the default here, and the
 * mode BASELINE.json's bit-exactness is stated for --, `std::time(0)` taken once at process start otherwise.  The engine is
 * thread_local and find_primitives runs on a fresh thread per frame (rgbd_slam.cpp:291), so EVERY frame restarts the sequence at
 * the seed: the handle keeps the first 40 000 doubles of mt19937(seed) + uniform_real_distribution on the device and this call
 * regenerates them (it waits for the handle's work in flight).  A caller that wants the reference's non-deterministic build passes
 * its own time(0). */
int cape_set_rng_seed(cape_handle h, uint32_t seed);

/* outputs::log / log_warning / log_error of the path (the reference's src/outputs/logger.hpp), for a caller that wants the
 * reference's own lines: level 0 = log, 1 = log_warning, 2 = log_error.  The messages find_primitives prints on the hot path,
 *   "Could not find a single plane segment: invalid seed"                      (warning, primitive_detection.cpp:302)
 *   "Plane segment is not planar after merge"                                  (log, :374 and :497; once per occurrence)
 *   "Could not find a correct boundary polygon, rejecting plane segment"       (warning, :618: a planar merge root with fewer than
 *                                                                               three boundary points)
 * are decided on the device and travel in the frame record (CAPE_FRAME_INVALID_SEED, CAPE_FRAME_NOT_PLANAR_COUNT, the segments'
 * boundary counts); the callback gets them when a batch's records first reach the host -- cape_copy_results with a records
 * pointer, or cape_host_results -- once per extracted batch, frame by frame, on the calling thread.  The library's own
 * capacity warnings (CAPE_FRAME_*_OVERFLOW) come the same way.  (:631 "Polyfit error" belongs to the polygon constructor:
 * a consumer of cape_copy_polygons sees CAPE_POLY_REJECTED.)  fn == NULL removes the callback; nothing is ever printed. */
typedef void (*cape_log_fn)(int32_t level, const char* message, int32_t frame, void* user);
int cape_set_log_callback(cape_handle h, cape_log_fn fn, void* user);
/* The same lines for records the caller holds (HOST memory, e.g. out of cape_copy_results): no handle, no device.  Returns the
 * number of lines (>= 0) or a negative cape_status. */
int cape_log_records(const cape_frame_record* records, int32_t n_frames, cape_log_fn fn, void* user);

const char* cape_last_error(void);
const char* cape_version(void);
/* CAPE_ABI_VERSION of the library that was loaded: a binding compares it with the header it was built against before anything else */
int32_t cape_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CAPE_HIP_H */
