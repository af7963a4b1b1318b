#include "primitive_detection.hpp"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <exception>
#include <functional>
#include <mutex>
#include <string>
#include <thread>

#include "outputs/logger.hpp"
#include "parameters.hpp"

namespace rgbd_slam::features::primitives {

namespace {
// contiguous block of `total` frames owned by shard `k` of `shards` (the first total % shards blocks get one more)
void block_of(int total, int k, int shards, int& first, int& count)
{
    const int q = total / shards, r = total % shards;
    first = k * q + (k < r ? k : r);
    count = q + (k < r ? 1 : 0);
}
} // namespace

// The reference's own find_primitives is multi-threaded (OpenCV's forEach over the cells, parameters.hpp: coreNumber = 8); what is
// left on the host here is the boundary polygon of every plane of a ONE-frame call -- ~7 us each, ~10 per frame, 70 of the call's
// 207 us on one core.  A few sleeping workers take them side by side with the calling thread, which works through the same list and
// waits only for workers that actually JOINED the job: a worker that wakes up after the list is drained finds the epoch closed and
// goes back to sleep (through round 4 the caller waited for every worker to wake, take the mutex and count itself out: ~15 us on a
// call whose polygons it had already built itself).  Batches build their polygons on the device and never come here.
struct PolygonPool
{
    explicit PolygonPool(unsigned workers)
    {
        for (unsigned w = 0; w < workers; ++w)
            threads.emplace_back([this]() { work(); });
    }
    ~PolygonPool()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        wake.notify_all();
        for (std::thread& t : threads)
            t.join();
    }
    // The caller knows a job is coming (it has just sent the frame to the device): the workers wake up NOW and watch for it
    // for at most half a millisecond, so that the ~15 us a sleeping thread needs to get going pass under the kernels.
    void prepare()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            expectUntil = std::chrono::steady_clock::now() + std::chrono::microseconds(500);
            ++hint;
        }
        wake.notify_all();
    }
    // f(i) for every i in [0, n), on the workers and on the caller
    void run(int n, const std::function<void(int)>& f)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            job = &f;
            count = n;
            next.store(0, std::memory_order_relaxed);
            busy = 0; // workers count themselves IN when they pick the job up
            ++epoch;
            published.store(epoch, std::memory_order_release);
        }
        wake.notify_all();
        for (int i; (i = next.fetch_add(1, std::memory_order_relaxed)) < n;)
            f(i);
        std::unique_lock<std::mutex> lk(m);
        closed = epoch; // the list is drained: whoever wakes up from now on has nothing to join
        idle.wait(lk, [this]() { return busy == 0; });
        job = nullptr;
    }

  private:
    void work()
    {
        unsigned long seen = 0, seenHint = 0;
        for (;;)
        {
            const std::function<void(int)>* f;
            int n;
            {
                std::unique_lock<std::mutex> lk(m);
                for (;;)
                {
                    wake.wait(lk, [&]() { return stop || epoch != seen || hint != seenHint; });
                    if (stop)
                        return;
                    if (epoch != seen)
                        break;
                    // a job was announced: watch for it without sleeping, until it comes or the announcement expires
                    seenHint = hint;
                    const auto until = expectUntil;
                    lk.unlock();
                    while (published.load(std::memory_order_acquire) == seen && std::chrono::steady_clock::now() < until)
                        ;
                    lk.lock();
                    if (stop)
                        return;
                    if (epoch != seen)
                        break;
                }
                seen = epoch;
                if (closed == epoch)
                    continue; // too late: the caller has drained the list (and may be gone with `job`)
                ++busy;
                f = job;
                n = count;
            }
            for (int i; (i = next.fetch_add(1, std::memory_order_relaxed)) < n;)
                (*f)(i);
            std::lock_guard<std::mutex> lk(m);
            if (--busy == 0)
                idle.notify_one();
        }
    }
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable wake, idle;
    const std::function<void(int)>* job = nullptr;
    int count = 0, busy = 0;
    unsigned long epoch = 0, hint = 0, closed = 0;
    std::atomic<unsigned long> published {0};
    std::chrono::steady_clock::time_point expectUntil {};
    std::atomic<int> next {0};
    bool stop = false;
};

namespace {
// cape_set_log_callback: the lines the reference's find_primitives logs on the hot path (decided on the device, carried by the
// frame record) and the library's capacity warnings, through the reference's logger
void forward_log(int32_t level, const char* message, int32_t /*frame*/, void* /*user*/)
{
    if (level >= 2)
        outputs::log_error(message);
    else if (level == 1)
        outputs::log_warning(message);
    else
        outputs::log(message);
}
} // namespace

void Primitive_Detection::set_detailed_statistics(bool on) noexcept
{
    _detailedStatistics = on;
    if (_single.handle)
        cape_enable_timing(_single.handle, on ? 1 : 0);
    for (Shard& s : _shards)
        cape_enable_timing(s.handle, on ? 1 : 0);
}

void Primitive_Detection::set_random_seed(uint32_t seed) noexcept
{
    _randomSeed = seed;
    if (_single.handle)
        cape_set_rng_seed(_single.handle, seed);
    for (Shard& s : _shards)
        cape_set_rng_seed(s.handle, seed);
}

Primitive_Detection::Primitive_Detection(const uint width, const uint height) : _width(width), _height(height)
{
    if (const char* env = std::getenv("CAPE_DETAILED_STATISTICS"))
        _detailedStatistics = env[0] == '1';
    Plane_Segment::set_static_members(parameters::detection::depthMapPatchSize_px,
                                      parameters::detection::depthMapPatchSize_px * parameters::detection::depthMapPatchSize_px);
    int devices = 0;
    if (cape_device_count(&devices) != CAPE_OK || devices <= 0 || !make_shard(_single, 0, 1))
        outputs::log_error(std::string("Primitive_Detection: ") + cape_last_error());
}

Primitive_Detection::~Primitive_Detection()
{
    _polygonPool.reset();
    cape_destroy(_single.handle);
    for (Shard& s : _shards)
    {
        cape_stream_destroy(s.handle, s.stream);
        cape_destroy(s.handle);
    }
}

bool Primitive_Detection::make_shard(Shard& s, int device, int maxBatch) noexcept
{
    try
    {
        if (!Parameters::is_valid())
            Parameters::load_defaut();
        cape_config cfg {};
        cfg.width = static_cast<int32_t>(_width);
        cfg.height = static_cast<int32_t>(_height);
        // camera intrinsics are read once, here (the reference caches them in function-local statics,
        // point_coordinates.cpp:81)
        cfg.fx = Parameters::get_camera_1_focal().x();
        cfg.fy = Parameters::get_camera_1_focal().y();
        cfg.cx = Parameters::get_camera_1_center().x();
        cfg.cy = Parameters::get_camera_1_center().y();
        cfg.flags = CAPE_FLAG_CYLINDERS; // the reference always runs the cylinder branch (primitive_detection.cpp:385-388)
        cfg.device = device;
        cfg.max_batch = maxBatch;
        s.device = device;
        s.maxBatch = maxBatch;
        if (cape_create(&cfg, &s.handle) != CAPE_OK)
        {
            s.handle = nullptr;
            return false;
        }
        cape_set_log_callback(s.handle, &forward_log, nullptr);
        if (_randomSeed != 0)
            cape_set_rng_seed(s.handle, _randomSeed);
        if (_detailedStatistics)
            cape_enable_timing(s.handle, 1);
        cape_layout lay {};
        cape_get_layout(s.handle, &lay);
        _cells = lay.cells;
        _boundaryCapacity = lay.boundary_capacity;
        if (maxBatch > 8) // results in HBM: the shard keeps host copies
        {
            if (cape_stream_create(s.handle, &s.stream) != CAPE_OK)
                s.stream = nullptr; // (the null stream works as well, without the overlap)
            s.recordCopy.resize(maxBatch);
            s.boundaryCopy.resize(static_cast<size_t>(maxBatch) * _boundaryCapacity * 3);
        }
        return true;
    }
    catch (const std::exception&)
    {
        return false;
    }
}

// batch shard i lives on device i % device_count; handles are created on demand and kept
bool Primitive_Detection::ensure_shards(int wanted) noexcept
{
    try
    {
        int devices = 0;
        if (cape_device_count(&devices) != CAPE_OK || devices <= 0)
            return false; // no GPU: this library has no CPU path
        while (static_cast<int>(_shards.size()) < wanted)
        {
            Shard s;
            if (!make_shard(s, static_cast<int>(_shards.size()) % devices, _maxBatch))
                return false;
            _shards.push_back(std::move(s));
        }
        return true;
    }
    catch (const std::exception&)
    {
        return false;
    }
}

// add_planes_to_primitives / add_cylinders_to_primitives (primitive_detection.cpp:562-648, 705-734), from the record
void Primitive_Detection::collect(const Shard& shard, int f, plane_container& planes, cylinder_container& cylinders) const
{
    planes.clear();
    cylinders.clear();
    const cape_frame_record& r = shard.records[f];
    // (the frame's log lines -- invalid seed, not planar after merge, rejected boundary, capacity -- came through forward_log when
    //  the record reached the host)
    planes.reserve(r.header.n_planes);
    // A frame is a CHAIN of records: its own, then -- beyond 64 plane segments or cylinder labels, the reference's vectors are
    // unbounded (primitive_detection.hpp:206) -- spill records of the handle's pool (cape_frame_header::next_record), each with a
    // boundary slab, a polygon row and a vertex slab of its own.  The spill records of a frame are fetched here, one by one: rare.
    struct Part
    {
        const cape_frame_record* rec;
        const double* bnd;           // boundary slab (null: not read back)
        const cape_polygon* pol;     // polygon row (null: no device polygons)
        const double* ver;           // vertex slab
    };
    struct SpillCopy
    {
        cape_frame_record rec;
        std::vector<double> bnd, ver;
        std::vector<cape_polygon> pol;
    };
    std::vector<Part> parts;
    std::vector<std::unique_ptr<SpillCopy>> spill;
    parts.push_back({&r, shard.boundary ? shard.boundary + static_cast<size_t>(f) * _boundaryCapacity * 3 : nullptr,
                     shard.devicePolygons ? &shard.polygonCopy[static_cast<size_t>(f) * CAPE_MAX_PLANES] : nullptr,
                     shard.devicePolygons ? shard.vertexCopy.data() + static_cast<size_t>(f) * _boundaryCapacity * 2 : nullptr});
    for (int next = r.header.next_record; next >= shard.maxBatch;)
    {
        auto sc = std::make_unique<SpillCopy>();
        sc->bnd.resize(static_cast<size_t>(_boundaryCapacity) * 3);
        bool ok = cape_copy_spill(shard.handle, next - shard.maxBatch, 1, &sc->rec, sc->bnd.data()) == CAPE_OK;
        if (ok && shard.devicePolygons)
        {
            sc->pol.resize(CAPE_MAX_PLANES);
            sc->ver.resize(static_cast<size_t>(_boundaryCapacity) * 2);
            ok = cape_copy_spill_polygons(shard.handle, next - shard.maxBatch, 1, sc->pol.data(), sc->ver.data()) == CAPE_OK;
        }
        if (!ok)
        {
            outputs::log_error(std::string("find_primitives: a frame's spill record could not be read: ") + cape_last_error());
            break;
        }
        parts.push_back({&sc->rec, sc->bnd.data(), shard.devicePolygons ? sc->pol.data() : nullptr, shard.devicePolygons ? sc->ver.data() : nullptr});
        next = sc->rec.header.next_record > next ? sc->rec.header.next_record : -1; // (a chain only points forward)
        spill.push_back(std::move(sc));
    }
    // the frame's plane segments in order, each with the slabs of the record that holds it
    struct SegRef
    {
        const cape_plane_segment* s;
        const cape_polygon* dp;
        const double* bnd;
        const double* ver;
    };
    std::vector<SegRef> segs;
    for (const Part& part : parts)
    {
        const int n = part.rec->header.n_plane_segments < CAPE_MAX_PLANES ? part.rec->header.n_plane_segments : CAPE_MAX_PLANES;
        for (int i = 0; i < n; ++i)
            segs.push_back({&part.rec->segments[i], part.pol ? part.pol + i : nullptr, part.bnd, part.ver});
    }
    // the output planes whose polygon the host class builds (one-frame calls, CAPE_POLY_OVERFLOW planes of a batch): built side
    // by side when there are several, then emplaced in segment order like the reference's loop (:577-631)
    struct HostPolygon
    {
        int segment;
        std::unique_ptr<CameraPolygon> polygon; // null: rejected (with `error`)
        std::string error;
    };
    std::vector<HostPolygon> hostPolygons;
    for (size_t i = 0; i < segs.size(); ++i)
        if (segs[i].s->is_output && !(segs[i].dp && !(segs[i].dp->flags & CAPE_POLY_OVERFLOW)))
            hostPolygons.push_back({static_cast<int>(i), nullptr, std::string()});
    auto build_host_polygon = [&](int k) {
        HostPolygon& hp = hostPolygons[static_cast<size_t>(k)];
        const SegRef& ref = segs[static_cast<size_t>(hp.segment)];
        const cape_plane_segment& s = *ref.s;
        try
        {
            if (!ref.bnd)
            {
                hp.error = "boundary points of the plane were not read back";
                return;
            }
            const Plane_Segment planeSegment(s);
            std::vector<vector3> orderedBoundary;
            orderedBoundary.reserve(s.boundary_count);
            const double* p = ref.bnd + static_cast<size_t>(s.boundary_offset) * 3;
            for (uint32_t q = 0; q < s.boundary_count; ++q)
                orderedBoundary.emplace_back(p[3 * q], p[3 * q + 1], p[3 * q + 2]);
            // :622 -- the SEGMENT's normal and centre, not the plane's re-normalised ones
            auto polygon = std::make_unique<CameraPolygon>(orderedBoundary, planeSegment.get_normal(), planeSegment.get_center());
            std::string debug;
            if (polygon->is_valid(debug) and polygon->boundary_length() >= 3)
                hp.polygon = std::move(polygon);
            else
                hp.error = debug;
        }
        catch (const std::exception& e)
        {
            hp.error = e.what();
        }
    };
    // (only the one-frame call: the shards of a batch run collect() on threads of their own, and the pool serves one caller)
    if (&shard == &_single)
        _expectHostPolygons = hostPolygons.size() >= 3;
    if (&shard == &_single && hostPolygons.size() >= 3 && std::thread::hardware_concurrency() > 1)
    {
        if (!_polygonPool)
        {
            const unsigned hw = std::thread::hardware_concurrency();
            _polygonPool = std::make_unique<PolygonPool>(hw > 4 ? 3u : hw - 1u);
        }
        _polygonPool->run(static_cast<int>(hostPolygons.size()), build_host_polygon);
    }
    else
        for (size_t k = 0; k < hostPolygons.size(); ++k)
            build_host_polygon(static_cast<int>(k));
    size_t nextHost = 0;
    for (const SegRef& ref : segs)
    {
        const cape_plane_segment& s = *ref.s;
        if (!s.is_output) // merged away, not planar, or fewer than 3 boundary points (:577-612)
            continue;
        const Plane_Segment planeSegment(s);
        try
        {
            // device polygon of this segment, if the batch built them: the ring goes through the reference's
            // Polygon(ring, xAxis, yAxis, center) constructor (polygon.cpp:236-266) -- no hull on the host.  A plane with more
            // boundary points than the device kernel takes (CAPE_POLY_OVERFLOW) was built by the host class above.
            const cape_polygon* dp = ref.dp;
            if (dp && !(dp->flags & CAPE_POLY_OVERFLOW))
            {
                if (!(dp->flags & CAPE_POLY_VALID) || dp->vertex_count < 3)
                {
                    outputs::log_error("Polyfit error: Geometry has invalid self-intersections or too few points");
                    continue;
                }
                const double* v = ref.ver + static_cast<size_t>(dp->vertex_offset) * 2;
                std::vector<vector2> ring;
                ring.reserve(dp->vertex_count);
                for (uint32_t k = 0; k < dp->vertex_count; ++k)
                    ring.emplace_back(v[2 * k], v[2 * k + 1]);
                const CameraPolygon polygon(ring, vector3(dp->x_axis[0], dp->x_axis[1], dp->x_axis[2]),
                                            vector3(dp->y_axis[0], dp->y_axis[1], dp->y_axis[2]),
                                            vector3(dp->center[0], dp->center[1], dp->center[2]));
                planes.emplace_back(planeSegment, polygon);
                continue;
            }
            HostPolygon& hp = hostPolygons[nextHost++];
            if (hp.polygon)
                planes.emplace_back(planeSegment, *hp.polygon);
            else
                outputs::log_error("Polyfit error: " + hp.error);
        }
        catch (const std::exception& e)
        {
            outputs::log_error(std::string("Polyfit error: ") + e.what());
        }
    }
    cylinders.reserve(r.header.n_cylinders);
    for (const Part& part : parts)
    {
        const int n = part.rec->header.n_cylinder_labels < CAPE_MAX_CYLINDERS ? part.rec->header.n_cylinder_labels : CAPE_MAX_CYLINDERS;
        for (int i = 0; i < n; ++i)
            if (part.rec->cylinders[i].kept)
                cylinders.emplace_back(Cylinder_Segment(part.rec->cylinders[i]));
    }
}

// one chunk (<= _maxBatch frames) through the C ABI: H2D copy + kernels + D2H of records and boundary points
bool Primitive_Detection::extract_chunk(Shard& shard, const float* depth, const uint16_t* raw, float scale, int m) const
{
    // raw sensor images cross PCIe at half the bytes and are converted on the device
    bool ok = (raw ? cape_extract_u16_host(shard.handle, raw, scale, m, shard.stream) : cape_extract_host(shard.handle, depth, m, shard.stream)) == CAPE_OK;
    shard.devicePolygons = false;
    if (ok && _devicePolygons && shard.maxBatch > 8)
    {
        // one wavefront per output plane builds the boundary polygons behind the extraction: the host class
        // costs ~12 us per plane, one core keeps up with ~80 k planes/s while a GPU emits millions
        shard.polygonCopy.resize(static_cast<size_t>(shard.maxBatch) * CAPE_MAX_PLANES);
        shard.vertexCopy.resize(static_cast<size_t>(shard.maxBatch) * _boundaryCapacity * 2);
        ok = cape_build_polygons(shard.handle, m, shard.stream) == CAPE_OK &&
             cape_copy_polygons(shard.handle, m, shard.polygonCopy.data(), shard.vertexCopy.data()) == CAPE_OK;
        shard.devicePolygons = ok;
    }
    if (ok && shard.maxBatch <= 8)
        ok = cape_host_results(shard.handle, &shard.records, nullptr, nullptr, &shard.boundary) == CAPE_OK; // in place
    else if (ok)
    {
        // with the polygons built on the device the boundary points (37 KB per frame) stay there -- unless a plane had more of
        // them than the device hull takes and falls back to the host class
        ok = cape_copy_results(shard.handle, m, shard.recordCopy.data(), nullptr, nullptr, nullptr) == CAPE_OK;
        shard.records = shard.recordCopy.data();
        bool needPoints = !shard.devicePolygons;
        for (int f = 0; ok && !needPoints && f < m; ++f)
        {
            const cape_frame_record& r = shard.recordCopy[static_cast<size_t>(f)];
            const int nSeg = r.header.n_plane_segments < CAPE_MAX_PLANES ? r.header.n_plane_segments : CAPE_MAX_PLANES; // (this record's)
            for (int i = 0; i < nSeg && !needPoints; ++i)
                needPoints = r.segments[i].is_output && (shard.polygonCopy[static_cast<size_t>(f) * CAPE_MAX_PLANES + i].flags & CAPE_POLY_OVERFLOW);
        }
        shard.boundary = nullptr;
        if (ok && needPoints)
        {
            ok = cape_copy_results(shard.handle, m, nullptr, nullptr, nullptr, shard.boundaryCopy.data()) == CAPE_OK;
            shard.boundary = shard.boundaryCopy.data();
        }
    }
    if (!ok)
        shard.error = cape_last_error(); // thread-local in the library: keep it for the caller's thread
    return ok;
}

void Primitive_Detection::run_shard(Shard& shard, const float* depth, const uint16_t* raw, float scale, int firstFrame, int n,
                                    std::vector<plane_container>& planes, std::vector<cylinder_container>& cylinders, bool& ok) const
{
    const size_t frameElems = static_cast<size_t>(_width) * _height;
    ok = true;
    for (int base = 0; base < n; base += shard.maxBatch)
    {
        const int m = (n - base < shard.maxBatch) ? n - base : shard.maxBatch;
        const size_t offset = static_cast<size_t>(firstFrame + base) * frameElems;
        if (!extract_chunk(shard, depth ? depth + offset : nullptr, raw ? raw + offset : nullptr, scale, m))
        {
            ok = false;
            return;
        }
        if (_matchInBatch)
        {
            // the chunk's frames matched on the device right behind their polygons; its first frame (its predecessor went through
            // another pass) and whatever the device flags are left to the host class once every shard is done (batch_impl)
            const int g0 = firstFrame + base;
            const uint32_t flags = (_matchAdvanced ? static_cast<uint32_t>(CAPE_MATCH_ADVANCED) : 0u) |
                                   (_matchIndexZero ? static_cast<uint32_t>(CAPE_MATCH_ALLOW_INDEX0) : 0u);
            const bool onDevice = shard.devicePolygons &&
                                  cape_match_polygons_pose(shard.handle, m, _matchPoses ? _matchPoses + static_cast<size_t>(16) * g0 : nullptr, flags,
                                                           shard.stream) == CAPE_OK &&
                                  cape_copy_polygon_matches(shard.handle, m, _batchMatches.data() + g0) == CAPE_OK;
            for (int f = 0; f < m; ++f)
                _matchOnHost[g0 + f] = !onDevice || f == 0 || (_batchMatches[g0 + f].flags & CAPE_MATCH_EXACT_OVERFLOW) != 0;
        }
        // containers are filled in place: a Plane copy would re-normalise its parametrisation once more
        for (int f = 0; f < m; ++f)
            collect(shard, f, planes[firstFrame + base + f], cylinders[firstFrame + base + f]);
    }
}

void Primitive_Detection::find_primitives_batch(const float* depth, int n_frames, std::vector<plane_container>& planes,
                                                std::vector<cylinder_container>& cylinders) noexcept
{
    batch_impl(depth, nullptr, 1.0f, n_frames, planes, cylinders);
}

void Primitive_Detection::find_primitives_batch(const uint16_t* raw, float scale, int n_frames, std::vector<plane_container>& planes,
                                                std::vector<cylinder_container>& cylinders) noexcept
{
    batch_impl(nullptr, raw, scale, n_frames, planes, cylinders);
}

void Primitive_Detection::batch_impl(const float* depth, const uint16_t* raw, float scale, int n_frames, std::vector<plane_container>& planes,
                                     std::vector<cylinder_container>& cylinders) noexcept
{
    try
    {
        // (Plane / Cylinder are copy-constructible but not assignable, like the reference's: no vector::assign here)
        planes.clear();
        cylinders.clear();
        planes.resize(n_frames > 0 ? n_frames : 0);
        cylinders.resize(n_frames > 0 ? n_frames : 0);
        if (n_frames <= 0 || (!depth && !raw))
            return;
        int wanted = _requestedShards;
        if (wanted <= 0)
        {
            // default: FOUR shards per visible device as soon as the batch holds two chunks per device, else one -- the shards' host
            // threads overlap one shard's PCIe copy with another's kernels, read-back and container building (each shard has its own
            // handle and stream, and a handle's read-back waits for that handle's work only).  Measured on one MI355X, round 6
            // (profiles/r06_overlay_batch_rate.txt): 512 raw frames 61 k frames/s with one shard, 64 k with two, 70 k with four;
            // 1 024: 61 k / 68 k / 78 k; below 384 frames the shards' fixed costs eat the overlap (256: 61 k / 52 k / 61 k).
            int devices = 0;
            if (cape_device_count(&devices) != CAPE_OK || devices <= 0)
                devices = 1;
            const int perDevice = n_frames / devices >= 2 * _maxBatch ? 4 : 1;
            wanted = devices * perDevice;
        }
        if (wanted > n_frames)
            wanted = n_frames;
        if (!ensure_shards(wanted) && _shards.empty())
        {
            outputs::log_error("find_primitives: no device extractor (cape_create failed); returning no primitives");
            return;
        }
        const int shards = static_cast<int>(_shards.size()) < wanted ? static_cast<int>(_shards.size()) : wanted;
        _lastBatchShards = shards;
        {
            // frames of shard 0's last chunk: what its handle still holds
            int first = 0, count = n_frames;
            if (shards > 1)
                block_of(n_frames, 0, shards, first, count);
            const int mb = _shards[0].maxBatch;
            _lastBatchResident = count <= 0 ? 0 : (count % mb == 0 ? mb : count % mb);
        }
        if (_matchInBatch)
        {
            _batchMatches.assign(static_cast<size_t>(n_frames), cape_frame_match_exact {});
            _matchOnHost.assign(static_cast<size_t>(n_frames), 1);
            // what these entries are computed with: match_consecutive_polygons hands them out only for the same question
            _batchMatchAdvanced = _matchAdvanced;
            _batchMatchIndexZero = _matchIndexZero;
            _batchMatchPoses = _matchPoses;
        }
        else
            _batchMatches.clear();
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<char> ok(shards, 1);
        if (shards == 1)
        {
            bool good = true;
            run_shard(_shards[0], depth, raw, scale, 0, n_frames, planes, cylinders, good);
            ok[0] = good;
        }
        else
        {
            // one host thread per shard: the copies, kernels and polygon fits of the devices overlap; every thread writes
            // its own block of the output vectors
            std::vector<std::thread> workers;
            workers.reserve(shards);
            for (int k = 0; k < shards; ++k)
                workers.emplace_back([&, k]() {
                    int first = 0, count = 0;
                    block_of(n_frames, k, shards, first, count);
                    bool good = true;
                    try
                    {
                        run_shard(_shards[k], depth, raw, scale, first, count, planes, cylinders, good);
                    }
                    catch (const std::exception&)
                    {
                        good = false;
                    }
                    ok[k] = good;
                });
            for (std::thread& w : workers)
                w.join();
        }
        for (int k = 0; k < shards; ++k)
            if (!ok[k])
                outputs::log_error("find_primitives: shard " + std::to_string(k) + " failed (" + _shards[k].error +
                                   "); its frames yield no primitives");
        if (_matchInBatch)
            for (int f = 0; f < n_frames; ++f)
                if (_matchOnHost[f])
                    host_match(f, planes); // stitches the chunk and shard boundaries, serves the frames beyond the device's capacities
        _meanPrimitiveTreatmentDuration += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    catch (const std::exception& e)
    {
        outputs::log_error(std::string("find_primitives: ") + e.what());
    }
}

void Primitive_Detection::find_primitives(const matrixf&, const depth_image& depthImage, plane_container& planeContainer,
                                          cylinder_container& primitiveContainer) noexcept
{
    try
    {
        const auto tr = std::chrono::steady_clock::now();
        planeContainer.clear();
        primitiveContainer.clear();
        _hostResetTime += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr).count(); // all reset_data() leaves to do
        if (depthImage.rows != static_cast<int>(_height) || depthImage.cols != static_cast<int>(_width))
        {
            outputs::log_error("find_primitives: depth image size differs from the configured size");
            return;
        }
        if (!_single.handle)
        {
            outputs::log_error("find_primitives: no device extractor (cape_create failed); returning no primitives");
            return;
        }
        const depth_image d = depthImage.isContinuous() ? depthImage : depthImage.clone();
        const auto t0 = std::chrono::steady_clock::now();
        if (_polygonPool && _expectHostPolygons)
            _polygonPool->prepare(); // the polygon workers wake up while the device works on the frame (only when the previous
                                     // frame gave them something to do: three cores spinning for a frame with two planes is waste)
        if (!extract_chunk(_single, d.ptr<float>(0), nullptr, 1.0f, 1))
        {
            outputs::log_error("find_primitives: " + _single.error);
            return;
        }
        const auto t1 = std::chrono::steady_clock::now();
        collect(_single, 0, planeContainer, primitiveContainer); // in place, like emplace_back at primitive_detection.cpp:627
        const auto t2 = std::chrono::steady_clock::now();
        _hostRefineTime += std::chrono::duration<double>(t2 - t1).count();
        _meanPrimitiveTreatmentDuration += std::chrono::duration<double>(t2 - t0).count();
    }
    catch (const std::exception& e)
    {
        outputs::log_error(std::string("find_primitives: ") + e.what());
    }
}

bool Primitive_Detection::match_consecutive(int n_frames, std::vector<cape_frame_match>& matches, bool useAdvancedSearch,
                                            bool allowIndexZero) noexcept
{
    try
    {
        matches.clear();
        if (_shards.empty() || n_frames < 0 || _lastBatchShards == 0)
            return false; // the batch handle of shard 0 holds the frames: find_primitives_batch comes first
        if (_lastBatchShards != 1)
        {
            // frame f-1 and frame f of a shard boundary live on different devices, and a shard's local indices are not the
            // batch's: refuse instead of returning matches that do not line up
            outputs::log_error("match_consecutive: the last batch was cut over " + std::to_string(_lastBatchShards) +
                               " shards; call set_shard_count(1) before find_primitives_batch");
            return false;
        }
        if (n_frames > _lastBatchResident)
        {
            outputs::log_error("match_consecutive: only the last chunk of the batch (" + std::to_string(_lastBatchResident) +
                               " frames) is still resident on the device");
            return false;
        }
        const uint32_t flags = (useAdvancedSearch ? static_cast<uint32_t>(CAPE_MATCH_ADVANCED) : 0u) |
                               (allowIndexZero ? static_cast<uint32_t>(CAPE_MATCH_ALLOW_INDEX0) : 0u);
        matches.resize(n_frames);
        if (cape_match_consecutive(_shards[0].handle, n_frames, flags, nullptr) != CAPE_OK ||
            cape_copy_matches(_shards[0].handle, n_frames, matches.data()) != CAPE_OK)
        {
            outputs::log_error(std::string("match_consecutive: ") + cape_last_error());
            matches.clear();
            return false;
        }
        return true;
    }
    catch (const std::exception&)
    {
        return false;
    }
}

bool Primitive_Detection::match_consecutive_polygons(int n_frames, std::vector<cape_frame_match_exact>& matches, bool useAdvancedSearch,
                                                     bool allowIndexZero, const double* prevToCur) noexcept
{
    try
    {
        matches.clear();
        if (_matchInBatch && n_frames >= 0 && static_cast<size_t>(n_frames) <= _batchMatches.size())
        {
            // set_batch_matching: the last batch was matched while it ran, stitched over its chunks and shards -- with the flags
            // and poses set_batch_matching held at that time.  Another question is not answered from that table (the
            // arguments used to be ignored): it goes to the device below if the batch is still resident, else it is refused.
            if (useAdvancedSearch == _batchMatchAdvanced && allowIndexZero == _batchMatchIndexZero && prevToCur == _batchMatchPoses)
            {
                matches.assign(_batchMatches.begin(), _batchMatches.begin() + n_frames);
                return true;
            }
            if (_lastBatchShards != 1 || n_frames > _lastBatchResident)
            {
                outputs::log_error("match_consecutive_polygons: the batch was matched with other flags / poses (set_batch_matching) and "
                                   "is no longer resident on one device: run the batch again with the settings wanted");
                return false;
            }
        }
        if (_shards.empty() || n_frames < 0 || _lastBatchShards != 1 || n_frames > _lastBatchResident || !_devicePolygons)
        {
            outputs::log_error("match_consecutive_polygons: needs a find_primitives_batch on ONE shard with device polygons, and at "
                               "most the frames of its last chunk");
            return false;
        }
        const uint32_t flags = (useAdvancedSearch ? static_cast<uint32_t>(CAPE_MATCH_ADVANCED) : 0u) |
                               (allowIndexZero ? static_cast<uint32_t>(CAPE_MATCH_ALLOW_INDEX0) : 0u);
        matches.resize(n_frames);
        if (cape_match_polygons_pose(_shards[0].handle, n_frames, prevToCur, flags, nullptr) != CAPE_OK ||
            cape_copy_polygon_matches(_shards[0].handle, n_frames, matches.data()) != CAPE_OK)
        {
            outputs::log_error(std::string("match_consecutive_polygons: ") + cape_last_error());
            matches.clear();
            return false;
        }
        return true;
    }
    catch (const std::exception&)
    {
        return false;
    }
}

// One frame's entry by the host class: MapPlane::find_matches of every plane of frame f-1 (through the pose, if any) against the
// planes of frame f, like the device does it (cape_match_polygons_pose) and with the same statements.
void Primitive_Detection::host_match(int f, const std::vector<plane_container>& planes) const
{
    cape_frame_match_exact& out = _batchMatches[static_cast<size_t>(f)];
    out = cape_frame_match_exact {};
    out.flags = CAPE_MATCH_EXACT_HOST;
    for (int k = 0; k < CAPE_MATCH_MAX_PLANES; ++k)
    {
        out.match[k] = out.seg_prev[k] = out.seg_cur[k] = -1;
        for (int i = 0; i < CAPE_MATCH_MAX_PLANES; ++i)
            out.inter_area[k][i] = -1.0;
    }
    if (f == 0)
        return;
    const plane_container &prev = planes[static_cast<size_t>(f) - 1], &cur = planes[static_cast<size_t>(f)];
    out.n_prev = static_cast<int32_t>(prev.size());
    out.n_cur = static_cast<int32_t>(cur.size());
    if (prev.size() > CAPE_MATCH_MAX_PLANES)
        out.flags |= CAPE_MATCH_EXACT_OVERFLOW; // the table holds 16 previous planes
    const double* T = _matchPoses ? _matchPoses + static_cast<size_t>(16) * f : nullptr;
    std::vector<bool> matched(cur.size(), false);
    for (size_t j = 0; j < prev.size() && j < CAPE_MATCH_MAX_PLANES; ++j)
    {
        int m = -1;
        if (T)
        {
            // to_camera_coordinates (plane_coordinates.cpp:20-24): the rotated normal goes into the constructor, which normalises it once
            const vector3 n = prev[j].get_normal();
            const double d = prev[j].get_d();
            const vector3 rn((T[0] * n[0] + T[1] * n[1]) + T[2] * n[2], (T[4] * n[0] + T[5] * n[1]) + T[6] * n[2], (T[8] * n[0] + T[9] * n[1]) + T[10] * n[2]);
            const double m0 = -((T[3] * T[0] + T[7] * T[4]) + T[11] * T[8]), m1 = -((T[3] * T[1] + T[7] * T[5]) + T[11] * T[9]),
                         m2 = -((T[3] * T[2] + T[7] * T[6]) + T[11] * T[10]);
            const PlaneCameraCoordinates projected(rn, ((m0 * n[0] + m1 * n[1]) + m2 * n[2]) + d);
            m = find_plane_match(cur, matched, projected, CameraPolygon(prev[j].get_boundary_polygon().to_camera_space(T)), _matchAdvanced,
                                 _matchIndexZero);
        }
        else
            m = find_plane_match(cur, matched, prev[j].get_parametrization(), prev[j].get_boundary_polygon(), _matchAdvanced, _matchIndexZero);
        if (m >= 0)
            matched[static_cast<size_t>(m)] = true;
        out.match[j] = m;
    }
}

void Primitive_Detection::show_statistics(const double meanFrameTreatmentDuration, const uint frameCount,
                                          const bool shouldDisplayDetails) const noexcept
{
    // primitive_detection.cpp:69-117
    auto percent = [](double t, double total) { return total <= 0 ? 0.0 : (t / total) * 100.0; };
    if (frameCount == 0)
        return;
    const double mean = _meanPrimitiveTreatmentDuration / static_cast<double>(frameCount);
    char buf[256];
    std::snprintf(buf, sizeof buf, "\tMean primitive extraction time is %.4f seconds (%.2f%%)", mean,
                  percent(mean, meanFrameTreatmentDuration));
    outputs::log(buf);
    if (!shouldDisplayDetails)
        return;
    // the reference's five buckets (:87-115), per frame: device seconds of every handle (HIP events; stage B's kernels split into
    // grow / merge / refine by the ticks their waves booked, cape_timings) + the host's own share of reset and refine
    double reset = _hostResetTime, init = 0.0, grow = 0.0, merge = 0.0, refine = _hostRefineTime;
    uint64_t timedCalls = 0;
    auto add = [&](cape_handle handle) {
        cape_timings t {};
        if (handle && cape_get_timings(handle, &t) == CAPE_OK)
        {
            reset += t.reset_s, init += t.init_s, grow += t.grow_phase_s, merge += t.merge_s, refine += t.refine_s;
            timedCalls += t.calls;
        }
    };
    add(_single.handle);
    for (const Shard& s : _shards)
        add(s.handle);
    if (timedCalls == 0)
    {
        outputs::log("\t\t(stage details need set_detailed_statistics(true) or CAPE_DETAILED_STATISTICS=1 before the frames are treated)");
        return;
    }
    const double n = static_cast<double>(frameCount);
    const std::pair<const char*, double> buckets[5] = {{"reset", reset / n}, {"init", init / n}, {"grow", grow / n}, {"merge", merge / n},
                                                       {"refine", refine / n}};
    for (const auto& b : buckets)
    {
        std::snprintf(buf, sizeof buf, "\t\tMean primitive %s time is %.4f seconds (%.2f%%)", b.first, b.second, percent(b.second, mean));
        outputs::log(buf);
    }
}

int find_plane_match(const plane_container& detectedPlanes, const std::vector<bool>& isDetectedFeatureMatched,
                     const PlaneCameraCoordinates& projectedPlane, const CameraPolygon& projectedPolygon, bool useAdvancedSearch,
                     bool allowIndexZero) noexcept
{
    const double projectedArea = projectedPolygon.get_area();
    const double planeMinimalOverlap = parameters::matching::minimumPlaneOverlapToConsiderMatch; // float widened, like the reference's static double
    const double areaSimilarityThreshold = useAdvancedSearch ? planeMinimalOverlap / 2 : planeMinimalOverlap;
    double greatestSimilarity = 0.0;
    if (projectedArea <= 0.0)
        return -1;
    int selectedIndex = -1;
    const int detectedPlaneSize = static_cast<int>(detectedPlanes.size());
    for (int planeIndex = 0; planeIndex < detectedPlaneSize; ++planeIndex)
    {
        if (planeIndex < static_cast<int>(isDetectedFeatureMatched.size()) && isDetectedFeatureMatched[planeIndex])
            continue;
        const Plane& shapePlane = detectedPlanes[planeIndex];
        if (not shapePlane.is_distance_similar(projectedPlane) or not shapePlane.is_normal_similar(projectedPlane))
            continue;
        const CameraPolygon& detectedPolygon = shapePlane.get_boundary_polygon();
        const double newPlaneArea = detectedPolygon.get_area();
        const double interArea = detectedPolygon.inter_area(projectedPolygon);
        if (interArea > greatestSimilarity and interArea / newPlaneArea >= areaSimilarityThreshold)
        {
            selectedIndex = planeIndex;
            greatestSimilarity = interArea;
        }
    }
    if (selectedIndex < 0 || (selectedIndex == 0 && !allowIndexZero)) // quirk of the reference: index 0 can never be returned
        return -1;
    return selectedIndex;
}

} // namespace rgbd_slam::features::primitives
