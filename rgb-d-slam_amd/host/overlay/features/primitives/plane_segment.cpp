#include "plane_segment.hpp"

namespace rgbd_slam::features::primitives {

Plane_Segment::Plane_Segment(const cape_plane_segment& r) :
    _pointCount(r.point_count),
    _score(r.score),
    _MSE(r.mse),
    _isPlanar(r.planar != 0),
    _centroid(r.centroid[0], r.centroid[1], r.centroid[2]),
    _Sx(r.sums[0]), _Sy(r.sums[1]), _Sz(r.sums[2]), _Sxs(r.sums[3]), _Sys(r.sums[4]), _Szs(r.sums[5]), _Sxy(r.sums[6]),
    _Syz(r.sums[7]), _Szx(r.sums[8])
{
    // direct member access instead of PlaneCoordinates(normal, d): the record already holds the normal as it stands in
    // the reference's _planeSegments vector (every normalisation the reference applied up to there was applied on the
    // device); one more here would change its last bits
    _parametrization.normal() = vector3(r.normal[0], r.normal[1], r.normal[2]);
    _parametrization.d() = r.d;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            _pointCloudCovariance(i, j) = r.cov[3 * i + j];
}

void Plane_Segment::expand_segment(const Plane_Segment& p) noexcept
{
    _Sx += p._Sx;
    _Sy += p._Sy;
    _Sz += p._Sz;
    _Sxs += p._Sxs;
    _Sys += p._Sys;
    _Szs += p._Szs;
    _Sxy += p._Sxy;
    _Syz += p._Syz;
    _Szx += p._Szx;
    _pointCount += p._pointCount;
}

bool Plane_Segment::can_be_merged(const Plane_Segment& p, const double maxMatchDistance) const noexcept
{
    static const double maximumMergeAngle = std::cos(static_cast<double>(parameters::detection::maximumPlaneAngleForMerge_d) * M_PI / 180.0);
    return get_cos_angle(p) > maximumMergeAngle and std::abs(get_point_distance(p.get_centroid())) < maxMatchDistance;
}

void Plane_Segment::clear_plane_parameters() noexcept
{
    *this = Plane_Segment();
}

} // namespace rgbd_slam::features::primitives
