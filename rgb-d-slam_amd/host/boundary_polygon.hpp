// Host-side boundary polygon of a plane ("next" row N1 of SURVEY.md 8f): the step that follows the boundary candidate
// points of compute_plane_segment_boundary in the reference (primitive_detection.cpp:622):
//   utils::Polygon(points, normal, center)   reference src/utils/polygon.cpp:168-229
//     get_plane_coordinate_system            :74-115   (select_correct_transform :50-68)
//     get_projected_plan_coordinates         :125-144
//     compute_concave_hull                   :283-318  -> third_party/concave_fitting.cpp (Moreira-Santos k-nearest
//                                             neighbours hull, k = 3,3,5,7,11,13,17,21)
//     compute_convex_hull fallback           :268-281
//     area / contains / simplify             :453-461, :320-323, :578-601
// Dependency-free (the reference uses Boost.Geometry + FLANN).  The polygon's VERTICES are not a parity target -- the
// reference feeds the hull a nondeterministically ordered point list through randomized kd-trees -- its validity,
// area and containment are (reference tests/test_polygons.cpp:6-89), and tests/host/test_polygon.cpp checks those.
#pragma once
#include <array>
#include <cstddef>
#include <string>
#include <utility>
#include <vector>

#include "compat/types.hpp" // rgbd_slam::vector2 / vector3 (the reference's src/types.hpp names)

namespace rgbd_slam::utils {

using rgbd_slam::vector2;
using rgbd_slam::vector3;

// get_plane_coordinate_system (polygon.cpp:74-115): two unit vectors spanning the plane of `normal`
std::pair<vector3, vector3> get_plane_coordinate_system(const vector3& normal);
vector2 get_projected_plan_coordinates(const vector3& point, const vector3& center, const vector3& xAxis, const vector3& yAxis);
vector3 get_point_from_plane_coordinates(const vector2& point, const vector3& center, const vector3& xAxis, const vector3& yAxis);

// PlaneWorldCoordinates::to_camera_coordinates (plane_coordinates.cpp:20-24) through compute_plane_world_to_camera_matrix
// (camera_transformation.cpp:53-71): for worldToCamera = [R t] the plane matrix is [R 0; -t^T R 1], and the
// PlaneCameraCoordinates constructor normalises the rotated normal once (plane_coordinates.hpp:19-23).
void plane_to_camera(const double* normal, double d, const double* worldToCamera, double* normalOut, double* dOut);

class Polygon
{
  public:
    Polygon() = default;
    // throws std::invalid_argument like the reference (normal not unit / fewer than 3 points)
    Polygon(const std::vector<vector3>& points, const vector3& normal, const vector3& center);

    [[nodiscard]] bool is_valid() const noexcept;               // simple (non self-intersecting) rings, outer of >= 3 vertices
    [[nodiscard]] bool is_valid(std::string& reason) const noexcept; // polygon.hpp:84-87: same, with the failure reason
    [[nodiscard]] size_t boundary_length() const noexcept { return _ring.size(); }
    [[nodiscard]] double area() const noexcept;                 // shoelace of the outer ring minus the interior rings, >= 0
    [[nodiscard]] double get_area() const noexcept { return _area; }
    [[nodiscard]] bool contains(const vector2& point) const noexcept; // strictly inside (boost::geometry::within): not in a hole
    [[nodiscard]] vector3 get_center() const noexcept { return _center; }
    [[nodiscard]] vector3 get_x_axis() const noexcept { return _xAxis; }
    [[nodiscard]] vector3 get_y_axis() const noexcept { return _yAxis; }
    [[nodiscard]] vector3 get_normal() const noexcept;          // xAxis x yAxis
    [[nodiscard]] const std::vector<vector2>& boundary() const noexcept { return _ring; }
    // interior rings (holes): only a merge_union can create them, like the reference's polygon::inners()
    [[nodiscard]] const std::vector<std::vector<vector2>>& interior_rings() const noexcept { return _inners; }
    [[nodiscard]] std::vector<vector3> get_unprojected_boundary() const;

    void simplify(double distanceThreshold = 10) noexcept;      // Douglas-Peucker, threshold max(area/1e5, distanceThreshold)

    // --- plane matching support ("next" row N2: MapPlane::find_matches, map_primitive.cpp:91-161) -----------------
    // Polygon::project (polygon.cpp:338-382): the same boundary expressed in another plane frame (orthogonal projection)
    [[nodiscard]] Polygon project(const vector3& nextNormal, const vector3& nextCenter) const;
    [[nodiscard]] Polygon project(const vector3& nextXAxis, const vector3& nextYAxis, const vector3& nextCenter) const;
    // inter_area / union_area / inter_over_union (polygon.cpp:525-576): `other` is first projected into this frame.
    // Exact areas of the intersection of two simple polygons by vertical-slab decomposition (no Boost).
    [[nodiscard]] double inter_area(const Polygon& other) const;
    [[nodiscard]] double union_area(const Polygon& other) const;
    [[nodiscard]] double inter_over_union(const Polygon& other) const;
    // Polygon::transform (polygon.cpp:384-451): the polygon moved rigidly with its frame -- every boundary point is
    // lifted to 3D, mapped by get_transformation_matrix (point_coordinates.cpp:24-70: R_to R_from^-1 p + c_to - c_from,
    // the rotation being about the origin like the reference) and re-expressed in the new frame
    [[nodiscard]] Polygon transform(const vector3& nextNormal, const vector3& nextCenter) const;
    [[nodiscard]] Polygon transform(const vector3& nextXAxis, const vector3& nextYAxis, const vector3& nextCenter) const;
    // WorldPolygon::to_camera_space (polygon_coordinates.cpp:135-165 over Polygon::transform_boundary, polygon.cpp:430-451):
    // the polygon seen from another camera.  worldToCamera = 16 doubles, row-major [R t; 0 0 0 1]: the centre goes through
    // the whole transform, the axes through its rotation (re-normalised), every boundary point is lifted to 3-D, moved and
    // re-expressed in the new frame.  Same statements as the device matcher (csrc/cape_match_polygon.hip).
    [[nodiscard]] Polygon to_camera_space(const double* worldToCamera) const;
    // Polygon::merge_union (polygon.cpp:325-336 over union_one :463-493): this polygon becomes (this U other), `other`
    // being projected into this frame first; two disjoint polygons leave the larger one (the reference keeps the biggest
    // piece of the multi-polygon), then simplify().  A region that the two outlines enclose without covering it becomes an
    // interior ring, like in the Boost polygon the reference assigns (area, contains and the intersection / union areas
    // honour it).  Of a hole one of the operands already had, what the other operand leaves uncovered stays a hole (the
    // whole of it, nothing, or the pieces of hole \ other: round 6).  Returns false (and
    // changes nothing) if the result is empty.  Dependency-free: face walks over the arrangement of the two outer rings,
    // touching vertices and collinear overlaps included.
    bool merge_union(const Polygon& other);
    // polygon from an explicit ring in a given frame (polygon.cpp:236-266)
    Polygon(const std::vector<vector2>& ring, const vector3& xAxis, const vector3& yAxis, const vector3& center);

    static std::vector<vector2> compute_concave_hull(const std::vector<vector2>& points) noexcept;
    static std::vector<vector2> compute_convex_hull(const std::vector<vector2>& points) noexcept;

  private:
    struct OpenRing
    {
    };
    // internal: `ring` is already open (no repeated closing vertex), so a degenerate ring whose first and last vertices
    // coincide -- e.g. a projection onto a perpendicular plane -- keeps all its vertices
    Polygon(OpenRing, const std::vector<vector2>& ring, const vector3& xAxis, const vector3& yAxis, const vector3& center);
    void add_hole(std::vector<vector2> hole);
    std::vector<vector2> _ring; // open ring (first vertex not repeated), clockwise like the reference
    std::vector<std::vector<vector2>> _inners; // holes, open rings, counter-clockwise (boost::geometry::correct)
    vector3 _center {0, 0, 0}, _xAxis {1, 0, 0}, _yAxis {0, 1, 0};
    double _area = 0.0;
};

} // namespace rgbd_slam::utils
