// flipvdb.h -- the small grid vocabulary the host classes are written against.
//
// The reference's public headers speak OpenVDB (openvdb::Vec3d, Coord, CoordBBox, math::Transform,
// FloatGrid::Ptr, Vec3dGrid::Ptr, MaskGrid::Ptr).  On the device every grid is a dense array over one box
// (include/flip_hip.h), so the host side needs only: vectors, integer coordinates, an inclusive box, the
// linear index<->world transform of main.cpp:107-108, and a dense grid with a background value whose
// getValue() behaves like an OpenVDB accessor (background outside the box).  Where OpenVDB is installed
// these types are the seam to replace: DenseGrid<T>::Ptr <-> openvdb::Grid<T>::Ptr conversion is a dense
// copy over evalActiveVoxelBoundingBox().
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <memory>
#include <type_traits>
#include <string>
#include <vector>

namespace flipvdb {

struct Vec3d {
    double v[3];
    Vec3d(): v{ 0, 0, 0 } {}
    explicit Vec3d(double a): v{ a, a, a } {}
    Vec3d(double x, double y, double z): v{ x, y, z } {}
    double &operator[](int i) { return v[i]; }
    const double &operator[](int i) const { return v[i]; }
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double z() const { return v[2]; }
    Vec3d operator+(const Vec3d &o) const { return Vec3d(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    Vec3d operator-(const Vec3d &o) const { return Vec3d(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
    Vec3d operator*(double s) const { return Vec3d(v[0] * s, v[1] * s, v[2] * s); }
    Vec3d operator/(double s) const { return Vec3d(v[0] / s, v[1] / s, v[2] / s); }
    Vec3d &operator+=(const Vec3d &o) { v[0] += o.v[0]; v[1] += o.v[1]; v[2] += o.v[2]; return *this; }
    bool operator==(const Vec3d &o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; }
    double length() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
};
inline Vec3d operator*(double s, const Vec3d &a) { return a * s; }
using Vec3R = Vec3d;
using Vec3f = Vec3d; // main.cpp only uses Vec3f for points that are immediately widened
using Real = double;

struct Coord {
    int32_t c[3];
    Coord(): c{ 0, 0, 0 } {}
    Coord(int32_t x, int32_t y, int32_t z): c{ x, y, z } {}
    int32_t &operator[](int i) { return c[i]; }
    const int32_t &operator[](int i) const { return c[i]; }
    int32_t x() const { return c[0]; }
    int32_t y() const { return c[1]; }
    int32_t z() const { return c[2]; }
    Coord offsetBy(int dx, int dy, int dz) const { return Coord(c[0] + dx, c[1] + dy, c[2] + dz); }
    Vec3d asVec3d() const { return Vec3d(c[0], c[1], c[2]); }
    bool operator==(const Coord &o) const { return c[0] == o.c[0] && c[1] == o.c[1] && c[2] == o.c[2]; }
    // openvdb::Coord::round: nearest integer per axis, floor(x + 0.5)
    static Coord round(const Vec3d &p) {
        return Coord((int32_t)std::floor(p[0] + 0.5), (int32_t)std::floor(p[1] + 0.5), (int32_t)std::floor(p[2] + 0.5));
    }
};

// Inclusive on both ends, like openvdb::CoordBBox.
struct CoordBBox {
    Coord mn, mx;
    CoordBBox(): mn(0, 0, 0), mx(-1, -1, -1) {}
    CoordBBox(const Coord &lo, const Coord &hi): mn(lo), mx(hi) {}
    const Coord &min() const { return mn; }
    const Coord &max() const { return mx; }
    bool empty() const { return mx[0] < mn[0] || mx[1] < mn[1] || mx[2] < mn[2]; }
    Coord dim() const { return Coord(mx[0] - mn[0] + 1, mx[1] - mn[1] + 1, mx[2] - mn[2] + 1); }
    int64_t volume() const { return empty() ? 0 : (int64_t)dim()[0] * dim()[1] * dim()[2]; }
    bool isInside(const Coord &p) const {
        return p[0] >= mn[0] && p[1] >= mn[1] && p[2] >= mn[2] && p[0] <= mx[0] && p[1] <= mx[1] && p[2] <= mx[2];
    }
    bool operator==(const CoordBBox &o) const { return mn == o.mn && mx == o.mx; }
};

struct BBoxd {
    Vec3d mn, mx;
    BBoxd(const Vec3d &lo, const Vec3d &hi): mn(lo), mx(hi) {}
};

// world = voxel_size * (index + pre_translation)   (createLinearTransform + preTranslate, main.cpp:107-108)
class Transform {
public:
    using Ptr = std::shared_ptr<Transform>;
    static Ptr createLinearTransform(double voxel_size) { return Ptr(new Transform(voxel_size)); }
    void preTranslate(const Vec3d &t) { _t += t; }
    Vec3d indexToWorld(const Vec3d &p) const { return (p + _t) * _vs; }
    Vec3d worldToIndex(const Vec3d &p) const { return p / _vs - _t; }
    double voxelSize() const { return _vs; }

private:
    explicit Transform(double vs): _vs(vs) {}
    double _vs;
    Vec3d _t;
};

// What openvdb::GridBase is to main.cpp:117-119,175-181: a common base so that grids of any value type can sit in
// one GridPtrVec.
class GridBase {
public:
    using Ptr = std::shared_ptr<GridBase>;
    virtual ~GridBase() {}
    virtual const char *valueTypeName() const = 0;
    virtual int64_t valueBytes() const = 0;
    virtual const void *rawBytes() const = 0;
    virtual CoordBBox gridBox() const = 0;
};
using GridPtrVec = std::vector<GridBase::Ptr>;
template <class T> struct GridTypeName { static const char *name() { return "bytes"; } };
template <> struct GridTypeName<float> { static const char *name() { return "float"; } };
template <> struct GridTypeName<int32_t> { static const char *name() { return "int32"; } };

// Dense grid over an inclusive box; x fastest: id = (i-ox) + nx*((j-oy) + ny*(k-oz)), the device layout.
template <class T>
class DenseGrid : public GridBase {
public:
    using Ptr = std::shared_ptr<DenseGrid<T>>;
    using ValueType = T;
    // Stands for openvdb::Grid::Accessor (headers/MacGrid.h:33): reads return the background outside the box, writes
    // land in the grid and mark it edited (the owner of a device-backed snapshot uploads it before the next step).
    class Accessor {
    public:
        explicit Accessor(DenseGrid *g): _g(g) {}
        const T &getValue(const Coord &c) const { return _g->getValue(c); }
        void setValue(const Coord &c, const T &v) { _g->setValue(c, v); _g->_edited = true; }
        const DenseGrid *grid() const { return _g; }
    private:
        DenseGrid *_g;
    };
    Accessor getAccessor() { return Accessor(this); }
    bool edited() const { return _edited; }
    void clearEdited() { _edited = false; }
    const char *valueTypeName() const override { return GridTypeName<T>::name(); }
    int64_t valueBytes() const override { return (int64_t)sizeof(T); }
    const void *rawBytes() const override { return _data.data(); }
    CoordBBox gridBox() const override { return _box; }
    static Ptr create(const CoordBBox &box, const T &background) { return Ptr(new DenseGrid(box, background)); }
    Ptr deepCopy() const { return Ptr(new DenseGrid(*this)); }

    const CoordBBox &evalActiveVoxelBoundingBox() const { return _box; }
    const T &background() const { return _bg; }
    void setBackground(const T &bg) { _bg = bg; }
    int64_t index(int i, int j, int k) const {
        const Coord d = _box.dim();
        return (int64_t)(i - _box.mn[0]) + (int64_t)d[0] * ((int64_t)(j - _box.mn[1]) + (int64_t)d[1] * (k - _box.mn[2]));
    }
    const T &getValue(const Coord &c) const { return _box.isInside(c) ? _data[index(c[0], c[1], c[2])] : _bg; }
    void setValue(const Coord &c, const T &v) {
        if (_box.isInside(c)) _data[index(c[0], c[1], c[2])] = v;
    }
    T *raw() { return _data.data(); }
    const T *raw() const { return _data.data(); }
    int64_t size() const { return (int64_t)_data.size(); }
    Transform::Ptr transform() const { return _xform; }
    void setTransform(Transform::Ptr t) { _xform = t; }

private:
    DenseGrid(const CoordBBox &box, const T &bg): _box(box), _bg(bg), _data((size_t)box.volume(), bg) {}
    CoordBBox _box;
    T _bg;
    std::vector<T> _data;
    Transform::Ptr _xform;
    bool _edited = false;
};

using FloatGrid = DenseGrid<float>;
using Vec3dGrid = DenseGrid<Vec3d>;
using MaskGrid = DenseGrid<uint8_t>;
using Int32Grid = DenseGrid<int32_t>;
using Vec3fGrid = Vec3dGrid; // Vec3f is Vec3d here

// Trilinear sample in index space (openvdb::tools::BoxSampler): z, then y, then x, each a + (b - a) * t.
float sampleBox(const FloatGrid &grid, const Vec3d &index_point);
// openvdb::tools::StaggeredBoxSampler: component c sampled at index_point + e_c / 2.
Vec3d sampleStaggered(const Vec3dGrid &grid, const Vec3d &index_point);
// openvdb::tools::BoxSampler on a Vec3d grid: all three components at index_point.
Vec3d sampleCollocated(const Vec3dGrid &grid, const Vec3d &index_point);

// Narrow-band signed distance grids, negative inside (the OpenVDB convention), world units, clipped to
// +-half_width voxels; the dense box is the band's bounding box.
FloatGrid::Ptr createLevelSetBox(const BBoxd &world_box, const Transform &xform, double half_width = 3.0);
FloatGrid::Ptr createLevelSetSphere(float radius, const Vec3d &center, float voxel_size, float half_width,
                                    const Transform &xform);
// openvdb::tools::createLevelSetCube: a cube of edge `scale` centred at `center` (world units).
FloatGrid::Ptr createLevelSetCube(float scale, const Vec3d &center, float voxel_size, const Transform &xform,
                                  float half_width = 3.0f);

// openvdb::tools::csgUnionCopy (main.cpp:178-180) on signed distances: the minimum, over the union of the two boxes.
FloatGrid::Ptr csgUnionCopy(const FloatGrid &a, const FloatGrid &b);

inline void initialize() {} // openvdb::initialize (main.cpp:103): nothing to register

// ---- the nested names main.cpp and the reference's public headers spell (openvdb::math::, ::tools::, ::io::) ----
namespace math {
using Transform = flipvdb::Transform;
using Coord = flipvdb::Coord;
using CoordBBox = flipvdb::CoordBBox;
}
namespace tools {
// createLevelSetBox<GridType, VecType>(bbox, transform, halfWidth): main.cpp:122-125
template <class GridT, class VecT>
typename GridT::Ptr createLevelSetBox(const BBoxd &bbox, const flipvdb::Transform &xform, double half_width = 3.0) {
    return flipvdb::createLevelSetBox(bbox, xform, half_width);
}
// createLevelSetCube<GridType>(scale, centre, voxelSize, halfWidth): main.cpp:83-84.  As in OpenVDB the grid is built
// on a plain linear transform of that voxel size (voxel (i,j,k) at world voxelSize * (i,j,k)); a later setTransform
// re-interprets the same voxels.
template <class GridT>
typename GridT::Ptr createLevelSetCube(float scale, const Vec3d &center, float voxel_size, float half_width = 3.0f) {
    auto plain = flipvdb::Transform::createLinearTransform(voxel_size);
    auto grid = flipvdb::createLevelSetCube(scale, center, voxel_size, *plain, half_width);
    grid->setTransform(plain);
    return grid;
}
template <class GridT>
typename GridT::Ptr createLevelSetSphere(float radius, const Vec3d &center, float voxel_size, float half_width = 3.0f) {
    auto plain = flipvdb::Transform::createLinearTransform(voxel_size);
    auto grid = flipvdb::createLevelSetSphere(radius, center, voxel_size, half_width, *plain);
    grid->setTransform(plain);
    return grid;
}
inline FloatGrid::Ptr csgUnionCopy(const FloatGrid &a, const FloatGrid &b) { return flipvdb::csgUnionCopy(a, b); }
// openvdb::tools::GridSampler<Accessor, Sampler> as headers/MacGrid.h:34-37 instantiates it: index- and world-space
// samples of a Vec3d grid through one of the two samplers below
struct BoxSampler {};
struct StaggeredBoxSampler {};
template <class AccessorT, class SamplerT>
class GridSampler {
public:
    GridSampler(const AccessorT &accessor, const flipvdb::Transform &xform): _grid(accessor.grid()), _xform(xform) {}
    Vec3d isSample(const Vec3d &index_point) const {
        if (std::is_same<SamplerT, StaggeredBoxSampler>::value) return flipvdb::sampleStaggered(*_grid, index_point);
        return flipvdb::sampleCollocated(*_grid, index_point);
    }
    Vec3d wsSample(const Vec3d &world_point) const { return isSample(_xform.worldToIndex(world_point)); }
private:
    const Vec3dGrid *_grid;
    flipvdb::Transform _xform;
};
}
namespace io {
// openvdb::io::File (main.cpp:111-113,193-199).  The .vdb container format is OpenVDB's own and not reproduced: write()
// stores every grid as a dense little-endian block behind a one-line text header (value type, box, voxel size), which
// is enough to look at a result and to compare two runs.
class File {
public:
    explicit File(const std::string &name): _name(name) {}
    void write(const GridPtrVec &grids);
    void close() {}
private:
    std::string _name;
};
}

} // namespace flipvdb
