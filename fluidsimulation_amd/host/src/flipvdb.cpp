// flipvdb.cpp -- samplers and analytic level-set builders of the host grid vocabulary.
#include "flipvdb.h"

#include <cstdio>

namespace flipvdb {

namespace {
template <class V, class Get>
V trilinear(const Vec3d &p, Get get) {
    const double fx = std::floor(p[0]), fy = std::floor(p[1]), fz = std::floor(p[2]);
    const int i = (int)fx, j = (int)fy, k = (int)fz;
    const double tu = p[0] - fx, tv = p[1] - fy, tw = p[2] - fz;
    auto lerp = [](V a, V b, double t) { return (V)(a + (V)((b - a) * t)); };
    return lerp(lerp(lerp(get(i, j, k), get(i, j, k + 1), tw), lerp(get(i, j + 1, k), get(i, j + 1, k + 1), tw), tv),
                lerp(lerp(get(i + 1, j, k), get(i + 1, j, k + 1), tw), lerp(get(i + 1, j + 1, k), get(i + 1, j + 1, k + 1), tw), tv),
                tu);
}
} // namespace

float sampleBox(const FloatGrid &grid, const Vec3d &p) {
    return trilinear<float>(p, [&](int i, int j, int k) { return grid.getValue(Coord(i, j, k)); });
}

Vec3d sampleStaggered(const Vec3dGrid &grid, const Vec3d &p) {
    Vec3d out;
    for (int c = 0; c < 3; ++c) {
        Vec3d q = p;
        q[c] += 0.5;
        out[c] = trilinear<double>(q, [&](int i, int j, int k) { return grid.getValue(Coord(i, j, k))[c]; });
    }
    return out;
}

Vec3d sampleCollocated(const Vec3dGrid &grid, const Vec3d &p) {
    Vec3d out;
    for (int c = 0; c < 3; ++c) out[c] = trilinear<double>(p, [&](int i, int j, int k) { return grid.getValue(Coord(i, j, k))[c]; });
    return out;
}

namespace {
// signed distance (index units) to an axis-aligned box, negative inside
double boxDistance(const Vec3d &p, const Vec3d &lo, const Vec3d &hi) {
    double q[3], outside2 = 0.0, inside = -1e300;
    for (int a = 0; a < 3; ++a) {
        q[a] = std::fabs(p[a] - 0.5 * (lo[a] + hi[a])) - 0.5 * (hi[a] - lo[a]);
        const double m = std::max(q[a], 0.0);
        outside2 += m * m;
        inside = std::max(inside, q[a]);
    }
    return std::sqrt(outside2) + std::min(inside, 0.0);
}

template <class F>
FloatGrid::Ptr bandGrid(const Vec3d &lo_index, const Vec3d &hi_index, double half_width, double voxel, F distance_index) {
    const int hw = (int)std::ceil(half_width);
    CoordBBox box(Coord((int)std::floor(lo_index[0]) - hw, (int)std::floor(lo_index[1]) - hw, (int)std::floor(lo_index[2]) - hw),
                  Coord((int)std::ceil(hi_index[0]) + hw, (int)std::ceil(hi_index[1]) + hw, (int)std::ceil(hi_index[2]) + hw));
    auto grid = FloatGrid::create(box, (float)(half_width * voxel));
    float *data = grid->raw();
#pragma omp parallel for schedule(static)
    for (int k = box.mn[2]; k <= box.mx[2]; ++k)
        for (int j = box.mn[1]; j <= box.mx[1]; ++j)
            for (int i = box.mn[0]; i <= box.mx[0]; ++i) {
                const double d = distance_index(Vec3d(i, j, k));
                data[grid->index(i, j, k)] = (float)(std::min(std::max(d, -half_width), half_width) * voxel);
            }
    return grid;
}
} // namespace

FloatGrid::Ptr createLevelSetBox(const BBoxd &world_box, const Transform &xform, double half_width) {
    const Vec3d lo = xform.worldToIndex(world_box.mn), hi = xform.worldToIndex(world_box.mx);
    // the voxel size as main.cpp:106 computes it: a float quotient widened to double
    const double voxel = (double)(float)xform.voxelSize();
    return bandGrid(lo, hi, half_width, voxel, [&](const Vec3d &p) { return boxDistance(p, lo, hi); });
}

FloatGrid::Ptr createLevelSetSphere(float radius, const Vec3d &center, float voxel_size, float half_width,
                                    const Transform &xform) {
    const double r = (double)radius / voxel_size;
    const Vec3d c = xform.worldToIndex(center);
    const Vec3d lo(c[0] - r, c[1] - r, c[2] - r), hi(c[0] + r, c[1] + r, c[2] + r);
    return bandGrid(lo, hi, half_width, voxel_size, [&](const Vec3d &p) { return (p - c).length() - r; });
}

FloatGrid::Ptr createLevelSetCube(float scale, const Vec3d &center, float voxel_size, const Transform &xform,
                                  float half_width) {
    const Vec3d h(0.5 * scale);
    const Vec3d lo = xform.worldToIndex(center - h), hi = xform.worldToIndex(center + h);
    return bandGrid(lo, hi, half_width, voxel_size, [&](const Vec3d &p) { return boxDistance(p, lo, hi); });
}

FloatGrid::Ptr csgUnionCopy(const FloatGrid &a, const FloatGrid &b) {
    const CoordBBox &ba = a.evalActiveVoxelBoundingBox(), &bb = b.evalActiveVoxelBoundingBox();
    CoordBBox box(Coord(std::min(ba.mn[0], bb.mn[0]), std::min(ba.mn[1], bb.mn[1]), std::min(ba.mn[2], bb.mn[2])),
                  Coord(std::max(ba.mx[0], bb.mx[0]), std::max(ba.mx[1], bb.mx[1]), std::max(ba.mx[2], bb.mx[2])));
    auto out = FloatGrid::create(box, std::min(a.background(), b.background()));
    out->setTransform(a.transform());
    float *data = out->raw();
#pragma omp parallel for schedule(static)
    for (int k = box.mn[2]; k <= box.mx[2]; ++k)
        for (int j = box.mn[1]; j <= box.mx[1]; ++j)
            for (int i = box.mn[0]; i <= box.mx[0]; ++i) {
                const Coord c(i, j, k);
                data[out->index(i, j, k)] = std::min(a.getValue(c), b.getValue(c));
            }
    return out;
}

void io::File::write(const GridPtrVec &grids) {
    FILE *f = std::fopen(_name.c_str(), "wb");
    if (!f) return; // like the reference, which ignores I/O trouble
    std::fprintf(f, "FLIPDENSE 1 grids %zu\n", grids.size());
    for (const GridBase::Ptr &g : grids) {
        const CoordBBox b = g->gridBox();
        std::fprintf(f, "grid %s bytes %lld box %d %d %d %d %d %d\n", g->valueTypeName(), (long long)g->valueBytes(), b.mn[0], b.mn[1],
                     b.mn[2], b.mx[0], b.mx[1], b.mx[2]);
        std::fwrite(g->rawBytes(), (size_t)g->valueBytes(), (size_t)b.volume(), f);
        std::fputc('\n', f);
    }
    std::fclose(f);
}

} // namespace flipvdb
