#include "LevelSet.h"

#include "ParticleSet.h"

using namespace flipvdb;

LevelSet::~LevelSet() {}
LevelSet::LevelSet(FloatGrid::Ptr level_set): _level_set(level_set) {}

void LevelSet::unionOfBalls(ParticleSet &points, float voxel_size, const CoordBBox &box, float half_width) {
    // index units throughout; the narrow band is +-half_width voxels
    auto grid = FloatGrid::create(box, half_width * voxel_size);
    grid->setTransform(_level_set ? _level_set->transform() : Transform::Ptr());
    std::vector<double> best((size_t)box.volume(), (double)half_width);
    const std::vector<Particle> &ps = points.hostView();
    for (const Particle &p : ps) {
        const double r = (double)p.radius() / voxel_size;
        const double reach = r + half_width;
        const Vec3d c = p.pos();
        const int i0 = std::max((int)std::ceil(c[0] - reach), box.mn[0]), i1 = std::min((int)std::floor(c[0] + reach), box.mx[0]);
        const int j0 = std::max((int)std::ceil(c[1] - reach), box.mn[1]), j1 = std::min((int)std::floor(c[1] + reach), box.mx[1]);
        const int k0 = std::max((int)std::ceil(c[2] - reach), box.mn[2]), k1 = std::min((int)std::floor(c[2] + reach), box.mx[2]);
        for (int k = k0; k <= k1; ++k)
            for (int j = j0; j <= j1; ++j) {
                const double dy = j - c[1], dz = k - c[2];
                int64_t id = grid->index(i0, j, k);
                for (int i = i0; i <= i1; ++i, ++id) {
                    const double dx = i - c[0];
                    const double d = std::sqrt(dx * dx + dy * dy + dz * dz) - r;
                    if (d < best[(size_t)id]) best[(size_t)id] = d;
                }
            }
    }
    float *out = grid->raw();
    const double vs = (double)voxel_size;
    for (size_t q = 0; q < best.size(); ++q)
        out[q] = (float)(std::min(std::max(best[q], -(double)half_width), (double)half_width) * vs);
    _level_set = grid;
    ++_version;
}

float LevelSet::valueInterpolatedI(Vec3R point) const { return sampleBox(*_level_set, point); }
float LevelSet::valueInterpolatedW(Vec3R point) const {
    return sampleBox(*_level_set, _level_set->transform()->worldToIndex(point));
}

namespace {
template <class Op>
FloatGrid::Ptr combine(const FloatGrid &a, const FloatGrid &b, Op op) {
    const CoordBBox &ba = a.evalActiveVoxelBoundingBox(), &bb = b.evalActiveVoxelBoundingBox();
    CoordBBox box(Coord(std::min(ba.mn[0], bb.mn[0]), std::min(ba.mn[1], bb.mn[1]), std::min(ba.mn[2], bb.mn[2])),
                  Coord(std::max(ba.mx[0], bb.mx[0]), std::max(ba.mx[1], bb.mx[1]), std::max(ba.mx[2], bb.mx[2])));
    auto out = FloatGrid::create(box, op(a.background(), b.background()));
    out->setTransform(a.transform());
    for (int k = box.mn[2]; k <= box.mx[2]; ++k)
        for (int j = box.mn[1]; j <= box.mx[1]; ++j)
            for (int i = box.mn[0]; i <= box.mx[0]; ++i)
                out->raw()[out->index(i, j, k)] = op(a.getValue(Coord(i, j, k)), b.getValue(Coord(i, j, k)));
    return out;
}
} // namespace

void LevelSet::unionLevelSetInPlace(LevelSet &o) {
    _level_set = combine(*_level_set, *o.getLevelSet(), [](float a, float b) { return std::min(a, b); });
    ++_version;
}
void LevelSet::intersectionLevelSetInPlace(LevelSet &o) {
    _level_set = combine(*_level_set, *o.getLevelSet(), [](float a, float b) { return std::max(a, b); });
    ++_version;
}
void LevelSet::differenceLevelSetInPlace(LevelSet &o) {
    _level_set = combine(*_level_set, *o.getLevelSet(), [](float a, float b) { return std::max(a, -b); });
    ++_version;
}

void LevelSet::invert() {
    float *d = _level_set->raw();
    for (int64_t q = 0; q < _level_set->size(); ++q) d[q] = -d[q];
    _level_set->setBackground(-_level_set->background());
    ++_version;
}
