#include "FlipDevice.h"

using namespace flipvdb;

FlipDevice::FlipDevice(const CoordBBox &box, float voxel_size, float phi_fluid_bg, float phi_solid_bg, float rest_density,
                       int device): _box(box) {
    const Coord d = box.dim();
    _desc = flip_grid_desc{ d[0], d[1], d[2], box.mn[0], box.mn[1], box.mn[2], voxel_size, phi_fluid_bg, phi_solid_bg, rest_density };
    if (flip_ctx_create(&_desc, device, nullptr, &_ctx) != 0)
        throw std::runtime_error(std::string("flip_ctx_create: ") + flip_last_error(nullptr));
}

FlipDevice::~FlipDevice() { flip_ctx_destroy(_ctx); }

void FlipDevice::check(int rc, const char *what) const {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + flip_last_error(_ctx));
}

void FlipDevice::uploadFloat(int field, const FloatGrid &grid) {
    if (grid.evalActiveVoxelBoundingBox() == _box) {
        check(flip_field_upload(_ctx, field, grid.raw()), "flip_field_upload");
        return;
    }
    // a grid over another box: resample onto the device box through the accessor semantics
    std::vector<float> tmp((size_t)cells());
    size_t q = 0;
    for (int k = _box.mn[2]; k <= _box.mx[2]; ++k)
        for (int j = _box.mn[1]; j <= _box.mx[1]; ++j)
            for (int i = _box.mn[0]; i <= _box.mx[0]; ++i) tmp[q++] = grid.getValue(Coord(i, j, k));
    check(flip_field_upload(_ctx, field, tmp.data()), "flip_field_upload");
}

FloatGrid::Ptr FlipDevice::downloadFloat(int field, float background) {
    auto g = FloatGrid::create(_box, background);
    check(flip_field_download(_ctx, field, g->raw()), "flip_field_download");
    return g;
}

Vec3dGrid::Ptr FlipDevice::downloadVec3d(int field_x) {
    auto g = Vec3dGrid::create(_box, Vec3d(0.0));
    std::vector<double> comp((size_t)cells());
    for (int c = 0; c < 3; ++c) {
        check(flip_field_download(_ctx, field_x + c, comp.data()), "flip_field_download");
        Vec3d *dst = g->raw();
        for (int64_t q = 0; q < cells(); ++q) dst[q][c] = comp[(size_t)q];
    }
    return g;
}

void FlipDevice::uploadVec3d(int field_x, const Vec3dGrid &grid) {
    std::vector<double> comp((size_t)cells());
    for (int c = 0; c < 3; ++c) {
        size_t q = 0;
        for (int k = _box.mn[2]; k <= _box.mx[2]; ++k)
            for (int j = _box.mn[1]; j <= _box.mx[1]; ++j)
                for (int i = _box.mn[0]; i <= _box.mx[0]; ++i) comp[q++] = grid.getValue(Coord(i, j, k))[c];
        check(flip_field_upload(_ctx, field_x + c, comp.data()), "flip_field_upload");
    }
}

MaskGrid::Ptr FlipDevice::downloadMask(int field) {
    auto g = MaskGrid::create(_box, 0);
    check(flip_field_download(_ctx, field, g->raw()), "flip_field_download");
    return g;
}

void FlipDevice::uploadMask(int field, const MaskGrid &grid) {
    std::vector<uint8_t> tmp((size_t)cells());
    size_t q = 0;
    for (int k = _box.mn[2]; k <= _box.mx[2]; ++k)
        for (int j = _box.mn[1]; j <= _box.mx[1]; ++j)
            for (int i = _box.mn[0]; i <= _box.mx[0]; ++i) tmp[q++] = grid.getValue(Coord(i, j, k)) ? 1 : 0;
    check(flip_field_upload(_ctx, field, tmp.data()), "flip_field_upload");
}
