// FlipDevice.h -- RAII owner of one flip_ctx (include/flip_hip.h).  Every host class that the reference
// backs with OpenVDB trees is backed by this context here; a failing entry point throws with the
// library's error text (the reference's methods return void, so there is no status to mirror).
#pragma once

#include "flipvdb.h"

#include <stdexcept>
#include <string>

#include "../../../include/flip_hip.h"

class FlipDevice {
public:
    FlipDevice(const flipvdb::CoordBBox &box, float voxel_size, float phi_fluid_bg, float phi_solid_bg,
               float rest_density, int device = 0);
    ~FlipDevice();
    FlipDevice(const FlipDevice &) = delete;
    FlipDevice &operator=(const FlipDevice &) = delete;

    flip_ctx *ctx() const { return _ctx; }
    const flipvdb::CoordBBox &box() const { return _box; }
    int64_t cells() const { return _box.volume(); }
    const flip_grid_desc &desc() const { return _desc; }

    // throws std::runtime_error with flip_last_error() when rc != 0
    void check(int rc, const char *what) const;

    void uploadFloat(int field, const flipvdb::FloatGrid &grid);
    flipvdb::FloatGrid::Ptr downloadFloat(int field, float background);
    flipvdb::Vec3dGrid::Ptr downloadVec3d(int field_x);
    void uploadVec3d(int field_x, const flipvdb::Vec3dGrid &grid);
    flipvdb::MaskGrid::Ptr downloadMask(int field);
    void uploadMask(int field, const flipvdb::MaskGrid &grid);

private:
    flip_ctx *_ctx = nullptr;
    flipvdb::CoordBBox _box;
    flip_grid_desc _desc;
};
