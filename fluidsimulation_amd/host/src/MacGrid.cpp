#include "MacGrid.h"

#include "FlipDevice.h"

using namespace flipvdb;

MacGrid::MacGrid(Transform::Ptr i2w_transform, float voxel_size): _i2w_transform(i2w_transform), _voxel_size(voxel_size) {}
MacGrid::~MacGrid() {}

void MacGrid::updatePreviousVelocityBuffer() { _device->check(flip_update_prev(_device->ctx()), "flip_update_prev"); }
void MacGrid::updateDiffBuffers() { _device->check(flip_update_diff(_device->ctx()), "flip_update_diff"); }

void MacGrid::swapVelocityBuffers() {
    push(); // pending writes belong to the buffer that is the front one now
    _device->check(flip_swap_velocity_buffers(_device->ctx()), "flip_swap_velocity_buffers");
    _front_snapshot.reset();
}
void MacGrid::swapValidMasks() { _device->check(flip_swap_valid_masks(_device->ctx()), "flip_swap_valid_masks"); }

Vec3dGrid::Ptr MacGrid::velFront() {
    if (_front_snapshot && _front_snapshot->edited()) return _front_snapshot; // unsaved writes: hand the same grid out again
    _front_snapshot = _device->downloadVec3d(FLIP_F_VEL_U);
    return _front_snapshot;
}
void MacGrid::push() {
    if (!_front_snapshot || !_front_snapshot->edited()) return;
    _device->uploadVec3d(FLIP_F_VEL_U, *_front_snapshot);
    _front_snapshot->clearEdited();
}

namespace {
template <class F>
void forBlock(const Coord &lo, F fn) { // the 2x2x2 voxels starting at lo, last axis fastest
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int c = 0; c < 2; ++c) fn(lo.offsetBy(a, b, c));
}
void setComponentBlock(MacGrid::Accessor &acc, const Coord &lo, int component, double vel) {
    Vec3d value = acc.getValue(lo);
    value[component] = vel;
    forBlock(lo, [&](const Coord &c) { acc.setValue(c, value); });
}
// weights of the eight voxels around a world-space point, index truncated toward zero as in the reference
template <class F>
void spread(const Transform &xform, float x, float y, float z, const Vec3d &value, F fn) {
    const Vec3d p = xform.worldToIndex(Vec3d(x, y, z));
    const int base[3] = { (int)p[0], (int)p[1], (int)p[2] };
    const double frac[3] = { p[0] - base[0], p[1] - base[1], p[2] - base[2] };
    for (int c = 0; c < 2; ++c)
        for (int b = 0; b < 2; ++b)
            for (int a = 0; a < 2; ++a) {
                const double wz = c ? frac[2] : 1 - frac[2], wy = b ? frac[1] : 1 - frac[1], wx = a ? frac[0] : 1 - frac[0];
                fn(Coord(base[0] + a, base[1] + b, base[2] + c), wx * (wy * (wz * value)));
            }
}
} // namespace

void MacGrid::setVelHalfIndexed(Accessor &acc, Coord coord, Vec3d vel) { acc.setValue(coord, vel); }
void MacGrid::setVelHalfIndexed(Accessor &acc, int k, int j, int i, Vec3d vel) { acc.setValue(Coord(k, j, i), vel); }
void MacGrid::setVelOneDimHalfIndexed(Accessor &acc, Coord coord, double vel, int idx) {
    Vec3d value = acc.getValue(coord);
    value[idx] = vel;
    acc.setValue(coord, value);
}
void MacGrid::setVelOneDimHalfIndexed(Accessor &acc, int k, int j, int i, double vel, int idx) {
    setVelOneDimHalfIndexed(acc, Coord(k, j, i), vel, idx);
}
void MacGrid::setVel(Accessor &acc, Coord coord, Vec3d vel) { forBlock(coord, [&](const Coord &c) { acc.setValue(c, vel); }); }
void MacGrid::setVel(Accessor &acc, int k, int j, int i, Vec3d vel) { setVel(acc, Coord(k, j, i), vel); }
void MacGrid::setVelX(Accessor &acc, Coord coord, double vel) { setComponentBlock(acc, coord, 0, vel); }
void MacGrid::setVelX(Accessor &acc, int k, int j, int i, double vel) { setComponentBlock(acc, Coord(k, j, i), 0, vel); }
void MacGrid::setVelY(Accessor &acc, Coord coord, double vel) { setComponentBlock(acc, coord, 1, vel); }
void MacGrid::setVelY(Accessor &acc, int k, int j, int i, double vel) { setComponentBlock(acc, Coord(k, j, i), 1, vel); }
void MacGrid::setVelZ(Accessor &acc, Coord coord, double vel) { setComponentBlock(acc, coord, 2, vel); }
void MacGrid::setVelZ(Accessor &acc, int k, int j, int i, double vel) { setComponentBlock(acc, Coord(k, j, i), 2, vel); }
void MacGrid::addToVelInterpolated(Accessor &acc, float x, float y, float z, Vec3d value) {
    spread(*_i2w_transform, x, y, z, value, [&](const Coord &c, const Vec3d &share) { acc.setValue(c, acc.getValue(c) + share); });
}
void MacGrid::setVelInterpolated(Accessor &acc, float x, float y, float z, Vec3d value) {
    spread(*_i2w_transform, x, y, z, value, [&](const Coord &c, const Vec3d &share) { acc.setValue(c, share); });
}
Vec3dGrid::Ptr MacGrid::velBack() { return _device->downloadVec3d(FLIP_F_VEL_BACK_U); }
MaskGrid::Ptr MacGrid::validUBack() { return _device->downloadMask(FLIP_F_VALID_BACK_U); }
MaskGrid::Ptr MacGrid::validVBack() { return _device->downloadMask(FLIP_F_VALID_BACK_V); }
MaskGrid::Ptr MacGrid::validWBack() { return _device->downloadMask(FLIP_F_VALID_BACK_W); }
void MacGrid::setValidUBack(MaskGrid::Ptr mask) { _device->uploadMask(FLIP_F_VALID_BACK_U, *mask); }
void MacGrid::setValidVBack(MaskGrid::Ptr mask) { _device->uploadMask(FLIP_F_VALID_BACK_V, *mask); }
void MacGrid::setValidWBack(MaskGrid::Ptr mask) { _device->uploadMask(FLIP_F_VALID_BACK_W, *mask); }
void MacGrid::setVelBack(Vec3dGrid::Ptr vel_back) { _device->uploadVec3d(FLIP_F_VEL_BACK_U, *vel_back); }
Vec3dGrid::Ptr MacGrid::velPrev() { return _device->downloadVec3d(FLIP_F_PREV_U); }
Vec3dGrid::Ptr MacGrid::velDiff() { return _device->downloadVec3d(FLIP_F_DIFF_U); }
FloatGrid::Ptr MacGrid::densityGrid() { return _device->downloadFloat(FLIP_F_DENSITY, 0.f); }
Vec3dGrid::Ptr MacGrid::displacementGrid() { return _device->downloadVec3d(FLIP_F_DISP_X); }
MaskGrid::Ptr MacGrid::validUFront() { return _device->downloadMask(FLIP_F_VALID_U); }
MaskGrid::Ptr MacGrid::validVFront() { return _device->downloadMask(FLIP_F_VALID_V); }
MaskGrid::Ptr MacGrid::validWFront() { return _device->downloadMask(FLIP_F_VALID_W); }
FloatGrid::Ptr MacGrid::uWeights() { return _device->downloadFloat(FLIP_F_WEIGHT_U, 0.f); }
FloatGrid::Ptr MacGrid::vWeights() { return _device->downloadFloat(FLIP_F_WEIGHT_V, 0.f); }
FloatGrid::Ptr MacGrid::wWeights() { return _device->downloadFloat(FLIP_F_WEIGHT_W, 0.f); }
FloatGrid::Ptr MacGrid::uFluidWeights() { return _device->downloadFloat(FLIP_F_FLUID_WEIGHT_U, 0.f); }
FloatGrid::Ptr MacGrid::vFluidWeights() { return _device->downloadFloat(FLIP_F_FLUID_WEIGHT_V, 0.f); }
FloatGrid::Ptr MacGrid::wFluidWeights() { return _device->downloadFloat(FLIP_F_FLUID_WEIGHT_W, 0.f); }

Vec3d MacGrid::velInterpolatedI(const Vec3dGrid &snapshot, Vec3d isPoint) const { return sampleStaggered(snapshot, isPoint); }
Vec3d MacGrid::velInterpolatedW(const Vec3dGrid &snapshot, Vec3d wsPoint) const {
    return sampleStaggered(snapshot, _i2w_transform->worldToIndex(wsPoint));
}
