#include "MacGrid.h"

#include "FlipDevice.h"

using namespace flipvdb;

MacGrid::MacGrid(Transform::Ptr i2w_transform, float voxel_size): _i2w_transform(i2w_transform), _voxel_size(voxel_size) {}
MacGrid::~MacGrid() {}

void MacGrid::updatePreviousVelocityBuffer() { _device->check(flip_update_prev(_device->ctx()), "flip_update_prev"); }
void MacGrid::updateDiffBuffers() { _device->check(flip_update_diff(_device->ctx()), "flip_update_diff"); }

Vec3dGrid::Ptr MacGrid::velFront() { return _device->downloadVec3d(FLIP_F_VEL_U); }
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
