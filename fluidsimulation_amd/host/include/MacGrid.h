// MacGrid.h -- headers/MacGrid.h:30-339 with the buffers resident on the device.
//
// Getters return a dense host snapshot of the device buffer (a download), where the reference returns the
// live OpenVDB tree; buffer operations (previous / difference buffers) run on the device.  The reference's
// per-voxel accessor setters (setVel*, addToVelInterpolated, ...) write single voxels of a host tree and
// are used only by its dead Eulerian code path; they have no device-resident meaning and are not kept.
#pragma once

#include "flipvdb.h"

#include <memory>

class FlipDevice;

class MacGrid {
public:
    MacGrid(flipvdb::Transform::Ptr i2w_transform, float voxel_size);
    ~MacGrid();

    void updatePreviousVelocityBuffer();   // src/MacGrid.cpp:55-61
    void updateDiffBuffers();              // src/MacGrid.cpp:63

    // Getters (headers/MacGrid.h:47-66): snapshots of the device buffers.
    flipvdb::Vec3dGrid::Ptr velFront();
    flipvdb::Vec3dGrid::Ptr velPrev();
    flipvdb::Vec3dGrid::Ptr velDiff();
    flipvdb::FloatGrid::Ptr densityGrid();
    flipvdb::Vec3dGrid::Ptr displacementGrid();
    flipvdb::MaskGrid::Ptr validUFront();
    flipvdb::MaskGrid::Ptr validVFront();
    flipvdb::MaskGrid::Ptr validWFront();
    flipvdb::FloatGrid::Ptr uWeights();
    flipvdb::FloatGrid::Ptr vWeights();
    flipvdb::FloatGrid::Ptr wWeights();
    flipvdb::FloatGrid::Ptr uFluidWeights();
    flipvdb::FloatGrid::Ptr vFluidWeights();
    flipvdb::FloatGrid::Ptr wFluidWeights();

    // headers/MacGrid.h:73-86 on a snapshot returned by velFront()
    flipvdb::Vec3d velInterpolatedI(const flipvdb::Vec3dGrid &snapshot, flipvdb::Vec3d isPoint) const;
    flipvdb::Vec3d velInterpolatedW(const flipvdb::Vec3dGrid &snapshot, flipvdb::Vec3d wsPoint) const;

    inline float voxelSize() { return _voxel_size; };

    void attach(std::shared_ptr<FlipDevice> device) { _device = device; }
    bool attached() const { return (bool)_device; }

private:
    flipvdb::Transform::Ptr _i2w_transform;
    float _voxel_size;
    std::shared_ptr<FlipDevice> _device;
};
