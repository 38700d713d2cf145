// MacGrid.h -- headers/MacGrid.h:30-339 with the buffers resident on the device.
//
// Getters return a dense host snapshot of the device buffer (a download), where the reference returns the
// live OpenVDB tree; buffer operations (previous / difference buffers) run on the device.  The per-voxel
// accessor setters (headers/MacGrid.h:96-306) write into the snapshot velFront() handed out; a snapshot that
// was written to is kept, handed out again by velFront(), and uploaded by push(), which
// FluidSimulator::advance_flip_pic calls before it touches the device.
#pragma once

#include "flipvdb.h"

#include <memory>

class FlipDevice;

class MacGrid {
public:
    MacGrid(flipvdb::Transform::Ptr i2w_transform, float voxel_size);
    ~MacGrid();

    using StaggeredSampler = flipvdb::tools::GridSampler<flipvdb::Vec3dGrid::Accessor, flipvdb::tools::StaggeredBoxSampler>;
    using Sampler = flipvdb::tools::GridSampler<flipvdb::Vec3dGrid::Accessor, flipvdb::tools::BoxSampler>;

    void updatePreviousVelocityBuffer();   // src/MacGrid.cpp:55-61
    void updateDiffBuffers();              // src/MacGrid.cpp:63

    // src/MacGrid.cpp:65-70: the front and back buffers change places (pointer swaps on the device)
    void swapVelocityBuffers();
    void swapValidMasks();

    // Getters (headers/MacGrid.h:47-66): snapshots of the device buffers.
    flipvdb::Vec3dGrid::Ptr velFront();
    flipvdb::Vec3dGrid::Ptr velBack();
    flipvdb::Vec3dGrid::Ptr velPrev();
    flipvdb::Vec3dGrid::Ptr velDiff();
    flipvdb::FloatGrid::Ptr densityGrid();
    flipvdb::Vec3dGrid::Ptr displacementGrid();
    flipvdb::MaskGrid::Ptr validUFront();
    flipvdb::MaskGrid::Ptr validUBack();
    flipvdb::MaskGrid::Ptr validVFront();
    flipvdb::MaskGrid::Ptr validVBack();
    flipvdb::MaskGrid::Ptr validWFront();
    flipvdb::MaskGrid::Ptr validWBack();
    flipvdb::FloatGrid::Ptr uWeights();
    flipvdb::FloatGrid::Ptr vWeights();
    flipvdb::FloatGrid::Ptr wWeights();
    flipvdb::FloatGrid::Ptr uFluidWeights();
    flipvdb::FloatGrid::Ptr vFluidWeights();
    flipvdb::FloatGrid::Ptr wFluidWeights();

    // headers/MacGrid.h:73-86: through a sampler built on a snapshot's accessor, as the reference's callers do
    // (StaggeredSampler s(grid.velFront()->getAccessor(), *transform)) ...
    inline flipvdb::Vec3d velInterpolatedI(StaggeredSampler &sampler, flipvdb::Vec3d isPoint) const { return sampler.isSample(isPoint); }
    inline flipvdb::Vec3d velInterpolatedI(StaggeredSampler &sampler, double x, double y, double z) const {
        return sampler.isSample(flipvdb::Vec3d(x, y, z));
    }
    inline flipvdb::Vec3d velInterpolatedW(StaggeredSampler &sampler, flipvdb::Vec3d wsPoint) const { return sampler.wsSample(wsPoint); }
    inline flipvdb::Vec3d velInterpolatedW(StaggeredSampler &sampler, double x, double y, double z) const {
        return sampler.wsSample(flipvdb::Vec3d(x, y, z));
    }
    // ... or directly on a snapshot returned by velFront()
    flipvdb::Vec3d velInterpolatedI(const flipvdb::Vec3dGrid &snapshot, flipvdb::Vec3d isPoint) const;
    flipvdb::Vec3d velInterpolatedW(const flipvdb::Vec3dGrid &snapshot, flipvdb::Vec3d wsPoint) const;

    // headers/MacGrid.h:91-94: a caller's grid becomes the back buffer (uploaded; the box must be the device's)
    void setValidUBack(flipvdb::MaskGrid::Ptr mask);
    void setValidVBack(flipvdb::MaskGrid::Ptr mask);
    void setValidWBack(flipvdb::MaskGrid::Ptr mask);
    void setVelBack(flipvdb::Vec3dGrid::Ptr vel_back);

    inline float voxelSize() { return _voxel_size; };

    // ---- per-voxel setters, headers/MacGrid.h:91-306 (same names, argument order and meaning) ----
    using Accessor = flipvdb::Vec3dGrid::Accessor;
    inline flipvdb::Vec3d velHalfIndexed(Accessor &accessor, flipvdb::Coord coord) const { return accessor.getValue(coord); }
    inline flipvdb::Vec3d velHalfIndexed(Accessor &accessor, int k, int j, int i) const { return accessor.getValue(flipvdb::Coord(k, j, i)); }
    void setVelHalfIndexed(Accessor &accessor, flipvdb::Coord coord, flipvdb::Vec3d vel);
    void setVelHalfIndexed(Accessor &accessor, int k, int j, int i, flipvdb::Vec3d vel);
    void setVelOneDimHalfIndexed(Accessor &accessor, flipvdb::Coord coord, double vel, int idx);
    void setVelOneDimHalfIndexed(Accessor &accessor, int k, int j, int i, double vel, int idx);
    // the 2x2x2 block of voxels whose low corner is the coordinate
    void setVel(Accessor &accessor, flipvdb::Coord coord, flipvdb::Vec3d vel);
    void setVel(Accessor &accessor, int k, int j, int i, flipvdb::Vec3d vel);
    void setVelX(Accessor &accessor, flipvdb::Coord coord, double vel);
    void setVelX(Accessor &accessor, int k, int j, int i, double vel);
    void setVelY(Accessor &accessor, flipvdb::Coord coord, double vel);
    void setVelY(Accessor &accessor, int k, int j, int i, double vel);
    void setVelZ(Accessor &accessor, flipvdb::Coord coord, double vel);
    void setVelZ(Accessor &accessor, int k, int j, int i, double vel);
    // trilinear spreading of a value given at a world-space point onto the eight voxels around it
    void addToVelInterpolated(Accessor &accessor, float x, float y, float z, flipvdb::Vec3d value);
    void setVelInterpolated(Accessor &accessor, float x, float y, float z, flipvdb::Vec3d value);

    // uploads the velFront() snapshot if a setter wrote to it since it was handed out
    void push();

    void attach(std::shared_ptr<FlipDevice> device) { _device = device; }
    bool attached() const { return (bool)_device; }

private:
    flipvdb::Transform::Ptr _i2w_transform;
    float _voxel_size;
    std::shared_ptr<FlipDevice> _device;
    flipvdb::Vec3dGrid::Ptr _front_snapshot;
};
