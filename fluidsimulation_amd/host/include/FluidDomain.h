// FluidDomain.h -- headers/FluidDomain.h:11-58.  Owns the MacGrid, both level sets and the ParticleSet by
// value as the reference does, plus the device context that backs them.
#pragma once

#include "FlipDevice.h"
#include "FluidSource.h"
#include "LevelSet.h"
#include "MacGrid.h"
#include "ParticleSet.h"
#include "SolidObject.h"
#include "flipvdb.h"

#include <memory>
#include <vector>

class FluidDomain {
public:
    FluidDomain(flipvdb::FloatGrid::Ptr boundary, flipvdb::Transform::Ptr i2w_transform, float voxel_size);
    ~FluidDomain();

    // Density in kg/m^3 (src/FluidDomain.cpp:12)
    static const float density;

    void addFluidSource(std::shared_ptr<FluidSource> fluid_source);
    void removeFluidSource(int i);

    void addSolidObj(std::shared_ptr<SolidObject> solid_obj);
    void removeSolidObj(int i);
    std::shared_ptr<SolidObject> getSolidObj(int i);

    void update(float dt);
    void clearParticleSet();
    void clearFluidSources();
    void clearSolidObjs();

    void constructFluidLevelSetFromMarkerParticles();

    inline MacGrid &grid() { return _grid; };
    // the level set is built on the device; the host grid is refreshed from it when someone asks for it
    LevelSet &fluidLevelSet();
    inline LevelSet &solidLevelSet() { return _solid_level_set; };
    inline std::vector<std::shared_ptr<SolidObject>> &solidObjs() { return _solid_objs; }
    inline ParticleSet &particleSet() { return particle_set; };
    inline std::vector<std::shared_ptr<FluidSource>> &fluidSources() { return _fluid_sources; };
    inline float voxelSize() { return _voxel_size; };

    void advectParticles(float dt);

    flipvdb::Transform::Ptr _i2w_transform;

    // ---- device residency (no counterpart in the reference) ----
    // Creates the context on first use (box = active box of the solid level set, as the reference's dense
    // loops use, src/FluidSimulator.cpp:205,1040,1122) and brings the device copies of the solid level set,
    // the rigid body and the particles up to date.
    FlipDevice &device();
    void syncSolidObject();           // level set + rigid state of solidObjs()[0] -> device
    void fetchSolidObjectVelocity();  // device -> solidObjs()[0] after the divergence projection
    void setDeviceIndex(int device) { _device_index = device; }

private:
    MacGrid _grid;
    LevelSet _fluid_level_set;
    LevelSet _solid_level_set;
    ParticleSet particle_set;
    std::vector<std::shared_ptr<FluidSource>> _fluid_sources;
    std::vector<std::shared_ptr<SolidObject>> _solid_objs;
    float _voxel_size;

    std::shared_ptr<FlipDevice> _device;
    int _device_index = 0;
    unsigned _solid_version_on_device = 0;
    unsigned _fluid_version_on_device = 0;
    bool _object_on_device = false;
    bool _fluid_device_newer = false;
};
