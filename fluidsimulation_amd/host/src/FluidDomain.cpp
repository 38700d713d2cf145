#include "FluidDomain.h"

#include <stdexcept>

using namespace flipvdb;

const float FluidDomain::density { 1000.f };

namespace {
const float kFluidHalfWidth = 3.0f; // openvdb::createLevelSet default half width (src/FluidDomain.cpp:18)
}

FluidDomain::FluidDomain(FloatGrid::Ptr boundary, Transform::Ptr i2w_transform, float voxel_size):
 _i2w_transform(i2w_transform),
 _grid(i2w_transform, voxel_size),
 _fluid_level_set(FloatGrid::create(boundary->evalActiveVoxelBoundingBox(), kFluidHalfWidth * voxel_size)),
 _solid_level_set(boundary), particle_set(i2w_transform), _voxel_size(voxel_size) {
    _fluid_level_set.getLevelSet()->setTransform(_i2w_transform);
    _solid_level_set.getLevelSet()->setTransform(_i2w_transform);
}
FluidDomain::~FluidDomain() {}

void FluidDomain::update(float dt) {
    for (auto fluid_source : _fluid_sources) { fluid_source->update(_grid, particle_set, dt); }
}

void FluidDomain::addFluidSource(std::shared_ptr<FluidSource> fluid_source) { _fluid_sources.push_back(fluid_source); }
void FluidDomain::removeFluidSource(int i) {
    if ((int)_fluid_sources.size() >= i + 1) _fluid_sources.erase(_fluid_sources.begin() + i);
}
void FluidDomain::addSolidObj(std::shared_ptr<SolidObject> solid_obj) {
    // the device couples ONE rigid body in the divergence solve (flip_solid_object_set); a second one would keep moving
    // while being left out of the solve and of the velocity constraints, so it is refused here
    if (!_solid_objs.empty())
        throw std::runtime_error("FluidDomain::addSolidObj: the device path supports one rigid body per domain");
    _solid_objs.push_back(solid_obj);
}
void FluidDomain::removeSolidObj(int i) {
    if ((int)_solid_objs.size() >= i + 1) _solid_objs.erase(_solid_objs.begin() + i);
}
std::shared_ptr<SolidObject> FluidDomain::getSolidObj(int i) { return _solid_objs[i]; }

void FluidDomain::clearParticleSet() { particle_set.clear(); }
void FluidDomain::clearFluidSources() { _fluid_sources.clear(); }
void FluidDomain::clearSolidObjs() { _solid_objs.clear(); }

FlipDevice &FluidDomain::device() {
    const FloatGrid &solid = *_solid_level_set.getLevelSet();
    const CoordBBox &box = solid.evalActiveVoxelBoundingBox();
    if (!_device || !(_device->box() == box) || _device->desc().phi_solid_bg != solid.background()) {
        particle_set.pull(); // keep what an old context holds
        _device = std::make_shared<FlipDevice>(box, _voxel_size, kFluidHalfWidth * _voxel_size, solid.background(), density,
                                               _device_index);
        _grid.attach(_device);
        particle_set.attach(_device);
        _solid_version_on_device = 0;
        _fluid_version_on_device = 0;
        _object_on_device = false;
        _fluid_device_newer = false;
    }
    if (_solid_version_on_device != _solid_level_set.version()) {
        _device->uploadFloat(FLIP_F_PHI_SOLID, solid);
        _solid_version_on_device = _solid_level_set.version();
    }
    if (!_fluid_device_newer && _fluid_version_on_device != _fluid_level_set.version()) {
        _device->uploadFloat(FLIP_F_PHI_FLUID, *_fluid_level_set.getLevelSet());
        _fluid_version_on_device = _fluid_level_set.version();
    }
    return *_device;
}

// src/FluidDomain.cpp:50-61: union of the particles' balls, merged with the solid so that the fluid surface
// continues into the wall (IntersectSolidFluidLevelSets).  Computed on the device from the sorted particles
// (flip_fluid_levelset_from_particles); LevelSet::unionOfBalls is the same computation on the host for callers that
// use a LevelSet on its own.
void FluidDomain::constructFluidLevelSetFromMarkerParticles() {
    if (particle_set.size() > 0) {
        FlipDevice &dev = device();
        particle_set.push();
        dev.check(flip_bin_particles(dev.ctx(), nullptr), "flip_bin_particles");
        dev.check(flip_fluid_levelset_from_particles(dev.ctx(), particle_set.getMinRadius(), kFluidHalfWidth),
                  "flip_fluid_levelset_from_particles");
        _fluid_device_newer = true;
    }
}

LevelSet &FluidDomain::fluidLevelSet() {
    if (_fluid_device_newer && _device) {
        auto g = _device->downloadFloat(FLIP_F_PHI_FLUID, kFluidHalfWidth * _voxel_size);
        g->setTransform(_i2w_transform);
        _fluid_level_set.setLevelSet(g);
        _fluid_version_on_device = _fluid_level_set.version();
        _fluid_device_newer = false;
    }
    return _fluid_level_set;
}

void FluidDomain::syncSolidObject() {
    FlipDevice &dev = device();
    if (_solid_objs.empty()) {
        if (_object_on_device) {
            dev.check(flip_solid_object_set(dev.ctx(), nullptr, 0.f, nullptr, nullptr, nullptr), "flip_solid_object_set");
            _object_on_device = false;
        }
        return;
    }
    SolidObject &obj = *_solid_objs[0]; // one strongly coupled body, as main.cpp:65 sets up
    const FloatGrid &phi = *obj.levelSet().getLevelSet();
    std::vector<float> dense((size_t)dev.cells());
    size_t q = 0;
    const CoordBBox &box = dev.box();
    for (int k = box.mn[2]; k <= box.mx[2]; ++k)
        for (int j = box.mn[1]; j <= box.mx[1]; ++j)
            for (int i = box.mn[0]; i <= box.mx[0]; ++i) dense[q++] = phi.getValue(Coord(i, j, k));
    const Vec3d com = obj.centerOfMass();
    const double c[3] = { com[0], com[1], com[2] };
    const std::array<double, 6> vel = obj.combinedVelocityVector();
    dev.check(flip_solid_object_set(dev.ctx(), dense.data(), phi.background(), c, obj.massMatrixInverse().data(), vel.data()),
              "flip_solid_object_set");
    _object_on_device = true;
}

void FluidDomain::fetchSolidObjectVelocity() {
    if (_solid_objs.empty() || !_object_on_device) return;
    std::array<double, 6> vel;
    _device->check(flip_solid_object_velocity(_device->ctx(), vel.data()), "flip_solid_object_velocity");
    _solid_objs[0]->setVelocityVector(vel);
}

// src/FluidDomain.cpp:63-75: the combined solid (container + bodies) is formed inside the device kernel.
void FluidDomain::advectParticles(float dt) {
    device();
    particle_set.advectAndEnsureOutsideObstacles(dt);
}
