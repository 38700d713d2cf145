#include "ParticleSet.h"

#include "FlipDevice.h"

using namespace flipvdb;

Particle::Particle() {}
Particle::Particle(Vec3d pos, Vec3d vel, float radius, float mass): _pos(pos), _vel(vel), _radius(radius), _mass(mass) {}
Particle::~Particle() {}

ParticleSet::ParticleSet() {}
ParticleSet::ParticleSet(Transform::Ptr i2w_transform): _i2w_transform(i2w_transform) {}
ParticleSet::~ParticleSet() {}

int ParticleSet::size() const {
    // the count changes on the device only through the particle exchange of a slab decomposition (pull() then resizes)
    if (_host_stale && _device) return (int)flip_particles_count(_device->ctx());
    return (int)_particles.size();
}

void ParticleSet::attach(std::shared_ptr<FlipDevice> device) {
    if (_device != device) {
        if (_device) pull();
        _device = device;
        _device_stale = true;
    }
}

void ParticleSet::pull() {
    if (!_host_stale || !_device) return;
    const size_t n = (size_t)flip_particles_count(_device->ctx());
    std::vector<double> pos(3 * n), vel(3 * n);
    _device->check(flip_particles_download(_device->ctx(), pos.data(), vel.data()), "flip_particles_download");
    if (n != _particles.size()) {
        // a particle exchange between slabs replaced the set: masses come from the device, the radius is the sources' one
        // radius for all (src/FluidSource.cpp:67), taken from a particle this set already had
        const float radius = _particles.empty() ? 0.f : _particles[0].radius();
        std::vector<float> mass(n);
        _device->check(flip_particles_download_mass(_device->ctx(), mass.data()), "flip_particles_download_mass");
        _particles.assign(n, Particle(Vec3d(0, 0, 0), Vec3d(0, 0, 0), radius, 0.f));
        for (size_t q = 0; q < n; ++q) _particles[q] = Particle(Vec3d(0, 0, 0), Vec3d(0, 0, 0), radius, mass[q]);
    }
    for (size_t q = 0; q < n; ++q) {
        _particles[q].setPosition(Vec3d(pos[3 * q], pos[3 * q + 1], pos[3 * q + 2]));
        _particles[q].setVelocity(Vec3d(vel[3 * q], vel[3 * q + 1], vel[3 * q + 2]));
    }
    _host_stale = false;
}

void ParticleSet::push() {
    if (!_device_stale || !_device) return;
    const size_t n = _particles.size();
    std::vector<double> pos(3 * n), vel(3 * n);
    std::vector<float> mass(n);
    for (size_t q = 0; q < n; ++q) {
        const Vec3d p = _particles[q].pos(), v = _particles[q].vel();
        for (int a = 0; a < 3; ++a) { pos[3 * q + a] = p[a]; vel[3 * q + a] = v[a]; }
        mass[q] = _particles[q].mass();
    }
    _device->check(flip_particles_upload(_device->ctx(), (int64_t)n, pos.data(), vel.data(), mass.data()), "flip_particles_upload");
    _device_stale = false;
    _host_stale = false;
}

void ParticleSet::advectAndEnsureOutsideObstacles(float dt) {
    push();
    _device->check(flip_advect_particles(_device->ctx(), dt), "flip_advect_particles");
    _host_stale = true;
}

void ParticleSet::advectAndEnsureOutsideObstacles(Vec3dGrid::Ptr, Vec3fGrid::Ptr, LevelSet &, float dt) {
    advectAndEnsureOutsideObstacles(dt);
}

void ParticleSet::getPos(size_t n, Vec3R &xyz) {
    pull();
    xyz = _i2w_transform->indexToWorld(_particles[n].pos());
}
void ParticleSet::getRadius(size_t n, ScalarType &radius) { radius = _zeroRadius ? 0 : _particles[n].radius(); }
void ParticleSet::getPosRad(size_t n, Vec3R &xyz, Real &radius) {
    getPos(n, xyz);
    radius = _zeroRadius ? 0 : _particles[n].radius();
}
