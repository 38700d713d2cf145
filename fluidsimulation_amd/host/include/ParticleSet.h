// ParticleSet.h -- same public interface as the reference's headers/ParticleSet.h:12-116.
//
// The reference keeps std::vector<Particle> as the only store.  Here the device holds the particles
// (SoA, cell-sorted, include/flip_hip.h) while a simulation runs, and this class keeps the host vector as
// the editable view of them: any accessor that can observe or change particles first pulls the device
// state back (download keeps upload order, so indices stay stable), and marks the device copy stale so the
// next step uploads again.  A frame loop that only calls FluidSimulator::advance_flip_pic never copies.
#pragma once

#include "flipvdb.h"

#include <climits>
#include <memory>
#include <vector>

class FlipDevice;
class LevelSet;

class Particle {
public:
    Particle();
    Particle(flipvdb::Vec3d pos, flipvdb::Vec3d vel, float radius, float mass);
    ~Particle();

    inline flipvdb::Vec3d pos() const { return _pos; };
    inline flipvdb::Vec3d vel() const { return _vel; };
    inline float radius() const { return _radius; };
    inline float mass() const { return _mass; };

    inline void setPosition(flipvdb::Vec3d pos) { _pos = pos; };
    inline void setVelocity(flipvdb::Vec3d vel) { _vel = vel; };

private:
    flipvdb::Vec3d _pos;
    flipvdb::Vec3d _vel;
    float _radius;
    float _mass;
};

class ParticleSet {
public:
    ParticleSet();
    ParticleSet(flipvdb::Transform::Ptr i2w_transform);
    ~ParticleSet();

    using PosType = flipvdb::Vec3R;
    using ScalarType = double;
    using iterator = std::vector<Particle>::iterator;

    // headers/ParticleSet.h:55-56; the velocity and closest-point grids live on the device, so the call
    // takes only the time step (FluidDomain::advectParticles passes it through).
    void advectAndEnsureOutsideObstacles(float dt);
    // The reference's signature (src/ParticleSet.cpp:18-19).  The three grids describe state the device already holds
    // (velFront, the closest-point transform of the solids, their union), so they are accepted and not read.
    void advectAndEnsureOutsideObstacles(flipvdb::Vec3dGrid::Ptr vel_grid, flipvdb::Vec3fGrid::Ptr cpt_grid, LevelSet &solid_level_set,
                                         float dt);

    inline void addParticle(const Particle &p) {
        pull();
        _particles.push_back(p);
        _minRadius = std::min(_minRadius, p.radius());
        _device_stale = true;
    };
    inline void pop_back() { pull(); _particles.pop_back(); _device_stale = true; };
    inline void removeParticle(ParticleSet::iterator p) { _particles.erase(p); _device_stale = true; };

    inline ParticleSet::iterator begin() { pull(); _device_stale = true; return _particles.begin(); };
    inline ParticleSet::iterator end() { return _particles.end(); };

    int size() const;

    inline float getMinRadius() { return _minRadius; };

    inline Particle &operator[](std::size_t index) { pull(); _device_stale = true; return _particles[index]; };

    // World-space position / radius of the nth particle (the OpenVDB particle-list concept).
    void getPos(size_t n, flipvdb::Vec3R &xyz);
    void getRadius(size_t n, ScalarType &radius);
    void getPosRad(size_t n, flipvdb::Vec3R &xyz, flipvdb::Real &radius);

    inline void clear() { _particles.clear(); _host_stale = false; _device_stale = true; };

    inline void setZeroRadius(bool zeroRadius) { _zeroRadius = zeroRadius; };
    flipvdb::Transform::Ptr _i2w_transform;

    // ---- device residency (no counterpart in the reference) ----
    void attach(std::shared_ptr<FlipDevice> device); // rebinding marks the device copy stale
    void pull();             // device -> host vector, if the device copy is newer
    void push();             // host vector -> device, if the host copy is newer
    void markDeviceNewer() { _host_stale = true; }
    // index-space positions without touching the velocity (used by the level-set construction)
    const std::vector<Particle> &hostView() { pull(); return _particles; }

private:
    std::vector<Particle> _particles;
    float _minRadius = INT_MAX;
    bool _zeroRadius = false;
    std::shared_ptr<FlipDevice> _device;
    bool _host_stale = false;   // the device holds newer positions/velocities
    bool _device_stale = true;  // the host vector was edited since the last upload
};
