// FluidSource.h -- headers/FluidSource.h / src/FluidSource.cpp:10-77: spawns particles in the cells of a
// level-set region.  The reference draws from a std::random_device-seeded generator (src/util.cpp:5); this
// one takes a seed so that runs repeat.
#pragma once

#include "LevelSet.h"
#include "MacGrid.h"
#include "ParticleSet.h"

#include <random>

class FluidSource {
public:
    FluidSource(LevelSet spawning_region, flipvdb::Vec3d vel, int particle_generation_rate, LevelSet &solidLevelSet,
                float max_spawning_duration, uint64_t seed = 12345);
    ~FluidSource();

    void update(MacGrid &grid, ParticleSet &particle_set, float dt);
    bool active() const { return _spawning_duration_left > 0; }

private:
    flipvdb::Vec3d _vel;
    LevelSet _spawning_region;
    int _particle_generation_rate;
    float _max_spawning_duration, _spawning_duration_left;
    std::mt19937_64 _rng;
};
