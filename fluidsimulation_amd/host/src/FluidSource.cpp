#include "FluidSource.h"

using namespace flipvdb;

namespace {
const float kRestDensity = 1000.f; // FluidDomain::density; the reference forward-declares it here too
}

FluidSource::FluidSource(LevelSet spawning_region, Vec3d vel, int particle_generation_rate, LevelSet &solidLevelSet,
                         float max_spawning_duration, uint64_t seed):
 _vel(vel),
 _spawning_region(spawning_region), _particle_generation_rate(particle_generation_rate),
 _max_spawning_duration(max_spawning_duration), _spawning_duration_left(max_spawning_duration), _rng(seed) {
    // src/FluidSource.cpp:17-18: clip the region against the solid
    auto g = _spawning_region.getLevelSet()->deepCopy();
    const FloatGrid &solid = *solidLevelSet.getLevelSet();
    const CoordBBox &box = g->evalActiveVoxelBoundingBox();
    for (int k = box.mn[2]; k <= box.mx[2]; ++k)
        for (int j = box.mn[1]; j <= box.mx[1]; ++j)
            for (int i = box.mn[0]; i <= box.mx[0]; ++i) {
                float &v = g->raw()[g->index(i, j, k)];
                v = IntersectSolidFluidLevelSets::intersect(v, solid.getValue(Coord(i, j, k)));
            }
    _spawning_region.setLevelSet(g);
}

FluidSource::~FluidSource() {}

// src/FluidSource.cpp:23-77
void FluidSource::update(MacGrid &grid, ParticleSet &particle_set, float dt) {
    if (_spawning_duration_left <= 0) return;
    const float spawn_amount = std::min(dt, _spawning_duration_left) * _particle_generation_rate;
    const int spawn_counter = static_cast<int>(spawn_amount);
    const float probability_threshold = spawn_amount - static_cast<float>(spawn_counter);
    std::uniform_real_distribution<double> jitter(-0.5, 0.5);

    const CoordBBox box = _spawning_region.getActiveCoordBBox();
    const FloatGrid &region = *_spawning_region.getLevelSet();
    Vec3dGrid::Ptr vel_snapshot = grid.attached() && particle_set.size() > 0 ? grid.velFront() : Vec3dGrid::Ptr();

    for (int i = box.mn[0]; i <= box.mx[0]; ++i)       // ZYX iteration of the reference: z fastest
        for (int j = box.mn[1]; j <= box.mx[1]; ++j)
            for (int k = box.mn[2]; k <= box.mx[2]; ++k) {
                const Coord cell(i, j, k);
                if (!(region.getValue(cell) < 0)) continue;
                int temp_spawn_counter = spawn_counter;
                if (jitter(_rng) < probability_threshold) temp_spawn_counter++;
                int tries = 0;
                while (temp_spawn_counter > 0 && tries < 8 * temp_spawn_counter) {
                    tries++;
                    const Vec3d new_pos = cell.asVec3d() + Vec3d(jitter(_rng), jitter(_rng), jitter(_rng));
                    if (_spawning_region.valueInterpolatedI(new_pos) < 0) {
                        Vec3d new_vel = _vel;
                        if (vel_snapshot) new_vel += grid.velInterpolatedI(*vel_snapshot, new_pos);
                        Particle p(new_pos, new_vel, 1.01f * grid.voxelSize() * std::sqrt(3) / 2.f,
                                   kRestDensity * std::pow(grid.voxelSize(), 3) / spawn_counter);
                        particle_set.addParticle(p);
                        temp_spawn_counter--;
                    }
                }
            }
    _spawning_duration_left -= dt;
}
