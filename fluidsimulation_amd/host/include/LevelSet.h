// LevelSet.h -- the part of headers/LevelSet.h:18-96 the FLIP path and main.cpp use, on a dense grid.
#pragma once

#include "flipvdb.h"

#include <memory>

class ParticleSet;

class LevelSet {
public:
    ~LevelSet();
    LevelSet(flipvdb::FloatGrid::Ptr level_set);

    // Union of the particles' balls (src/LevelSet.cpp:17-23), clipped to the grid's narrow band, over `box`.
    // Runs on the host: it is an input of the hot path, not part of it.
    void unionOfBalls(ParticleSet &points, float voxel_size, const flipvdb::CoordBBox &box, float half_width = 3.0f);

    // Trilinear value in index / world space (src/LevelSet.cpp:25-37).
    float valueInterpolatedI(flipvdb::Vec3R point) const;
    float valueInterpolatedW(flipvdb::Vec3R point) const;

    // CSG on signed distances (src/LevelSet.cpp:39-47): union = min, intersection = max, difference = max(a,-b).
    void unionLevelSetInPlace(LevelSet &level_set);
    void intersectionLevelSetInPlace(LevelSet &level_set);
    void differenceLevelSetInPlace(LevelSet &level_set);

    // src/LevelSet.cpp:57-61: turns the shape inside out (values and background change sign).
    void invert();

    inline flipvdb::CoordBBox getActiveCoordBBox() { return _level_set->evalActiveVoxelBoundingBox(); };
    inline flipvdb::FloatGrid::Ptr getLevelSet() const { return _level_set; };
    inline void setLevelSet(flipvdb::FloatGrid::Ptr const &level_set) { _level_set = level_set; ++_version; };

    // bumped by every mutation; FluidDomain re-uploads a level set whose version moved
    unsigned version() const { return _version; }
    void touch() { ++_version; }

private:
    flipvdb::FloatGrid::Ptr _level_set;
    unsigned _version = 1;
};

// headers/LevelSet.h:83-96: fluid value a, solid value b -> b <= 0 ? max(a, -b) : a
struct IntersectSolidFluidLevelSets {
    static inline float intersect(float a, float b) { return b <= 0 ? std::max(a, -b) : a; }
};
