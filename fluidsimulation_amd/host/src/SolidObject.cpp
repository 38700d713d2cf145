#include "SolidObject.h"

using namespace flipvdb;

// src/SolidObject.cpp:12-40.  The mass keeps the reference's integer 4/3 (= 1) and its float arithmetic.
SolidObject::SolidObject(float radius, float voxel_size, double density, Transform::Ptr xform_, Vec3d center,
                         Vec3d vel_translational, Vec3d vel_angular):
 _center_of_mass(center / voxel_size),
 _radius(radius), _voxel_size(voxel_size), xform(xform_), _vel_translational(vel_translational), _vel_angular(vel_angular),
 _density(density), _level_set(createLevelSetSphere(radius, center, voxel_size, 5, *xform_)) {
    _level_set.getLevelSet()->setTransform(xform);
    const float mass = (float)(4 / 3 * 3.14 * std::pow(radius, 3) * _density);
    const double inertia = (double)(2 * mass * (radius * radius) / 5);
    _mass_matrix_inverse = { 1.0 / mass, 1.0 / mass, 1.0 / mass, 1.0 / inertia, 1.0 / inertia, 1.0 / inertia };
}

std::array<double, 6> SolidObject::combinedVelocityVector() {
    return { _vel_translational[0], _vel_translational[1], _vel_translational[2], _vel_angular[0], _vel_angular[1], _vel_angular[2] };
}

void SolidObject::setVelocityVector(const std::array<double, 6> &v) {
    _vel_translational = Vec3d(v[0], v[1], v[2]);
    _vel_angular = Vec3d(v[3], v[4], v[5]);
}

void SolidObject::rebuildLevelSet() {
    // the sphere's signed distance about the moved centre; the reference advects the level-set grid
    // (LevelSetAdvection, src/SolidObject.cpp:100), which for a rigid translation is the same shape
    const Vec3d world_center = _center_of_mass * (double)_voxel_size;
    auto g = createLevelSetSphere(_radius, world_center, _voxel_size, 5, *xform);
    g->setTransform(xform);
    _level_set.setLevelSet(g);
}

void SolidObject::update_positions_orientations(float dt) {
    Vec3d scaled_com = _center_of_mass * (double)_voxel_size;
    scaled_com += (double)dt * _vel_translational;
    _center_of_mass = scaled_com / (double)_voxel_size;
    rebuildLevelSet();
}

void SolidObject::add_forces(float dt) { _vel_translational[1] -= 9.81f * dt; }
