// SolidObject.h -- headers/SolidObject.h:9-62: the rigid sphere that is strongly coupled in the
// divergence solve.  Face fractions and the J rows are computed on the device from the object's level set
// (flip_face_fractions, flip_build_system); the host keeps the level set, the rigid state and M^-1.
#pragma once

#include "LevelSet.h"
#include "flipvdb.h"

#include <array>

class SolidObject {
public:
    SolidObject(float radius, float voxel_size, double density, flipvdb::Transform::Ptr xform, flipvdb::Vec3d center,
                flipvdb::Vec3d vel_translational, flipvdb::Vec3d vel_angular);

    inline LevelSet &levelSet() { return _level_set; }
    inline flipvdb::Vec3d &velTranslational() { return _vel_translational; }
    inline flipvdb::Vec3d &velAngular() { return _vel_angular; }
    inline flipvdb::Vec3d centerOfMass() { return _center_of_mass; }
    // diagonal of the 6x6 inverse mass matrix (src/SolidObject.cpp:24-39)
    inline const std::array<double, 6> &massMatrixInverse() const { return _mass_matrix_inverse; }

    void update_positions_orientations(float dt);   // src/SolidObject.cpp:83-101
    std::array<double, 6> combinedVelocityVector(); // src/SolidObject.cpp:42-53
    void setVelocityVector(const std::array<double, 6> &vel_vector);
    void add_forces(float dt);                      // src/SolidObject.cpp:154-157

private:
    void rebuildLevelSet();
    flipvdb::Vec3d _center_of_mass; // index space
    float _radius, _voxel_size;
    flipvdb::Transform::Ptr xform;
    flipvdb::Vec3d _vel_translational, _vel_angular;
    double _density;
    LevelSet _level_set;
    std::array<double, 6> _mass_matrix_inverse;
};
