// FluidSimulator.cpp -- the substep orchestration of src/FluidSimulator.cpp:307-446 over libflip_hip.so.
#include "FluidSimulator.h"

#include <chrono>
#include <cstdio>
#include <stdexcept>
#include <string>

using namespace flipvdb;
using namespace std::chrono;

namespace {
const float epsilon = 10e-37; // src/FluidSimulator.cpp:32
double hat(double x) {        // headers/util.h:470-478
    if (-1. <= x && x <= 0.) return 1 + x;
    if (0. <= x && x <= 1.) return 1 - x;
    return 0;
}
} // namespace

FluidSimulator::FluidSimulator(): _bbox() {}
FluidSimulator::FluidSimulator(CoordBBox bbox): _bbox(bbox) {}
FluidSimulator::~FluidSimulator() {}

void FluidSimulator::check_dist(int rc, const char *what) const {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + flip_dist_last_error(_dist));
}

// src/FluidSimulator.cpp:1347-1371
double FluidSimulator::calculate_kernel_function(Vec3d vec) { return calculate_kernel_function(vec[0], vec[1], vec[2]); }
double FluidSimulator::calculate_kernel_function(double x, double y, double z) { return hat(x) * hat(y) * hat(z); }
Vec3d FluidSimulator::calculate_kernel_function_staggered(Vec3d d) {
    const float x0 = hat(d[0]), y0 = hat(d[1]), z0 = hat(d[2]);
    const float x1 = hat(d[0] + 0.5), y1 = hat(d[1] + 0.5), z1 = hat(d[2] + 0.5);
    return Vec3d(x1 * y0 * z0, x0 * y1 * z0, x0 * y0 * z1);
}

void FluidSimulator::print_velocity_field(MacGrid &, const char *) {} // debugging aid, empty in the reference too

void FluidSimulator::advance_flip_pic(FluidDomain &domain, float t_frame, float flip_pic_ratio) {
    float t = 0;
    while (t < t_frame) {
        domain.device();
        domain.particleSet().push();
        domain.grid().push(); // velocities written through the MacGrid setters since the last step
        float timestep = compute_cfl(domain);
        if (timestep + t >= t_frame) { timestep = t_frame - t; }
        if (verbose) printf("Taking substep of size %f (to %0.3f%% of the frame)\n", timestep, 100 * (t + timestep) / t_frame);
        auto t0 = high_resolution_clock::now();

        // Fluid sources update: host side, touches the particle vector only while a source is still active
        bool spawning = false;
        for (auto &src : domain.fluidSources()) spawning = spawning || src->active();
        if (spawning) domain.update(timestep);

        domain.constructFluidLevelSetFromMarkerParticles();
        FlipDevice &dev = domain.device(); // uploads the new level set
        domain.particleSet().push();
        domain.syncSolidObject();
        auto t1 = high_resolution_clock::now();

        transfer_from_particles_to_grid(domain);
        dev.check(flip_mark_valid_from_active(dev.ctx()), "flip_mark_valid_from_active"); // :343-345
        extrapolate_data(domain, 2);
        constrain_velocity(domain);
        if (!domain.solidObjs().empty()) constrain_velocity(domain, *domain.solidObjs()[0]);
        domain.grid().updatePreviousVelocityBuffer();
        add_forces(domain, timestep);
        prepare_pressure_solve(domain);
        solve_pressure_divirgence(domain, timestep);
        if (density_constraint) solve_pressure_density(domain, timestep);
        extrapolate_data(domain, 2);
        constrain_velocity(domain);
        if (!domain.solidObjs().empty()) constrain_velocity(domain, *domain.solidObjs()[0]);
        domain.grid().updateDiffBuffers();
        if (fused_particle_update) {
            // :420, :426, :429 as one pass over the particles (flip_update_particles): same results, a third of the traffic
            dev.check(flip_update_particles(dev.ctx(), flip_pic_ratio, timestep, advect ? 1 : 0, density_constraint ? 1 : 0), "flip_update_particles");
            domain.particleSet().markDeviceNewer();
        } else {
            transfer_from_grid_to_particles(domain, flip_pic_ratio);
            if (advect) domain.advectParticles(timestep);
            if (density_constraint) apply_particle_displacements(domain);
        }
        if (_dist) check_dist(flip_dist_exchange_particles(_dist, nullptr, nullptr), "flip_dist_exchange_particles");
        for (auto solidObj : domain.solidObjs()) { solidObj->update_positions_orientations(timestep); }
        dev.check(flip_synchronize(dev.ctx()), "flip_synchronize");
        auto t2 = high_resolution_clock::now();
        if (verbose)
            printf("host stages %.3f s, device substep %.3f s, divergence solve %d it (res %.2e), density solve %d it\n",
                   duration_cast<microseconds>(t1 - t0).count() / 1e6, duration_cast<microseconds>(t2 - t1).count() / 1e6,
                   last_divergence_solve.iterations, last_divergence_solve.rel_residual, last_density_solve.iterations);
        t += timestep;
        ++substeps_taken;
    }
}

// src/FluidSimulator.cpp:448-458
float FluidSimulator::compute_cfl(FluidDomain &domain) {
    FlipDevice &dev = domain.device();
    double m[3];
    dev.check(flip_velocity_max(dev.ctx(), m), "flip_velocity_max");
    Vec3d maxVector(m[0], m[1], m[2]);
    for (auto solidObject : domain.solidObjs()) {
        const Vec3d a = solidObject->velTranslational();
        for (int c = 0; c < 3; ++c) maxVector[c] = std::max(maxVector[c], std::fabs(a[c]));
    }
    if (_dist) { // the substep size is one number for all slabs
        double m3[3] = { maxVector[0], maxVector[1], maxVector[2] };
        check_dist(flip_dist_all_reduce(_dist, m3, 3, FLIP_DIST_MAX), "flip_dist_all_reduce");
        maxVector = Vec3d(m3[0], m3[1], m3[2]);
    }
    return domain.voxelSize() / (std::max(maxVector[0], std::max(maxVector[1], maxVector[2])) + epsilon);
}

// src/FluidSimulator.cpp:71-118
void FluidSimulator::transfer_from_particles_to_grid(FluidDomain &domain) {
    FlipDevice &dev = domain.device();
    dev.check(flip_bin_particles(dev.ctx(), nullptr), "flip_bin_particles");
    dev.check(flip_p2g(dev.ctx()), "flip_p2g");
}

// src/FluidSimulator.cpp:460-480
void FluidSimulator::add_forces(FluidDomain &domain, float dt) {
    FlipDevice &dev = domain.device();
    dev.check(flip_add_forces(dev.ctx(), dt), "flip_add_forces");
    for (auto solidObj : domain.solidObjs()) { solidObj->add_forces(dt); }
    if (!domain.solidObjs().empty()) domain.syncSolidObject();
}

// src/FluidSimulator.cpp:181-265
void FluidSimulator::extrapolate_data(FluidDomain &domain, int iterations_n) {
    FlipDevice &dev = domain.device();
    dev.check(flip_extrapolate(dev.ctx(), iterations_n), "flip_extrapolate");
}

// src/FluidSimulator.cpp:1180-1247 and :1249-1321 (the body's state is the one the device already holds)
void FluidSimulator::constrain_velocity(FluidDomain &domain) {
    if (!constrain) return;
    FlipDevice &dev = domain.device();
    dev.check(flip_constrain_velocity(dev.ctx(), 0), "flip_constrain_velocity");
}
void FluidSimulator::constrain_velocity(FluidDomain &domain, SolidObject &) {
    if (!constrain) return;
    FlipDevice &dev = domain.device();
    dev.check(flip_constrain_velocity(dev.ctx(), 1), "flip_constrain_velocity");
}

// src/FluidSimulator.cpp:482-492
void FluidSimulator::prepare_pressure_solve(FluidDomain &domain) {
    index_fluid_cells(domain);
    compute_face_fractions(domain);
    compute_density(domain);
}
void FluidSimulator::index_fluid_cells(FluidDomain &domain) {
    FlipDevice &dev = domain.device();
    dev.check(flip_index_cells(dev.ctx(), nullptr), "flip_index_cells");
}
void FluidSimulator::compute_face_fractions(FluidDomain &domain) {
    FlipDevice &dev = domain.device();
    dev.check(flip_face_fractions(dev.ctx()), "flip_face_fractions");
}
void FluidSimulator::compute_density(FluidDomain &domain) {
    FlipDevice &dev = domain.device();
    dev.check(flip_density(dev.ctx(), nullptr), "flip_density");
}

namespace {
FlipSolveReport report(const flip_pcg_info &i) {
    FlipSolveReport r;
    r.converged = i.converged; r.iterations = i.iterations; r.rows = i.rows;
    r.rel_residual = i.rel_residual_2; r.seconds = i.seconds;
    return r;
}
} // namespace

// src/FluidSimulator.cpp:494-508 (assembly :654-816, CG :822-823, projection :1017-1102)
void FluidSimulator::solve_pressure_divirgence(FluidDomain &domain, float dt) {
    FlipDevice &dev = domain.device();
    flip_pcg_info info;
    _density_already_solved = false;
    if (_dist) {
        // over all slabs; flip_project_velocity leaves the body alone on a slab context, its update takes the J p of every slab
        if (density_constraint && concurrent_solves) {
            // both slab solves side by side: the collectives of one travel under the kernels of the other
            flip_pcg_info info_density;
            check_dist(flip_dist_solve_pair(_dist, dt, solver_tolerance, solver_max_iterations, preconditioner, &info, &info_density),
                       "flip_dist_solve_pair");
            last_density_solve = report(info_density);
            _density_already_solved = true;
        } else {
            check_dist(flip_dist_solve(_dist, FLIP_SOLVE_DIVERGENCE, dt, solver_tolerance, solver_max_iterations, preconditioner, &info), "flip_dist_solve");
        }
        last_divergence_solve = report(info);
        dev.check(flip_project_velocity(dev.ctx(), dt), "flip_project_velocity");
        if (!domain.solidObjs().empty()) {
            double jp[6];
            dev.check(flip_solid_coupling_partial(dev.ctx(), jp), "flip_solid_coupling_partial");
            check_dist(flip_dist_all_reduce(_dist, jp, 6, FLIP_DIST_SUM), "flip_dist_all_reduce");
            dev.check(flip_solid_velocity_apply(dev.ctx(), dt, jp), "flip_solid_velocity_apply");
        }
        domain.fetchSolidObjectVelocity();
        return;
    }
    if (density_constraint && concurrent_solves) {
        flip_pcg_info info_density;
        dev.check(flip_solve_pair(dev.ctx(), dt, solver_tolerance, solver_max_iterations, preconditioner, &info, &info_density),
                  "flip_solve_pair");
        last_density_solve = report(info_density);
        _density_already_solved = true;
    } else {
        dev.check(flip_build_system(dev.ctx(), FLIP_SOLVE_DIVERGENCE, dt), "flip_build_system");
        dev.check(flip_pcg_solve(dev.ctx(), solver_tolerance, solver_max_iterations, preconditioner, &info), "flip_pcg_solve");
    }
    last_divergence_solve = report(info);
    dev.check(flip_project_velocity(dev.ctx(), dt), "flip_project_velocity");
    domain.fetchSolidObjectVelocity(); // :1095-1100
}

// src/FluidSimulator.cpp:510-524 (assembly :832-990, CG :1006-1007, projection :1104-1178)
void FluidSimulator::solve_pressure_density(FluidDomain &domain, float dt) {
    FlipDevice &dev = domain.device();
    if (_density_already_solved) {
        dev.check(flip_select_system(dev.ctx(), FLIP_SOLVE_DENSITY), "flip_select_system");
        dev.check(flip_project_displacement(dev.ctx(), dt), "flip_project_displacement");
        dev.check(flip_select_system(dev.ctx(), FLIP_SOLVE_DIVERGENCE), "flip_select_system");
        return;
    }
    flip_pcg_info info;
    if (_dist) {
        check_dist(flip_dist_solve(_dist, FLIP_SOLVE_DENSITY, dt, solver_tolerance, solver_max_iterations, preconditioner, &info), "flip_dist_solve");
        last_density_solve = report(info);
        dev.check(flip_project_displacement(dev.ctx(), dt), "flip_project_displacement");
        return;
    }
    dev.check(flip_build_system(dev.ctx(), FLIP_SOLVE_DENSITY, dt), "flip_build_system");
    dev.check(flip_pcg_solve(dev.ctx(), solver_tolerance, solver_max_iterations, preconditioner, &info), "flip_pcg_solve");
    last_density_solve = report(info);
    dev.check(flip_project_displacement(dev.ctx(), dt), "flip_project_displacement");
}

// src/FluidSimulator.cpp:120-135
void FluidSimulator::transfer_from_grid_to_particles(FluidDomain &domain, float flip_pic_ratio) {
    FlipDevice &dev = domain.device();
    dev.check(flip_g2p(dev.ctx(), flip_pic_ratio), "flip_g2p");
    domain.particleSet().markDeviceNewer();
}

// src/FluidSimulator.cpp:137-151
void FluidSimulator::apply_particle_displacements(FluidDomain &domain) {
    FlipDevice &dev = domain.device();
    dev.check(flip_apply_displacements(dev.ctx()), "flip_apply_displacements");
    domain.particleSet().markDeviceNewer();
}
