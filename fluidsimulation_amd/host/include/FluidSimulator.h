// FluidSimulator.h -- headers/FluidSimulator.h:12-61.  Same public interface; the private step methods keep
// the reference's names and order, and each body is one or two calls into libflip_hip.so.
#pragma once

#include "FluidDomain.h"
#include "flipvdb.h"

#include "../../../include/flip_dist.h"

struct FlipSolveReport {
    int converged = 0, iterations = 0, rows = 0;
    double rel_residual = 0.0, seconds = 0.0;
};

class FluidSimulator {
public:
    FluidSimulator();
    FluidSimulator(flipvdb::CoordBBox bbox);
    ~FluidSimulator();

    void advance_flip_pic(FluidDomain &domain, float t_frame, float flip_pic_ratio);
    const float grav = 9.81f;

    void print_velocity_field(MacGrid &grid, const char *variable_name);

    static double calculate_kernel_function(flipvdb::Vec3d vec);
    static double calculate_kernel_function(double x, double y, double z);
    static flipvdb::Vec3d calculate_kernel_function_staggered(flipvdb::Vec3d difference);

    // ---- knobs the reference hard-codes (src/FluidSimulator.cpp:31,659-660) ----
    double solver_tolerance = 1e-15;   // Eigen CG tolerance of both solves
    int solver_max_iterations = 0;     // <= 0: 2 * rows, the Eigen default
    int preconditioner = FLIP_PRECOND_MULTIGRID;
    bool density_constraint = true;    // DENSITY_CONSTRAINT_PPE
    bool concurrent_solves = true;     // both systems assembled first and solved side by side (flip_solve_pair): the
                                       // density system does not depend on the divergence solve; same results
    bool fused_particle_update = true; // the three particle updates that close a substep (src/FluidSimulator.cpp:420-429)
                                       // as one device pass; false: the three calls of the reference's loop body
    bool advect = true;                // FluidDomain::advectParticles inside the substep
    bool constrain = true;             // the two constrain_velocity calls of the substep
    bool verbose = true;               // the reference prints per-substep timings
    int substeps_taken = 0;
    FlipSolveReport last_divergence_solve, last_density_solve;

    // ---- several GPUs: one process per GPU, each with the FluidDomain of its z-slab (no counterpart in the reference).
    // attach() hands over the slab handle of include/flip_dist.h built on this domain's context; from then on the two
    // solves run over all slabs (RCCL all-reduces and plane exchanges on the context's stream), compute_cfl takes the
    // maximum over all ranks, the rigid body's velocity update sums J p over all slabs, and every substep ends with the
    // exchange of the particles near the slab faces.  nullptr (the default): one context, as before.
    void attach(flip_dist *slab) { _dist = slab; }
    flip_dist *attached() const { return _dist; }

private:
    float compute_cfl(FluidDomain &domain);
    void transfer_from_particles_to_grid(FluidDomain &domain);
    void add_forces(FluidDomain &domain, float dt);
    void extrapolate_data(FluidDomain &domain, int iterations_n);
    void constrain_velocity(FluidDomain &domain);
    void constrain_velocity(FluidDomain &domain, SolidObject &solidObject);
    void prepare_pressure_solve(FluidDomain &domain);
    void solve_pressure_divirgence(FluidDomain &domain, float dt);
    void solve_pressure_density(FluidDomain &domain, float dt);
    void index_fluid_cells(FluidDomain &domain);
    void compute_face_fractions(FluidDomain &domain);
    void compute_density(FluidDomain &domain);
    void transfer_from_grid_to_particles(FluidDomain &domain, float flip_pic_ratio);
    void apply_particle_displacements(FluidDomain &domain);
    const flipvdb::CoordBBox _bbox;
    flip_dist *_dist = nullptr;
    void check_dist(int rc, const char *what) const;
    bool _density_already_solved = false;  // this substep's density system was solved together with the divergence one
};
