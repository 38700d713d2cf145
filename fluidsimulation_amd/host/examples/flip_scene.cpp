// flip_scene.cpp -- the scene of the reference's main.cpp:101-200 (container box, cube source or a seeded
// jittered fill, optional floating sphere) stepped through the host classes, without the SFML window and
// the .vdb output.
//
//   flip_scene [--n 64] [--frames 3] [--t-frame 0.01] [--ratio 0.95] [--fill PPC | --source RATE]
//              [--dam] [--sphere] [--no-advect] [--no-constrain] [--no-density] [--jacobi] [--sequential] [--seed S] [--dump DIR] [--quiet]
//
// --dump DIR writes the inputs of the first substep and the particle state after the last frame as raw
// little-endian arrays plus a small text header, for checks against an independent implementation.
#include "FluidSimulator.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>

using namespace flipvdb;

namespace {
template <class T>
void dump(const std::string &path, const T *data, size_t n) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(data), (std::streamsize)(n * sizeof(T)));
}

void dumpParticles(const std::string &dir, const char *tag, ParticleSet &ps) {
    const size_t n = (size_t)ps.size();
    std::vector<double> pos(3 * n), vel(3 * n);
    std::vector<float> mass(n);
    for (size_t q = 0; q < n; ++q) {
        const Particle &p = ps[q];
        for (int a = 0; a < 3; ++a) { pos[3 * q + a] = p.pos()[a]; vel[3 * q + a] = p.vel()[a]; }
        mass[q] = p.mass();
    }
    dump(dir + "/pos_" + tag + ".bin", pos.data(), pos.size());
    dump(dir + "/vel_" + tag + ".bin", vel.data(), vel.size());
    dump(dir + "/mass.bin", mass.data(), mass.size());
}
} // namespace

int main(int argc, char **argv) {
    int n = 64, frames = 3, fill_ppc = 8, source_rate = 0, device = 0;
    float t_frame = 0.01f, ratio = 0.95f;
    bool dam = false, sphere = false, constrain = true, advect = true, density_on = true, jacobi = false, quiet = false, sequential = false;
    uint64_t seed = 12345;
    std::string dump_dir;
    for (int a = 1; a < argc; ++a) {
        auto is = [&](const char *s) { return std::strcmp(argv[a], s) == 0; };
        auto next = [&]() -> const char * { if (a + 1 >= argc) { fprintf(stderr, "missing value for %s\n", argv[a]); exit(2); } return argv[++a]; };
        if (is("--n")) n = atoi(next());
        else if (is("--frames")) frames = atoi(next());
        else if (is("--t-frame")) t_frame = (float)atof(next());
        else if (is("--ratio")) ratio = (float)atof(next());
        else if (is("--fill")) { fill_ppc = atoi(next()); source_rate = 0; }
        else if (is("--source")) { source_rate = atoi(next()); fill_ppc = 0; }
        else if (is("--dam")) dam = true;
        else if (is("--sphere")) sphere = true;
        else if (is("--no-advect")) advect = false;
        else if (is("--no-constrain")) constrain = false;
        else if (is("--no-density")) density_on = false;
        else if (is("--jacobi")) jacobi = true;
        else if (is("--sequential")) sequential = true;
        else if (is("--quiet")) quiet = true;
        else if (is("--seed")) seed = strtoull(next(), nullptr, 10);
        else if (is("--device")) device = atoi(next());
        else if (is("--dump")) dump_dir = next();
        else { fprintf(stderr, "unknown option %s\n", argv[a]); return 2; }
    }
    const int nx = n, ny = n, nz = n;
    const float width = 1;

    // main.cpp:106-108
    double voxel_size = width / std::max(nx, std::max(ny, nz));
    auto i2w_transform = Transform::createLinearTransform(voxel_size);
    i2w_transform->preTranslate(Vec3d(0.5, 0.5, 0.5));
    Vec3d center_point(width / 2.f, width / 2.f, width / 2.f);

    // main.cpp:122-133: the container, turned inside out
    auto solidBBox = createLevelSetBox(BBoxd(i2w_transform->indexToWorld(Vec3d(nx / 4, 1, nz / 4)),
                                             i2w_transform->indexToWorld(Vec3d(3 * nx / 4, ny - 2, 3 * nz / 4))),
                                       *i2w_transform, 5);
    solidBBox->setTransform(i2w_transform);
    FluidSimulator sim(CoordBBox(Coord(0, 0, 0), Coord(nx, ny, nz)));
    FluidDomain fluidDomain(solidBBox, i2w_transform, (float)voxel_size);
    fluidDomain.setDeviceIndex(device);
    fluidDomain.solidLevelSet().invert();

    sim.advect = advect;
    sim.constrain = constrain;
    sim.density_constraint = density_on;
    sim.concurrent_solves = !sequential;
    sim.verbose = !quiet;
    if (jacobi) sim.preconditioner = FLIP_PRECOND_JACOBI;

    if (sphere) { // main.cpp:60-65
        fluidDomain.addSolidObj(std::make_shared<SolidObject>(0.1f, (float)voxel_size, 250., i2w_transform, center_point,
                                                              Vec3d(0, 0, 0), Vec3d(0)));
    }
    if (source_rate > 0) { // main.cpp:83-95
        auto l = createLevelSetCube(0.5f, center_point - Vec3d(0, width / 4, 0), (float)voxel_size, *i2w_transform);
        l->setTransform(i2w_transform);
        fluidDomain.addFluidSource(std::make_shared<FluidSource>(LevelSet(l->deepCopy()), Vec3d(0, 0., 0.f), source_rate,
                                                                 fluidDomain.solidLevelSet(), t_frame, seed));
    } else {
        // seeded jittered fill: the cube of main.cpp:83-87, or a column against the -x wall (--dam)
        std::mt19937_64 rng(seed);
        std::uniform_real_distribution<double> jitter(-0.5, 0.5);
        const int lo[3] = { nx / 4, 1, nz / 4 };
        const int hi[3] = { dam ? nx / 2 : 3 * nx / 4 + 1, dam ? 3 * ny / 4 : ny / 2, 3 * nz / 4 + 1 };
        const double wall_lo[3] = { nx / 4 + 0.05, 1.05, nz / 4 + 0.05 };
        const double wall_hi[3] = { 3 * nx / 4 - 0.05, ny - 2 - 0.05, 3 * nz / 4 - 0.05 };
        const float radius = 1.01f * (float)voxel_size * std::sqrt(3) / 2.f;      // src/FluidSource.cpp:67
        const float mass = FluidDomain::density * std::pow((float)voxel_size, 3) / fill_ppc; // :68
        for (int k = lo[2]; k < hi[2]; ++k)
            for (int j = lo[1]; j < hi[1]; ++j)
                for (int i = lo[0]; i < hi[0]; ++i)
                    for (int q = 0; q < fill_ppc; ++q) {
                        const Vec3d p(i + jitter(rng), j + jitter(rng), k + jitter(rng));
                        bool inside = true;
                        for (int a = 0; a < 3; ++a) inside = inside && p[a] > wall_lo[a] && p[a] < wall_hi[a];
                        if (inside) fluidDomain.particleSet().addParticle(Particle(p, Vec3d(0.0), radius, mass));
                    }
    }

    if (!dump_dir.empty()) {
        // inputs of the first substep, exactly as the step will see them
        if (source_rate > 0) fluidDomain.update(t_frame);
        fluidDomain.constructFluidLevelSetFromMarkerParticles();
        const FloatGrid &ps = *fluidDomain.solidLevelSet().getLevelSet();
        const FloatGrid &pf = *fluidDomain.fluidLevelSet().getLevelSet();
        const CoordBBox box = ps.evalActiveVoxelBoundingBox();
        dump(dump_dir + "/phi_solid.bin", ps.raw(), (size_t)ps.size());
        dump(dump_dir + "/phi_fluid.bin", pf.raw(), (size_t)pf.size());
        dumpParticles(dump_dir, "before", fluidDomain.particleSet());
        std::ofstream h(dump_dir + "/header.txt");
        h.precision(9);
        h << box.dim()[0] << " " << box.dim()[1] << " " << box.dim()[2] << " " << box.mn[0] << " " << box.mn[1] << " " << box.mn[2]
          << " " << (float)voxel_size << " " << pf.background() << " " << ps.background() << " " << FluidDomain::density << " "
          << fluidDomain.particleSet().size() << " " << t_frame << " " << ratio << "\n";
    }

    auto t_start = std::chrono::high_resolution_clock::now();
    for (int frame = 1; frame <= frames; ++frame) {
        auto t_frame_start = std::chrono::high_resolution_clock::now();
        sim.advance_flip_pic(fluidDomain, t_frame, ratio);
        auto t_end = std::chrono::high_resolution_clock::now();
        printf("Frame %d. Last frame time %.3f. Average time per frame %.3f. Particles %d. Substeps so far %d\n", frame,
               std::chrono::duration_cast<std::chrono::milliseconds>(t_end - t_frame_start).count() / 1000.0,
               std::chrono::duration_cast<std::chrono::milliseconds>(t_end - t_start).count() / 1000.0 / frame,
               fluidDomain.particleSet().size(), sim.substeps_taken);
    }

    // a summary an outside check can read: mean height and speed of the particles
    ParticleSet &ps = fluidDomain.particleSet();
    double mean_y = 0, max_speed = 0;
    for (int q = 0; q < ps.size(); ++q) {
        mean_y += ps[q].pos()[1];
        max_speed = std::max(max_speed, ps[q].vel().length());
    }
    printf("summary particles %d substeps %d mean_y %.9f max_speed %.9f div_converged %d div_iterations %d div_residual %.3e\n",
           ps.size(), sim.substeps_taken, ps.size() ? mean_y / ps.size() : 0.0, max_speed, sim.last_divergence_solve.converged,
           sim.last_divergence_solve.iterations, sim.last_divergence_solve.rel_residual);
    if (!dump_dir.empty()) {
        dumpParticles(dump_dir, "after", ps);
        if (sphere) {
            auto v = fluidDomain.getSolidObj(0)->combinedVelocityVector();
            dump(dump_dir + "/object_vel.bin", v.data(), v.size());
        }
    }
    return 0;
}
