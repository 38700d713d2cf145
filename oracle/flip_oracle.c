/*
 * flip_oracle.c -- CPU restatement of the FLIP hot path of PanagiotisP/FluidSimulation.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the GPU library is compared with.
 * Nothing in the product path (fluidsimulation_amd/) links, loads or calls it; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * The reference stores every field in sparse OpenVDB trees.  Here every field is a dense
 * array over one axis-aligned box of cells ("the box"), x fastest:
 *     id(i,j,k) = i + nx*(j + ny*k),   index-space coordinate of (i,j,k) = (ox+i, oy+j, oz+k)
 * A read outside the box returns the grid's background value, which is what an OpenVDB
 * accessor returns for an unallocated voxel.  Particle positions are index-space doubles,
 * as in the reference (headers/ParticleSet.h:36-40).
 *
 * Arithmetic types follow the reference expression by expression (float where the
 * reference computes in float, double where it computes in double); compile with
 * -ffp-contract=off so no multiply-add is fused.
 *
 * Every function cites the reference lines (relative to the reference checkout) it follows.
 *
 * Parity status: headers/util.h and headers/pcgsolver/ are checked against the real
 * reference headers (oracle/ref_shim.cpp -> oracle/_ref/, golden vectors in tests/golden/).
 * Everything that depends on OpenVDB / Eigen internals (binning tie-breaks, the box
 * sampler, the resampled corner values, the mask that pre-allocates the weight grids, the
 * Eigen CG reduction order) restates their published algorithms: parity unpinned, because
 * neither library exists in the build container and the reference has no tests or fixtures.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int nx, ny, nz;     /* extent of the box in cells                               */
    int ox, oy, oz;     /* index-space coordinate of local cell (0,0,0)             */
    float dx;           /* voxel size, world units (main.cpp:106)                   */
    float phi_fluid_bg; /* fluid level-set value outside the box                    */
    float phi_solid_bg; /* solid level-set value outside the box                    */
} orc_grid;

#define ORC_EXPORT __attribute__((visibility("default")))

static inline int64_t ncell(const orc_grid *g) { return (int64_t)g->nx * g->ny * g->nz; }
static inline int inside(const orc_grid *g, int i, int j, int k) {
    return i >= 0 && j >= 0 && k >= 0 && i < g->nx && j < g->ny && k < g->nz;
}
static inline int64_t lid(const orc_grid *g, int i, int j, int k) {
    return (int64_t)i + (int64_t)g->nx * ((int64_t)j + (int64_t)g->ny * k);
}
static inline float getf(const orc_grid *g, const float *a, int i, int j, int k, float bg) {
    return inside(g, i, j, k) ? a[lid(g, i, j, k)] : bg;
}
static inline double getd(const orc_grid *g, const double *a, int i, int j, int k) {
    return inside(g, i, j, k) ? a[lid(g, i, j, k)] : 0.0;
}

ORC_EXPORT void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}
ORC_EXPORT int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------ */
/* headers/util.h scalar helpers                                                        */

/* headers/util.h:470-478, T = double */
static inline double hat_d(double x) {
    if (-1. <= x && x <= 0.) return 1 + x;
    else if (0. <= x && x <= 1.) return 1 - x;
    else return 0;
}
ORC_EXPORT double orc_trilinear_hat(double x) { return hat_d(x); }

/* headers/util.h:200-208, T = float */
static inline float clamp_f(float a, float lower, float upper) {
    if (a < lower) return lower;
    else if (a > upper) return upper;
    else return a;
}
ORC_EXPORT float orc_clamp_f(float a, float lo, float hi) { return clamp_f(a, lo, hi); }

/* headers/util.h:493-501, T = float */
static inline float frac2(float phi_left, float phi_right) {
    if (phi_left < 0 && phi_right < 0) return 1;
    if (phi_left < 0 && phi_right >= 0) return phi_left / (phi_left - phi_right);
    if (phi_left >= 0 && phi_right < 0) return phi_right / (phi_right - phi_left);
    else return 0;
}
ORC_EXPORT float orc_fraction_inside2(float l, float r) { return frac2(l, r); }

/* headers/util.h:485-490 */
static inline void cycle4(float *arr) {
    float t = arr[0];
    arr[0] = arr[1]; arr[1] = arr[2]; arr[2] = arr[3]; arr[3] = t;
}

/* headers/util.h:505-580, T = float */
static float frac4(float phi_bl, float phi_br, float phi_tl, float phi_tr) {
    int inside_count = (phi_bl < 0 ? 1 : 0) + (phi_tl < 0 ? 1 : 0) + (phi_br < 0 ? 1 : 0) + (phi_tr < 0 ? 1 : 0);
    float list[4] = { phi_bl, phi_br, phi_tr, phi_tl };
    float return_value;
    if (inside_count == 4) return 1;
    else if (inside_count == 3) {
        while (list[0] < 0) cycle4(list);
        float side0 = 1 - frac2(list[0], list[3]);
        float side1 = 1 - frac2(list[0], list[1]);
        return_value = 1 - 0.5f * side0 * side1;
    } else if (inside_count == 2) {
        while (list[0] >= 0 || !(list[1] < 0 || list[2] < 0)) cycle4(list);
        if (list[1] < 0) {
            float side_left = frac2(list[0], list[3]);
            float side_right = frac2(list[1], list[2]);
            return_value = 0.5f * (side_left + side_right);
        } else {
            float middle_point = 0.25f * (list[0] + list[1] + list[2] + list[3]);
            if (middle_point < 0) {
                float area = 0;
                float side1 = 1 - frac2(list[0], list[3]);
                float side3 = 1 - frac2(list[2], list[3]);
                area += 0.5f * side1 * side3;
                float side2 = 1 - frac2(list[2], list[1]);
                float side0 = 1 - frac2(list[0], list[1]);
                area += 0.5f * side0 * side2;
                return_value = 1 - area;
            } else {
                float area = 0;
                float side0 = frac2(list[0], list[1]);
                float side1 = frac2(list[0], list[3]);
                area += 0.5f * side0 * side1;
                float side2 = frac2(list[2], list[1]);
                float side3 = frac2(list[2], list[3]);
                area += 0.5f * side2 * side3;
                return_value = area;
            }
        }
    } else if (inside_count == 1) {
        while (list[0] >= 0) cycle4(list);
        float side0 = frac2(list[0], list[3]);
        float side1 = frac2(list[0], list[1]);
        return_value = 0.5f * side0 * side1;
    } else
        return_value = 0;

    if ((double)return_value > 0.9) return 1;
    if ((double)return_value < 0.1) return 0;
    else return clamp_f(return_value, 0.f, 1.f);
}
ORC_EXPORT float orc_fraction_inside4(float bl, float br, float tl, float tr) { return frac4(bl, br, tl, tr); }

/* src/FluidSimulator.cpp:1361-1371: three hats rounded to float, float products */
ORC_EXPORT void orc_kernel_staggered(const double d[3], double out[3]) {
    float val_x_0 = (float)hat_d(d[0]);
    float val_y_0 = (float)hat_d(d[1]);
    float val_z_0 = (float)hat_d(d[2]);
    float val_x_1 = (float)hat_d(d[0] + 0.5);
    float val_y_1 = (float)hat_d(d[1] + 0.5);
    float val_z_1 = (float)hat_d(d[2] + 0.5);
    out[0] = (double)(val_x_1 * val_y_0 * val_z_0);
    out[1] = (double)(val_x_0 * val_y_1 * val_z_0);
    out[2] = (double)(val_x_0 * val_y_0 * val_z_1);
}

/* src/FluidSimulator.cpp:1347-1352: three double hats, double product */
ORC_EXPORT double orc_kernel(const double d[3]) {
    double val1 = hat_d(d[0]);
    double val2 = hat_d(d[1]);
    double val3 = hat_d(d[2]);
    return val1 * val2 * val3;
}

/* ------------------------------------------------------------------------------------ */
/* Cell binning.  The reference rebuilds an OpenVDB ParticleAtlas / point mask           */
/* (src/FluidSimulator.cpp:83-90, 281-283, 628-630, 686-688); a particle belongs to the  */
/* voxel Coord::round(index position) = floor(pos + 0.5) per axis.                      */

static inline int cell_of(double p) { return (int)floor(p + 0.5); }

/* Stable counting sort by cell id.  cell_id[p] = linear id or -1 when outside the box.
 * cell_start has B+1 entries; slots [cell_start[B], n) of perm hold the outside particles.
 * Returns the number of particles outside the box. */
ORC_EXPORT int64_t orc_bin_particles(const orc_grid *g, int64_t n, const double *pos, int32_t *cell_id,
                                     int32_t *cell_start, int32_t *perm) {
    const int64_t B = ncell(g);
    memset(cell_start, 0, (size_t)(B + 1) * sizeof(int32_t));
    int64_t n_out = 0;
    for (int64_t p = 0; p < n; ++p) {
        int i = cell_of(pos[3 * p + 0]) - g->ox;
        int j = cell_of(pos[3 * p + 1]) - g->oy;
        int k = cell_of(pos[3 * p + 2]) - g->oz;
        if (inside(g, i, j, k)) {
            cell_id[p] = (int32_t)lid(g, i, j, k);
            cell_start[cell_id[p] + 1]++;
        } else {
            cell_id[p] = -1;
            ++n_out;
        }
    }
    for (int64_t c = 0; c < B; ++c) cell_start[c + 1] += cell_start[c];
    int32_t *fill = (int32_t *)malloc((size_t)(B + 1) * sizeof(int32_t));
    memcpy(fill, cell_start, (size_t)(B + 1) * sizeof(int32_t));
    for (int64_t p = 0; p < n; ++p) {
        int32_t c = cell_id[p] >= 0 ? cell_id[p] : (int32_t)B;
        perm[fill[c]++] = (int32_t)p;
    }
    free(fill);
    return n_out;
}

/* Cells that hold at least one particle, dilated `radius` cells in the 26-neighbourhood
 * sense (createPointMask + dilateActiveValues(NN_FACE_EDGE_VERTEX), e.g.
 * src/FluidSimulator.cpp:87-90), clipped to the box. */
ORC_EXPORT void orc_particle_mask(const orc_grid *g, const int32_t *cell_start, int radius, uint8_t *mask) {
    const int64_t B = ncell(g);
    memset(mask, 0, (size_t)B);
    for (int k = 0; k < g->nz; ++k)
        for (int j = 0; j < g->ny; ++j)
            for (int i = 0; i < g->nx; ++i) {
                int64_t c = lid(g, i, j, k);
                if (cell_start[c + 1] == cell_start[c]) continue;
                for (int kk = k - radius; kk <= k + radius; ++kk)
                    for (int jj = j - radius; jj <= j + radius; ++jj)
                        for (int ii = i - radius; ii <= i + radius; ++ii)
                            if (inside(g, ii, jj, kk)) mask[lid(g, ii, jj, kk)] = 1;
            }
}

/* Visit order of the particles one cell gathers.  The reference walks whatever list OpenVDB's ParticleAtlas   */
/* hands back for the query box (src/Functors.cpp:21-26): an order internal to OpenVDB and to the TBB schedule, */
/* so any fixed order restates it equally well (parity unpinned).  The order used here and by the GPU library:  */
/* neighbour grid rows (kk, jj) ascending; inside a row by half cell -- z half, then y half, then position      */
/* along x in half cells (2*ii + x half) -- and by upload index inside one half cell.  "Upper half" of a cell    */
/* n is decided by the expression the gather of cell n+1 evaluates for the lower edge of its query box,        */
/* pos - (n+1) >= -1, so that a half cell is either wholly inside or wholly outside that box.                   */
static inline int upper_half(double p, int n_index) { return (p - (double)(n_index + 1)) >= -1.0 ? 1 : 0; }

typedef struct { int32_t key; int32_t p; } orc_cand;
static int cand_cmp(const void *a, const void *b) {
    const orc_cand *x = (const orc_cand *)a, *y = (const orc_cand *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->p < y->p ? -1 : (x->p > y->p ? 1 : 0);
}
/* The particles of cells (i-1..i+1, jj, kk) in visit order.  *buf grows as needed. */
static int32_t row_candidates(const orc_grid *g, const double *pos, const int32_t *cell_start, const int32_t *perm,
                              int i, int jj, int kk, orc_cand **buf, int32_t *cap) {
    int32_t m = 0;
    for (int ii = i - 1; ii <= i + 1; ++ii) {
        if (!inside(g, ii, jj, kk)) continue;
        const int64_t nb = lid(g, ii, jj, kk);
        for (int32_t s = cell_start[nb]; s < cell_start[nb + 1]; ++s) {
            const int32_t p = perm[s];
            if (m == *cap) {
                *cap = *cap ? 2 * *cap : 64;
                *buf = (orc_cand *)realloc(*buf, (size_t)*cap * sizeof(orc_cand));
            }
            const int hx = upper_half(pos[3 * (int64_t)p], g->ox + ii);
            const int hy = upper_half(pos[3 * (int64_t)p + 1], g->oy + jj);
            const int hz = upper_half(pos[3 * (int64_t)p + 2], g->oz + kk);
            (*buf)[m].key = (hz * 2 + hy) * (2 * g->nx) + 2 * ii + hx;
            (*buf)[m].p = p;
            ++m;
        }
    }
    if (m > 1) qsort(*buf, (size_t)m, sizeof(orc_cand), cand_cmp);
    return m;
}

/* ------------------------------------------------------------------------------------ */
/* P2G gather: src/FluidSimulator.cpp:71-118 + src/Functors.cpp:10-45.                  */
/* For every cell of the band (particle cells dilated 2) visit the particles with        */
/* |pos - cell| <= 1 per axis (the atlas query box of Functors.cpp:21-24 with zero       */
/* radii), accumulate weight and weight*velocity in double starting from the grids'      */
/* backgrounds (weight: float 10e-37 widened; velocity: 0), then divide component-wise   */
/* (FluidSimulator.cpp:108).  Visit order: see row_candidates.                            */
ORC_EXPORT void orc_p2g(const orc_grid *g, int64_t n, const double *pos, const double *vel,
                        const int32_t *cell_start, const int32_t *perm, const uint8_t *band,
                        double *u, double *v, double *w) {
    (void)n;
    const float epsilon = 10e-37f; /* FluidSimulator.cpp:32 */
#pragma omp parallel
    {
    orc_cand *buf = NULL;
    int32_t cap = 0;
#pragma omp for schedule(dynamic, 1)
    for (int k = 0; k < g->nz; ++k)
        for (int j = 0; j < g->ny; ++j)
            for (int i = 0; i < g->nx; ++i) {
                const int64_t c = lid(g, i, j, k);
                if (!band[c]) { u[c] = v[c] = w[c] = 0.0; continue; }
                double ws[3] = { (double)epsilon, (double)epsilon, (double)epsilon };
                double vs[3] = { 0.0, 0.0, 0.0 };
                const double cx = (double)(g->ox + i), cy = (double)(g->oy + j), cz = (double)(g->oz + k);
                for (int kk = k - 1; kk <= k + 1; ++kk)
                    for (int jj = j - 1; jj <= j + 1; ++jj)
                        {
                            const int32_t m = row_candidates(g, pos, cell_start, perm, i, jj, kk, &buf, &cap);
                            for (int32_t s = 0; s < m; ++s) {
                                const int64_t p = buf[s].p;
                                double d[3] = { pos[3 * p] - cx, pos[3 * p + 1] - cy, pos[3 * p + 2] - cz };
                                if (d[0] < -1. || d[0] > 1. || d[1] < -1. || d[1] > 1. || d[2] < -1. || d[2] > 1.)
                                    continue; /* outside the atlas query box */
                                double wt[3];
                                orc_kernel_staggered(d, wt);
                                if (wt[0] != 0.0 || wt[1] != 0.0 || wt[2] != 0.0) { /* !weight_vec.isZero() */
                                    for (int a = 0; a < 3; ++a) {
                                        ws[a] = ws[a] + wt[a];
                                        vs[a] = vs[a] + wt[a] * vel[3 * p + a];
                                    }
                                }
                            }
                        }
                u[c] = vs[0] / ws[0];
                v[c] = vs[1] / ws[1];
                w[c] = vs[2] / ws[2];
            }
    free(buf);
    }
}

/* ------------------------------------------------------------------------------------ */
/* OpenVDB 9 tools::BoxSampler::trilinearInterpolation (openvdb/tools/Interpolation.h)   */
/* as called through StaggeredBoxSampler by MacGrid::StaggeredSampler                    */
/* (headers/MacGrid.h:36-37): interpolate along z, then y, then x, each as a+(b-a)*w.    */
static inline double lerp_d(double a, double b, double wgt) {
    const double temp = (b - a) * wgt;
    return a + temp;
}
static double box_sample_d(const orc_grid *g, const double *f, double x, double y, double z) {
    const double fx = floor(x), fy = floor(y), fz = floor(z);
    const int i = (int)fx - g->ox, j = (int)fy - g->oy, k = (int)fz - g->oz;
    const double tu = x - fx, tv = y - fy, tw = z - fz;
    double d[2][2][2];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int c = 0; c < 2; ++c) d[a][b][c] = getd(g, f, i + a, j + b, k + c);
    return lerp_d(lerp_d(lerp_d(d[0][0][0], d[0][0][1], tw), lerp_d(d[0][1][0], d[0][1][1], tw), tv),
                  lerp_d(lerp_d(d[1][0][0], d[1][0][1], tw), lerp_d(d[1][1][0], d[1][1][1], tw), tv), tu);
}
/* StaggeredBoxSampler::sample: component c is sampled at pos + 0.5*e_c */
static void staggered_sample(const orc_grid *g, const double *u, const double *v, const double *w,
                             const double p[3], double out[3]) {
    out[0] = box_sample_d(g, u, p[0] + 0.5, p[1], p[2]);
    out[1] = box_sample_d(g, v, p[0], p[1] + 0.5, p[2]);
    out[2] = box_sample_d(g, w, p[0], p[1], p[2] + 0.5);
}
ORC_EXPORT void orc_staggered_sample(const orc_grid *g, const double *u, const double *v, const double *w,
                                     const double p[3], double out[3]) {
    staggered_sample(g, u, v, w, p, out);
}

/* G2P: src/FluidSimulator.cpp:120-135.  (1 - flip_pic_ratio) is a float expression. */
ORC_EXPORT void orc_g2p(const orc_grid *g, int64_t n, const double *pos, double *vel, const double *u,
                        const double *v, const double *w, const double *du, const double *dv, const double *dw,
                        float flip_pic_ratio) {
    const float one_minus = 1 - flip_pic_ratio;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        double pic[3], dif[3];
        staggered_sample(g, u, v, w, pos + 3 * p, pic);
        staggered_sample(g, du, dv, dw, pos + 3 * p, dif);
        for (int a = 0; a < 3; ++a) {
            const double flip = vel[3 * p + a] + dif[a];
            vel[3 * p + a] = pic[a] * (double)one_minus + flip * (double)flip_pic_ratio;
        }
    }
}

/* src/FluidSimulator.cpp:137-151 */
ORC_EXPORT void orc_apply_displacements(const orc_grid *g, int64_t n, double *pos, const double *dispx,
                                        const double *dispy, const double *dispz) {
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        double d[3];
        staggered_sample(g, dispx, dispy, dispz, pos + 3 * p, d);
        for (int a = 0; a < 3; ++a) pos[3 * p + a] = pos[3 * p + a] + d[a];
    }
}

/* ------------------------------------------------------------------------------------ */
/* Face area fractions: src/FluidSimulator.cpp:527-623, src/SolidObject.cpp:103-152.    */

/* The mask that pre-allocates the weight grids (FluidSimulator.cpp:562-566):            */
/* extractIsosurfaceMask(phi,0) U extractEnclosedRegion(phi,0), dilated once over the    */
/* 26-neighbourhood.  Restated as: voxels with phi<0, plus voxels with phi>=0 that have  */
/* a face neighbour with phi<0 (the isovalue-crossing voxels on the outer side), then    */
/* dilated.  OpenVDB internals: parity unpinned.                                        */
ORC_EXPORT void orc_weight_mask(const orc_grid *g, const float *phi, float bg, uint8_t *mask) {
    const int64_t B = ncell(g);
    uint8_t *seed = (uint8_t *)calloc((size_t)B, 1);
    for (int k = 0; k < g->nz; ++k)
        for (int j = 0; j < g->ny; ++j)
            for (int i = 0; i < g->nx; ++i) {
                const int64_t c = lid(g, i, j, k);
                if (phi[c] < 0) { seed[c] = 1; continue; }
                if (getf(g, phi, i - 1, j, k, bg) < 0 || getf(g, phi, i + 1, j, k, bg) < 0
                    || getf(g, phi, i, j - 1, k, bg) < 0 || getf(g, phi, i, j + 1, k, bg) < 0
                    || getf(g, phi, i, j, k - 1, bg) < 0 || getf(g, phi, i, j, k + 1, bg) < 0)
                    seed[c] = 1;
            }
    memset(mask, 0, (size_t)B);
    for (int k = 0; k < g->nz; ++k)
        for (int j = 0; j < g->ny; ++j)
            for (int i = 0; i < g->nx; ++i) {
                if (!seed[lid(g, i, j, k)]) continue;
                for (int kk = k - 1; kk <= k + 1; ++kk)
                    for (int jj = j - 1; jj <= j + 1; ++jj)
                        for (int ii = i - 1; ii <= i + 1; ++ii)
                            if (inside(g, ii, jj, kk)) mask[lid(g, ii, jj, kk)] = 1;
            }
    free(seed);
}

/* BoxSampler on a float grid: (b-a)*weight is a double product narrowed back to float. */
static inline float lerp_f(float a, float b, double wgt) {
    const double temp = (double)(b - a) * wgt;
    return a + (float)temp;
}
/* Level-set value at the low corner (i-1/2, j-1/2, k-1/2) of cell (i,j,k): the trilinear */
/* resample of FluidSimulator.cpp:554-560 evaluated half a voxel below the cell centre.   */
static float corner_value(const orc_grid *g, const float *phi, float bg, int i, int j, int k) {
    float d[2][2][2];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int c = 0; c < 2; ++c) d[a][b][c] = getf(g, phi, i - 1 + a, j - 1 + b, k - 1 + c, bg);
    return lerp_f(lerp_f(lerp_f(d[0][0][0], d[0][0][1], 0.5), lerp_f(d[0][1][0], d[0][1][1], 0.5), 0.5),
                  lerp_f(lerp_f(d[1][0][0], d[1][0][1], 0.5), lerp_f(d[1][1][0], d[1][1][1], 0.5), 0.5), 0.5);
}
ORC_EXPORT float orc_corner_value(const orc_grid *g, const float *phi, float bg, int i, int j, int k) {
    return corner_value(g, phi, bg, i, j, k);
}

/* FluidSimulator.cpp:576-597: three fractions per masked cell; zero elsewhere. */
ORC_EXPORT void orc_face_fractions(const orc_grid *g, const float *phi, float bg, const uint8_t *mask, float *wu,
                                   float *wv, float *ww) {
#pragma omp parallel for schedule(static)
    for (int k = 0; k < g->nz; ++k)
        for (int j = 0; j < g->ny; ++j)
            for (int i = 0; i < g->nx; ++i) {
                const int64_t c = lid(g, i, j, k);
                if (!mask[c]) { wu[c] = wv[c] = ww[c] = 0.f; continue; }
                const float p000 = corner_value(g, phi, bg, i, j, k);
                const float p001 = corner_value(g, phi, bg, i, j, k + 1);
                const float p010 = corner_value(g, phi, bg, i, j + 1, k);
                const float p011 = corner_value(g, phi, bg, i, j + 1, k + 1);
                const float p100 = corner_value(g, phi, bg, i + 1, j, k);
                const float p101 = corner_value(g, phi, bg, i + 1, j, k + 1);
                const float p110 = corner_value(g, phi, bg, i + 1, j + 1, k);
                wu[c] = frac4(p000, p010, p001, p011);
                wv[c] = frac4(p000, p001, p100, p101);
                ww[c] = frac4(p000, p010, p100, p110);
            }
}

/* FluidSimulator.cpp:603-619: add a solid object's fractions, then clamp to [0,1]. */
ORC_EXPORT void orc_combine_weights(int64_t B, float *w, const float *w_obj) {
    for (int64_t c = 0; c < B; ++c) w[c] = clamp_f(w_obj ? w[c] + w_obj[c] : w[c], 0.f, 1.f);
}

/* ------------------------------------------------------------------------------------ */
/* Cell labels: src/FluidSimulator.cpp:267-305.  pmask1 = particle cells dilated 1.      */
/* The weights are whatever the weight grids hold when the function is called (the       */
/* reference calls it before compute_face_fractions, FluidSimulator.cpp:486-488).        */
/* Rows are numbered in ascending linear id (the reference's order is OpenVDB's          */
/* iteration order; only the SET of labelled cells is compared).                         */
ORC_EXPORT int32_t orc_index_cells(const orc_grid *g, const uint8_t *pmask1, const float *phi_fluid,
                                   const float *phi_solid, const float *wu, const float *wv, const float *ww,
                                   uint8_t *label, int32_t *index) {
    const int64_t B = ncell(g);
    int32_t n = 0;
    for (int64_t c = 0; c < B; ++c) {
        label[c] = 0;
        index[c] = -1;
        if (!pmask1[c]) continue;
        const float uf = 1 - wu[c], vf = 1 - wv[c], wf = 1 - ww[c];
        if (phi_fluid[c] < 0
            || (((0 < uf && uf < 1) || (0 < vf && vf < 1) || (0 < wf && wf < 1)) && phi_solid[c] <= 0)) {
            label[c] = 1;
            index[c] = n++;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------ */
/* Density: src/FluidSimulator.cpp:625-652 + src/Functors.cpp:216-266.                  */
/* band2 = particle cells dilated 2.  vs3 = pow((double)dx, 3).  The redistribution of   */
/* over-dense cells (Functors.cpp:277-322) is restated separately (orc_redistribute);    */
/* n_over returns how many cells exceed 1.5*rest_density so that a                        */
/* harness can tell whether the reference would have redistributed (orc_redistribute).    */
ORC_EXPORT int64_t orc_density(const orc_grid *g, int64_t n, const double *pos, const float *mass,
                               const int32_t *cell_start, const int32_t *perm, const uint8_t *band2,
                               const float *phi, float rest_density, float *density) {
    (void)n;
    const float bg = g->phi_fluid_bg;
    const double vs3 = pow((double)g->dx, 3);
    int64_t n_over = 0;
#pragma omp parallel reduction(+ : n_over)
    {
    orc_cand *buf = NULL;
    int32_t cap = 0;
#pragma omp for schedule(dynamic, 1)
    for (int k = 0; k < g->nz; ++k)
        for (int j = 0; j < g->ny; ++j)
            for (int i = 0; i < g->nx; ++i) {
                const int64_t c = lid(g, i, j, k);
                if (!band2[c]) { density[c] = 0.f; continue; }
                const float phi_c = phi[c];
                const float fluid_weight = frac2(getf(g, phi, i - 1, j, k, bg), phi_c)
                                           + frac2(getf(g, phi, i, j - 1, k, bg), phi_c)
                                           + frac2(getf(g, phi, i, j, k - 1, bg), phi_c)
                                           + frac2(getf(g, phi, i + 1, j, k, bg), phi_c)
                                           + frac2(getf(g, phi, i, j + 1, k, bg), phi_c)
                                           + frac2(getf(g, phi, i, j, k + 1, bg), phi_c);
                float dens = 0.f;
                const double cx = (double)(g->ox + i), cy = (double)(g->oy + j), cz = (double)(g->oz + k);
                for (int kk = k - 1; kk <= k + 1; ++kk)
                    for (int jj = j - 1; jj <= j + 1; ++jj)
                        {
                            const int32_t m = row_candidates(g, pos, cell_start, perm, i, jj, kk, &buf, &cap);
                            for (int32_t s = 0; s < m; ++s) {
                                const int64_t p = buf[s].p;
                                double d[3] = { pos[3 * p] - cx, pos[3 * p + 1] - cy, pos[3 * p + 2] - cz };
                                if (d[0] < -1. || d[0] > 1. || d[1] < -1. || d[1] > 1. || d[2] < -1. || d[2] > 1.)
                                    continue;
                                const double weight = orc_kernel(d);
                                if (weight > 0 && fluid_weight > 0) {
                                    /* density += grid value (0.f) + weight*mass / (fluid_weight/6 * vs^3) */
                                    const double add =
                                        (double)0.f + weight * (double)mass[p] / ((double)(fluid_weight / 6) * vs3);
                                    dens = (float)((double)dens + add);
                                }
                            }
                        }
                if (0 < fluid_weight && fluid_weight < 6) dens = dens > rest_density ? dens : rest_density;
                density[c] = dens;
                if ((double)dens > 1.5 * (double)rest_density) ++n_over;
            }
    free(buf);
    }
    return n_over;
}

/* ------------------------------------------------------------------------------------ */
/* Redistribution of the particles of over-dense cells: src/Functors.cpp:268-322.  The     */
/* reference draws from a std::random_device-seeded global mt19937 inside a TBB loop        */
/* (src/util.cpp:5), so not even two runs of it agree; this restatement keeps the algorithm */
/* and replaces the generator by the stateless hash of headers/util.h:259-266 (randhash),   */
/* keyed on (seed, cell, draw number), so that the result depends on nothing but its inputs */
/* (parity unpinned, necessarily).  Per cell whose density exceeds 1.5 rest density:        */
/*   n particles of the cell (:269-272, :285; the atlas box [c-1/2, c+1/2] is restated as   */
/*   "binned in cell c"), f = floor(pow(n, 1./3.)) as written at :288 -- the largest integer */
/*   with f^3 <= n, except that 1./3. is below a third, so a perfect cube n = k^3 with       */
/*   k >= 4 gives k - 1 (64 particles: a 3^3 lattice) --, a shuffled list of                 */
/*   the f^3 sub-cells (:292-296; std::shuffle restated as Fisher-Yates from the top), and  */
/*   the m-th particle in slot order goes to sub-cell list[m % f^3] at a uniform random      */
/*   offset: corner + (i,j,k)/f + U/f with linearTo3d's index order (:305-315).             */
/* All densities are taken from the field as computed before any particle moved.           */
static inline uint32_t randhash_u(uint32_t seed) { /* headers/util.h:259-266 */
    uint32_t i = (seed ^ 0xA3C59AC3u) * 2654435769u;
    i ^= (i >> 16);
    i *= 2654435769u;
    i ^= (i >> 16);
    i *= 2654435769u;
    return i;
}
static inline uint32_t redist_base(uint32_t seed, int64_t cell) { return randhash_u(randhash_u(seed) ^ (uint32_t)cell); }
static inline double redist_uniform(uint32_t base, uint32_t q) { /* 53-bit uniform in [0, 1) from two hashes */
    const uint32_t hi = randhash_u(base + 0x10000000u + 2u * q) >> 5, lo = randhash_u(base + 0x10000000u + 2u * q + 1u) >> 6;
    return ((double)hi * 67108864.0 + (double)lo) * (1.0 / 9007199254740992.0);
}
ORC_EXPORT int64_t orc_redistribute(const orc_grid *g, int64_t n, double *pos, const int32_t *cell_start, const int32_t *perm,
                                    const float *density, float rest_density, uint32_t seed, int64_t *n_moved) {
    (void)n;
    const int64_t B = ncell(g);
    int64_t cells = 0, moved = 0;
    int32_t *list = NULL;
    int32_t cap = 0;
    for (int64_t c = 0; c < B; ++c) {
        if (!((double)density[c] > 1.5 * (double)rest_density)) continue;
        const int32_t s0 = cell_start[c], m = cell_start[c + 1] - s0;
        if (m == 0) continue;
        const int f = (int)floor(pow((double)m, 1. / 3.));
        const int32_t sub = f * f * f;
        if (sub > cap) { cap = sub; list = (int32_t *)realloc(list, (size_t)cap * sizeof(int32_t)); }
        for (int32_t a = 0; a < sub; ++a) list[a] = a;
        const uint32_t base = redist_base(seed, c);
        for (int32_t a = sub - 1; a >= 1; --a) {
            const int32_t b = (int32_t)(randhash_u(base + (uint32_t)a) % (uint32_t)(a + 1));
            const int32_t t = list[a]; list[a] = list[b]; list[b] = t;
        }
        const int k = (int)(c / ((int64_t)g->nx * g->ny)), j = (int)((c / g->nx) % g->ny), i = (int)(c % g->nx);
        const double corner[3] = { (double)(g->ox + i) - 0.5, (double)(g->oy + j) - 0.5, (double)(g->oz + k) - 0.5 };
        for (int32_t q = 0; q < m; ++q) {
            const int32_t idx = list[q % sub];
            /* linearTo3d(idx, f, f, i, j, k): headers/util.h:451-456 */
            const int sc[3] = { idx / (f * f), (idx / f) % f, idx % f };
            const int64_t p = perm[s0 + q];
            for (int a = 0; a < 3; ++a)
                pos[3 * p + a] = (corner[a] + (double)sc[a] / (double)f) + redist_uniform(base, 3u * (uint32_t)q + (uint32_t)a) / (double)f;
        }
        ++cells;
        moved += m;
    }
    free(list);
    if (n_moved) *n_moved = moved;
    return cells;
}

/* ------------------------------------------------------------------------------------ */
/* System assembly: src/FluidSimulator.cpp:654-806 (kind 0, divergence constraint) and   */
/* :832-990 (kind 1, density constraint).  Output is the 7-point stencil of every        */
/* labelled cell on the dense box: off[f][c] = A(c, neighbour f) for the faces           */
/* f = 0:-x 1:+x 2:-y 3:+y 4:-z 5:+z (0 where the reference inserts nothing), diag[c],   */
/* rhs[c]; zero for unlabelled cells.  kind 0 includes the dx^2 scaling of :805-806.     */
ORC_EXPORT void orc_build_system(const orc_grid *g, int kind, float dt, float rho, const uint8_t *label,
                                 const float *phi_fluid, const float *phi_solid, const float *wu, const float *wv,
                                 const float *ww, const double *u, const double *v, const double *w,
                                 const float *density, double *off, double *diag, double *rhs) {
    const int64_t B = ncell(g);
    const float vs = g->dx;
    const float scale = kind == 0 ? dt / (rho * vs) : dt / (rho * (vs * vs));
    const float fbg = g->phi_fluid_bg, sbg = g->phi_solid_bg;
    static const int e[6][3] = { { -1, 0, 0 }, { 1, 0, 0 }, { 0, -1, 0 }, { 0, 1, 0 }, { 0, 0, -1 }, { 0, 0, 1 } };
#pragma omp parallel for schedule(static)
    for (int k = 0; k < g->nz; ++k)
        for (int j = 0; j < g->ny; ++j)
            for (int i = 0; i < g->nx; ++i) {
                const int64_t c = lid(g, i, j, k);
                for (int f = 0; f < 6; ++f) off[(int64_t)f * B + c] = 0.0;
                diag[c] = 0.0;
                rhs[c] = 0.0;
                if (!label[c]) continue;
                const float phi_c = phi_fluid[c];
                double diagonal = 0, r = 0;
                float wsum = 0;
                for (int f = 0; f < 6; ++f) {
                    const int ni = i + e[f][0], nj = j + e[f][1], nk = k + e[f][2];
                    const float phi_n = getf(g, phi_fluid, ni, nj, nk, fbg);
                    const float sphi_n = getf(g, phi_solid, ni, nj, nk, sbg);
                    /* low faces read the weight stored at the cell, high faces at the neighbour */
                    const float *wa = f < 2 ? wu : (f < 4 ? wv : ww);
                    const float wval = (f & 1) ? getf(g, wa, ni, nj, nk, 0.f) : wa[c];
                    const float weight = 1 - wval;
                    const double term = (double)(weight * scale);
                    if (phi_n < 0) {
                        off[(int64_t)f * B + c] = -term;
                        diagonal += term;
                    } else if (sphi_n > 0) {
                        const float th = (f & 1) ? frac2(phi_c, phi_n) : frac2(phi_n, phi_c);
                        const double theta = (double)(0.001f > th ? 0.001f : th);
                        diagonal += term * ((double)1.f / theta);
                    }
                    if (kind == 0) {
                        const double *va = f < 2 ? u : (f < 4 ? v : w);
                        if (f & 1) r -= (double)weight * getd(g, va, ni, nj, nk);
                        else r += (double)weight * va[c];
                    } else {
                        wsum = wsum + weight;
                    }
                }
                if (kind == 0) {
                    const double s2 = (double)(vs * vs);
                    for (int f = 0; f < 6; ++f) off[(int64_t)f * B + c] *= s2;
                    diag[c] = diagonal * s2;
                    rhs[c] = r * s2;
                } else {
                    diag[c] = diagonal;
                    float rr = -1 / dt * (1 - clamp_f(density[c] / rho, 0.5f, 1.5f));
                    if (wsum == 0) rr = 0;
                    rhs[c] = (double)rr;
                }
            }
}

/* Pack the stencil into CSR with ascending columns.  An off-diagonal whose neighbour has  */
/* no row is dropped (the reference would index the matrix with -1 there; it cannot        */
/* happen when phi_fluid comes from the particles).  Returns nnz. */
ORC_EXPORT int64_t orc_stencil_to_csr(const orc_grid *g, const int32_t *index, const double *off,
                                      const double *diag, const double *rhs, int32_t *rowptr, int32_t *col,
                                      double *val, double *rhs_out) {
    const int64_t B = ncell(g);
    /* ascending linear id of the neighbours: -z, -y, -x, diag, +x, +y, +z */
    static const int order[7] = { 4, 2, 0, -1, 1, 3, 5 };
    static const int e[6][3] = { { -1, 0, 0 }, { 1, 0, 0 }, { 0, -1, 0 }, { 0, 1, 0 }, { 0, 0, -1 }, { 0, 0, 1 } };
    int64_t nnz = 0;
    int32_t row = 0;
    for (int k = 0; k < g->nz; ++k)
        for (int j = 0; j < g->ny; ++j)
            for (int i = 0; i < g->nx; ++i) {
                const int64_t c = lid(g, i, j, k);
                if (index[c] < 0) continue;
                rowptr[row] = (int32_t)nnz;
                for (int q = 0; q < 7; ++q) {
                    const int f = order[q];
                    if (f < 0) {
                        col[nnz] = index[c];
                        val[nnz++] = diag[c];
                    } else {
                        const double a = off[(int64_t)f * B + c];
                        if (a == 0.0) continue;
                        const int ni = i + e[f][0], nj = j + e[f][1], nk = k + e[f][2];
                        if (!inside(g, ni, nj, nk)) continue;
                        const int32_t nc = index[lid(g, ni, nj, nk)];
                        if (nc < 0) continue;
                        col[nnz] = nc;
                        val[nnz++] = a;
                    }
                }
                rhs_out[row] = rhs[c];
                ++row;
            }
    rowptr[row] = (int32_t)nnz;
    return nnz;
}

/* Rigid-body coupling rows: src/SolidObject.cpp:159-185.  J is 6 x n (row-major), built  */
/* from the object's own face fractions; com is the centre of mass in index space.        */
ORC_EXPORT void orc_build_J(const orc_grid *g, const int32_t *index, const float *owu, const float *owv,
                            const float *oww, const double com[3], int32_t n, double *J) {
    const float vs = g->dx;
    const float vs2 = vs * vs;
    for (int k = 0; k < g->nz; ++k)
        for (int j = 0; j < g->ny; ++j)
            for (int i = 0; i < g->nx; ++i) {
                const int64_t c = lid(g, i, j, k);
                const int32_t idx = index[c];
                if (idx < 0) continue;
                const float du = getf(g, owu, i + 1, j, k, 0.f) - owu[c];
                const float dv = getf(g, owv, i, j + 1, k, 0.f) - owv[c];
                const float dw = getf(g, oww, i, j, k + 1, 0.f) - oww[c];
                const double rx = (double)(g->ox + i) - com[0];
                const double ry = (double)(g->oy + j) - com[1];
                const double rz = (double)(g->oz + k) - com[2];
                J[0 * (int64_t)n + idx] = (double)(vs2 * du);
                J[1 * (int64_t)n + idx] = (double)(vs2 * dv);
                J[2 * (int64_t)n + idx] = (double)(vs2 * dw);
                J[3 * (int64_t)n + idx] = (double)vs2 * ((double)(-vs) * rz * (double)dv + (double)vs * ry * (double)dw);
                J[4 * (int64_t)n + idx] = (double)vs2 * ((double)(-vs) * rx * (double)dw + (double)vs * rz * (double)du);
                J[5 * (int64_t)n + idx] = (double)vs2 * ((double)(-vs) * ry * (double)du + (double)vs * rx * (double)dv);
            }
}

/* Sphere inverse mass matrix diagonal: src/SolidObject.cpp:24-39 (4/3 is integer division). */
ORC_EXPORT void orc_sphere_minv(float radius, double density, double minv[6]) {
    float mass = (float)(4 / 3 * 3.14 * pow((double)radius, 3) * density);
    const double inertia = (double)(2 * mass * (radius * radius) / 5);
    minv[0] = minv[1] = minv[2] = 1.0 / (double)mass;
    minv[3] = minv[4] = minv[5] = 1.0 / inertia;
}

/* ------------------------------------------------------------------------------------ */
/* Linear solvers on CSR.                                                               */

/* y = A x + dt * J^T (Minv (J x))   (the product of FluidSimulator.cpp:813, applied      */
/* factor by factor instead of forming the dense block) */
static void apply_op(int32_t n, const int32_t *rowptr, const int32_t *col, const double *val, const double *J,
                     const double *minv, double dt, const double *x, double *y) {
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < n; ++i) {
        double s = 0;
        for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) s += val[q] * x[col[q]];
        y[i] = s;
    }
    if (J) {
        double jx[6];
        for (int a = 0; a < 6; ++a) {
            double s = 0;
            for (int32_t i = 0; i < n; ++i) s += J[(int64_t)a * n + i] * x[i];
            jx[a] = dt * minv[a] * s;
        }
#pragma omp parallel for schedule(static)
        for (int32_t i = 0; i < n; ++i) {
            double s = 0;
            for (int a = 0; a < 6; ++a) s += J[(int64_t)a * n + i] * jx[a];
            y[i] += s;
        }
    }
}
ORC_EXPORT void orc_apply_operator(int32_t n, const int32_t *rowptr, const int32_t *col, const double *val,
                                   const double *J, const double *minv, double dt, const double *x, double *y) {
    apply_op(n, rowptr, col, val, J, minv, dt, x, y);
}

static double dot(int32_t n, const double *a, const double *b) {
    double s = 0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (int32_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* Eigen 3.4.0 ConjugateGradient with DiagonalPreconditioner as the reference runs it     */
/* (src/FluidSimulator.cpp:659-660, 822-823, 837-838, 1006-1007): x0 = 0, stop when       */
/* |r|^2 < tol^2 |b|^2, at most max_iter (Eigen default 2n) iterations, inverse diagonal  */
/* 1 where the diagonal is 0.  Restates Eigen/src/IterativeLinearSolvers/                 */
/* ConjugateGradient.h (not in the reference tree): parity unpinned.                      */
/* drop_null_rows != 0: the right-hand side of a row whose diagonal is exactly zero (its whole     */
/* stencil is then zero) is taken as zero; such a row makes the system inconsistent and the    */
/* iteration above diverge, while every other unknown's exact solution does not depend on it   */
/* (headers/pcgsolver skips the same rows, pcg_solver.h:121).  0 = as the reference runs it.   */
/* Returns the iteration count; *err = sqrt(|r|^2/|b|^2). */
ORC_EXPORT int32_t orc_cg_jacobi(int32_t n, const int32_t *rowptr, const int32_t *col, const double *val,
                                 const double *J, const double *minv, double dt, const double *rhs_in, double tol,
                                 int32_t max_iter, int drop_null_rows, double *x, double *err) {
    double *rhs = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *invd = (double *)malloc(sizeof(double) * (size_t)n);
    double *r = (double *)malloc(sizeof(double) * (size_t)n);
    double *p = (double *)malloc(sizeof(double) * (size_t)n);
    double *z = (double *)malloc(sizeof(double) * (size_t)n);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        double d = 0;
        for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q)
            if (col[q] == i) d += val[q];
        if (J)
            for (int a = 0; a < 6; ++a) d += dt * minv[a] * J[(int64_t)a * n + i] * J[(int64_t)a * n + i];
        invd[i] = d != 0 ? 1.0 / d : 1.0;
        rhs[i] = (drop_null_rows && d == 0) ? 0.0 : rhs_in[i];
        x[i] = 0;
        r[i] = rhs[i];
    }
    int32_t it = 0;
    const double rhsNorm2 = dot(n, rhs, rhs);
    double residualNorm2 = 0;
    if (rhsNorm2 == 0) {
        *err = 0;
        goto done;
    }
    {
        const double considerAsZero = 2.2250738585072014e-308;
        double threshold = tol * tol * rhsNorm2;
        if (threshold < considerAsZero) threshold = considerAsZero;
        residualNorm2 = dot(n, r, r);
        if (residualNorm2 < threshold) {
            *err = sqrt(residualNorm2 / rhsNorm2);
            goto done;
        }
        for (int32_t i = 0; i < n; ++i) p[i] = invd[i] * r[i];
        double absNew = dot(n, r, p);
        while (it < max_iter) {
            apply_op(n, rowptr, col, val, J, minv, dt, p, tmp);
            const double alpha = absNew / dot(n, p, tmp);
#pragma omp parallel for schedule(static)
            for (int32_t i = 0; i < n; ++i) {
                x[i] += alpha * p[i];
                r[i] -= alpha * tmp[i];
            }
            residualNorm2 = dot(n, r, r);
            if (residualNorm2 < threshold) break;
#pragma omp parallel for schedule(static)
            for (int32_t i = 0; i < n; ++i) z[i] = invd[i] * r[i];
            const double absOld = absNew;
            absNew = dot(n, r, z);
            const double beta = absNew / absOld;
#pragma omp parallel for schedule(static)
            for (int32_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
            ++it;
        }
        *err = sqrt(residualNorm2 / rhsNorm2);
    }
done:
    free(invd); free(r); free(p); free(z); free(tmp); free(rhs);
    return it;
}

/* NOT a restatement of reference code: the same system A p = b solved by block elimination.  */
/* A row of a labelled cell with phi_fluid >= 0 (dep[i] != 0) refers to liquid rows, but no row  */
/* refers to it, so A = [L 0; C D] with L symmetric and D diagonal.  CG (the iteration above)    */
/* runs on L; the dependent rows follow by substitution.  Where the as-run iteration converges   */
/* both give the same p; where it cannot (non-symmetric matrix), this is the checker.  A rigid    */
/* body's J must vanish on the dependent rows (the caller checks).  Returns the CG iterations.  */
ORC_EXPORT int32_t orc_cg_eliminated(int32_t n, const int32_t *rowptr, const int32_t *col, const double *val,
                                     const uint8_t *dep, const double *J, const double *minv, double dt,
                                     const double *rhs, double tol, int32_t max_iter, int drop_null_rows, double *x,
                                     double *err) {
    int32_t *map = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int32_t m = 0;
    for (int32_t i = 0; i < n; ++i) map[i] = dep[i] ? -1 : m++;
    int32_t *rp = (int32_t *)malloc(sizeof(int32_t) * ((size_t)m + 1));
    int32_t *cc = (int32_t *)malloc(sizeof(int32_t) * (size_t)(rowptr[n] > 0 ? rowptr[n] : 1));
    double *vv = (double *)malloc(sizeof(double) * (size_t)(rowptr[n] > 0 ? rowptr[n] : 1));
    double *bb = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    double *xx = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    double *JJ = J ? (double *)malloc(sizeof(double) * 6 * (size_t)(m > 0 ? m : 1)) : NULL;
    int32_t nnz = 0;
    for (int32_t i = 0; i < n; ++i) {
        if (dep[i]) continue;
        rp[map[i]] = nnz;
        for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
            if (map[col[q]] < 0) continue; /* cannot happen: a liquid row refers to liquid rows only */
            cc[nnz] = map[col[q]];
            vv[nnz++] = val[q];
        }
        bb[map[i]] = rhs[i];
        if (J)
            for (int a = 0; a < 6; ++a) JJ[(int64_t)a * m + map[i]] = J[(int64_t)a * n + i];
    }
    rp[m] = nnz;
    int32_t it = 0;
    *err = 0;
    if (m > 0) it = orc_cg_jacobi(m, rp, cc, vv, JJ, minv, dt, bb, tol, max_iter, drop_null_rows, xx, err);
    for (int32_t i = 0; i < n; ++i)
        if (!dep[i]) x[i] = xx[map[i]];
    for (int32_t i = 0; i < n; ++i) {
        if (!dep[i]) continue;
        double d = 0, s = 0;
        for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
            if (col[q] == i) d += val[q];
            else s += val[q] * x[col[q]];
        }
        x[i] = d != 0 ? (rhs[i] - s) / d : 0.0;
    }
    free(map); free(rp); free(cc); free(vv); free(bb); free(xx); free(JJ);
    return it;
}

/* headers/pcgsolver/pcg_solver.h:80-182: modified incomplete Cholesky level 0 of the     */
/* lower triangle, columns stored CSC.  The matrix is CSR with ascending columns and is    */
/* assumed structurally symmetric, as the header assumes.                                 */
typedef struct {
    int32_t n;
    double *invdiag, *adiag, *value;
    int32_t *rowindex, *colstart;
} mic0_factor;

static void mic0_build(int32_t n, const int32_t *rowptr, const int32_t *col, const double *val, mic0_factor *F,
                       double modification_parameter, double min_diagonal_ratio) {
    F->n = n;
    F->invdiag = (double *)calloc((size_t)n, sizeof(double));
    F->adiag = (double *)calloc((size_t)n, sizeof(double));
    F->colstart = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    int64_t cap = rowptr[n];
    F->value = (double *)malloc(sizeof(double) * (size_t)(cap > 0 ? cap : 1));
    F->rowindex = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cap > 0 ? cap : 1));
    int32_t cnt = 0;
    for (int32_t i = 0; i < n; ++i) {
        F->colstart[i] = cnt;
        for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
            if (col[q] > i) {
                F->rowindex[cnt] = col[q];
                F->value[cnt++] = val[q];
            } else if (col[q] == i) {
                F->invdiag[i] = F->adiag[i] = val[q];
            }
        }
    }
    F->colstart[n] = cnt;
    for (int32_t k = 0; k < n; ++k) {
        if (F->adiag[k] == 0) continue;
        if (F->invdiag[k] < min_diagonal_ratio * F->adiag[k]) F->invdiag[k] = 1 / sqrt(F->adiag[k]);
        else F->invdiag[k] = 1 / sqrt(F->invdiag[k]);
        for (int32_t p = F->colstart[k]; p < F->colstart[k + 1]; ++p) F->value[p] *= F->invdiag[k];
        for (int32_t p = F->colstart[k]; p < F->colstart[k + 1]; ++p) {
            const int32_t j = F->rowindex[p];
            const double multiplier = F->value[p];
            double missing = 0;
            int32_t a = F->colstart[k];
            int32_t b = 0;
            const int32_t jn = rowptr[j + 1] - rowptr[j];
            const int32_t *jidx = col + rowptr[j];
            while (a < F->colstart[k + 1] && F->rowindex[a] < j) {
                while (b < jn) {
                    if (jidx[b] < F->rowindex[a]) ++b;
                    else if (jidx[b] == F->rowindex[a]) break;
                    else {
                        missing += F->value[a];
                        break;
                    }
                }
                ++a;
            }
            if (a < F->colstart[k + 1] && F->rowindex[a] == j) F->invdiag[j] -= multiplier * F->value[a];
            ++a;
            b = F->colstart[j];
            while (a < F->colstart[k + 1] && b < F->colstart[j + 1]) {
                if (F->rowindex[b] < F->rowindex[a]) ++b;
                else if (F->rowindex[b] == F->rowindex[a]) {
                    F->value[b] -= multiplier * F->value[a];
                    ++a;
                    ++b;
                } else {
                    missing += F->value[a];
                    ++a;
                }
            }
            while (a < F->colstart[k + 1]) {
                missing += F->value[a];
                ++a;
            }
            F->invdiag[j] -= modification_parameter * multiplier * missing;
        }
    }
}
static void mic0_free(mic0_factor *F) {
    free(F->invdiag); free(F->adiag); free(F->value); free(F->rowindex); free(F->colstart);
}
/* pcg_solver.h:188-214 */
static void mic0_apply(const mic0_factor *F, const double *x, double *result) {
    const int32_t n = F->n;
    memcpy(result, x, sizeof(double) * (size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        result[i] *= F->invdiag[i];
        for (int32_t j = F->colstart[i]; j < F->colstart[i + 1]; ++j)
            result[F->rowindex[j]] -= F->value[j] * result[i];
    }
    int32_t i = n;
    do {
        --i;
        for (int32_t j = F->colstart[i]; j < F->colstart[i + 1]; ++j) result[i] -= F->value[j] * result[F->rowindex[j]];
        result[i] *= F->invdiag[i];
    } while (i != 0);
}
static double abs_max(int32_t n, const double *x) { /* blas_wrapper.h:24-42 */
    double m = 0;
    for (int32_t i = 0; i < n; ++i)
        if (fabs(x[i]) > m) m = fabs(x[i]);
    return m;
}
static double dot_serial(int32_t n, const double *a, const double *b) { /* blas_wrapper.h:12-20 */
    double s = 0;
    for (int32_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* headers/pcgsolver/pcg_solver.h:234-275 (PCGSolver<double>::solve).  The factor always   */
/* uses the 0.97 / 0.25 defaults (pcg_solver.h:290-292).  Returns 1 on convergence.        */
ORC_EXPORT int orc_pcg_mic0(int32_t n, const int32_t *rowptr, const int32_t *col, const double *val,
                            const double *rhs, double tolerance_factor, int32_t max_iterations, double *result,
                            double *residual_out, int32_t *iterations_out, double *invdiag_out) {
    if (tolerance_factor < 1e-30) tolerance_factor = 1e-30;
    double *s = (double *)malloc(sizeof(double) * (size_t)n);
    double *z = (double *)malloc(sizeof(double) * (size_t)n);
    double *r = (double *)malloc(sizeof(double) * (size_t)n);
    int ok = 0;
    mic0_factor F;
    memset(&F, 0, sizeof F);
    for (int32_t i = 0; i < n; ++i) { result[i] = 0; r[i] = rhs[i]; }
    *residual_out = abs_max(n, r);
    if (*residual_out == 0) {
        *iterations_out = 0;
        free(s); free(z); free(r);
        return 1;
    }
    const double tol = tolerance_factor * *residual_out;
    mic0_build(n, rowptr, col, val, &F, 0.97, 0.25);
    if (invdiag_out) memcpy(invdiag_out, F.invdiag, sizeof(double) * (size_t)n);
    mic0_apply(&F, r, z);
    double rho = dot_serial(n, z, r);
    if (rho == 0 || rho != rho) {
        *iterations_out = 0;
        goto done;
    }
    memcpy(s, z, sizeof(double) * (size_t)n);
    int32_t iteration;
    for (iteration = 0; iteration < max_iterations; ++iteration) {
        for (int32_t i = 0; i < n; ++i) { /* multiply(fixed_matrix, s, z) */
            double acc = 0;
            for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) acc += val[q] * s[col[q]];
            z[i] = acc;
        }
        const double alpha = rho / dot_serial(n, s, z);
        for (int32_t i = 0; i < n; ++i) result[i] += alpha * s[i];
        for (int32_t i = 0; i < n; ++i) r[i] += -alpha * z[i];
        *residual_out = abs_max(n, r);
        if (*residual_out <= tol) {
            *iterations_out = iteration + 1;
            ok = 1;
            goto done;
        }
        mic0_apply(&F, r, z);
        const double rho_new = dot_serial(n, z, r);
        const double beta = rho_new / rho;
        for (int32_t i = 0; i < n; ++i) z[i] += beta * s[i]; /* add_scaled(beta, s, z); s.swap(z) */
        double *t = s; s = z; z = t;
        rho = rho_new;
    }
    *iterations_out = iteration;
done:
    mic0_free(&F);
    free(s); free(z); free(r);
    return ok;
}

/* ------------------------------------------------------------------------------------ */
/* Projections.                                                                         */

/* src/FluidSimulator.cpp:1017-1102: every cell of the box; pressure narrowed to float,   */
/* gradient arithmetic in float; valid masks cleared then set where a face is updated.    */
ORC_EXPORT void orc_project_velocity(const orc_grid *g, float dt, float rho, const int32_t *index, const double *p,
                                     const float *phi, const float *wu, const float *wv, const float *ww,
                                     const double *u, const double *v, const double *w, double *u_out,
                                     double *v_out, double *w_out, uint8_t *valid_u, uint8_t *valid_v,
                                     uint8_t *valid_w) {
    const float vs = g->dx, bg = g->phi_fluid_bg;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < g->nz; ++k)
        for (int j = 0; j < g->ny; ++j)
            for (int i = 0; i < g->nx; ++i) {
                const int64_t c = lid(g, i, j, k);
                const int32_t idx = index[c];
                const int32_t idx_i = inside(g, i - 1, j, k) ? index[lid(g, i - 1, j, k)] : -1;
                const int32_t idx_j = inside(g, i, j - 1, k) ? index[lid(g, i, j - 1, k)] : -1;
                const int32_t idx_k = inside(g, i, j, k - 1) ? index[lid(g, i, j, k - 1)] : -1;
                const float pc = idx >= 0 ? (float)p[idx] : 0;
                const float p_i = idx_i >= 0 ? (float)p[idx_i] : 0;
                const float p_j = idx_j >= 0 ? (float)p[idx_j] : 0;
                const float p_k = idx_k >= 0 ? (float)p[idx_k] : 0;
                const float phi_c = phi[c];
                const float phi_i = getf(g, phi, i - 1, j, k, bg);
                const float phi_j = getf(g, phi, i, j - 1, k, bg);
                const float phi_k = getf(g, phi, i, j, k - 1, bg);
                double nu = u[c], nv = v[c], nw = w[c];
                valid_u[c] = valid_v[c] = valid_w[c] = 0;
                if ((1 - wu[c]) > 0 && (phi_c < 0 || phi_i < 0)) {
                    float theta = 1;
                    if (phi_c >= 0 || phi_i >= 0) { float t = frac2(phi_i, phi_c); theta = 0.001f > t ? 0.001f : t; }
                    nu -= (double)(dt / rho * (pc - p_i) / vs / theta);
                    valid_u[c] = 1;
                }
                if ((1 - wv[c]) > 0 && (phi_c < 0 || phi_j < 0)) {
                    float theta = 1;
                    if (phi_c >= 0 || phi_j >= 0) { float t = frac2(phi_j, phi_c); theta = 0.001f > t ? 0.001f : t; }
                    nv -= (double)(dt / rho * (pc - p_j) / vs / theta);
                    valid_v[c] = 1;
                }
                if ((1 - ww[c]) > 0 && (phi_c < 0 || phi_k < 0)) {
                    float theta = 1;
                    if (phi_c >= 0 || phi_k >= 0) { float t = frac2(phi_k, phi_c); theta = 0.001f > t ? 0.001f : t; }
                    nw -= (double)(dt / rho * (pc - p_k) / vs / theta);
                    valid_w[c] = 1;
                }
                u_out[c] = nu; v_out[c] = nv; w_out[c] = nw;
            }
}

/* src/FluidSimulator.cpp:1095-1100: V += dt * Minv * (J p) */
ORC_EXPORT void orc_solid_velocity_update(int32_t n, const double *J, const double *minv, float dt, const double *p,
                                          double V[6]) {
    for (int a = 0; a < 6; ++a) {
        double s = 0;
        for (int32_t i = 0; i < n; ++i) s += J[(int64_t)a * n + i] * p[i];
        V[a] = V[a] + (double)dt * (minv[a] * s);
    }
}

/* src/FluidSimulator.cpp:1104-1178: displacement in index units for every cell of the box */
ORC_EXPORT void orc_project_displacement(const orc_grid *g, float dt, float rho, const int32_t *index,
                                         const double *p, const float *phi, const float *wu, const float *wv,
                                         const float *ww, double *dispx, double *dispy, double *dispz) {
    const float vs = g->dx, bg = g->phi_fluid_bg;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < g->nz; ++k)
        for (int j = 0; j < g->ny; ++j)
            for (int i = 0; i < g->nx; ++i) {
                const int64_t c = lid(g, i, j, k);
                const int32_t idx = index[c];
                const int32_t idx_i = inside(g, i - 1, j, k) ? index[lid(g, i - 1, j, k)] : -1;
                const int32_t idx_j = inside(g, i, j - 1, k) ? index[lid(g, i, j - 1, k)] : -1;
                const int32_t idx_k = inside(g, i, j, k - 1) ? index[lid(g, i, j, k - 1)] : -1;
                const float pc = idx >= 0 ? (float)p[idx] : 0;
                const float p_i = idx_i >= 0 ? (float)p[idx_i] : 0;
                const float p_j = idx_j >= 0 ? (float)p[idx_j] : 0;
                const float p_k = idx_k >= 0 ? (float)p[idx_k] : 0;
                const float phi_c = phi[c];
                const float phi_i = getf(g, phi, i - 1, j, k, bg);
                const float phi_j = getf(g, phi, i, j - 1, k, bg);
                const float phi_k = getf(g, phi, i, j, k - 1, bg);
                double d0 = 0, d1 = 0, d2 = 0;
                if ((1 - wu[c]) > 0 && (phi_c < 0 || phi_i < 0)) {
                    float theta = 1;
                    if (phi_c >= 0 || phi_i >= 0) { float t = frac2(phi_i, phi_c); theta = 0.001f > t ? 0.001f : t; }
                    d0 = (double)(-dt * dt / rho * (pc - p_i) / vs / theta);
                }
                if ((1 - wv[c]) > 0 && (phi_c < 0 || phi_j < 0)) {
                    float theta = 1;
                    if (phi_c >= 0 || phi_j >= 0) { float t = frac2(phi_j, phi_c); theta = 0.001f > t ? 0.001f : t; }
                    d1 = (double)(-dt * dt / rho * (pc - p_j) / vs / theta);
                }
                if ((1 - ww[c]) > 0 && (phi_c < 0 || phi_k < 0)) {
                    float theta = 1;
                    if (phi_c >= 0 || phi_k >= 0) { float t = frac2(phi_k, phi_c); theta = 0.001f > t ? 0.001f : t; }
                    d2 = (double)(-dt * dt / rho * (pc - p_k) / vs / theta);
                }
                dispx[c] = d0 / (double)vs;
                dispy[c] = d1 / (double)vs;
                dispz[c] = d2 / (double)vs;
            }
}

/* src/MacGrid.cpp:57-63 */
ORC_EXPORT void orc_update_diff(int64_t B, const double *front, const double *prev, double *diff) {
    for (int64_t c = 0; c < B; ++c) diff[c] = front[c] - prev[c];
}

/* ------------------------------------------------------------------------------------ */
/* Grid-side glue between P2G and the solves.                                           */

/* src/FluidSimulator.cpp:181-265, one sweep: Jacobi averaging of valid face neighbours    */
/* into non-valid faces, the sum kept in float (new_vel is a float, :212). */
ORC_EXPORT void orc_extrapolate_sweep(const orc_grid *g, const double *f_in, const uint8_t *valid_in, double *f_out,
                                      uint8_t *valid_out) {
    static const int e[6][3] = { { -1, 0, 0 }, { 0, -1, 0 }, { 0, 0, -1 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
#pragma omp parallel for schedule(static)
    for (int k = 0; k < g->nz; ++k)
        for (int j = 0; j < g->ny; ++j)
            for (int i = 0; i < g->nx; ++i) {
                const int64_t c = lid(g, i, j, k);
                f_out[c] = f_in[c];
                valid_out[c] = valid_in[c];
                if (valid_in[c]) continue;
                float new_vel = 0;
                int n_valid = 0;
                for (int q = 0; q < 6; ++q) {
                    const int ni = i + e[q][0], nj = j + e[q][1], nk = k + e[q][2];
                    if (!inside(g, ni, nj, nk)) continue;
                    const int64_t nb = lid(g, ni, nj, nk);
                    if (valid_in[nb]) {
                        new_vel = (float)((double)new_vel + f_in[nb]);
                        ++n_valid;
                    }
                }
                if (n_valid > 0) {
                    f_out[c] = (double)(new_vel / n_valid);
                    valid_out[c] = 1;
                }
            }
}

/* src/FluidSimulator.cpp:460-473: gravity on the y component of every active voxel */
ORC_EXPORT void orc_add_gravity(int64_t B, const uint8_t *active, double *v, float grav, float dt) {
    for (int64_t c = 0; c < B; ++c)
        if (active[c]) v[c] = v[c] - (double)(grav * dt);
}

/* src/FluidSimulator.cpp:448-458: compute_cfl.  The reference takes grid.velFront()->evalMinMax (OpenVDB 9, absent
 * here: parity unpinned) and, per rigid body, std::max(maxVector, Abs(velTranslational)).  On Vec3d neither is a
 * component-wise maximum in OpenVDB: evalMinMax keeps a value only when it exceeds the running maximum in ALL three
 * components (math::cwiseGreaterThan, visiting order unspecified and threaded), and std::max uses the tuple's
 * lexicographic operator<.  ASSUMED HERE, and what the device path computes: the component-wise maximum over the active
 * voxels (signed, no absolute value: a purely falling fluid yields the rest of the frame), and the component-wise maximum
 * with |V_translational| -- the one reading that does not depend on the order in which voxels are visited.  Types as
 * the reference: float voxel size, double maxima, float epsilon 10e-37 widened, float result. */
ORC_EXPORT float orc_compute_cfl(int64_t B, const uint8_t *active, const double *u, const double *v, const double *w, float voxel_size,
                                 const double *solid_translational, double max_out[3]) {
    double m[3] = { -1e300, -1e300, -1e300 };
    for (int64_t c = 0; c < B; ++c)
        if (active[c]) {
            if (u[c] > m[0]) m[0] = u[c];
            if (v[c] > m[1]) m[1] = v[c];
            if (w[c] > m[2]) m[2] = w[c];
        }
    for (int a = 0; a < 3; ++a) {
        if (m[a] == -1e300) m[a] = 0.0; /* no active voxel: evalMinMax returns zero */
        if (solid_translational && fabs(solid_translational[a]) > m[a]) m[a] = fabs(solid_translational[a]);
        if (max_out) max_out[a] = m[a];
    }
    const float epsilon = 10e-37f; /* src/FluidSimulator.cpp:32 */
    double top = m[0] > m[1] ? m[0] : m[1];
    top = top > m[2] ? top : m[2];
    return (float)((double)voxel_size / (top + (double)epsilon));
}

/* ------------------------------------------------------------------------------------ */
/* constrain_velocity: src/FluidSimulator.cpp:1180-1247 (container) and :1249-1321 (rigid  */
/* body).  On every face whose solid fraction is exactly 1 the sampled velocity loses its  */
/* component along the level set's normal (and, for the body, takes the body's normal      */
/* velocity instead).  Restates openvdb::tools::gradient (second-order central differences */
/* scaled to world units, float), BoxSampler on the Vec3f gradient, Vec3::normalize (multi- */
/* plies by the reciprocal length, leaves a vector shorter than 1e-7 alone) and            */
/* Vec3::projection (eps 1e-10 on the squared length): OpenVDB internals, parity unpinned. */
/* Reads (u,v,w), writes (ou,ov,ow); every other face is copied.                            */
static void level_set_gradient(const orc_grid *g, const float *phi, float bg, int i, int j, int k, float out[3]) {
    const double inv_dx = 1.0 / (double)g->dx;
    const float dxp = (getf(g, phi, i + 1, j, k, bg) - getf(g, phi, i - 1, j, k, bg)) * 0.5f;
    const float dyp = (getf(g, phi, i, j + 1, k, bg) - getf(g, phi, i, j - 1, k, bg)) * 0.5f;
    const float dzp = (getf(g, phi, i, j, k + 1, bg) - getf(g, phi, i, j, k - 1, bg)) * 0.5f;
    out[0] = (float)((double)dxp * inv_dx);
    out[1] = (float)((double)dyp * inv_dx);
    out[2] = (float)((double)dzp * inv_dx);
}
static double constrained_component(const orc_grid *g, const float *phi, float bg, const double *u, const double *v,
                                    const double *w, int i, int j, int k, int axis, int has_obj, const double *vt,
                                    const double *va) {
    /* face position: cell centre minus half a voxel along the axis */
    int lo[3] = { i, j, k };
    lo[axis] -= 1;
    float ga[3], gb[3], nrm[3];
    level_set_gradient(g, phi, bg, lo[0], lo[1], lo[2], ga);
    level_set_gradient(g, phi, bg, i, j, k, gb);
    for (int a = 0; a < 3; ++a) nrm[a] = lerp_f(ga[a], gb[a], 0.5);
    if (!(nrm[0] == 0.f && nrm[1] == 0.f && nrm[2] == 0.f)) {
        const float len = sqrtf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
        if (!(fabsf(len) <= 1e-7f)) {
            const float r = 1.0f / len;
            for (int a = 0; a < 3; ++a) nrm[a] = nrm[a] * r;
        }
    }
    double p[3] = { (double)(g->ox + i), (double)(g->oy + j), (double)(g->oz + k) };
    p[axis] -= 0.5;
    double vel[3];
    staggered_sample(g, u, v, w, p, vel);
    const double n[3] = { (double)nrm[0], (double)nrm[1], (double)nrm[2] };
    const double l = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    double proj = 0.0;
    if (!(fabs(l) <= 1e-10)) proj = n[axis] * ((vel[0] * n[0] + vel[1] * n[1] + vel[2] * n[2]) * (1.0 / l));
    double out = vel[axis] - proj;
    if (has_obj) {
        /* solidVel = vel_trans + voxel * (vel_ang x (coord - e/2)), src/FluidSimulator.cpp:1287 */
        const double vs = (double)g->dx;
        const double c[3] = { vs * va[1] * p[2] - vs * va[2] * p[1], vs * va[2] * p[0] - vs * va[0] * p[2],
                              vs * va[0] * p[1] - vs * va[1] * p[0] };
        const double sv[3] = { vt[0] + c[0], vt[1] + c[1], vt[2] + c[2] };
        out = out + (n[0] * sv[0] + n[1] * sv[1] + n[2] * sv[2]) * n[axis];
    }
    return out;
}
ORC_EXPORT void orc_constrain_velocity(const orc_grid *g, const float *phi, float phi_bg, const float *wu, const float *wv,
                                       const float *ww, const double *u, const double *v, const double *w, double *ou,
                                       double *ov, double *ow, int has_obj, const double *vel_trans, const double *vel_ang) {
#pragma omp parallel for schedule(static)
    for (int k = 0; k < g->nz; ++k)
        for (int j = 0; j < g->ny; ++j)
            for (int i = 0; i < g->nx; ++i) {
                const int64_t c = lid(g, i, j, k);
                ou[c] = wu[c] == 1.f ? constrained_component(g, phi, phi_bg, u, v, w, i, j, k, 0, has_obj, vel_trans, vel_ang) : u[c];
                ov[c] = wv[c] == 1.f ? constrained_component(g, phi, phi_bg, u, v, w, i, j, k, 1, has_obj, vel_trans, vel_ang) : v[c];
                ow[c] = ww[c] == 1.f ? constrained_component(g, phi, phi_bg, u, v, w, i, j, k, 2, has_obj, vel_trans, vel_ang) : w[c];
            }
}

/* ------------------------------------------------------------------------------------ */
/* Particle advection: src/FluidDomain.cpp:63-75 + src/ParticleSet.cpp:18-45.  The        */
/* reference calls OpenVDB (tools::PointAdvect order 3, tools::cpt + BoxSampler); neither */
/* is available, so this restates their published algorithms (parity unpinned):           */
/*   VelocityIntegrator::rungeKutta<3>: V0 = v(P), V1 = v(P + dt/2 V0),                   */
/*   V2 = v(P + dt (2 V1 - V0)), P += dt (V0 + 4 V1 + V2) / 6, velocities sampled with    */
/*   the staggered box sampler, positions kept in index space (world = dx (index + 1/2)); */
/*   a particle that ends where the combined solid level set (container, csgUnion = min   */
/*   with the rigid body) is <= 0 goes to the closest point of the surface; for a signed  */
/*   distance that point is P - phi grad(phi) / |grad(phi)|, the gradient taken by        */
/*   central differences of the sampled level set half a cell either side.                */
static float box_sample_f(const orc_grid *g, const float *f, float bg, double x, double y, double z) {
    const double fx = floor(x), fy = floor(y), fz = floor(z);
    const int i = (int)fx - g->ox, j = (int)fy - g->oy, k = (int)fz - g->oz;
    const double tu = x - fx, tv = y - fy, tw = z - fz;
    float d[2][2][2];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int c = 0; c < 2; ++c) d[a][b][c] = getf(g, f, i + a, j + b, k + c, bg);
    return lerp_f(lerp_f(lerp_f(d[0][0][0], d[0][0][1], tw), lerp_f(d[0][1][0], d[0][1][1], tw), tv),
                  lerp_f(lerp_f(d[1][0][0], d[1][0][1], tw), lerp_f(d[1][1][0], d[1][1][1], tw), tv), tu);
}
static float solid_sample(const orc_grid *g, const float *phi_s, float bg_s, const float *phi_o, float bg_o, double x, double y,
                          double z) {
    float v = box_sample_f(g, phi_s, bg_s, x, y, z);
    if (phi_o) {
        const float o = box_sample_f(g, phi_o, bg_o, x, y, z);
        v = o < v ? o : v;
    }
    return v;
}
ORC_EXPORT void orc_advect_particles(const orc_grid *g, int64_t n, double *pos, const double *u, const double *v, const double *w,
                                     const float *phi_solid, const float *phi_object, float phi_object_bg, float dt) {
    const double inv_dx = 1.0 / (double)g->dx;
    const double dt_over_dx = (double)dt * inv_dx;
    const float bg_s = g->phi_solid_bg;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        const double x = pos[3 * p], y = pos[3 * p + 1], z = pos[3 * p + 2];
        double a[3], b[3], c[3];
        const double P0[3] = { x, y, z };
        staggered_sample(g, u, v, w, P0, a);
        const double h = 0.5 * dt_over_dx;
        const double P1[3] = { x + h * a[0], y + h * a[1], z + h * a[2] };
        staggered_sample(g, u, v, w, P1, b);
        const double P2[3] = { x + dt_over_dx * (2.0 * b[0] - a[0]), y + dt_over_dx * (2.0 * b[1] - a[1]),
                               z + dt_over_dx * (2.0 * b[2] - a[2]) };
        staggered_sample(g, u, v, w, P2, c);
        double nx = x + dt_over_dx * (a[0] + 4.0 * b[0] + c[0]) / 6.0;
        double ny = y + dt_over_dx * (a[1] + 4.0 * b[1] + c[1]) / 6.0;
        double nz = z + dt_over_dx * (a[2] + 4.0 * b[2] + c[2]) / 6.0;
        const float phi = solid_sample(g, phi_solid, bg_s, phi_object, phi_object_bg, nx, ny, nz);
        if (phi <= 0.f) {
            const double gx = (double)solid_sample(g, phi_solid, bg_s, phi_object, phi_object_bg, nx + 0.5, ny, nz)
                              - (double)solid_sample(g, phi_solid, bg_s, phi_object, phi_object_bg, nx - 0.5, ny, nz);
            const double gy = (double)solid_sample(g, phi_solid, bg_s, phi_object, phi_object_bg, nx, ny + 0.5, nz)
                              - (double)solid_sample(g, phi_solid, bg_s, phi_object, phi_object_bg, nx, ny - 0.5, nz);
            const double gz = (double)solid_sample(g, phi_solid, bg_s, phi_object, phi_object_bg, nx, ny, nz + 0.5)
                              - (double)solid_sample(g, phi_solid, bg_s, phi_object, phi_object_bg, nx, ny, nz - 0.5);
            const double len = sqrt(gx * gx + gy * gy + gz * gz);
            if (len > 0.0) {
                const double d = -(double)phi * inv_dx / len;
                nx += d * gx; ny += d * gy; nz += d * gz;
            }
        }
        pos[3 * p] = nx; pos[3 * p + 1] = ny; pos[3 * p + 2] = nz;
    }
}

/* ------------------------------------------------------------------------------------ */
/* Fluid level set from the marker particles: src/FluidDomain.cpp:50-61 with               */
/* LevelSet::unionOfBalls (src/LevelSet.cpp:17-23) and IntersectSolidFluidLevelSets       */
/* (headers/LevelSet.h:83-96).  OpenVDB's ParticlesToLevelSet rasterises the spheres      */
/* onto the narrow band; restated as the exact union of balls (parity unpinned):          */
/* phi(c) = min_p |c - p| - radius in index units, clipped to +-half_width voxels, times  */
/* dx, as float; then, where the solid level set is <= 0, max(phi, -phi_solid).  A cell   */
/* with no particle within radius + half_width keeps +half_width dx.                      */
ORC_EXPORT void orc_fluid_levelset(const orc_grid *g, int64_t n, const double *pos, const int32_t *cell_start, const int32_t *perm,
                                   const float *phi_solid, float radius_world, float half_width_voxels, float *phi_fluid) {
    (void)n;
    const double radius = (double)radius_world / (double)g->dx, half_width = (double)half_width_voxels;
    const int reach = (int)ceil(radius + half_width);
#pragma omp parallel for schedule(dynamic, 1)
    for (int k = 0; k < g->nz; ++k)
        for (int j = 0; j < g->ny; ++j)
            for (int i = 0; i < g->nx; ++i) {
                const double cx = (double)(g->ox + i), cy = (double)(g->oy + j), cz = (double)(g->oz + k);
                double best2 = 1e300;
                for (int kk = k - reach; kk <= k + reach; ++kk)
                    for (int jj = j - reach; jj <= j + reach; ++jj) {
                        if (kk < 0 || kk >= g->nz || jj < 0 || jj >= g->ny) continue;
                        const int xa = i - reach > 0 ? i - reach : 0, xb = i + reach < g->nx - 1 ? i + reach : g->nx - 1;
                        const int64_t row = (int64_t)g->nx * ((int64_t)jj + (int64_t)g->ny * kk);
                        for (int32_t s = cell_start[row + xa]; s < cell_start[row + xb + 1]; ++s) {
                            const int64_t p = perm[s];
                            const double dx = cx - pos[3 * p], dy = cy - pos[3 * p + 1], dz = cz - pos[3 * p + 2];
                            const double d2 = dx * dx + dy * dy + dz * dz;
                            if (d2 < best2) best2 = d2;
                        }
                    }
                const int64_t c = lid(g, i, j, k);
                double d = half_width;
                if (best2 < 1e299) d = sqrt(best2) - radius;
                d = d < -half_width ? -half_width : (d > half_width ? half_width : d);
                float f = (float)(d * (double)g->dx);
                const float ps = phi_solid[c];
                if (ps <= 0) f = f > -ps ? f : -ps;
                phi_fluid[c] = f;
            }
}
