// flip_particles.hip -- particle-side kernels: counting-sort binning, occupancy bands, P2G gather,
// G2P, displacement sampling and the density field.
#include "flip_internal.h"

#include <cstdlib>

// ---------------------------------------------------------------------------------------------
// Binning.  cell(p) = floor(pos + 0.5) per axis (OpenVDB Coord::round of the index position).
// Outside particles go to the extra bin B.

// The particles arrive in the order of the previous sort and move less than a cell per substep, so neighbouring lanes
// mostly hold the same cell: a run of equal cells within a wave issues one atomic for the whole run instead of one per
// particle (same-address atomics serialise).  Returns the run's first lane and its length to every lane of the run.
__device__ __forceinline__ void d_wave_run(int32_t c, int &start, int &len) {
    const int lane = threadIdx.x & (FLIP_WAVE - 1);
    const int32_t prev = __shfl_up(c, 1, FLIP_WAVE);
    const bool head = lane == 0 || prev != c;
    const unsigned long long heads = __ballot(head);
    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull); // bits 0..lane
    start = 63 - __clzll((long long)(heads & upto));
    const unsigned long long above = heads & ~upto;
    const int end = above ? __ffsll((long long)above) - 1 : FLIP_WAVE;
    len = end - start;
}

__global__ void k_cell_ids(GridDims g, long long n, const double *__restrict__ x, const double *__restrict__ y,
                           const double *__restrict__ z, int32_t *__restrict__ cell_id, int32_t *__restrict__ counts) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int32_t c = -1;
    if (s < n) {
        const int i = (int)floor(x[s] + 0.5) - g.ox;
        const int j = (int)floor(y[s] + 0.5) - g.oy;
        const int k = (int)floor(z[s] + 0.5) - g.oz;
        c = d_inside(g, i, j, k) ? (int32_t)d_lid(g, i, j, k) : (int32_t)g.B;
        cell_id[s] = c;
    }
    int start, len;
    d_wave_run(c, start, len);
    // integer: the result does not depend on the order of arrival
    if (c >= 0 && (int)(threadIdx.x & (FLIP_WAVE - 1)) == start) atomicAdd(&counts[c], len);
}

__global__ void k_scatter_slots(long long n, const int32_t *__restrict__ cell_id, const int32_t *__restrict__ pid,
                                const int32_t *__restrict__ cell_start, int32_t *__restrict__ fill,
                                int32_t *__restrict__ slot_src, int32_t *__restrict__ slot_key) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t c = s < n ? cell_id[s] : -1;
    int start, len;
    d_wave_run(c, start, len);
    const int lane = threadIdx.x & (FLIP_WAVE - 1);
    int32_t base = 0;
    if (c >= 0 && lane == start) base = cell_start[c] + atomicAdd(&fill[c], len);
    base = __shfl(base, start, FLIP_WAVE);
    if (c < 0) return;
    const int32_t slot = base + (lane - start);
    slot_src[slot] = (int32_t)s;
    slot_key[slot] = pid[s];
}

struct ParticlePtrs {
    double *a[6];
    float *mass;
    int32_t *pid;
};
// the particles outside the box (the last bin): plain reorder; they take part in nothing until they return
__global__ void k_reorder_outside(long long n, const int32_t *__restrict__ first, const int32_t *__restrict__ slot_src,
                                  const int32_t *__restrict__ slot_key, ParticlePtrs src, ParticlePtrs dst, flip_ppos *__restrict__ pcell) {
    const long long s = (long long)*first + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int32_t o = slot_src[s];
    double v[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) { v[q] = src.a[q][o]; dst.a[q][s] = v[q]; }
    const float m = src.mass[o];
    dst.mass[s] = m;
    dst.pid[s] = slot_key[s];
    flip_ppos a;
    a.x = v[0]; a.y = v[1]; a.z = v[2]; a.mass = m; a.pad0 = 0.f;
    pcell[s] = a;
}

// ---------------------------------------------------------------------------------------------
// Gather order.  A cell c gathers the particles with |pos - c| <= 1 per axis (the atlas query box of
// src/Functors.cpp:21-24).  Cut every cell in two along each axis: the box of c is then exactly 4 x 4 x 4 half cells
// -- the upper half of the cells at c-1, both halves at c, the lower half at c+1 -- except for a particle at exactly
// c+1, whose weights are all zero (hat(1) = hat(1.5) = 0) and which the reference therefore skips as well.  So the
// gather kernels read a second copy of the particle records that is sorted by half cell: grid row (k, j), then z half,
// then y half, then 2*i + x half, then upload index.  The support of a cell is 16 contiguous runs of that copy
// (sub_start holds the run boundaries, 8 entries per cell), every record in them contributes, and no record outside
// them does: no support test, no rejected visit.  The primary SoA order (cell, upload index) is untouched; a grid row
// occupies the same slot range in both orders.
//
// "Upper half" uses the expression the gather of the cell above evaluates for the lower edge of its box,
// pos - (n+1) >= -1, so the classification agrees with that test to the last bit.  (The mirrored statement for the
// upper edge, pos - (n-1) < 1 for every particle of a lower half, can only fail for a coordinate within 2^-53 of an
// integer below 1 in magnitude, where the subtraction rounds; such a particle carries a weight of 2^-53.)
__device__ __forceinline__ int d_upper_half(double p, int n_index) { return (p - (double)(n_index + 1)) >= -1.0 ? 1 : 0; }

__device__ __forceinline__ int d_sub_index(const GridDims &g, double x, double y, double z, int ci, int j, int k) {
    const int hx = d_upper_half(x, g.ox + ci), hy = d_upper_half(y, g.oy + j), hz = d_upper_half(z, g.oz + k);
    return (hz * 2 + hy) * (2 * g.nx) + 2 * ci + hx;
}

// One workgroup per grid row (j, k) does the whole reordering of the row's particles.  The counting sort left every cell's
// particles in its slot range in the order of arrival (atomic ranks, which depend on timing); here every particle finds its
// final place by counting, among the particles of its cell, those with a smaller upload index -- that makes the order the
// stable one -- and, among those, the ones in its own half cell, which is its rank in the half-cell order.  The upload
// indices and half-cell numbers of the row are parked in LDS for those counts (a row too crowded for that reads them
// from global memory).  The workgroup gathers the row's particles from the previous order (slot_src), counts them per
// half cell in LDS, scans, writes the row's 8 nx entries of sub_start, and stores every particle three times -- SoA in the
// primary (cell, upload index) order, the position record of that order, and the position and velocity records in the
// half-cell order.
__global__ void __launch_bounds__(FLIP_BLOCK) k_reorder_pack(GridDims g, const int32_t *__restrict__ cell_start,
                                                            const int32_t *__restrict__ slot_src, const int32_t *__restrict__ slot_key,
                                                            ParticlePtrs src, ParticlePtrs dst, int32_t *__restrict__ sub_start,
                                                            int32_t *__restrict__ sub_slot, int32_t *__restrict__ tmp_q,
                                                            flip_ppos *__restrict__ pcell, flip_ppos *__restrict__ pm,
                                                            flip_pvel *__restrict__ pv, int qcap) {
    extern __shared__ __align__(16) int32_t lds_words[];
    int32_t *lds = lds_words; // [8 nx] counts, then exclusive offsets; [4] wave totals; [qcap] upload indices; [qcap / 2] half-cell numbers (16 bit)
    const int nsub = 8 * g.nx;
    const long long row = blockIdx.x;
    const long long c0 = row * g.nx;
    const int32_t r0 = cell_start[c0], r1 = cell_start[c0 + g.nx];
    int32_t *__restrict__ out = sub_start + row * nsub;
    const int tid = threadIdx.x;
    const bool last = row == (long long)gridDim.x - 1;
    if (r0 == r1) {
        for (int q = tid; q < nsub; q += FLIP_BLOCK) out[q] = r0;
        if (last && tid == 0) out[nsub] = r1;
        return;
    }
    for (int q = tid; q < nsub; q += FLIP_BLOCK) lds[q] = 0;
    int32_t *kbuf = lds + nsub + 4;
    uint16_t *qbuf = reinterpret_cast<uint16_t *>(kbuf + qcap);
    const bool in_lds = r1 - r0 <= qcap; // uniform over the workgroup
    __syncthreads();
    const int j = (int)(row % g.ny), k = (int)(row / g.ny);
    for (int32_t s = r0 + tid; s < r1; s += FLIP_BLOCK) {
        const int32_t o = slot_src[s];
        const double x = src.a[0][o], y = src.a[1][o], z = src.a[2][o];
        const int ci = (int)floor(x + 0.5) - g.ox;
        const int q = d_sub_index(g, x, y, z, ci, j, k);
        atomicAdd(&lds[q], 1); // integer: the totals do not depend on the order of arrival
        if (in_lds) { qbuf[s - r0] = (uint16_t)q; kbuf[s - r0] = slot_key[s]; }
        else tmp_q[s] = q;
    }
    __syncthreads();
    // exclusive scan of lds[0 .. nsub): a run of consecutive entries per thread, wave scan of the run totals
    const int items = (nsub + FLIP_BLOCK - 1) / FLIP_BLOCK;
    const int b0 = tid * items, b1 = b0 + items < nsub ? b0 + items : nsub;
    int32_t sum = 0;
    for (int q = b0; q < b1; ++q) sum += lds[q];
    const int lane = tid & 63, wave = tid >> 6;
    int32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t t = __shfl_up(incl, off, FLIP_WAVE);
        if (lane >= off) incl += t;
    }
    int32_t *wave_tot = lds + nsub;
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int32_t run = r0 + incl - sum;
    for (int w = 0; w < wave; ++w) run += wave_tot[w];
    for (int q = b0; q < b1; ++q) {
        const int32_t v = lds[q];
        lds[q] = run;
        run += v;
    }
    __syncthreads();
    for (int q = tid; q < nsub; q += FLIP_BLOCK) out[q] = lds[q];
    if (last && tid == 0) out[nsub] = r1;
    for (int32_t sa = r0 + tid; sa < r1; sa += FLIP_BLOCK) {
        const int32_t o = slot_src[sa];
        double v[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) v[a] = src.a[a][o];
        const float m = src.mass[o];
        const int ci = (int)floor(v[0] + 0.5) - g.ox;
        const int32_t cs = cell_start[c0 + ci], ce = cell_start[c0 + ci + 1];
        int q, rank_cell = 0, rank_sub = 0;
        int32_t key;
        if (in_lds) {
            q = qbuf[sa - r0];
            key = kbuf[sa - r0];
            for (int32_t t = cs - r0; t < ce - r0; ++t) {
                const int less = kbuf[t] < key ? 1 : 0;
                rank_cell += less;
                rank_sub += (qbuf[t] == q) ? less : 0;
            }
        } else {
            q = tmp_q[sa];
            key = slot_key[sa];
            for (int32_t t = cs; t < ce; ++t) {
                const int less = slot_key[t] < key ? 1 : 0;
                rank_cell += less;
                rank_sub += (tmp_q[t] == q) ? less : 0;
            }
        }
        const int32_t s = cs + rank_cell;  // place in the primary order
        const int32_t d = lds[q] + rank_sub; // place in the half-cell order
#pragma unroll
        for (int a = 0; a < 6; ++a) dst.a[a][s] = v[a];
        dst.mass[s] = m;
        dst.pid[s] = key;
        sub_slot[s] = d;
        flip_ppos pa;
        pa.x = v[0]; pa.y = v[1]; pa.z = v[2]; pa.mass = m; pa.pad0 = 0.f;
        flip_pvel pb;
        pb.vx = v[3]; pb.vy = v[4]; pb.vz = v[5]; pb.pad1 = 0.0;
        pcell[s] = pa;
        pm[2 * (long long)d] = pa;   // position and velocity halves of a record sit side by side (see packed_pos)
        pv[2 * (long long)d] = pb;
    }
}
// refresh of the packed records when the SoA arrays changed without a re-sort (velocities after G2P); n_inside_ptr:
// the number of particles inside the box, the only ones the half-cell order holds
__global__ void k_pack_records(long long n, const int32_t *__restrict__ n_inside_ptr, ParticlePtrs src,
                               const int32_t *__restrict__ sub_slot, flip_ppos *__restrict__ pm, flip_pvel *__restrict__ pv,
                               flip_ppos *__restrict__ pcell) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    flip_ppos a;
    a.x = src.a[0][s]; a.y = src.a[1][s]; a.z = src.a[2][s]; a.mass = src.mass[s]; a.pad0 = 0.f;
    pcell[s] = a;
    if (s >= *n_inside_ptr) return;
    flip_pvel b;
    b.vx = src.a[3][s]; b.vy = src.a[4][s]; b.vz = src.a[5][s]; b.pad1 = 0.0;
    const int32_t d = sub_slot[s];
    pm[2 * (long long)d] = a;
    pv[2 * (long long)d] = b;
}
static ParticlePtrs particle_ptrs(flip_ctx *ctx, int buffer) {
    ParticlePtrs ps;
    for (int q = 0; q < 6; ++q) ps.a[q] = ctx->p[buffer][q];
    ps.mass = ctx->mass[buffer];
    ps.pid = ctx->pid[buffer];
    return ps;
}

// The gathers' copy is an array of 64-byte records, position half then velocity half: record d's halves are element
// 2 d of these two views.  Every record the P2G walk visits is inside the support, so it needs both halves, and one
// record is one cache line access instead of two.
static inline flip_ppos *packed_pos(flip_ctx *ctx) { return reinterpret_cast<flip_ppos *>(ctx->prec); }
static inline flip_pvel *packed_vel(flip_ctx *ctx) { return reinterpret_cast<flip_pvel *>(reinterpret_cast<flip_ppos *>(ctx->prec) + 1); }

// Occupancy bands: bit0 = within 1 cell of a particle cell, bit1 = within 2 (Chebyshev distance).
// Box dilation is separable: one pass per axis.
__global__ void k_band_x(GridDims g, const int32_t *__restrict__ cell_start, uint8_t *__restrict__ out) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.B) return;
    int i, j, k;
    d_ijk(g, c, i, j, k);
    uint8_t m = 0;
    for (int d = -2; d <= 2; ++d) {
        const int ii = i + d;
        if (ii < 0 || ii >= g.nx) continue;
        const long long n = c + d;
        if (cell_start[n + 1] > cell_start[n]) m |= (d >= -1 && d <= 1) ? 3 : 2;
    }
    out[c] = m;
}
__global__ void k_band_axis(GridDims g, int axis, const uint8_t *__restrict__ in, uint8_t *__restrict__ out) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.B) return;
    int i, j, k;
    d_ijk(g, c, i, j, k);
    const int pos = axis == 1 ? j : k, lim = axis == 1 ? g.ny : g.nz;
    const long long stride = axis == 1 ? g.nx : (long long)g.nx * g.ny;
    uint8_t m = 0;
    for (int d = -2; d <= 2; ++d) {
        if (pos + d < 0 || pos + d >= lim) continue;
        const uint8_t v = in[c + d * stride];
        if (d >= -1 && d <= 1) m |= v;
        else m |= (v & 2);
    }
    // bit0 needs distance <= 1 on every axis, bit1 distance <= 2: bit0 of a +-2 neighbour does not
    // count, and bit1 of any neighbour within 2 does
    out[c] = m;
}

#define GT_T 4 // edge of the LDS-staged gather's tiles in cells (see k_gather_tile)
static int gather_column();
static int gather_planes();
// tiles with a cell of the band (bit 1): flag, then compacted by a scan into the list the gather's workgroups share out
__global__ void k_gather_tile_flags(GridDims g, int tiles_x, int tiles_y, long long ntiles, const uint8_t *__restrict__ band,
                                    int32_t *__restrict__ flag) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > ntiles) return;
    if (t == ntiles) { flag[t] = 0; return; } // terminator of the scan
    const int i0 = (int)(t % tiles_x) * GT_T;
    const long long tq = t / tiles_x;
    const int j0 = (int)(tq % tiles_y) * GT_T, k0 = (int)(tq / tiles_y) * GT_T;
    int any = 0;
    for (int dk = 0; dk < GT_T && k0 + dk < g.nz; ++dk)
        for (int dj = 0; dj < GT_T && j0 + dj < g.ny; ++dj) {
            const long long row = d_lid(g, 0, j0 + dj, k0 + dk);
            for (int di = 0; di < GT_T && i0 + di < g.nx; ++di) any |= band[row + i0 + di] & 2;
        }
    flag[t] = any ? 1 : 0;
}
// Work list of the per-lane gather: a slot is a column of ry cells in each of rz planes (the cells of one thread of
// k_gather); it is listed iff one of its cells lies in the band.  The gather's lanes take consecutive list entries, so
// that a wave holds 64 slots with work whatever the shape of the liquid (slot numbering = x fastest, as the cells).
__global__ void k_gather_slot_flags(GridDims g, int ry, int rz, int nyb, long long nslots, const uint8_t *__restrict__ band,
                                    int32_t *__restrict__ scan) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > nslots) return;
    if (t == nslots) { scan[t] = 0; return; }
    const int i = (int)(t % g.nx);
    const long long tq = t / g.nx;
    const int j0 = (int)(tq % nyb) * ry, k0 = (int)(tq / nyb) * rz;
    int flag = 0;
    for (int dk = 0; dk < rz; ++dk)
        for (int dj = 0; dj < ry; ++dj)
            if (j0 + dj < g.ny && k0 + dk < g.nz && (band[d_lid(g, i, j0 + dj, k0 + dk)] & 2)) flag = 1;
    scan[t] = flag;
}
__global__ void k_gather_tile_list(long long ntiles, const int32_t *__restrict__ scan, int32_t *__restrict__ list) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    if (scan[t + 1] > scan[t]) list[scan[t]] = (int32_t)t;
}

// The work list of the form of the gather the context uses (after the band passes of the binning, or when the form
// was switched since): list_shape says which one the arrays hold.
static int build_gather_lists(flip_ctx *ctx) {
    const GridDims g = flip_dims(ctx);
    hipStream_t st = ctx->stream;
    if (ctx->gather_mode != 1) {   // the slots of the per-lane gather that hold a cell of the band
        const int ry = gather_column(), rz = gather_planes();
        const int nyb = (ctx->g.ny + ry - 1) / ry, nzb = (ctx->g.nz + rz - 1) / rz;
        const long long nslots = (long long)ctx->g.nx * nyb * nzb;
        k_gather_slot_flags<<<flip_grid_blocks(nslots + 1), FLIP_BLOCK, 0, st>>>(g, ry, rz, nyb, nslots, ctx->band, ctx->slot_scan);
        FLIP_LAUNCH_CHECK();
        if (int rc = flip_exclusive_scan(ctx, ctx->slot_scan, nslots + 1)) return rc;
        k_gather_tile_list<<<flip_grid_blocks(nslots), FLIP_BLOCK, 0, st>>>(nslots, ctx->slot_scan, ctx->slot_list);
        FLIP_LAUNCH_CHECK();
        ctx->slot_list_shape = ry * 16 + rz;
    } else {   // the tiles of the LDS-staged gather that hold a cell of the band
        ctx->slot_list_shape = -1;
        const int tx = (ctx->g.nx + GT_T - 1) / GT_T, ty = (ctx->g.ny + GT_T - 1) / GT_T;
        k_gather_tile_flags<<<flip_grid_blocks(ctx->n_tiles + 1), FLIP_BLOCK, 0, st>>>(g, tx, ty, ctx->n_tiles, ctx->band, ctx->tile_scan);
        FLIP_LAUNCH_CHECK();
        if (int rc = flip_exclusive_scan(ctx, ctx->tile_scan, ctx->n_tiles + 1)) return rc;
        k_gather_tile_list<<<flip_grid_blocks(ctx->n_tiles), FLIP_BLOCK, 0, st>>>(ctx->n_tiles, ctx->tile_scan, ctx->tile_list);
        FLIP_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int flip_bin_particles(flip_ctx *ctx, int64_t *n_outside) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_CHECK(hipSetDevice(ctx->device));
    const GridDims g = flip_dims(ctx);
    const int64_t n = ctx->n, B = ctx->B;
    hipStream_t st = ctx->stream;
    if (ctx->binned) { // nothing moved since the last sort
        if (n_outside) {
            int32_t ends[2];
            FLIP_CHECK(hipMemcpyAsync(ends, ctx->cell_start + B, 8, hipMemcpyDeviceToHost, st));
            FLIP_CHECK(hipStreamSynchronize(st));
            ctx->n_outside = (int64_t)ends[1] - ends[0];
            *n_outside = ctx->n_outside;
        }
        return 0;
    }
    FLIP_CHECK(hipMemsetAsync(ctx->cell_start, 0, (size_t)(B + 2) * 4, st));
    FLIP_CHECK(hipMemsetAsync(ctx->cell_fill, 0, (size_t)(B + 1) * 4, st));
    const int src = ctx->cur, dst = 1 - src;
    ctx->density_valid = false;
    if (n > 0) {
        k_cell_ids<<<flip_grid_blocks(n), FLIP_BLOCK, 0, st>>>(g, n, ctx->p[src][0], ctx->p[src][1], ctx->p[src][2],
                                                              ctx->cell_id, ctx->cell_start);
        FLIP_LAUNCH_CHECK();
    }
    if (int rc = flip_exclusive_scan(ctx, ctx->cell_start, B + 2)) return rc;
    if (n > 0) {
        k_scatter_slots<<<flip_grid_blocks(n), FLIP_BLOCK, 0, st>>>(n, ctx->cell_id, ctx->pid[src], ctx->cell_start,
                                                                   ctx->cell_fill, ctx->slot_src, ctx->slot_key);
        FLIP_LAUNCH_CHECK();
        ctx->cur = dst;
    }
    // every grid row gathers its particles from the previous order and stores them in both orders (also with no particles:
    // the gathers read sub_start); the particles outside the box, behind the last row, keep the plain reorder
    // LDS room for the upload indices and half-cell numbers of a row: four times the mean row, within [1024, FLIP_ROW_QCAP]
    // (a row above it reads them from global memory instead); the smaller the table, the more rows a CU works on at once,
    // and the kernel lives on the latency of its gathers
    const long long rows = (long long)ctx->g.ny * ctx->g.nz;
    long long qcap = ((4 * n / rows + 511) / 512) * 512;
    qcap = qcap < 1024 ? 1024 : (qcap > FLIP_ROW_QCAP ? FLIP_ROW_QCAP : qcap);
    k_reorder_pack<<<(int)rows, FLIP_BLOCK, flip_subrow_lds_bytes_for(ctx->g.nx, (int)qcap), st>>>(
        g, ctx->cell_start, ctx->slot_src, ctx->slot_key, particle_ptrs(ctx, src), particle_ptrs(ctx, dst), ctx->sub_start, ctx->sub_slot,
        ctx->cell_id, ctx->pcell, packed_pos(ctx), packed_vel(ctx), (int)qcap);
    FLIP_LAUNCH_CHECK();
    if (n > 0) {
        k_reorder_outside<<<flip_grid_blocks(n), FLIP_BLOCK, 0, st>>>(n, ctx->cell_start + B, ctx->slot_src, ctx->slot_key, particle_ptrs(ctx, src),
                                                                     particle_ptrs(ctx, dst), ctx->pcell);
        FLIP_LAUNCH_CHECK();
    }
    ctx->prec_valid = true;
    k_band_x<<<flip_grid_blocks(B), FLIP_BLOCK, 0, st>>>(g, ctx->cell_start, ctx->band);
    FLIP_LAUNCH_CHECK();
    k_band_axis<<<flip_grid_blocks(B), FLIP_BLOCK, 0, st>>>(g, 1, ctx->band, ctx->tmp_u8);
    FLIP_LAUNCH_CHECK();
    k_band_axis<<<flip_grid_blocks(B), FLIP_BLOCK, 0, st>>>(g, 2, ctx->tmp_u8, ctx->band);
    FLIP_LAUNCH_CHECK();
    if (int rc = build_gather_lists(ctx)) return rc;
    ctx->binned = true;
    ctx->slots_valid = true;
    if (n_outside) {
        int32_t ends[2];
        FLIP_CHECK(hipMemcpyAsync(ends, ctx->cell_start + B, 8, hipMemcpyDeviceToHost, st));
        FLIP_CHECK(hipStreamSynchronize(st));
        ctx->n_outside = (int64_t)ends[1] - ends[0];
        *n_outside = ctx->n_outside;
    }
    return 0;
}

__global__ void k_cell_id_by_pid(long long n, long long B, const int32_t *__restrict__ cell_start_unused,
                                 const int32_t *__restrict__ pid, const double *__restrict__ x,
                                 const double *__restrict__ y, const double *__restrict__ z, GridDims g,
                                 int32_t *__restrict__ out) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int i = (int)floor(x[s] + 0.5) - g.ox;
    const int j = (int)floor(y[s] + 0.5) - g.oy;
    const int k = (int)floor(z[s] + 0.5) - g.oz;
    out[pid[s]] = d_inside(g, i, j, k) ? (int32_t)d_lid(g, i, j, k) : -1;
    (void)B; (void)cell_start_unused;
}

extern "C" int flip_binning_download(flip_ctx *ctx, int32_t *cell_id, int32_t *cell_start, int32_t *perm) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(ctx->binned, "flip_binning_download: call flip_bin_particles first");
    FLIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t n = ctx->n;
    if (cell_id && n > 0) {
        k_cell_id_by_pid<<<flip_grid_blocks(n), FLIP_BLOCK, 0, st>>>(n, ctx->B, nullptr, ctx->pid[ctx->cur],
                                                                    ctx->p[ctx->cur][0], ctx->p[ctx->cur][1],
                                                                    ctx->p[ctx->cur][2], flip_dims(ctx), ctx->slot_src);
        FLIP_LAUNCH_CHECK();
        FLIP_CHECK(hipMemcpyAsync(cell_id, ctx->slot_src, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    }
    if (cell_start) FLIP_CHECK(hipMemcpyAsync(cell_start, ctx->cell_start, (size_t)(ctx->B + 1) * 4, hipMemcpyDeviceToHost, st));
    if (perm && n > 0) FLIP_CHECK(hipMemcpyAsync(perm, ctx->pid[ctx->cur], (size_t)n * 4, hipMemcpyDeviceToHost, st));
    FLIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// P2G gather (src/FluidSimulator.cpp:71-118, src/Functors.cpp:10-45, kernel :1361-1371) and the density field
// (src/FluidSimulator.cpp:625-652, src/Functors.cpp:216-266; the level set does not change between the two reference
// calls, so one walk over the particles serves both; flip_density() launches the density-only form when the fused
// result is stale).
//
// One thread per slot of the work list: a column of RY cells in each of RZ planes (2 x 2 by default), all of which lie
// in the band or the thread skips them.  The particles of a cell are the 16 runs of the half-cell order described above.
// The runs of a lane are 0 to ~10 records long and differ from lane to lane, so a loop per run would idle most of the
// wave most of the time; instead every lane walks ONE flat sequence -- its non-empty runs back to back, their
// (first, end) pairs parked in LDS because the run number is a per-lane variable -- and the wave iterates
// max-over-lanes of the total.  Records are requested two steps before they are consumed.
// Accumulation order = run order = the order stated at d_upper_half.
struct DensityArgs {
    const float *phi;
    float *density;
    int32_t *n_over;
    float phi_bg, rest;
    double vs3;
};

// RY x RZ cells per thread (a y-column of RY cells in each of RZ planes): the runs around them are walked once for all of
// them ((2 RY + 2) x (2 RZ + 2) runs instead of 16 RY RZ), which removes loads and loop overhead, not arithmetic; every
// cell still meets its particles in its own order (the run order restricted to the runs of one cell's support is that
// cell's (dk, dj, hz, hy) order, because the cell offsets only shift dk and dj).  Which cells a run feeds travels in the
// spare top two bits of each half of its (first, end) pair, which is why a context holds fewer than 2^30 particles.
// The density term divides by a per-cell constant: a / b is evaluated as q0 = a * y, q = fma(fma(-q0, b, a), y, q0) with
// y = 1 / b rounded once per cell, which yields the correctly rounded quotient (the residual a - q0 b is exact in the
// fused operation, and one correction of a product that is already within an ulp lands on the nearest double) at three
// instructions instead of the fifteen of a division.
template <bool P2G, int RY, int RZ, int DEPTH>
__device__ __forceinline__ void d_gather_slot(const GridDims &g, int nyb, long long t, int2 (*runs)[FLIP_BLOCK], const int32_t *__restrict__ sub_start,
                                              const uint8_t *__restrict__ band, const flip_ppos *__restrict__ pm, const flip_pvel *__restrict__ pv,
                                              double *__restrict__ u, double *__restrict__ v, double *__restrict__ w,
                                              uint8_t *__restrict__ active, const DensityArgs &da) {
    constexpr int NC = RY * RZ;
    static_assert(NC <= 4, "the feed mask has four bits");
    const int tid = threadIdx.x;
    const int i = (int)(t % g.nx);
    const long long tq = t / g.nx;
    const int j0 = (int)(tq % nyb) * RY, k0 = (int)(tq / nyb) * RZ;
    const float *__restrict__ phi = da.phi;
    bool live[NC], with_density[NC];
    float fw[NC], dens[NC];
    double denom[NC], rden[NC];
    bool any_live = false, any_walk = false;
#pragma unroll
    for (int r = 0; r < NC; ++r) {
        const int j = j0 + r % RY, k = k0 + r / RY;
        live[r] = false; with_density[r] = false;
        fw[r] = 0.f; dens[r] = 0.f; denom[r] = 1.0; rden[r] = 1.0;
        if (j >= g.ny || k >= g.nz) continue;
        const long long c = d_lid(g, i, j, k);
        if (!(band[c] & 2)) continue;   // its zeros are k_gather_clear's
        live[r] = true;
        any_live = true;
        const float pc = phi[c];
        fw[r] = d_frac2(d_getf(g, phi, i - 1, j, k, da.phi_bg), pc) + d_frac2(d_getf(g, phi, i, j - 1, k, da.phi_bg), pc)
                + d_frac2(d_getf(g, phi, i, j, k - 1, da.phi_bg), pc) + d_frac2(d_getf(g, phi, i + 1, j, k, da.phi_bg), pc)
                + d_frac2(d_getf(g, phi, i, j + 1, k, da.phi_bg), pc) + d_frac2(d_getf(g, phi, i, j, k + 1, da.phi_bg), pc);
        denom[r] = (double)(fw[r] / 6) * da.vs3;
        with_density[r] = fw[r] > 0;
        if (with_density[r]) rden[r] = 1.0 / denom[r];
        any_walk = any_walk || P2G || with_density[r];
    }
    if (!any_live) return;
    int nr = 0;
    if (any_walk) {
        const int sxa = i > 0 ? 2 * i - 1 : 0, sxb = i < g.nx - 1 ? 2 * i + 2 : 2 * g.nx - 1;
        const int nx2 = 2 * g.nx;
#pragma unroll
        for (int dk = -1; dk <= RZ; ++dk) {
            const int kk = k0 + dk;
            if (kk < 0 || kk >= g.nz) continue;
#pragma unroll
            for (int dj = -1; dj <= RY; ++dj) {
                const int jj = j0 + dj;
                if (jj < 0 || jj >= g.ny) continue;
                const long long rowbase = ((long long)kk * g.ny + jj) * (4LL * nx2);
#pragma unroll
                for (int hz = (dk < 0 ? 1 : 0); hz <= (dk >= RZ ? 0 : 1); ++hz) {
#pragma unroll
                    for (int hy = (dj < 0 ? 1 : 0); hy <= (dj >= RY ? 0 : 1); ++hy) {
                        // half-cell row 2 dj + hy (relative to the thread's first cell) lies in the support of a cell at
                        // offset ry iff 2 ry - 1 <= 2 dj + hy <= 2 ry + 2; the same along z
                        const int rely = 2 * dj + hy, relz = 2 * dk + hz;
                        unsigned feeds = 0;
#pragma unroll
                        for (int r = 0; r < NC; ++r) {
                            const int ry = r % RY, rz = r / RY;
                            if (live[r] && rely >= 2 * ry - 1 && rely <= 2 * ry + 2 && relz >= 2 * rz - 1 && relz <= 2 * rz + 2) feeds |= 1u << r;
                        }
                        if (!feeds) continue;
                        const int32_t *row = sub_start + rowbase + (long long)(hz * 2 + hy) * nx2;
                        const int32_t s0 = row[sxa], s1 = row[sxb + 1];
                        if (s0 < s1) {
                            runs[nr][tid] = make_int2((int)((unsigned)s0 | ((feeds & 3u) << 30)), (int)((unsigned)s1 | ((feeds >> 2) << 30)));
                            ++nr;
                        }
                    }
                }
            }
        }
    }
    const double eps = (double)10e-37f; // background of the weight grid (src/FluidSimulator.cpp:32,77)
    double ws0[NC], ws1[NC], ws2[NC], a0[NC], a1[NC], a2[NC];
#pragma unroll
    for (int r = 0; r < NC; ++r) {
        ws0[r] = ws1[r] = ws2[r] = eps;
        a0[r] = a1[r] = a2[r] = 0.0;
    }
    double cy[RY], cz[RZ];
#pragma unroll
    for (int r = 0; r < RY; ++r) cy[r] = (double)(g.oy + j0 + r);
#pragma unroll
    for (int r = 0; r < RZ; ++r) cz[r] = (double)(g.oz + k0 + r);
    const double cx = (double)(g.ox + i);
    if (nr > 0) {
        // cursor of the flat sequence: the record about to be requested
        int r_idx = 0;
        int2 cur = runs[0][tid];
        unsigned feeds = ((unsigned)cur.x >> 30) | (((unsigned)cur.y >> 30) << 2);
        int32_t s = cur.x & 0x3fffffff, e = cur.y & 0x3fffffff;
        bool more = true;
        // DEPTH records are in flight per lane: slot d is requested DEPTH steps before it is consumed, so the latency of a
        // record overlaps the arithmetic of the DEPTH - 1 before it (one record in flight leaves the lane idle for most
        // of a round trip; the kernel is bound by that latency, not by the number of requests)
        double2 b0[DEPTH], b1[DEPTH], b2[DEPTH], b3[DEPTH];
        unsigned bf[DEPTH];
        bool have[DEPTH];
        auto request = [&](int d) {
            have[d] = more;
            if (more) {
                const double2 *rec = reinterpret_cast<const double2 *>(pm + 2 * (long long)s);
                b0[d] = rec[0]; b1[d] = rec[1]; // (x, y) (z, mass)
                if (P2G) {
                    const double2 *vel = reinterpret_cast<const double2 *>(pv + 2 * (long long)s);
                    b2[d] = vel[0]; b3[d] = vel[1]; // (vx, vy) (vz, -)
                }
                bf[d] = feeds;
                ++s;
                if (s == e) {
                    ++r_idx;
                    more = r_idx < nr;
                    if (more) {
                        cur = runs[r_idx][tid];
                        feeds = ((unsigned)cur.x >> 30) | (((unsigned)cur.y >> 30) << 2);
                        s = cur.x & 0x3fffffff; e = cur.y & 0x3fffffff;
                    }
                }
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            b0[d] = b1[d] = b2[d] = b3[d] = make_double2(0.0, 0.0);
            bf[d] = 0;
            request(d);
        }
        while (have[0]) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (!have[d]) break;
                const double2 r0 = b0[d], r1 = b1[d], r2 = b2[d], r3 = b3[d];
                const unsigned fd = bf[d];
                request(d);
                const double d0 = r0.x - cx;
                // inside the support the hat function is 1 - |x| (1 + x for a negative x is the same IEEE operation), and the
                // shifted argument x + 1/2 lies in [-1/2, 3/2], so only its upper cut-off needs a test
                const double hx = d_hat_in(d0);
                const float mass = __int_as_float((int)(__double_as_longlong(r1.y) & 0xffffffffLL));
                const float x0 = (float)hx;
                const float x1 = (float)d_hat_up(d0 + 0.5);
                double hyv[RY], hzv[RZ];
                float y0v[RY], y1v[RY], z0v[RZ], z1v[RZ];
#pragma unroll
                for (int q = 0; q < RY; ++q) {
                    const double d1 = r0.y - cy[q];
                    hyv[q] = d_hat_in(d1);
                    y0v[q] = (float)hyv[q];
                    y1v[q] = (float)d_hat_up(d1 + 0.5);
                }
#pragma unroll
                for (int q = 0; q < RZ; ++q) {
                    const double d2 = r1.x - cz[q];
                    hzv[q] = d_hat_in(d2);
                    z0v[q] = (float)hzv[q];
                    z1v[q] = (float)d_hat_up(d2 + 0.5);
                }
#pragma unroll
                for (int r = 0; r < NC; ++r) {
                    if (NC > 1 && !((fd >> r) & 1u)) continue;
                    const int ry = r % RY, rz = r / RY;
                    const double wgt = hx * hyv[ry] * hzv[rz];
                    if (with_density[r] && wgt > 0) {
                        const double a = wgt * (double)mass;
                        const double qa = a * rden[r];
                        const double quot = fma(fma(-qa, denom[r], a), rden[r], qa); // = a / denom[r], correctly rounded
                        const double add = (double)0.f + quot;
                        dens[r] = (float)((double)dens[r] + add);
                    }
                    if (P2G) {
                        const float y0 = y0v[ry], y1 = y1v[ry], z0 = z0v[rz], z1 = z1v[rz];
                        const double w0 = (double)(x1 * y0 * z0), w1 = (double)(x0 * y1 * z0), w2 = (double)(x0 * y0 * z1);
                        if (w0 != 0.0 || w1 != 0.0 || w2 != 0.0) {
                            ws0[r] = ws0[r] + w0; ws1[r] = ws1[r] + w1; ws2[r] = ws2[r] + w2;
                            a0[r] = a0[r] + w0 * r2.x; a1[r] = a1[r] + w1 * r2.y; a2[r] = a2[r] + w2 * r3.x;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NC; ++r) {
        if (!live[r]) continue;
        const long long c = d_lid(g, i, j0 + r % RY, k0 + r / RY);
        if (P2G) {
            u[c] = a0[r] / ws0[r];
            v[c] = a1[r] / ws1[r];
            w[c] = a2[r] / ws2[r];
            active[c] = 1;
        }
        float dn = dens[r];
        if (0 < fw[r] && fw[r] < 6) dn = dn > da.rest ? dn : da.rest;
        da.density[c] = dn;
        if ((double)dn > 1.5 * (double)da.rest) atomicAdd(da.n_over, 1);
    }
}

// The cells outside the band: zero velocity, inactive, zero density (the reference's grids have no voxel there).
template <bool P2G>
__global__ void k_gather_clear(long long B, const uint8_t *__restrict__ band, double *__restrict__ u, double *__restrict__ v,
                               double *__restrict__ w, uint8_t *__restrict__ active, float *__restrict__ density) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B || (band[c] & 2)) return;
    if (P2G) { u[c] = 0.0; v[c] = 0.0; w[c] = 0.0; active[c] = 0; }
    density[c] = 0.f;
}
// Persistent workgroups over the work list (k_gather_slot_flags): a workgroup claims 256 consecutive entries at a time,
// a few neighbouring rows of slots, whose waves read many of the same records while they are in the CU's L1 (waves that
// claim on their own end up on different CUs: 857 against 786 us at 256^3).  The kernel is bound by the L1 misses a CU
// can have in flight, not by its request or arithmetic rate, which is also why more than two or three workgroups per CU
// make it slower.  The list is cut into eight contiguous parts, one per XCD (workgroups with the same blockIdx % 8 share an
// XCD and its L2 under round-robin dispatch), each with a claim counter of its own.  A workgroup whose part is used up
// goes on to the following parts, at most once round, and leaves: no workgroup waits for another.
template <bool P2G, int RY, int RZ, int DEPTH>
__global__ void __launch_bounds__(FLIP_BLOCK) k_gather(GridDims g, int nyb, const int32_t *__restrict__ list, const int32_t *__restrict__ n_list,
                                                      int32_t *__restrict__ claim, const int32_t *__restrict__ sub_start,
                                                      const uint8_t *__restrict__ band, const flip_ppos *__restrict__ pm,
                                                      const flip_pvel *__restrict__ pv, double *__restrict__ u, double *__restrict__ v,
                                                      double *__restrict__ w, uint8_t *__restrict__ active, DensityArgs da) {
    constexpr int NRUN = (2 * RY + 2) * (2 * RZ + 2);
    __shared__ int2 runs[NRUN][FLIP_BLOCK];
    __shared__ int block_claim;
    const int n = *n_list;
    const int home = blockIdx.x & 7;
    const int chunks = (n + FLIP_BLOCK - 1) / FLIP_BLOCK;
    for (int hop = 0; hop < 8; ++hop) {
        const int part = (home + hop) & 7;
        const int lo = (int)(((long long)chunks * part) >> 3), hi = (int)(((long long)chunks * (part + 1)) >> 3);
        for (;;) {
            __syncthreads();   // every thread has read the previous claim
            if (threadIdx.x == 0) block_claim = lo + atomicAdd(claim + part, 1);
            __syncthreads();
            const int c = block_claim;
            if (c >= hi) break;
            const int q = c * FLIP_BLOCK + threadIdx.x;
            if (q < n) d_gather_slot<P2G, RY, RZ, DEPTH>(g, nyb, list[q], runs, sub_start, band, pm, pv, u, v, w, active, da);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-staged form of the gather (flip_set_gather_mode 1; same bits as k_gather, measured slower).  A workgroup owns a tile of 4 x 4 x 4 cells.  The support of the tile
// is 10 x 10 (z, y) half-cell rows, and within each of them ONE contiguous span of the half-cell order (the half cells
// 2 i0 - 1 .. 2 i0 + 8 of that sub-row): the workgroup copies those 100 spans into LDS with coalesced 16-byte loads
// -- every record once per tile instead of once per cell and lane -- as separate arrays x[], y[], z[], vx[], vy[],
// vz[] (8-byte elements, so that neighbouring lanes, which read neighbouring records, fall on different banks) and
// mass[].  Every cell then meets its records in the order stated at d_upper_half, run by run, exactly as k_gather
// does from global memory, so the sums are the same bit for bit; the LDS positions of a cell's records are written
// once into a per-cell list (16-bit entries), so that the walk itself is a plain counted loop.
// The four sums of a cell (u, v, w of P2G and the density) are independent chains over the same records: the four
// waves of the workgroup each take one of them for the tile's 64 cells (wave 0: u, 1: v, 2: w, 3: density).  That
// repeats the three differences and the loop control but gives the CU four times the waves per LDS byte, and LDS
// capacity (two tiles per CU), not arithmetic, limits the number of tiles in flight.
// The workgroups are persistent and share out the list of tiles that hold a cell of the band (built by the binning).
// A tile whose spans hold more than GT_CAP records, or a cell more than GT_LMAX (a crowded corner of the scene), walks
// global memory instead, run by run as k_gather does.
#define GT_CELLS (GT_T * GT_T * GT_T)        // 64 = one wave
#define GT_HR (2 * GT_T + 2)                 // half-cell rows of a tile's support per axis
#define GT_NSPAN (GT_HR * GT_HR)
#define GT_SST (2 * GT_T + 3)                // boundaries of the 2 T + 2 half cells of a span
#define GT_CAP 1152                          // records a tile can stage (about 1000 at 8 particles per cell)
#define GT_STRIDE (GT_CAP + 4)               // elements between the arrays: a quarter of the banks apart
#define GT_LMAX 120                          // records a cell can list (64 at 8 particles per cell)
#define GT_SPG ((GT_NSPAN + 15) / 16)        // spans per 16-lane group of the staging copy
template <bool P2G>
struct GatherTileLds {
    static constexpr int NARR = P2G ? 6 : 3;
    static constexpr size_t rec = 0;                                        // double [NARR][GT_STRIDE]; unstaged tiles: int2 runs[16][64]
    static constexpr size_t mass = rec + (size_t)NARR * GT_STRIDE * 8;      // float [GT_CAP]
    static constexpr size_t idx = mass + (size_t)GT_CAP * 4;                // uint16 [GT_LMAX][64]: LDS position of a cell's t-th record
    static constexpr size_t sst = idx + (size_t)GT_LMAX * GT_CELLS * 2;     // int32 [GT_NSPAN][GT_SST]
    static constexpr size_t off = sst + (size_t)GT_NSPAN * GT_SST * 4;      // int32 [128]: exclusive scan of the span lengths
    static constexpr size_t total = off + 128 * 4;
};
static_assert(GT_NSPAN <= 128 && GT_CAP < 65536 && 3 * GT_STRIDE * 8 >= 16 * GT_CELLS * 8, "tile tables");

// One record of the walk: position, and the role's fourth value (a velocity component; the mass as float).
template <int ROLE, bool STAGED, bool P2G>
__device__ __forceinline__ void d_gt_fetch(const char *lds, const double *__restrict__ prec, int32_t s, double &x, double &y, double &z, double &q,
                                           float &m) {
    if (STAGED) {
        const double *r = reinterpret_cast<const double *>(lds + GatherTileLds<P2G>::rec);
        x = r[s]; y = r[GT_STRIDE + s]; z = r[2 * GT_STRIDE + s];
        if (ROLE < 3) q = r[(3 + (ROLE < 3 ? ROLE : 0)) * GT_STRIDE + s];
        else m = reinterpret_cast<const float *>(lds + GatherTileLds<P2G>::mass)[s];
    } else {
        const double *r = prec + 8 * (long long)s;
        x = r[0]; y = r[1]; z = r[2];
        if (ROLE < 3) q = r[4 + (ROLE < 3 ? ROLE : 0)];
        else m = __int_as_float((int)(__double_as_longlong(r[3]) & 0xffffffffLL));
    }
}
// The contribution of one record to one role's sums.  ROLE 0..2: ws += w_c, acc += w_c v_c (src/Functors.cpp:37-40
// with the weights of src/FluidSimulator.cpp:1361-1371); ROLE 3: the density sum (src/Functors.cpp:252-258).  Adding
// a zero weight leaves both sums unchanged, so the reference's "skip if the weight is zero" needs no test here.
template <int ROLE>
__device__ __forceinline__ void d_gt_add(double rx, double ry, double rz, double rq, float rm, double cx, double cy, double cz, double denom,
                                         double rden, double &ws, double &acc, float &dens) {
    const double d0 = rx - cx, d1 = ry - cy, d2 = rz - cz;
    if (ROLE == 3) {
        const double wgt = d_hat_in(d0) * d_hat_in(d1) * d_hat_in(d2);
        if (wgt > 0) {
            const double a = wgt * (double)rm;
            const double qa = a * rden;
            const double quot = fma(fma(-qa, denom, a), rden, qa); // = a / denom, correctly rounded (see k_gather)
            const double add = (double)0.f + quot;
            dens = (float)((double)dens + add);
        }
    } else {
        const float fx = ROLE == 0 ? (float)d_hat_up(d0 + 0.5) : (float)d_hat_in(d0);
        const float fy = ROLE == 1 ? (float)d_hat_up(d1 + 0.5) : (float)d_hat_in(d1);
        const float fz = ROLE == 2 ? (float)d_hat_up(d2 + 0.5) : (float)d_hat_in(d2);
        const double wc = (double)(fx * fy * fz);
        ws = ws + wc;
        acc = acc + wc * rq;
    }
}
// Staged walk: the cell's records by its list; the list entry is requested two records ahead, the record one ahead.
template <int ROLE, bool P2G>
__device__ __forceinline__ void d_gt_walk_list(const char *lds, int lane, int total, double cx, double cy, double cz, double denom, double rden,
                                               double &ws, double &acc, float &dens) {
    if (total <= 0) return;
    const uint16_t *il = reinterpret_cast<const uint16_t *>(lds + GatherTileLds<P2G>::idx) + lane;
    const int last = total - 1;
    double x, y, z, q = 0.0;
    float m = 0.f;
    d_gt_fetch<ROLE, true, P2G>(lds, nullptr, il[0], x, y, z, q, m);
    int ib = il[(last < 1 ? last : 1) * GT_CELLS];
    for (int t = 0; t < total; ++t) {
        const double rx = x, ry = y, rz = z, rq = q;
        const float rm = m;
        const int t2 = t + 2 < last ? t + 2 : last;
        const int ic = il[t2 * GT_CELLS];
        d_gt_fetch<ROLE, true, P2G>(lds, nullptr, ib, x, y, z, q, m); // past the end: the last record again, not used
        ib = ic;
        d_gt_add<ROLE>(rx, ry, rz, rq, rm, cx, cy, cz, denom, rden, ws, acc, dens);
    }
}
// Unstaged walk: the non-empty runs (first, end) of the cell in global record numbers, parked in LDS.
template <int ROLE, bool P2G>
__device__ __forceinline__ void d_gt_walk_runs(const char *lds, const double *__restrict__ prec, int lane, int nr, double cx, double cy, double cz,
                                               double denom, double rden, double &ws, double &acc, float &dens) {
    if (nr <= 0) return;
    const int2 *runs = reinterpret_cast<const int2 *>(lds + GatherTileLds<P2G>::rec) + lane;
    int2 cur = runs[0];
    int32_t s = cur.x, e = cur.y;
    double x, y, z, q = 0.0;
    float m = 0.f;
    d_gt_fetch<ROLE, false, P2G>(lds, prec, s, x, y, z, q, m);
    int ri = 0;
    bool more = true;
    while (more) {
        const double rx = x, ry = y, rz = z, rq = q;
        const float rm = m;
        ++s;
        if (s == e) {
            ++ri;
            more = ri < nr;
            if (more) { cur = runs[ri * GT_CELLS]; s = cur.x; e = cur.y; }
        }
        if (more) d_gt_fetch<ROLE, false, P2G>(lds, prec, s, x, y, z, q, m);
        d_gt_add<ROLE>(rx, ry, rz, rq, rm, cx, cy, cz, denom, rden, ws, acc, dens);
    }
}

template <int ROLE, bool P2G>
__device__ __forceinline__ void d_gt_role(const char *lds, const double *__restrict__ prec, bool staged, const GridDims &g, int lane, int count,
                                          bool live, int i, int j, int k, long long c, float fw, double *__restrict__ out,
                                          uint8_t *__restrict__ active, const DensityArgs &da) {
    if (!live) return;
    const double cx = (double)(g.ox + i), cy = (double)(g.oy + j), cz = (double)(g.oz + k);
    double ws = (double)10e-37f, acc = 0.0; // background of the weight grid (src/FluidSimulator.cpp:32,77)
    float dens = 0.f;
    double denom = 1.0, rden = 1.0;
    if (ROLE == 3) {
        denom = (double)(fw / 6) * da.vs3;
        if (fw > 0) rden = 1.0 / denom;
        else count = 0; // no density term without a liquid fraction (src/Functors.cpp:240)
    }
    if (staged) d_gt_walk_list<ROLE, P2G>(lds, lane, count, cx, cy, cz, denom, rden, ws, acc, dens);
    else d_gt_walk_runs<ROLE, P2G>(lds, prec, lane, count, cx, cy, cz, denom, rden, ws, acc, dens);
    if (ROLE == 3) {
        float dn = dens;
        if (0 < fw && fw < 6) dn = dn > da.rest ? dn : da.rest;
        da.density[c] = dn;
        if ((double)dn > 1.5 * (double)da.rest) atomicAdd(da.n_over, 1);
    } else {
        out[c] = acc / ws;
        if (ROLE == 2) active[c] = 1;
    }
}

// Run r of a cell (the order (z cell, y cell, z half, y half) of the walk) lies in the half-cell row 2 iz + zoff(r) of the
// tile's support in z and 2 iy + yoff(r) in y; two bits per run.
__device__ __forceinline__ int d_gt_zoff(int r) { return (int)((0xFF9A5900u >> (2 * r)) & 3u); } // 0 0 0 0 1 2 1 1 2 2 1 2 3 3 3 3
__device__ __forceinline__ int d_gt_yoff(int r) { return (int)((0xE4F990E4u >> (2 * r)) & 3u); } // 0 1 2 3 0 0 1 2 1 2 3 3 0 1 2 3
// one 16-byte piece of a record into the staged arrays
template <bool P2G>
__device__ __forceinline__ void d_gt_store(double *rec, float *mass, int d, int part, double2 t) {
    if (part == 0) { rec[d] = t.x; rec[GT_STRIDE + d] = t.y; }
    else if (part == 1) { rec[2 * GT_STRIDE + d] = t.x; mass[d] = __int_as_float((int)(__double_as_longlong(t.y) & 0xffffffffLL)); }
    else if (P2G && part == 2) { rec[3 * GT_STRIDE + d] = t.x; rec[4 * GT_STRIDE + d] = t.y; }
    else if (P2G) rec[5 * GT_STRIDE + d] = t.x;
}
// What the stages of one tile hand on to the next ones (per lane: the lane's cell).
struct GtTile {
    int id;        // tile number, -1: none
    int flags;     // bit 0: the cell is in the band, bit 1: the tile's records are staged in LDS
    float fw;      // sum of the six liquid fractions around the cell (density role)
    int count;     // records of the cell (staged) or non-empty runs (unstaged)
    int pos0;      // records of the cell before the four runs this wave lists
};
// Loads of a tile's first stage, in flight while the previous tile is walked.
struct GtPre {
    int32_t sv[(GT_NSPAN * GT_SST + 255) / 256];
    float phi[7];
    int32_t bnd;
};

// The workgroup is a pipeline over its tiles: while tile n is walked (arithmetic on LDS), the copies of tile n + 1's
// records are on their way from memory into registers, and so are the span boundaries, band byte and level-set
// values of tile n + 2.  Per tile that leaves the LDS writes, three barriers and the walk on the critical path.
template <bool P2G>
__global__ void __launch_bounds__(256, 2) k_gather_tile(GridDims g, int tiles_x, int tiles_y, const int32_t *__restrict__ tile_list,
                                                        const int32_t *__restrict__ n_tiles_ptr, const int32_t *__restrict__ sub_start,
                                                        const uint8_t *__restrict__ band, const double *__restrict__ prec, double *__restrict__ u,
                                                        double *__restrict__ v, double *__restrict__ w, uint8_t *__restrict__ active, DensityArgs da,
                                                        int32_t *__restrict__ stats) {
    using L = GatherTileLds<P2G>;
    extern __shared__ __align__(16) int32_t lds_words[]; // one name and type for the dynamic LDS of this file
    char *lds = reinterpret_cast<char *>(lds_words);
    constexpr int NT = 256, NSV = (GT_NSPAN * GT_SST + NT - 1) / NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int role = P2G ? wave : 3;
    const bool walker = P2G || wave == 0; // density alone: one wave walks, the others only help to stage
    const int ix = lane & (GT_T - 1), iy = (lane / GT_T) & (GT_T - 1), iz = lane / (GT_T * GT_T);
    int32_t *sst = reinterpret_cast<int32_t *>(lds + L::sst);
    int32_t *soff = reinterpret_cast<int32_t *>(lds + L::off);
    double *rec = reinterpret_cast<double *>(lds + L::rec);
    float *mass = reinterpret_cast<float *>(lds + L::mass);
    const double2 *__restrict__ src = reinterpret_cast<const double2 *>(prec);
    const int nx2 = 2 * g.nx;
    const int gq = tid >> 4, rsub = (tid >> 2) & 3, part = tid & 3; // the copy: a 16-lane group per span, 4 records x 4 pieces per step
    const bool wanted = P2G || part < 2;
    // consecutive workgroups go to the eight XCDs in turn: every XCD gets one contiguous piece of the list (tiles in the
    // order x, y, z), so that tiles which share spans meet in the same L2
    const int n_live = *n_tiles_ptr;
    const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
    const int t_lo = (int)((long long)n_live * xcd / 8), t_hi = (int)((long long)n_live * (xcd + 1) / 8);
    int tt = t_lo + (int)(blockIdx.x >> 3);
    int id0 = tt < t_hi ? tile_list[tt] : -1;
    if (id0 < 0) return;
    int id1 = tt + per_xcd < t_hi ? tile_list[tt + per_xcd] : -1;
    int id2 = tt + 2 * per_xcd < t_hi ? tile_list[tt + 2 * per_xcd] : -1;
    tt += 3 * per_xcd;
    int n_unstaged = 0, n_staged = 0;

    // stage 1: everything a tile needs from global memory before its copy can be addressed
    auto issue_pre = [&](int id, GtPre &p) {
        const int tix = id % tiles_x, tq = id / tiles_x;
        const int i0 = tix * GT_T, j0 = (tq % tiles_y) * GT_T, k0 = (tq / tiles_y) * GT_T;
        const int i = i0 + ix, j = j0 + iy, k = k0 + iz;
        const bool inside = i < g.nx && j < g.ny && k < g.nz;
        const long long c = inside ? d_lid(g, i, j, k) : 0;
        p.bnd = inside ? (int32_t)band[c] : 0;
#pragma unroll
        for (int a = 0; a < NSV; ++a) {
            int q = tid + a * NT;
            asm volatile("" : "+v"(q)); // keeps the index arithmetic below inside the loop (hoisted, it holds thirty registers per lane)
            int32_t val = 0;
            if (q < GT_NSPAN * GT_SST) {
                // boundaries of the half cells of every span (clamped to the row; rows outside the box are empty)
                const int span = q / GT_SST, e = q - span * GT_SST;
                const int zh = span / GT_HR, yh = span - zh * GT_HR;
                const int kk = k0 - 1 + ((zh + 1) >> 1), hz = (zh + 1) & 1;
                const int jj = j0 - 1 + ((yh + 1) >> 1), hy = (yh + 1) & 1;
                if (kk >= 0 && kk < g.nz && jj >= 0 && jj < g.ny) {
                    int hc = 2 * i0 - 1 + e;
                    hc = hc < 0 ? 0 : (hc > nx2 ? nx2 : hc);
                    val = sub_start[((long long)kk * g.ny + jj) * (4LL * nx2) + (long long)(hz * 2 + hy) * nx2 + hc];
                }
            }
            p.sv[a] = val;
        }
        if (role == 3 && walker) {
            const float *__restrict__ phi = da.phi;
            p.phi[0] = inside ? phi[c] : 0.f;
            p.phi[1] = inside ? d_getf(g, phi, i - 1, j, k, da.phi_bg) : 0.f;
            p.phi[2] = inside ? d_getf(g, phi, i, j - 1, k, da.phi_bg) : 0.f;
            p.phi[3] = inside ? d_getf(g, phi, i, j, k - 1, da.phi_bg) : 0.f;
            p.phi[4] = inside ? d_getf(g, phi, i + 1, j, k, da.phi_bg) : 0.f;
            p.phi[5] = inside ? d_getf(g, phi, i, j + 1, k, da.phi_bg) : 0.f;
            p.phi[6] = inside ? d_getf(g, phi, i, j, k + 1, da.phi_bg) : 0.f;
        }
    };
    auto write_sst = [&](const GtPre &p) {
#pragma unroll
        for (int a = 0; a < NSV; ++a)
            if (tid + a * NT < GT_NSPAN * GT_SST) sst[tid + a * NT] = p.sv[a];
    };
    // stage 2 (the span boundaries are in LDS): scan of the span lengths -- every wave for itself, all write the same values
    // --, the lane's cell and its record count
    auto setup = [&](int id, const GtPre &p, GtTile &t) {
        t.id = id;
        const int tix = id % tiles_x, tq = id / tiles_x;
        const int i = tix * GT_T + ix, j = (tq % tiles_y) * GT_T + iy, k = (tq / tiles_y) * GT_T + iz;
        const bool inside = i < g.nx && j < g.ny && k < g.nz;
        const long long c = inside ? d_lid(g, i, j, k) : 0;
        const bool live = (p.bnd & 2) != 0;
        t.fw = 0.f;
        if (role == 3 && walker)
            t.fw = d_frac2(p.phi[1], p.phi[0]) + d_frac2(p.phi[2], p.phi[0]) + d_frac2(p.phi[3], p.phi[0]) + d_frac2(p.phi[4], p.phi[0])
                   + d_frac2(p.phi[5], p.phi[0]) + d_frac2(p.phi[6], p.phi[0]);
        if (inside && !live && walker) {
            if (P2G && role == 0) u[c] = 0.0;
            if (P2G && role == 1) v[c] = 0.0;
            if (P2G && role == 2) { w[c] = 0.0; active[c] = 0; }
            if (role == 3) da.density[c] = 0.f;
        }
        const int sa = 2 * lane, sb = 2 * lane + 1;
        const int32_t la = sa < GT_NSPAN ? sst[sa * GT_SST + GT_SST - 1] - sst[sa * GT_SST] : 0;
        const int32_t lb = sb < GT_NSPAN ? sst[sb * GT_SST + GT_SST - 1] - sst[sb * GT_SST] : 0;
        int32_t incl = la + lb;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t x = __shfl_up(incl, off, FLIP_WAVE);
            if (lane >= off) incl += x;
        }
        if (sb <= GT_NSPAN) { soff[sa] = incl - (la + lb); soff[sb] = incl - lb; }
        const int32_t R = __shfl(incl, 63, FLIP_WAVE);
        {   // the lowest record number among the non-empty spans (the base of the copy's addresses)
            int32_t lo = 0x7fffffff;
            if (la > 0) lo = sst[sa * GT_SST];
            if (lb > 0) { const int32_t q = sst[sb * GT_SST]; lo = q < lo ? q : lo; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const int32_t q = __shfl_xor(lo, off, FLIP_WAVE); lo = q < lo ? q : lo; }
            if (lane == 0) { soff[GT_NSPAN + 4] = lo == 0x7fffffff ? 0 : lo; soff[GT_NSPAN + 5] = 0; }
        }
        // the 16 runs of the cell in the order (z cell, y cell, z half, y half): total length, and the part of it that lies
        // before the four runs this wave will list
        t.count = 0; t.pos0 = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int span = (2 * iz + d_gt_zoff(r)) * GT_HR + 2 * iy + d_gt_yoff(r);
            const int32_t len = live ? sst[span * GT_SST + 2 * ix + 4] - sst[span * GT_SST + 2 * ix] : 0;
            t.count += len;
            if ((r >> 2) < wave) t.pos0 += len;
        }
        const bool staged = R <= GT_CAP && !__any(t.count > GT_LMAX); // the same in every wave: lanes are cells
        t.flags = (live ? 1 : 0) | (staged ? 2 : 0);
    };
    // stage 3a: request the copy (x y | z mass | vx vy | vz - : a 16-byte piece per lane)
    auto issue_stage = [&](const GtTile &t, double2 (&val)[GT_SPG][4]) {
        if (!(t.flags & 2)) return;
        // addresses as a wave-uniform base (the first record of the tile's first non-empty span) plus a 32-bit byte offset per
        // lane: the spans of a tile lie within six grid rows of z, far less than 4 GB apart
        const long long base_rec = __builtin_amdgcn_readfirstlane(soff[GT_NSPAN + 4]) | ((long long)__builtin_amdgcn_readfirstlane(soff[GT_NSPAN + 5]) << 32);
        const char *__restrict__ base = reinterpret_cast<const char *>(prec) + base_rec * 64;
#pragma unroll
        for (int m = 0; m < GT_SPG; ++m) {
            const int sp = gq + 16 * m;
            int32_t sL = 0;
            long long sG = base_rec;
            if (sp < GT_NSPAN) {
                sG = sst[sp * GT_SST];
                sL = sst[sp * GT_SST + GT_SST - 1] - (int32_t)sG;
            }
            const uint32_t o = (uint32_t)((sG - base_rec) * 64) + (uint32_t)(part * 16);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int r = rsub + 4 * st;
                val[m][st] = make_double2(0.0, 0.0);
                if (wanted && r < sL) val[m][st] = *reinterpret_cast<const double2 *>(base + (o + (uint32_t)(r * 64)));
            }
        }
    };
    // stage 3b (the previous tile's walk is over): the cells' lists and the records into LDS -- or, unstaged, the run table
    auto write_stage = [&](GtTile &t, const double2 (&val)[GT_SPG][4]) {
        const bool live = (t.flags & 1) != 0;
        if (t.flags & 2) {
            ++n_staged;
            uint16_t *il = reinterpret_cast<uint16_t *>(lds + L::idx) + lane;
            int pos = t.pos0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { // wave q lists the runs 4 q .. 4 q + 3 of every cell
                const int r = 4 * wave + q;
                const int span = (2 * iz + d_gt_zoff(r)) * GT_HR + 2 * iy + d_gt_yoff(r);
                const int32_t a = sst[span * GT_SST + 2 * ix];
                const int32_t len = live ? sst[span * GT_SST + 2 * ix + 4] - a : 0;
                const int32_t first = soff[span] + a - sst[span * GT_SST];
                for (int32_t e = 0; e < len; ++e) il[(pos + e) * GT_CELLS] = (uint16_t)(first + e);
                pos += len;
            }
#pragma unroll
            for (int m = 0; m < GT_SPG; ++m) {
                const int sp = gq + 16 * m;
                int32_t sL = 0, sG = 0, sO = 0;
                if (sp < GT_NSPAN) {
                    sG = sst[sp * GT_SST];
                    sL = sst[sp * GT_SST + GT_SST - 1] - sG;
                    sO = soff[sp];
                }
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int r = rsub + 4 * st;
                    if (wanted && r < sL) d_gt_store<P2G>(rec, mass, sO + r, part, val[m][st]);
                }
                // a span longer than the 16 records of the four steps: the rest one by one
                for (int r = 16 + rsub; wanted && r < sL; r += 4) d_gt_store<P2G>(rec, mass, sO + r, part, src[(long long)(sG + r) * 4 + part]);
            }
        } else {
            ++n_unstaged;
            // the non-empty runs as (first, end) in global record numbers; every wave writes the same table
            int2 *runs = reinterpret_cast<int2 *>(lds + L::rec) + lane;
            int nr = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int span = (2 * iz + d_gt_zoff(r)) * GT_HR + 2 * iy + d_gt_yoff(r);
                const int32_t a = sst[span * GT_SST + 2 * ix], b = sst[span * GT_SST + 2 * ix + 4];
                if (live && a < b) { runs[nr * GT_CELLS] = make_int2(a, b); ++nr; }
            }
            t.count = nr;
        }
    };
    auto walk = [&](const GtTile &t) {
        if (!walker) return;
        const int tix = t.id % tiles_x, tq = t.id / tiles_x;
        const int i = tix * GT_T + ix, j = (tq % tiles_y) * GT_T + iy, k = (tq / tiles_y) * GT_T + iz;
        const bool live = (t.flags & 1) != 0, staged = (t.flags & 2) != 0;
        const long long c = live ? d_lid(g, i, j, k) : 0;
        if (P2G && role < 3) {
            // P2G is a template constant: the density-only kernel has no velocity arrays and does not instantiate these
            if (role == 0) d_gt_role<0, true>(lds, prec, staged, g, lane, t.count, live, i, j, k, c, t.fw, u, active, da);
            else if (role == 1) d_gt_role<1, true>(lds, prec, staged, g, lane, t.count, live, i, j, k, c, t.fw, v, active, da);
            else d_gt_role<2, true>(lds, prec, staged, g, lane, t.count, live, i, j, k, c, t.fw, w, active, da);
        } else {
            d_gt_role<3, P2G>(lds, prec, staged, g, lane, t.count, live, i, j, k, c, t.fw, nullptr, active, da);
        }
    };

    GtPre pre;
    GtTile cur, nxt;
    double2 val[GT_SPG][4];
    issue_pre(id0, pre);
    write_sst(pre);
    __syncthreads();
    setup(id0, pre, cur);
    issue_stage(cur, val);
    if (id1 >= 0) issue_pre(id1, pre);
    for (;;) {
        __syncthreads(); // the walk of the previous tile is over: its records may go
        write_stage(cur, val);
        __syncthreads(); // nobody reads the span boundaries of this tile any more
        if (id1 >= 0) write_sst(pre);
        __syncthreads();
        if (id1 >= 0) {
            setup(id1, pre, nxt);
            issue_stage(nxt, val);
        }
        if (id2 >= 0) issue_pre(id2, pre);
        walk(cur);
        if (id1 < 0) break;
        cur = nxt;
        id1 = id2;
        id2 = tt < t_hi ? tile_list[tt] : -1;
        tt += per_xcd;
    }
    if (tid == 0) {
        if (n_staged) atomicAdd(stats, n_staged);
        if (n_unstaged) atomicAdd(stats + 1, n_unstaged);
    }
}

// cells of a y-column per thread in the gathers: FLIP_GATHER_COLUMN=1|2 (tuning aid)
static int gather_column() {
    static const int v = [] { const char *e = getenv("FLIP_GATHER_COLUMN"); const int q = e ? atoi(e) : 2; return q == 1 ? 1 : 2; }();
    return v;
}
// planes per thread in the gathers: FLIP_GATHER_PLANES=1|2 (tuning aid)
static int gather_planes() {
    static const int v = [] { const char *e = getenv("FLIP_GATHER_PLANES"); const int q = e ? atoi(e) : 2; return q == 1 ? 1 : 2; }();
    return v;
}
// records in flight per lane in the gathers: FLIP_GATHER_DEPTH=1|2|3 (tuning aid)
static int gather_depth() {
    static const int v = [] { const char *e = getenv("FLIP_GATHER_DEPTH"); const int q = e ? atoi(e) : 2; return q < 1 ? 1 : (q > 3 ? 3 : q); }();
    return v;
}
// Which form of the gather a context uses: flip_set_gather_mode; FLIP_GATHER_TILE=1 makes the LDS-staged form the default
// of new contexts (tuning aid).
int flip_default_gather_mode() {
    static const int v = [] { const char *e = getenv("FLIP_GATHER_TILE"); return (e && atoi(e) == 1) ? 1 : 0; }();
    return v;
}
extern "C" int flip_set_gather_mode(flip_ctx *ctx, int mode) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(mode == 0 || mode == 1, "flip_set_gather_mode: 0 (per-lane walk of global memory) or 1 (LDS-staged tiles)");
    ctx->gather_mode = mode;
    return 0;
}
template <bool P2G>
static hipError_t launch_gather_tile(flip_ctx *ctx, const DensityArgs &da, double *u, double *v, double *w, uint8_t *act) {
    static bool attr_set[64] = {};
    hipError_t e = hipSuccess;
    if (!attr_set[ctx->device & 63]) { // more than 64 KB of dynamic LDS has to be asked for, once per device
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gather_tile<P2G>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)GatherTileLds<P2G>::total);
        if (e != hipSuccess) return e;
        attr_set[ctx->device & 63] = true;
    }
    const int tx = (ctx->g.nx + GT_T - 1) / GT_T, ty = (ctx->g.ny + GT_T - 1) / GT_T;
    // as many workgroups as fit on the device at once (LDS: two per CU with the velocities staged, three without)
    const int grid = ((ctx->n_cu * (P2G ? 2 : 3) + 7) / 8) * 8;
    k_gather_tile<P2G><<<grid, 256, GatherTileLds<P2G>::total, ctx->stream>>>(
        flip_dims(ctx), tx, ty, ctx->tile_list, ctx->tile_scan + ctx->n_tiles, ctx->sub_start, ctx->band,
        reinterpret_cast<const double *>(ctx->prec), u, v, w, act, da, ctx->gather_stats);
    return hipGetLastError();
}
template <bool P2G>
static void launch_gather(flip_ctx *ctx, const DensityArgs &da) {
    double *u = P2G ? ctx->vel[1][0] : nullptr, *v = P2G ? ctx->vel[1][1] : nullptr, *w = P2G ? ctx->vel[1][2] : nullptr;
    uint8_t *act = P2G ? ctx->active : nullptr;
    if (ctx->gather_mode == 1) {
        if (ctx->slot_list_shape != -1 && build_gather_lists(ctx) != 0) return;   // the form was switched since the binning
        if (launch_gather_tile<P2G>(ctx, da, u, v, w, act) == hipSuccess) return;
        // fall through to the other form; the caller's launch check reports a failure of that one
    }
    const int ry = gather_column(), rz = gather_planes();
    if (ctx->slot_list_shape != ry * 16 + rz && build_gather_lists(ctx) != 0) return;   // the form was switched since the binning
    const int nyb = (ctx->g.ny + ry - 1) / ry, nzb = (ctx->g.nz + rz - 1) / rz;
    const long long nslots = (long long)ctx->g.nx * nyb * nzb;
    k_gather_clear<P2G><<<flip_grid_blocks(ctx->B), FLIP_BLOCK, 0, ctx->stream>>>(ctx->B, ctx->band, u, v, w, act, da.density);
    if (hipMemsetAsync(ctx->gather_claim, 0, 8 * sizeof(int32_t), ctx->stream) != hipSuccess) return;
    // as many workgroups as stay resident at once (the run table in LDS allows three per CU, two with four cells per thread)
    static const int per_cu_env = [] { const char *e = getenv("FLIP_GATHER_BLOCKS_PER_CU"); return e ? atoi(e) : 0; }();
    const int per_cu = per_cu_env > 0 ? per_cu_env : (ry * rz > 2 ? 2 : 3);
    const unsigned blocks = (unsigned)(((ctx->n_cu * per_cu + 7) / 8) * 8);
#define FLIP_GATHER_LAUNCH(RY, RZ, D)                                                                                          \
    k_gather<P2G, RY, RZ, D><<<blocks, FLIP_BLOCK, 0, ctx->stream>>>(flip_dims(ctx), nyb, ctx->slot_list, ctx->slot_scan + nslots, ctx->gather_claim, \
                                                                    ctx->sub_start, ctx->band, packed_pos(ctx), packed_vel(ctx), u, v, w, act, da)
    const int depth = gather_depth();
    if (ry == 2 && rz == 2) { if (depth == 1) FLIP_GATHER_LAUNCH(2, 2, 1); else if (depth == 2) FLIP_GATHER_LAUNCH(2, 2, 2); else FLIP_GATHER_LAUNCH(2, 2, 3); }
    else if (ry == 2) { if (depth == 1) FLIP_GATHER_LAUNCH(2, 1, 1); else if (depth == 2) FLIP_GATHER_LAUNCH(2, 1, 2); else FLIP_GATHER_LAUNCH(2, 1, 3); }
    else if (rz == 2) { if (depth == 1) FLIP_GATHER_LAUNCH(1, 2, 1); else if (depth == 2) FLIP_GATHER_LAUNCH(1, 2, 2); else FLIP_GATHER_LAUNCH(1, 2, 3); }
    else { if (depth == 1) FLIP_GATHER_LAUNCH(1, 1, 1); else if (depth == 2) FLIP_GATHER_LAUNCH(1, 1, 2); else FLIP_GATHER_LAUNCH(1, 1, 3); }
#undef FLIP_GATHER_LAUNCH
}

static int ensure_records(flip_ctx *ctx) {
    if (ctx->prec_valid || ctx->n == 0) return 0;
    k_pack_records<<<flip_grid_blocks(ctx->n), FLIP_BLOCK, 0, ctx->stream>>>(ctx->n, ctx->cell_start + ctx->B, particle_ptrs(ctx, ctx->cur),
                                                                            ctx->sub_slot, packed_pos(ctx), packed_vel(ctx), ctx->pcell);
    FLIP_LAUNCH_CHECK();
    ctx->prec_valid = true;
    return 0;
}

extern "C" int flip_p2g(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(ctx->binned, "flip_p2g: call flip_bin_particles first");
    FLIP_CHECK(hipSetDevice(ctx->device));
    if (int rc = ensure_records(ctx)) return rc;
    // the gather writes the back buffer, then the buffers swap (src/FluidSimulator.cpp:74,117)
    DensityArgs da{ ctx->phi_fluid, ctx->density, ctx->d_counts + 2, ctx->g.phi_fluid_bg, ctx->g.rest_density,
                    pow((double)ctx->g.dx, 3) };
    FLIP_CHECK(hipMemsetAsync(ctx->d_counts + 2, 0, 4, ctx->stream));
    launch_gather<true>(ctx, da);
    FLIP_LAUNCH_CHECK();
    ctx->density_valid = true;
    for (int a = 0; a < 3; ++a) std::swap(ctx->vel[0][a], ctx->vel[1][a]);
    return 0;
}

extern "C" int flip_gather_stats(flip_ctx *ctx, int64_t out[2]) {
    FLIP_REQUIRE(ctx && out, "flip_gather_stats: null argument");
    FLIP_CHECK(hipSetDevice(ctx->device));
    int32_t h[2] = { 0, 0 };
    FLIP_CHECK(hipMemcpyAsync(h, ctx->gather_stats, 8, hipMemcpyDeviceToHost, ctx->stream));
    FLIP_CHECK(hipStreamSynchronize(ctx->stream));
    out[0] = h[0]; out[1] = h[1];
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Staggered trilinear sampling (OpenVDB StaggeredBoxSampler as used by MacGrid::StaggeredSampler,
// headers/MacGrid.h:36-37): component c at pos + e_c/2; z, then y, then x; each step a + (b-a)*t.

__global__ void k_g2p(GridDims g, long long n, const double *__restrict__ px, const double *__restrict__ py,
                      const double *__restrict__ pz, double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz,
                      const double *__restrict__ u, const double *__restrict__ v, const double *__restrict__ w,
                      const double *__restrict__ du, const double *__restrict__ dv, const double *__restrict__ dw,
                      float ratio) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const double x = px[s], y = py[s], z = pz[s];
    const float one_minus = 1 - ratio; // float expression in the reference (src/FluidSimulator.cpp:132)
    const double pic0 = d_box_sample(g, u, x + 0.5, y, z);
    const double pic1 = d_box_sample(g, v, x, y + 0.5, z);
    const double pic2 = d_box_sample(g, w, x, y, z + 0.5);
    const double dif0 = d_box_sample(g, du, x + 0.5, y, z);
    const double dif1 = d_box_sample(g, dv, x, y + 0.5, z);
    const double dif2 = d_box_sample(g, dw, x, y, z + 0.5);
    const double f0 = vx[s] + dif0, f1 = vy[s] + dif1, f2 = vz[s] + dif2;
    vx[s] = pic0 * (double)one_minus + f0 * (double)ratio;
    vy[s] = pic1 * (double)one_minus + f1 * (double)ratio;
    vz[s] = pic2 * (double)one_minus + f2 * (double)ratio;
}

extern "C" int flip_g2p(flip_ctx *ctx, float ratio) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->n == 0) return 0;
    double **q = ctx->p[ctx->cur];
    k_g2p<<<flip_grid_blocks(ctx->n), FLIP_BLOCK, 0, ctx->stream>>>(flip_dims(ctx), ctx->n, q[0], q[1], q[2], q[3], q[4], q[5],
                                                                   ctx->vel[0][0], ctx->vel[0][1], ctx->vel[0][2],
                                                                   ctx->vel[3][0], ctx->vel[3][1], ctx->vel[3][2], ratio);
    FLIP_LAUNCH_CHECK();
    ctx->prec_valid = false; // velocities changed
    return 0;
}

__global__ void k_apply_disp(GridDims g, long long n, double *__restrict__ px, double *__restrict__ py, double *__restrict__ pz,
                             const double *__restrict__ dx, const double *__restrict__ dy, const double *__restrict__ dz) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const double x = px[s], y = py[s], z = pz[s];
    const double e0 = d_box_sample(g, dx, x + 0.5, y, z);
    const double e1 = d_box_sample(g, dy, x, y + 0.5, z);
    const double e2 = d_box_sample(g, dz, x, y, z + 0.5);
    px[s] = x + e0;
    py[s] = y + e1;
    pz[s] = z + e2;
}

extern "C" int flip_apply_displacements(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->n == 0) return 0;
    double **q = ctx->p[ctx->cur];
    k_apply_disp<<<flip_grid_blocks(ctx->n), FLIP_BLOCK, 0, ctx->stream>>>(flip_dims(ctx), ctx->n, q[0], q[1], q[2], ctx->disp[0],
                                                                          ctx->disp[1], ctx->disp[2]);
    FLIP_LAUNCH_CHECK();
    ctx->binned = false; // positions moved: the sort is stale
    ctx->density_valid = false;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Particle exchange of the z-slab decomposition (no counterpart in the reference).  A rank is authoritative
// for the particles that the last sort placed in cells of its owned planes: one contiguous run of slots,
// because the sort is by cell id and z is the slowest index.  After the particle update it (1) packs those of
// them that now lie in a neighbour's extended slab into 64-byte records (migrants and the neighbour's future
// ghost copies alike), in slot order, and (2) rebuilds its own set from the authoritative particles still inside
// its box plus the records received.  Old ghost copies are dropped: their owners send fresh ones.
// Particles in the outside bin (beyond the box in x or y) stay with the rank whose owned planes contain them.

__device__ __forceinline__ int d_plane_of(double z, int oz) { return (int)floor(z + 0.5) - oz; }

__global__ void k_flag_send(long long n, const double *__restrict__ z, const int32_t *__restrict__ cell_start, long long c_lo,
                            long long c_hi, int oz, int plane_lo, int plane_hi, int32_t *__restrict__ flag) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n) return;
    if (s == n) { flag[s] = 0; return; }
    const int pl = d_plane_of(z[s], oz);
    flag[s] = (s >= cell_start[c_lo] && s < cell_start[c_hi] && pl >= plane_lo && pl < plane_hi) ? 1 : 0;
}
__global__ void k_pack_send(long long n, const int32_t *__restrict__ scan, ParticlePtrs src, flip_prec *__restrict__ out,
                            long long capacity) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int32_t o = scan[s];
    if (scan[s + 1] == o || o >= capacity) return;
    flip_prec r;
    r.x = src.a[0][s]; r.y = src.a[1][s]; r.z = src.a[2][s]; r.vx = src.a[3][s]; r.vy = src.a[4][s]; r.vz = src.a[5][s];
    r.mass = src.mass[s]; r.pad0 = 0.f; r.pad1 = 0.0;
    out[o] = r;
}
__global__ void k_flag_keep(long long n, const double *__restrict__ z, const int32_t *__restrict__ cell_start, long long c_lo,
                            long long c_hi, long long B, int oz, int k0, int k1, int nz, int has_lower, int has_upper,
                            int32_t *__restrict__ flag) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n) return;
    if (s == n) { flag[s] = 0; return; }
    const int pl = d_plane_of(z[s], oz);
    int keep = 0;
    if (s >= cell_start[c_lo] && s < cell_start[c_hi]) keep = (pl >= 0 || !has_lower) && (pl < nz || !has_upper);
    else if (s >= cell_start[B]) keep = (pl >= k0 || !has_lower) && (pl < k1 || !has_upper);
    flag[s] = keep;
}
__global__ void k_compact_keep(long long n, const int32_t *__restrict__ scan, ParticlePtrs src, ParticlePtrs dst) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int32_t o = scan[s];
    if (scan[s + 1] == o) return;
#pragma unroll
    for (int q = 0; q < 6; ++q) dst.a[q][o] = src.a[q][s];
    dst.mass[o] = src.mass[s];
    dst.pid[o] = o;
}
__global__ void k_unpack_recv(long long m, const flip_prec *__restrict__ in, ParticlePtrs dst, long long offset) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= m) return;
    const flip_prec r = in[s];
    const long long o = offset + s;
    dst.a[0][o] = r.x; dst.a[1][o] = r.y; dst.a[2][o] = r.z; dst.a[3][o] = r.vx; dst.a[4][o] = r.vy; dst.a[5][o] = r.vz;
    dst.mass[o] = r.mass;
    dst.pid[o] = (int32_t)o;
}

extern "C" int flip_particles_pack(flip_ctx *ctx, int32_t plane_lo, int32_t plane_hi, void *device_out, int64_t capacity,
                                   int64_t *count) {
    FLIP_REQUIRE(ctx && count, "flip_particles_pack: null argument");
    FLIP_REQUIRE(ctx->slots_valid, "flip_particles_pack: no valid sort (call flip_bin_particles earlier in the substep)");
    FLIP_REQUIRE(device_out || capacity == 0, "flip_particles_pack: null buffer");
    FLIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t n = ctx->n;
    *count = 0;
    if (n == 0) return 0;
    const long long plane = (long long)ctx->g.nx * ctx->g.ny;
    const int k0 = ctx->slab_on ? ctx->slab_k0 : 0, k1 = ctx->slab_on ? ctx->slab_k1 : ctx->g.nz;
    k_flag_send<<<flip_grid_blocks(n + 1), FLIP_BLOCK, 0, st>>>(n, ctx->p[ctx->cur][2], ctx->cell_start, k0 * plane, k1 * plane,
                                                               ctx->g.oz, plane_lo, plane_hi, ctx->slot_src);
    FLIP_LAUNCH_CHECK();
    if (int rc = flip_exclusive_scan(ctx, ctx->slot_src, n + 1)) return rc;
    k_pack_send<<<flip_grid_blocks(n), FLIP_BLOCK, 0, st>>>(n, ctx->slot_src, particle_ptrs(ctx, ctx->cur), (flip_prec *)device_out,
                                                           capacity);
    FLIP_LAUNCH_CHECK();
    int32_t total = 0;
    FLIP_CHECK(hipMemcpyAsync(&total, ctx->slot_src + n, 4, hipMemcpyDeviceToHost, st));
    FLIP_CHECK(hipStreamSynchronize(st));
    FLIP_REQUIRE(total <= capacity, "flip_particles_pack: the send buffer is too small");
    *count = total;
    return 0;
}

extern "C" int flip_particles_exchange_apply(flip_ctx *ctx, const void *recv_lo, int64_t n_lo, const void *recv_hi, int64_t n_hi) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(ctx->slots_valid || ctx->n == 0, "flip_particles_exchange_apply: no valid sort");
    FLIP_REQUIRE(n_lo >= 0 && n_hi >= 0 && (recv_lo || n_lo == 0) && (recv_hi || n_hi == 0), "flip_particles_exchange_apply: bad receive buffers");
    FLIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t n = ctx->n;
    const long long plane = (long long)ctx->g.nx * ctx->g.ny;
    const int k0 = ctx->slab_on ? ctx->slab_k0 : 0, k1 = ctx->slab_on ? ctx->slab_k1 : ctx->g.nz;
    int32_t n_keep = 0;
    const int src = ctx->cur, dst = 1 - src;
    if (n > 0) {
        k_flag_keep<<<flip_grid_blocks(n + 1), FLIP_BLOCK, 0, st>>>(n, ctx->p[src][2], ctx->cell_start, k0 * plane, k1 * plane, ctx->B,
                                                                   ctx->g.oz, k0, k1, ctx->g.nz, k0 > 0 ? 1 : 0,
                                                                   k1 < ctx->g.nz ? 1 : 0, ctx->slot_src);
        FLIP_LAUNCH_CHECK();
        if (int rc = flip_exclusive_scan(ctx, ctx->slot_src, n + 1)) return rc;
        FLIP_CHECK(hipMemcpyAsync(&n_keep, ctx->slot_src + n, 4, hipMemcpyDeviceToHost, st));
        FLIP_CHECK(hipStreamSynchronize(st));
    }
    const int64_t n_new = (int64_t)n_keep + n_lo + n_hi;
    FLIP_REQUIRE(n_new <= ctx->cap, "flip_particles_exchange_apply: particle capacity exceeded (flip_particles_reserve more before the upload)");
    if (n > 0) {
        k_compact_keep<<<flip_grid_blocks(n), FLIP_BLOCK, 0, st>>>(n, ctx->slot_src, particle_ptrs(ctx, src), particle_ptrs(ctx, dst));
        FLIP_LAUNCH_CHECK();
    }
    if (n_lo > 0) {
        k_unpack_recv<<<flip_grid_blocks(n_lo), FLIP_BLOCK, 0, st>>>(n_lo, (const flip_prec *)recv_lo, particle_ptrs(ctx, dst), n_keep);
        FLIP_LAUNCH_CHECK();
    }
    if (n_hi > 0) {
        k_unpack_recv<<<flip_grid_blocks(n_hi), FLIP_BLOCK, 0, st>>>(n_hi, (const flip_prec *)recv_hi, particle_ptrs(ctx, dst), n_keep + n_lo);
        FLIP_LAUNCH_CHECK();
    }
    ctx->cur = dst;
    ctx->n = n_new;
    ctx->binned = false;
    ctx->slots_valid = false;
    ctx->prec_valid = false;
    ctx->density_valid = false;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Fluid level set from the marker particles (src/FluidDomain.cpp:50-61, src/LevelSet.cpp:17-23): the union of the
// particles' balls, phi(c) = min_p |c - p| - radius, clipped to the narrow band, then merged with the solid so that
// the surface continues into the wall (IntersectSolidFluidLevelSets, headers/LevelSet.h:83-96).  The reference
// rasterises with OpenVDB's ParticlesToLevelSet (parity unpinned); this is the exact distance to the nearest ball.
// One thread per cell searches the sorted particles shell by shell (Chebyshev distance s = 0, 1, ... in cells): a
// particle of shell s is at least s - 1/2 away, so the search stops as soon as the best distance found cannot be
// beaten by the next shell, and a cell with no particle within reach keeps the background value.

// Cells with a particle cell within `reach` cells (Chebyshev): box dilation of the occupancy, one pass per axis.
// A cell that is not marked keeps the background without a search.
__global__ void k_near_x(GridDims g, int reach, const int32_t *__restrict__ cell_start, uint8_t *__restrict__ out) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.B) return;
    int i, j, k;
    d_ijk(g, c, i, j, k);
    const int xa = i - reach > 0 ? i - reach : 0, xb = i + reach < g.nx - 1 ? i + reach : g.nx - 1;
    const long long row = c - i;
    out[c] = cell_start[row + xb + 1] > cell_start[row + xa] ? 1 : 0;
}
__global__ void k_near_axis(GridDims g, int axis, int reach, const uint8_t *__restrict__ in, uint8_t *__restrict__ out) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.B) return;
    int i, j, k;
    d_ijk(g, c, i, j, k);
    const int pos = axis == 1 ? j : k, lim = axis == 1 ? g.ny : g.nz;
    const long long stride = axis == 1 ? g.nx : (long long)g.nx * g.ny;
    uint8_t m = 0;
    for (int d = -reach; d <= reach; ++d)
        if (pos + d >= 0 && pos + d < lim) m |= in[c + d * stride];
    out[c] = m;
}

// One thread owns R cells of a y-column (the kernel is bound by the records it reads, and the shells of neighbouring
// cells overlap almost entirely).  Shell s of the column is the box [i-s, i+s] x [j0-s, j0+R-1+s] x [k-s, k+s] without
// the box of shell s-1: after it every particle within Chebyshev distance s of each of the R cells has been visited,
// so the stopping rule holds per cell, and visiting a few particles more than a cell needs does not change a minimum.
template <int R>
__device__ __forceinline__ void d_ls_range(const flip_ppos *__restrict__ pm, int32_t s0, int32_t s1, double cx, const double (&cy)[R],
                                           double cz, double (&best2)[R]) {
    for (int32_t s = s0; s < s1; ++s) {
        const double2 *rec = reinterpret_cast<const double2 *>(pm + s);
        const double2 r0 = rec[0];
        const double rz = rec[1].x;
        const double dx = cx - r0.x, dz = cz - rz;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double dy = cy[r] - r0.y;
            const double d2 = dx * dx + dy * dy + dz * dz;
            best2[r] = d2 < best2[r] ? d2 : best2[r];
        }
    }
}
// The shell search of one column of R cells up to shell `last`; returns whether every cell of the column is finished
// (its minimum cannot be beaten by a particle not yet visited).
template <int R>
__device__ __forceinline__ bool d_ls_search(const GridDims &g, int i, int j0, int k, int last, const int32_t *__restrict__ cell_start,
                                            const flip_ppos *__restrict__ pm, double (&best2)[R]) {
    const int j1 = j0 + R - 1 < g.ny - 1 ? j0 + R - 1 : g.ny - 1; // last cell of the column inside the box
    const double cx = (double)(g.ox + i), cz = (double)(g.oz + k);
    double cy[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        cy[r] = (double)(g.oy + j0 + r);
        best2[r] = 1e300;
    }
    for (int s = 0; s <= last; ++s) {
        const int ya = j0 - s, yb = j1 + s;
        for (int dz = -s; dz <= s; ++dz) {
            const int kk = k + dz;
            if (kk < 0 || kk >= g.nz) continue;
            const double bz = dz == 0 ? 0.0 : (double)(dz < 0 ? -dz : dz) - 0.5;
            for (int jj = ya; jj <= yb; ++jj) {
                if (jj < 0 || jj >= g.ny) continue;
                // Pruning: a particle of cell (ii, jj, kk) is at least (|ii - i| - 1/2, by, bz) away from every cell of the
                // column, so a cell of this row can only improve a minimum if that bound is below the largest of the
                // column's current minima: |ii - i| < 1/2 + sqrt(worst - by^2 - bz^2).  Cells far from the liquid would
                // otherwise walk every particle within their last shell; the minimum itself is unchanged (the cells
                // skipped hold no candidate).
                double worst = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) worst = (j0 + r < g.ny && best2[r] > worst) ? best2[r] : worst;
                const int dyc = jj < j0 ? j0 - jj : (jj > j1 ? jj - j1 : 0);
                const double by = dyc == 0 ? 0.0 : (double)dyc - 0.5;
                const double room = worst - by * by - bz * bz;
                if (!(room > 0.0)) continue;
                const int xr = room > 1e12 ? s : (int)(0.5 + sqrt(room) + 1e-9); // cells |ii - i| <= xr may hold a candidate
                const long long row = (long long)g.nx * ((long long)jj + (long long)g.ny * kk);
                const bool face = dz == -s || dz == s || jj == ya || jj == yb; // the whole x run lies on the shell
                if (face) {
                    const int h = xr < s ? xr : s;
                    const int xa = i - h > 0 ? i - h : 0, xb = i + h < g.nx - 1 ? i + h : g.nx - 1;
                    d_ls_range<R>(pm, cell_start[row + xa], cell_start[row + xb + 1], cx, cy, cz, best2);
                } else if (s <= xr) {
                    if (i - s >= 0) d_ls_range<R>(pm, cell_start[row + i - s], cell_start[row + i - s + 1], cx, cy, cz, best2);
                    if (s > 0 && i + s < g.nx) d_ls_range<R>(pm, cell_start[row + i + s], cell_start[row + i + s + 1], cx, cy, cz, best2);
                }
            }
        }
        // every particle not visited yet is at least s + 1/2 away from each cell of the column
        bool done = true;
#pragma unroll
        for (int r = 0; r < R; ++r) done = done && (j0 + r >= g.ny || best2[r] <= ((double)s + 0.5) * ((double)s + 0.5));
        if (done) return true;
    }
    return false;
}
__device__ __forceinline__ void d_ls_store(const GridDims &g, long long c, double best2, float vs, double radius, double half_width,
                                           const float *__restrict__ phi_solid, float *__restrict__ phi_fluid) {
    double d = half_width;
    // a particle beyond the reach of the band cannot bring the value below the band's edge
    if (best2 < 1e299) d = sqrt(best2) - radius;
    d = d < -half_width ? -half_width : (d > half_width ? half_width : d);
    float f = (float)(d * (double)vs);
    const float ps = phi_solid[c];
    if (ps <= 0) f = f > -ps ? f : -ps;
    phi_fluid[c] = f;
}

// First pass, every cell: shells 0 and 1 only.  Inside the liquid that settles the cell (a particle of the cell itself
// or of a neighbour is nearer than any unvisited one); the cells that are not settled -- air and wall cells up to
// `reach` cells away from the liquid, a thin layer, scattered a few to a wave -- are flagged for the second pass, which
// runs over their compacted list with full waves instead of leaving 60 lanes of a wave idle behind 4 long searches.
template <int R>
__global__ void k_fluid_levelset_near(GridDims g, int nyb, float vs, double radius, double half_width, int reach,
                                      const int32_t *__restrict__ cell_start, const flip_ppos *__restrict__ pm,
                                      const uint8_t *__restrict__ near, const float *__restrict__ phi_solid,
                                      float *__restrict__ phi_fluid, int32_t *__restrict__ far_flag) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) far_flag[g.B] = 0; // terminator of the scan
    if (t >= (long long)g.nx * nyb * g.nz) return;
    const int i = (int)(t % g.nx);
    const long long tq = t / g.nx;
    const int j0 = (int)(tq % nyb) * R, k = (int)(tq / nyb);
    double best2[R];
    bool search = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        best2[r] = 1e300;
        if (j0 + r < g.ny) search = search || near[d_lid(g, i, j0 + r, k)];
    }
    bool done = true;
    if (search) done = d_ls_search<R>(g, i, j0, k, reach < 1 ? reach : 1, cell_start, pm, best2) || reach <= 1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (j0 + r >= g.ny) continue;
        const long long c = d_lid(g, i, j0 + r, k);
        // a cell of an unfinished column may itself be finished: what shell 1 leaves is final if it is within 1.5
        const bool settled = done || best2[r] <= 2.25;
        far_flag[c] = settled ? 0 : 1;
        if (settled) d_ls_store(g, c, best2[r], vs, radius, half_width, phi_solid, phi_fluid);
    }
}
__global__ void k_fluid_levelset_list(long long B, const int32_t *__restrict__ scan, int32_t *__restrict__ list) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    if (scan[c + 1] > scan[c]) list[scan[c]] = (int32_t)c;
}
// Second pass: one thread per flagged cell, the whole search.
__global__ void k_fluid_levelset_far(GridDims g, float vs, double radius, double half_width, int reach, const int32_t *__restrict__ cell_start,
                                     const flip_ppos *__restrict__ pm, const int32_t *__restrict__ list, const int32_t *__restrict__ n_list,
                                     const float *__restrict__ phi_solid, float *__restrict__ phi_fluid) {
    const long long n = *n_list;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
        const long long c = list[q];
        int i, j, k;
        d_ijk(g, c, i, j, k);
        double best2[1];
        d_ls_search<1>(g, i, j, k, reach, cell_start, pm, best2);
        d_ls_store(g, c, best2[0], vs, radius, half_width, phi_solid, phi_fluid);
    }
}

extern "C" int flip_fluid_levelset_from_particles(flip_ctx *ctx, float radius_world, float half_width_voxels) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(radius_world > 0.f && half_width_voxels > 0.f, "flip_fluid_levelset_from_particles: bad radius or band");
    FLIP_REQUIRE(ctx->binned, "flip_fluid_levelset_from_particles: call flip_bin_particles first");
    FLIP_CHECK(hipSetDevice(ctx->device));
    if (int rc = ensure_records(ctx)) return rc;
    const double radius = (double)radius_world / (double)ctx->g.dx; // index units
    const int reach = (int)ceil(radius + (double)half_width_voxels);
    const GridDims g = flip_dims(ctx);
    const int nb = flip_grid_blocks(ctx->B);
    // scratch: both byte arrays are rewritten by the face-fraction stage before anything reads them
    k_near_x<<<nb, FLIP_BLOCK, 0, ctx->stream>>>(g, reach, ctx->cell_start, ctx->tmp_u8);
    k_near_axis<<<nb, FLIP_BLOCK, 0, ctx->stream>>>(g, 1, reach, ctx->tmp_u8, ctx->seed);
    k_near_axis<<<nb, FLIP_BLOCK, 0, ctx->stream>>>(g, 2, reach, ctx->seed, ctx->tmp_u8);
    constexpr int R = 3; // cells per thread in the first pass
    const int nyb = (ctx->g.ny + R - 1) / R;
    if (!ctx->ls_list) FLIP_CHECK(hipMalloc((void **)&ctx->ls_list, (size_t)ctx->B * 4));
    k_fluid_levelset_near<R><<<flip_grid_blocks((long long)ctx->g.nx * nyb * ctx->g.nz), FLIP_BLOCK, 0, ctx->stream>>>(
        g, nyb, ctx->g.dx, radius, (double)half_width_voxels, reach, ctx->cell_start, ctx->pcell, ctx->tmp_u8, ctx->phi_solid,
        ctx->phi_fluid, ctx->cg_scan);
    FLIP_LAUNCH_CHECK();
    if (reach > 1) {
        // cg_scan is scratch of the system assembly, free here
        if (int rc = flip_exclusive_scan(ctx, ctx->cg_scan, ctx->B + 1)) return rc;
        k_fluid_levelset_list<<<nb, FLIP_BLOCK, 0, ctx->stream>>>(ctx->B, ctx->cg_scan, ctx->ls_list);
        k_fluid_levelset_far<<<2048, FLIP_BLOCK, 0, ctx->stream>>>(g, ctx->g.dx, radius, (double)half_width_voxels, reach, ctx->cell_start,
                                                                  ctx->pcell, ctx->ls_list, ctx->cg_scan + ctx->B, ctx->phi_solid,
                                                                  ctx->phi_fluid);
    }
    FLIP_LAUNCH_CHECK();
    ctx->density_valid = false;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Particle advection (src/FluidDomain.cpp:63-75, src/ParticleSet.cpp:18-45): third-order Runge-Kutta
// through the staggered velocity field, then particles that ended inside a solid are moved to the solid's
// surface.  The reference does both with OpenVDB (PointAdvect order 3 and a closest-point-transform grid),
// neither of which is available to check against: the integrator below is the published RK3 scheme of
// openvdb::tools::VelocityIntegrator (P + dt (V0 + 4 V1 + V2) / 6 with V1 at P + dt/2 V0 and V2 at
// P + dt (2 V1 - V0)), and the push-out follows the level set's gradient for its own value, which equals
// the closest-point transform wherever the level set is a signed distance.

__device__ __forceinline__ float d_lerpf(float a, float b, double t) {
    const double tmp = (double)(b - a) * t;
    return a + (float)tmp;
}
__device__ float d_box_sample_f(const GridDims &g, const float *__restrict__ f, float bg, double x, double y, double z) {
    const double fx = floor(x), fy = floor(y), fz = floor(z);
    const int i = (int)fx - g.ox, j = (int)fy - g.oy, k = (int)fz - g.oz;
    const double tu = x - fx, tv = y - fy, tw = z - fz;
    const float d000 = d_getf(g, f, i, j, k, bg), d100 = d_getf(g, f, i + 1, j, k, bg);
    const float d010 = d_getf(g, f, i, j + 1, k, bg), d110 = d_getf(g, f, i + 1, j + 1, k, bg);
    const float d001 = d_getf(g, f, i, j, k + 1, bg), d101 = d_getf(g, f, i + 1, j, k + 1, bg);
    const float d011 = d_getf(g, f, i, j + 1, k + 1, bg), d111 = d_getf(g, f, i + 1, j + 1, k + 1, bg);
    return d_lerpf(d_lerpf(d_lerpf(d000, d001, tw), d_lerpf(d010, d011, tw), tv),
                   d_lerpf(d_lerpf(d100, d101, tw), d_lerpf(d110, d111, tw), tv), tu);
}
__device__ __forceinline__ float d_solid_sample(const GridDims &g, const float *phi_s, float bg_s, const float *phi_o,
                                                float bg_o, double x, double y, double z) {
    float v = d_box_sample_f(g, phi_s, bg_s, x, y, z);
    if (phi_o) {
        const float o = d_box_sample_f(g, phi_o, bg_o, x, y, z);
        v = o < v ? o : v; // csgUnion of the container and the rigid body
    }
    return v;
}

__global__ void k_advect(GridDims g, long long n, double dt_over_dx, double inv_dx, double *__restrict__ px,
                         double *__restrict__ py, double *__restrict__ pz, const double *__restrict__ u,
                         const double *__restrict__ v, const double *__restrict__ w, const float *__restrict__ phi_s,
                         float bg_s, const float *__restrict__ phi_o, float bg_o) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const double x = px[s], y = py[s], z = pz[s];
    const double a0 = d_box_sample(g, u, x + 0.5, y, z), a1 = d_box_sample(g, v, x, y + 0.5, z), a2 = d_box_sample(g, w, x, y, z + 0.5);
    const double h = 0.5 * dt_over_dx;
    const double x1 = x + h * a0, y1 = y + h * a1, z1 = z + h * a2;
    const double b0 = d_box_sample(g, u, x1 + 0.5, y1, z1), b1 = d_box_sample(g, v, x1, y1 + 0.5, z1), b2 = d_box_sample(g, w, x1, y1, z1 + 0.5);
    const double x2 = x + dt_over_dx * (2.0 * b0 - a0), y2 = y + dt_over_dx * (2.0 * b1 - a1), z2 = z + dt_over_dx * (2.0 * b2 - a2);
    const double c0 = d_box_sample(g, u, x2 + 0.5, y2, z2), c1 = d_box_sample(g, v, x2, y2 + 0.5, z2), c2 = d_box_sample(g, w, x2, y2, z2 + 0.5);
    double nx_ = x + dt_over_dx * (a0 + 4.0 * b0 + c0) / 6.0;
    double ny_ = y + dt_over_dx * (a1 + 4.0 * b1 + c1) / 6.0;
    double nz_ = z + dt_over_dx * (a2 + 4.0 * b2 + c2) / 6.0;
    const float phi = d_solid_sample(g, phi_s, bg_s, phi_o, bg_o, nx_, ny_, nz_);
    if (phi <= 0.f) {
        const double gx = (double)d_solid_sample(g, phi_s, bg_s, phi_o, bg_o, nx_ + 0.5, ny_, nz_)
                          - (double)d_solid_sample(g, phi_s, bg_s, phi_o, bg_o, nx_ - 0.5, ny_, nz_);
        const double gy = (double)d_solid_sample(g, phi_s, bg_s, phi_o, bg_o, nx_, ny_ + 0.5, nz_)
                          - (double)d_solid_sample(g, phi_s, bg_s, phi_o, bg_o, nx_, ny_ - 0.5, nz_);
        const double gz = (double)d_solid_sample(g, phi_s, bg_s, phi_o, bg_o, nx_, ny_, nz_ + 0.5)
                          - (double)d_solid_sample(g, phi_s, bg_s, phi_o, bg_o, nx_, ny_, nz_ - 0.5);
        const double len = sqrt(gx * gx + gy * gy + gz * gz);
        if (len > 0.0) {
            const double d = -(double)phi * inv_dx / len; // distance to the surface in index units
            nx_ += d * gx; ny_ += d * gy; nz_ += d * gz;
        }
    }
    px[s] = nx_; py[s] = ny_; pz[s] = nz_;
}

extern "C" int flip_advect_particles(flip_ctx *ctx, float dt) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->n == 0) return 0;
    double **q = ctx->p[ctx->cur];
    const double inv_dx = 1.0 / (double)ctx->g.dx;
    k_advect<<<flip_grid_blocks(ctx->n), FLIP_BLOCK, 0, ctx->stream>>>(
        flip_dims(ctx), ctx->n, (double)dt * inv_dx, inv_dx, q[0], q[1], q[2], ctx->vel[0][0], ctx->vel[0][1], ctx->vel[0][2],
        ctx->phi_solid, ctx->g.phi_solid_bg, ctx->has_obj ? ctx->phi_obj : nullptr, ctx->phi_obj_bg);
    FLIP_LAUNCH_CHECK();
    ctx->binned = false;
    ctx->density_valid = false;
    ctx->prec_valid = false;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// The particle update of a substep in one pass: transfer_from_grid_to_particles, FluidDomain::advectParticles and
// apply_particle_displacements (src/FluidSimulator.cpp:420, :426, :429) follow each other without anything in between
// that touches the particles, every one of them is a loop over independent particles, and the three kernels above spend
// their time reading and writing the same positions.  This kernel evaluates the same expressions in the same order on
// every particle -- the values handed from one stage to the next stay in registers instead of going through memory, and
// the first velocity sample of the integrator is the PIC sample of the transfer (same grid, same point) -- so the result
// is the three calls' result bit for bit.
template <bool ADVECT, bool DISPLACE>
__global__ void __launch_bounds__(FLIP_BLOCK) k_update_particles(GridDims g, long long n, double *__restrict__ px, double *__restrict__ py, double *__restrict__ pz,
                                   double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz,
                                   const double *__restrict__ u, const double *__restrict__ v, const double *__restrict__ w,
                                   const double *__restrict__ du, const double *__restrict__ dv, const double *__restrict__ dw,
                                   const double *__restrict__ ex, const double *__restrict__ ey, const double *__restrict__ ez,
                                   float ratio, double dt_over_dx, double inv_dx, const float *__restrict__ phi_s, float bg_s,
                                   const float *__restrict__ phi_o, float bg_o) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    double x = px[s], y = py[s], z = pz[s];
    // k_g2p
    const float one_minus = 1 - ratio;
    const double pic0 = d_box_sample(g, u, x + 0.5, y, z);
    const double pic1 = d_box_sample(g, v, x, y + 0.5, z);
    const double pic2 = d_box_sample(g, w, x, y, z + 0.5);
    const double dif0 = d_box_sample(g, du, x + 0.5, y, z);
    const double dif1 = d_box_sample(g, dv, x, y + 0.5, z);
    const double dif2 = d_box_sample(g, dw, x, y, z + 0.5);
    const double f0 = vx[s] + dif0, f1 = vy[s] + dif1, f2 = vz[s] + dif2;
    vx[s] = pic0 * (double)one_minus + f0 * (double)ratio;
    vy[s] = pic1 * (double)one_minus + f1 * (double)ratio;
    vz[s] = pic2 * (double)one_minus + f2 * (double)ratio;
    if (ADVECT) { // k_advect
        const double a0 = pic0, a1 = pic1, a2 = pic2;
        const double h = 0.5 * dt_over_dx;
        const double x1 = x + h * a0, y1 = y + h * a1, z1 = z + h * a2;
        const double b0 = d_box_sample(g, u, x1 + 0.5, y1, z1), b1 = d_box_sample(g, v, x1, y1 + 0.5, z1), b2 = d_box_sample(g, w, x1, y1, z1 + 0.5);
        const double x2 = x + dt_over_dx * (2.0 * b0 - a0), y2 = y + dt_over_dx * (2.0 * b1 - a1), z2 = z + dt_over_dx * (2.0 * b2 - a2);
        const double c0 = d_box_sample(g, u, x2 + 0.5, y2, z2), c1 = d_box_sample(g, v, x2, y2 + 0.5, z2), c2 = d_box_sample(g, w, x2, y2, z2 + 0.5);
        double nx_ = x + dt_over_dx * (a0 + 4.0 * b0 + c0) / 6.0;
        double ny_ = y + dt_over_dx * (a1 + 4.0 * b1 + c1) / 6.0;
        double nz_ = z + dt_over_dx * (a2 + 4.0 * b2 + c2) / 6.0;
        const float phi = d_solid_sample(g, phi_s, bg_s, phi_o, bg_o, nx_, ny_, nz_);
        if (phi <= 0.f) {
            const double gx = (double)d_solid_sample(g, phi_s, bg_s, phi_o, bg_o, nx_ + 0.5, ny_, nz_)
                              - (double)d_solid_sample(g, phi_s, bg_s, phi_o, bg_o, nx_ - 0.5, ny_, nz_);
            const double gy = (double)d_solid_sample(g, phi_s, bg_s, phi_o, bg_o, nx_, ny_ + 0.5, nz_)
                              - (double)d_solid_sample(g, phi_s, bg_s, phi_o, bg_o, nx_, ny_ - 0.5, nz_);
            const double gz = (double)d_solid_sample(g, phi_s, bg_s, phi_o, bg_o, nx_, ny_, nz_ + 0.5)
                              - (double)d_solid_sample(g, phi_s, bg_s, phi_o, bg_o, nx_, ny_, nz_ - 0.5);
            const double len = sqrt(gx * gx + gy * gy + gz * gz);
            if (len > 0.0) {
                const double d = -(double)phi * inv_dx / len;
                nx_ += d * gx; ny_ += d * gy; nz_ += d * gz;
            }
        }
        x = nx_; y = ny_; z = nz_;
    }
    if (DISPLACE) { // k_apply_disp
        const double e0 = d_box_sample(g, ex, x + 0.5, y, z);
        const double e1 = d_box_sample(g, ey, x, y + 0.5, z);
        const double e2 = d_box_sample(g, ez, x, y, z + 0.5);
        x = x + e0; y = y + e1; z = z + e2;
    }
    if (ADVECT || DISPLACE) { px[s] = x; py[s] = y; pz[s] = z; }
}

extern "C" int flip_update_particles(flip_ctx *ctx, float flip_pic_ratio, float dt, int advect, int displace) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(!advect || dt > 0.f, "flip_update_particles: dt must be positive");
    FLIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->n == 0) return 0;
    double **q = ctx->p[ctx->cur];
    const double inv_dx = 1.0 / (double)ctx->g.dx;
#define FLIP_UPDATE_LAUNCH(A, D)                                                                                                      \
    k_update_particles<A, D><<<flip_grid_blocks(ctx->n), FLIP_BLOCK, 0, ctx->stream>>>(                                               \
        flip_dims(ctx), ctx->n, q[0], q[1], q[2], q[3], q[4], q[5], ctx->vel[0][0], ctx->vel[0][1], ctx->vel[0][2], ctx->vel[3][0],   \
        ctx->vel[3][1], ctx->vel[3][2], ctx->disp[0], ctx->disp[1], ctx->disp[2], flip_pic_ratio, (double)dt * inv_dx, inv_dx,        \
        ctx->phi_solid, ctx->g.phi_solid_bg, ctx->has_obj ? ctx->phi_obj : nullptr, ctx->phi_obj_bg)
    if (advect && displace) FLIP_UPDATE_LAUNCH(true, true);
    else if (advect) FLIP_UPDATE_LAUNCH(true, false);
    else if (displace) FLIP_UPDATE_LAUNCH(false, true);
    else FLIP_UPDATE_LAUNCH(false, false);
#undef FLIP_UPDATE_LAUNCH
    FLIP_LAUNCH_CHECK();
    ctx->prec_valid = false; // velocities changed
    if (advect || displace) {
        ctx->binned = false; // positions moved: the sort is stale
        ctx->density_valid = false;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Density alone (src/FluidSimulator.cpp:625-652, src/Functors.cpp:216-266): the density-only form of k_gather.

extern "C" int flip_density(flip_ctx *ctx, int64_t *n_overdense) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(ctx->binned, "flip_density: call flip_bin_particles first");
    FLIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->density_valid) {
        if (int rc = ensure_records(ctx)) return rc;
        DensityArgs da{ ctx->phi_fluid, ctx->density, ctx->d_counts + 2, ctx->g.phi_fluid_bg, ctx->g.rest_density,
                        pow((double)ctx->g.dx, 3) };
        FLIP_CHECK(hipMemsetAsync(ctx->d_counts + 2, 0, 4, ctx->stream));
        launch_gather<false>(ctx, da);
        FLIP_LAUNCH_CHECK();
        ctx->density_valid = true;
    }
    if (n_overdense) {
        int32_t v = 0;
        FLIP_CHECK(hipMemcpyAsync(&v, ctx->d_counts + 2, 4, hipMemcpyDeviceToHost, ctx->stream));
        FLIP_CHECK(hipStreamSynchronize(ctx->stream));
        *n_overdense = v;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Redistribution of the particles of over-dense cells (src/Functors.cpp:268-322): a cell whose density exceeds 1.5 rest
// density has its n particles spread over the f^3 sub-cells of a lattice, f = floor(pow(n, 1./3.)) (:288), in a
// shuffled order, each at a uniform random offset inside its sub-cell.  That expression is the largest integer with
// f^3 <= n except at the perfect cubes n = k^3 with k >= 4, where the correctly rounded power with the exponent 1./3.
// (below a third) falls just short of k and the floor gives k - 1; it is restated in integers here (checked against the
// C library's pow for every n below 2^31 that is a cube, and for all n up to 2 * 10^5).  The reference draws from a random_device-seeded
// global generator inside a TBB loop, so it does not reproduce itself; here every draw is the stateless hash of
// headers/util.h:259-266 (randhash) keyed on (seed, cell, draw number), and every density is the value computed before
// any particle moved.  One thread per cell; the shuffled list lives in the cell's own slots of a scratch array.
__device__ __forceinline__ uint32_t d_randhash(uint32_t seed) {
    uint32_t i = (seed ^ 0xA3C59AC3u) * 2654435769u;
    i ^= (i >> 16);
    i *= 2654435769u;
    i ^= (i >> 16);
    i *= 2654435769u;
    return i;
}
__device__ __forceinline__ double d_redist_uniform(uint32_t base, uint32_t q) {
    const uint32_t hi = d_randhash(base + 0x10000000u + 2u * q) >> 5, lo = d_randhash(base + 0x10000000u + 2u * q + 1u) >> 6;
    return ((double)hi * 67108864.0 + (double)lo) * (1.0 / 9007199254740992.0);
}
__global__ void k_redistribute(GridDims g, float rest, uint32_t seed, const float *__restrict__ density, const int32_t *__restrict__ cell_start,
                               int32_t *__restrict__ list_scratch, double *__restrict__ px, double *__restrict__ py,
                               double *__restrict__ pz, int32_t *__restrict__ counters) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.B) return;
    if (!((double)density[c] > 1.5 * (double)rest)) return;
    const int32_t s0 = cell_start[c], m = cell_start[c + 1] - s0;
    if (m == 0) return;
    int f = 1;
    while ((long long)(f + 1) * (f + 1) * (f + 1) <= m) ++f;
    if (f >= 4 && f * f * f == m) --f;
    const int32_t sub = f * f * f;
    int32_t *list = list_scratch + s0; // sub <= m entries
    for (int32_t a = 0; a < sub; ++a) list[a] = a;
    const uint32_t base = d_randhash(d_randhash(seed) ^ (uint32_t)c);
    for (int32_t a = sub - 1; a >= 1; --a) {
        const int32_t b = (int32_t)(d_randhash(base + (uint32_t)a) % (uint32_t)(a + 1));
        const int32_t t = list[a]; list[a] = list[b]; list[b] = t;
    }
    int i, j, k;
    d_ijk(g, c, i, j, k);
    const double cx = (double)(g.ox + i) - 0.5, cy = (double)(g.oy + j) - 0.5, cz = (double)(g.oz + k) - 0.5;
    const double fd = (double)f;
    for (int32_t q = 0; q < m; ++q) {
        const int32_t idx = list[q % sub];
        const int sx = idx / (f * f), sy = (idx / f) % f, sz = idx % f; // linearTo3d, headers/util.h:451-456
        px[s0 + q] = (cx + (double)sx / fd) + d_redist_uniform(base, 3u * (uint32_t)q) / fd;
        py[s0 + q] = (cy + (double)sy / fd) + d_redist_uniform(base, 3u * (uint32_t)q + 1u) / fd;
        pz[s0 + q] = (cz + (double)sz / fd) + d_redist_uniform(base, 3u * (uint32_t)q + 2u) / fd;
    }
    atomicAdd(counters, 1);
    atomicAdd(counters + 1, m);
}

extern "C" int flip_redistribute_overdense(flip_ctx *ctx, uint32_t seed, int64_t *n_cells, int64_t *n_particles) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(ctx->binned && ctx->slots_valid, "flip_redistribute_overdense: call flip_bin_particles first (the slots must describe the cells)");
    FLIP_REQUIRE(ctx->density_valid, "flip_redistribute_overdense: call flip_p2g or flip_density first");
    FLIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (ctx->n > 0) {
        FLIP_CHECK(hipMemsetAsync(ctx->cell_fill, 0, 8, st)); // scratch of the sort, free between sorts: [0] cells, [1] particles
        double **q = ctx->p[ctx->cur];
        k_redistribute<<<flip_grid_blocks(ctx->B), FLIP_BLOCK, 0, st>>>(flip_dims(ctx), ctx->g.rest_density, seed, ctx->density, ctx->cell_start,
                                                                       ctx->slot_src, q[0], q[1], q[2], ctx->cell_fill);
        FLIP_LAUNCH_CHECK();
    }
    int32_t h[2] = { 0, 0 };
    if (ctx->n > 0) {
        FLIP_CHECK(hipMemcpyAsync(h, ctx->cell_fill, 8, hipMemcpyDeviceToHost, st));
        FLIP_CHECK(hipStreamSynchronize(st));
    }
    if (n_cells) *n_cells = h[0];
    if (n_particles) *n_particles = h[1];
    if (h[1] > 0) {
        // positions changed inside their cells: the packed copies and the half-cell order are stale, the cell ranges are not
        ctx->binned = false;
        ctx->prec_valid = false;
        ctx->density_valid = false;
    }
    return 0;
}
