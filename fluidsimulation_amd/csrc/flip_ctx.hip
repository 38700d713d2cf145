// flip_ctx.hip -- context life cycle, dense-field and particle transfers, the integer scan.
#include "flip_internal.h"

#include <utility>

#include <algorithm>

static std::string g_create_error;

int flip_fail(flip_ctx *ctx, const char *what, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s (%s:%d)", what, file, line);
    if (ctx) ctx->err = buf;
    else g_create_error = buf;
    return 1;
}
int flip_fail_hip(flip_ctx *ctx, hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) in %s (%s:%d)", (int)e, hipGetErrorString(e), what, file, line);
    if (ctx) ctx->err = buf;
    else g_create_error = buf;
    return 2;
}

extern "C" const char *flip_last_error(const flip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

hipError_t flip_zero_new(void *p, size_t bytes) {
    hipError_t e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return e;
}

template <class T>
static hipError_t dalloc(T **p, int64_t count) {
    hipError_t e = hipMalloc((void **)p, (size_t)(count > 0 ? count : 1) * sizeof(T));
    if (e != hipSuccess) { *p = nullptr; return e; }
    e = flip_zero_new(*p, (size_t)(count > 0 ? count : 1) * sizeof(T));
    if (e != hipSuccess) { (void)hipFree(*p); *p = nullptr; } // callers offset or free only pointers that are fully set up
    return e;
}
template <class T>
static void dfree(T *&p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

static void free_particles(flip_ctx *c) {
    for (int b = 0; b < 2; ++b) {
        for (int a = 0; a < 6; ++a) dfree(c->p[b][a]);
        dfree(c->mass[b]);
        dfree(c->pid[b]);
    }
    dfree(c->cell_id);
    dfree(c->slot_src);
    dfree(c->slot_key);
    dfree(c->stage);
    dfree(c->prec);
    dfree(c->pcell);
    dfree(c->sub_slot);
    c->cap = 0;
}

// The arrays of one assembled system and its solve, allocated into the context's active fields.
int flip_alloc_system(flip_ctx *c) {
    flip_ctx *ctx = c;
    const int64_t B = c->B;
    hipError_t e = hipSuccess;
#define A(ptr, count)                                  \
    if (e == hipSuccess) e = dalloc(&(ptr), (count));
    for (int a = 0; a < 3; ++a) A(c->term[a], B)
    A(c->flags, B) A(c->diag, B) A(c->invdiag, B) A(c->rhs, B)
    {
        // the seven solver vectors are one allocation (x first), so that one fill zeroes them all before a solve
        // (flip_zero_solver_vectors); x - pad is the pointer hipMalloc returned
        double **vec[] = { &c->x, &c->r, &c->r2, &c->z, &c->pv[0], &c->pv[1], &c->ap };
        const int64_t span = B + 2 * c->pad;
        double *slab = nullptr;
        A(slab, 7 * span)
        for (int q = 0; q < 7; ++q) *vec[q] = e == hipSuccess ? slab + q * span + c->pad : nullptr;
        double **dp[] = { &c->cg_diag, &c->cg_inv };
        for (double **q : dp) {
            A(*q, B + 2 * c->pad)
            if (e == hipSuccess) *q += c->pad;
        }
        for (int a = 0; a < 3; ++a) {
            A(c->cg_c[a], B + 2 * c->pad)
            if (e == hipSuccess) c->cg_c[a] += c->pad;
        }
    }
    A(c->cg_rows, B) A(c->reg, B)
    A(c->chunk_flag, c->n_chunks + 1) A(c->chunk_list, c->n_chunks + 1)
    A(c->d_counts, 8) A(c->partial, 16 * FLIP_MAX_PCG_BLOCKS) A(c->scal, 1)
#undef A
    if (e != hipSuccess) return flip_fail_hip(ctx, e, "hipMalloc of the system arrays", __FILE__, __LINE__);
    return 0;
}

// x (unless keep_x), r, r2, z, both direction buffers and Ap to zero, pads included, with one fill
int flip_zero_solver_vectors(flip_ctx *ctx, bool keep_x, hipStream_t st) {
    const int64_t span = ctx->B + 2 * ctx->pad;
    double *from = (keep_x ? ctx->r : ctx->x) - ctx->pad;
    FLIP_CHECK(hipMemsetAsync(from, 0, (size_t)(keep_x ? 6 : 7) * (size_t)span * 8, st));
    return 0;
}

void flip_free_system(flip_ctx *c) {
    flip_mg_free(c);
    for (int a = 0; a < 3; ++a) dfree(c->term[a]);
    dfree(c->flags); dfree(c->diag); dfree(c->invdiag); dfree(c->rhs);
    {   // padded solver arrays: the stored pointers are offset by c->pad
        if (c->x) (void)hipFree(c->x - c->pad);   // the seven vectors are one allocation
        c->x = c->r = c->r2 = c->z = c->pv[0] = c->pv[1] = c->ap = nullptr;
        double **dp[] = { &c->cg_diag, &c->cg_inv };
        for (double **q : dp)
            if (*q) { (void)hipFree(*q - c->pad); *q = nullptr; }
        for (int a = 0; a < 3; ++a)
            if (c->cg_c[a]) { (void)hipFree(c->cg_c[a] - c->pad); c->cg_c[a] = nullptr; }
    }
    dfree(c->cg_rows); dfree(c->reg);
    dfree(c->chunk_flag); dfree(c->chunk_list); dfree(c->d_counts); dfree(c->partial); dfree(c->scal);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    c->events.clear();
    for (hipGraphExec_t &e : c->batch_exec)
        if (e) { (void)hipGraphExecDestroy(e); e = nullptr; }
    // the slab solve's arrays of this set
    if (c->mg_cz) { (void)hipFree(c->mg_cz - c->pad); c->mg_cz = nullptr; }
    dfree(c->dist_sums); dfree(c->mg_rows); dfree(c->dist_recv[0]); dfree(c->dist_recv[1]);
    if (c->dist_rw) { (void)hipFree(c->dist_rw - c->pad); c->dist_rw = nullptr; }
}

void flip_swap_systems(flip_ctx *c) {
    flip_system_set &o = c->other;
    std::swap(c->sys_kind, o.sys_kind); std::swap(c->x_kind, o.x_kind); std::swap(c->sys_dt, o.sys_dt); std::swap(c->sys_s2, o.sys_s2);
    std::swap(c->sys_coupled, o.sys_coupled); std::swap(c->sys_eliminated, o.sys_eliminated);
    for (int a = 0; a < 3; ++a) { std::swap(c->term[a], o.term[a]); std::swap(c->cg_c[a], o.cg_c[a]); }
    std::swap(c->flags, o.flags); std::swap(c->diag, o.diag); std::swap(c->invdiag, o.invdiag); std::swap(c->rhs, o.rhs);
    std::swap(c->x, o.x); std::swap(c->r, o.r); std::swap(c->r2, o.r2); std::swap(c->z, o.z); std::swap(c->pv[0], o.pv[0]); std::swap(c->pv[1], o.pv[1]);
    std::swap(c->ap, o.ap); std::swap(c->cg_diag, o.cg_diag); std::swap(c->cg_inv, o.cg_inv);
    std::swap(c->reg, o.reg); std::swap(c->reg_value, o.reg_value);
    std::swap(c->cg_rows, o.cg_rows); std::swap(c->chunk_flag, o.chunk_flag); std::swap(c->chunk_list, o.chunk_list);
    std::swap(c->d_counts, o.d_counts); std::swap(c->n_active_chunks, o.n_active_chunks);
    std::swap(c->partial, o.partial); std::swap(c->scal, o.scal);
    c->events.swap(o.events); c->mg.swap(o.mg);
    std::swap(c->mg_diag_f, o.mg_diag_f); std::swap(c->mg_inv_f, o.mg_inv_f); std::swap(c->mg_narrowed, o.mg_narrowed);
    for (int q = 0; q < 2; ++q) { std::swap(c->batch_exec[q], o.batch_exec[q]); std::swap(c->batch_key[q], o.batch_key[q]); }
    std::swap(c->cg_scan, o.cg_scan); std::swap(c->scan_tmp, o.scan_tmp); std::swap(c->scan_tmp_len, o.scan_tmp_len);
    std::swap(c->mg_cz, o.mg_cz); std::swap(c->dist_sums, o.dist_sums); std::swap(c->dist_active, o.dist_active);
    std::swap(c->dist_mg, o.dist_mg); std::swap(c->dist_max_iter, o.dist_max_iter); std::swap(c->dist_prof, o.dist_prof);
    std::swap(c->mg_rows, o.mg_rows); std::swap(c->dist_recv[0], o.dist_recv[0]); std::swap(c->dist_recv[1], o.dist_recv[1]);
    std::swap(c->dist_recv_planes, o.dist_recv_planes); std::swap(c->dist_rw, o.dist_rw);
}

extern "C" void flip_ctx_destroy(flip_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->side_selected) { c->stream = c->side_main_stream; c->side_selected = false; }
    if (c->side_stream) {
        (void)hipStreamSynchronize(c->side_stream);
        (void)hipStreamDestroy(c->side_stream);
    }
    for (hipEvent_t e : c->side_event)
        if (e) (void)hipEventDestroy(e);
    (void)hipStreamSynchronize(c->stream);
    free_particles(c);
    flip_free_system(c);
    if (c->other.allocated) {
        flip_swap_systems(c);
        flip_free_system(c);
    }
    for (int q = 0; q < 2; ++q)
        if (c->pair_stream[q]) (void)hipStreamDestroy(c->pair_stream[q]);
    for (int q = 0; q < 3; ++q)
        if (c->pair_event[q]) (void)hipEventDestroy(c->pair_event[q]);
    for (hipEvent_t e : c->kprof)
        if (e) (void)hipEventDestroy(e);
    dfree(c->cell_start); dfree(c->cell_fill); dfree(c->sub_start); dfree(c->scan_tmp); dfree(c->ls_list);
    dfree(c->phi_fluid); dfree(c->phi_solid); dfree(c->phi_obj);
    for (int b = 0; b < 4; ++b)
        for (int a = 0; a < 3; ++a) dfree(c->vel[b][a]);
    for (int a = 0; a < 3; ++a) {
        dfree(c->disp[a]); dfree(c->wgt[a]); dfree(c->fwgt[a]); dfree(c->owgt[a]); dfree(c->wgt_back[a]); dfree(c->owgt_back[a]);
        dfree(c->valid[0][a]); dfree(c->valid[1][a]);
    }
    for (int a = 0; a < 6; ++a) dfree(c->J[a]);
    dfree(c->density); dfree(c->active); dfree(c->active_back); dfree(c->label); dfree(c->band);
    dfree(c->seed); dfree(c->tmp_u8); dfree(c->row_index);
    dfree(c->cg_scan); dfree(c->other.cg_scan); dfree(c->other.scan_tmp);
    dfree(c->d_vmax);
    dfree(c->gather_stats); dfree(c->tile_scan); dfree(c->tile_list); dfree(c->slot_scan); dfree(c->slot_list); dfree(c->gather_claim);
    delete c;
}

extern "C" int flip_ctx_create(const flip_grid_desc *desc, int device, void *stream, flip_ctx **out) {
    flip_ctx *ctx = nullptr; // so that the macros record into the creation error
    FLIP_REQUIRE(desc && out, "flip_ctx_create: null argument");
    FLIP_REQUIRE(desc->nx > 0 && desc->ny > 0 && desc->nz > 0, "flip_ctx_create: empty box");
    FLIP_REQUIRE((int64_t)desc->nx * desc->ny * desc->nz < (int64_t)2000000000, "flip_ctx_create: box too large for 32-bit cell ids");
    FLIP_REQUIRE(desc->dx > 0.f, "flip_ctx_create: dx must be positive");
    FLIP_REQUIRE(flip_subrow_lds_bytes(desc->nx) <= 64 * 1024, "flip_ctx_create: box too wide (nx > 895) for the per-row half-cell index in 64 KB of LDS");
    int ndev = 0;
    FLIP_CHECK(hipGetDeviceCount(&ndev));
    FLIP_REQUIRE(ndev > 0, "flip_ctx_create: no HIP device (this library has no CPU path)");
    FLIP_REQUIRE(device >= 0 && device < ndev, "flip_ctx_create: device index out of range");
    FLIP_CHECK(hipSetDevice(device));

    flip_ctx *c = new flip_ctx();
    c->g = *desc;
    c->B = (int64_t)desc->nx * desc->ny * desc->nz;
    c->n_chunks = flip_grid_blocks(c->B);
    c->device = device;
    c->gather_mode = flip_default_gather_mode();
    c->stream = (hipStream_t)stream;
    if (const char *e = getenv("FLIP_SIDE_STREAM")) c->side_enabled = atoi(e) != 0;   // A/B measurements
    const int64_t B = c->B;

    hipError_t e = hipSuccess;
#define A(ptr, count)                                  \
    if (e == hipSuccess) e = dalloc(&(ptr), (count));
    A(c->cell_start, B + 2) A(c->cell_fill, B + 1) A(c->sub_start, 8 * B + 1)
    A(c->phi_fluid, B) A(c->phi_solid, B) A(c->phi_obj, B)
    for (int b = 0; b < 4; ++b)
        for (int a = 0; a < 3; ++a) A(c->vel[b][a], B)
    for (int a = 0; a < 3; ++a) {
        A(c->disp[a], B) A(c->wgt[a], B) A(c->fwgt[a], B) A(c->owgt[a], B) A(c->wgt_back[a], B) A(c->owgt_back[a], B)
        A(c->valid[0][a], B) A(c->valid[1][a], B)
    }
    for (int a = 0; a < 6; ++a) A(c->J[a], B)
    A(c->density, B) A(c->active, B) A(c->active_back, B) A(c->label, B) A(c->band, B) A(c->seed, B)
    A(c->tmp_u8, B) A(c->row_index, B + 1)
    {
        const int64_t plane = (int64_t)desc->nx * desc->ny;
        c->pad = ((plane + FLIP_BLOCK + 63) / 64) * 64;
    }
    A(c->cg_scan, B + 1)
    A(c->d_vmax, 3 * FLIP_MAX_PCG_BLOCKS)
    A(c->gather_stats, 4)
    c->n_tiles = (int64_t)((desc->nx + 3) / 4) * ((desc->ny + 3) / 4) * ((desc->nz + 3) / 4);
    A(c->tile_scan, c->n_tiles + 1) A(c->tile_list, c->n_tiles)
    A(c->slot_scan, B + 1) A(c->slot_list, B) A(c->gather_claim, 8)
#undef A
    if (e != hipSuccess) {
        flip_ctx_destroy(c);
        return flip_fail_hip(nullptr, e, "hipMalloc of the dense fields", __FILE__, __LINE__);
    }
    if (flip_alloc_system(c) != 0) { // keep the text flip_alloc_system recorded
        g_create_error = c->err;
        flip_ctx_destroy(c);
        return 2;
    }
    // phi fields start at their backgrounds ("no fluid", "all open") until uploaded
    {
        std::vector<float> bgf((size_t)B, desc->phi_fluid_bg);
        e = hipMemcpy(c->phi_fluid, bgf.data(), (size_t)B * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            std::fill(bgf.begin(), bgf.end(), desc->phi_solid_bg);
            e = hipMemcpy(c->phi_solid, bgf.data(), (size_t)B * 4, hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) {
            flip_ctx_destroy(c);
            return flip_fail_hip(nullptr, e, "initialising the level sets", __FILE__, __LINE__);
        }
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
        c->n_cu = prop.multiProcessorCount;
        int nb = prop.multiProcessorCount * 4;
        nb = (nb / 8) * 8;
        if (nb < 8) nb = 8;
        if (nb > FLIP_MAX_PCG_BLOCKS) nb = FLIP_MAX_PCG_BLOCKS;
        c->pcg_blocks = nb;
        if (const char *env = getenv("FLIP_PCG_BLOCKS")) { // tuning aid: workgroups of the PCG kernels
            const int v = atoi(env);
            if (v >= 8 && v <= FLIP_MAX_PCG_BLOCKS) c->pcg_blocks = (v / 8) * 8;
        }
    }
    *out = c;
    return 0;
}

extern "C" int flip_synchronize(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}
extern "C" int flip_swap_velocity_buffers(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx, "null context");
    for (int a = 0; a < 3; ++a) std::swap(ctx->vel[0][a], ctx->vel[1][a]);
    return 0;
}
extern "C" int flip_swap_valid_masks(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx, "null context");
    for (int a = 0; a < 3; ++a) std::swap(ctx->valid[0][a], ctx->valid[1][a]);
    return 0;
}
extern "C" int flip_get_stream(const flip_ctx *ctx, void **stream, int *device) {
    if (!ctx || !stream) return 1;
    *stream = (void *)ctx->stream;
    if (device) *device = ctx->device;
    return 0;
}
extern "C" int flip_get_desc(const flip_ctx *ctx, flip_grid_desc *out) {
    if (!ctx || !out) return 1;
    *out = ctx->g;
    return 0;
}
extern "C" int flip_set_profiling(flip_ctx *ctx, int enabled) {
    FLIP_REQUIRE(ctx, "null context");
    ctx->profiling = enabled != 0;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// dense fields

static int field_ptr(flip_ctx *c, int field, void **ptr, int *elem) {
    void *p = nullptr;
    int es = 0;
    switch (field) {
    case FLIP_F_PHI_FLUID: p = c->phi_fluid; es = 4; break;
    case FLIP_F_PHI_SOLID: p = c->phi_solid; es = 4; break;
    case FLIP_F_PHI_OBJECT: p = c->phi_obj; es = 4; break;
    case FLIP_F_VEL_U: case FLIP_F_VEL_V: case FLIP_F_VEL_W: p = c->vel[0][field - FLIP_F_VEL_U]; es = 8; break;
    case FLIP_F_PREV_U: case FLIP_F_PREV_V: case FLIP_F_PREV_W: p = c->vel[2][field - FLIP_F_PREV_U]; es = 8; break;
    case FLIP_F_DIFF_U: case FLIP_F_DIFF_V: case FLIP_F_DIFF_W: p = c->vel[3][field - FLIP_F_DIFF_U]; es = 8; break;
    case FLIP_F_DISP_X: case FLIP_F_DISP_Y: case FLIP_F_DISP_Z: p = c->disp[field - FLIP_F_DISP_X]; es = 8; break;
    case FLIP_F_WEIGHT_U: case FLIP_F_WEIGHT_V: case FLIP_F_WEIGHT_W: p = c->wgt[field - FLIP_F_WEIGHT_U]; es = 4; break;
    case FLIP_F_FLUID_WEIGHT_U: case FLIP_F_FLUID_WEIGHT_V: case FLIP_F_FLUID_WEIGHT_W:
        p = c->fwgt[field - FLIP_F_FLUID_WEIGHT_U]; es = 4; break;
    case FLIP_F_OBJ_WEIGHT_U: case FLIP_F_OBJ_WEIGHT_V: case FLIP_F_OBJ_WEIGHT_W:
        p = c->owgt[field - FLIP_F_OBJ_WEIGHT_U]; es = 4; break;
    case FLIP_F_DENSITY: p = c->density; es = 4; break;
    case FLIP_F_PRESSURE: p = c->x; es = 8; break;
    case FLIP_F_VALID_U: case FLIP_F_VALID_V: case FLIP_F_VALID_W: p = c->valid[0][field - FLIP_F_VALID_U]; es = 1; break;
    case FLIP_F_VEL_BACK_U: case FLIP_F_VEL_BACK_V: case FLIP_F_VEL_BACK_W: p = c->vel[1][field - FLIP_F_VEL_BACK_U]; es = 8; break;
    case FLIP_F_VALID_BACK_U: case FLIP_F_VALID_BACK_V: case FLIP_F_VALID_BACK_W: p = c->valid[1][field - FLIP_F_VALID_BACK_U]; es = 1; break;
    case FLIP_F_ACTIVE: p = c->active; es = 1; break;
    case FLIP_F_LABEL: p = c->label; es = 1; break;
    case FLIP_F_ROW_INDEX: p = c->row_index; es = 4; break;
    case FLIP_F_BAND: p = c->band; es = 1; break;
    default: return 1;
    }
    *ptr = p;
    *elem = es;
    return 0;
}

extern "C" int64_t flip_field_bytes(const flip_ctx *ctx, int field) {
    void *p; int es;
    if (!ctx || field_ptr(const_cast<flip_ctx *>(ctx), field, &p, &es)) return -1;
    return ctx->B * es;
}
extern "C" int flip_field_upload(flip_ctx *ctx, int field, const void *host) {
    FLIP_REQUIRE(ctx && host, "flip_field_upload: null argument");
    void *p; int es;
    FLIP_REQUIRE(field_ptr(ctx, field, &p, &es) == 0, "flip_field_upload: unknown field");
    FLIP_CHECK(hipSetDevice(ctx->device));
    if (field == FLIP_F_PHI_FLUID || field == FLIP_F_DENSITY) ctx->density_valid = false;
    if (field == FLIP_F_PRESSURE) ctx->x_kind = -1; // no longer the solver's own result
    FLIP_CHECK(hipMemcpyAsync(p, host, (size_t)ctx->B * es, hipMemcpyHostToDevice, ctx->stream));
    FLIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}
extern "C" int flip_field_download(flip_ctx *ctx, int field, void *host) {
    FLIP_REQUIRE(ctx && host, "flip_field_download: null argument");
    void *p; int es;
    FLIP_REQUIRE(field_ptr(ctx, field, &p, &es) == 0, "flip_field_download: unknown field");
    FLIP_CHECK(hipSetDevice(ctx->device));
    FLIP_CHECK(hipMemcpyAsync(host, p, (size_t)ctx->B * es, hipMemcpyDeviceToHost, ctx->stream));
    FLIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// particles

__global__ void k_unpack_particles(long long n, const double *__restrict__ aos_pos, const double *__restrict__ aos_vel,
                                   double *x, double *y, double *z, double *vx, double *vy, double *vz, int32_t *pid) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    x[i] = aos_pos[3 * i]; y[i] = aos_pos[3 * i + 1]; z[i] = aos_pos[3 * i + 2];
    vx[i] = aos_vel[3 * i]; vy[i] = aos_vel[3 * i + 1]; vz[i] = aos_vel[3 * i + 2];
    pid[i] = (int32_t)i;
}
// out[pid[s]] = (a,b,c)[s] : back to upload order
__global__ void k_pack_particles(long long n, const double *__restrict__ a, const double *__restrict__ b,
                                 const double *__restrict__ c, const int32_t *__restrict__ pid, double *aos) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const long long i = pid[s];
    aos[3 * i] = a[s]; aos[3 * i + 1] = b[s]; aos[3 * i + 2] = c[s];
}

static int ensure_particle_capacity(flip_ctx *ctx, int64_t n) {
    if (n <= ctx->cap) return 0;
    free_particles(ctx);
    const int64_t cap = n + n / 8 + 1024;
    hipError_t e = hipSuccess;
    for (int b = 0; b < 2 && e == hipSuccess; ++b) {
        for (int a = 0; a < 6 && e == hipSuccess; ++a) e = dalloc(&ctx->p[b][a], cap);
        if (e == hipSuccess) e = dalloc(&ctx->mass[b], cap);
        if (e == hipSuccess) e = dalloc(&ctx->pid[b], cap);
    }
    if (e == hipSuccess) e = dalloc(&ctx->cell_id, cap);
    if (e == hipSuccess) e = dalloc(&ctx->slot_src, cap + 1);
    if (e == hipSuccess) e = dalloc(&ctx->slot_key, cap + 1);
    if (e == hipSuccess) e = dalloc(&ctx->stage, cap * 6);
    if (e == hipSuccess) e = dalloc(&ctx->prec, cap);
    if (e == hipSuccess) e = dalloc(&ctx->pcell, cap);
    if (e == hipSuccess) e = dalloc(&ctx->sub_slot, cap);
    if (e != hipSuccess) return flip_fail_hip(ctx, e, "hipMalloc of the particle arrays", __FILE__, __LINE__);
    ctx->cap = cap;
    return 0;
}

// Room for `capacity` particles before the first upload, so that a later exchange can grow the set in place.
extern "C" int flip_particles_reserve(flip_ctx *ctx, int64_t capacity) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(capacity >= 0 && capacity < (int64_t)2000000000, "flip_particles_reserve: capacity out of range");
    FLIP_REQUIRE(ctx->n == 0 || capacity <= ctx->cap, "flip_particles_reserve: call it before the particles are uploaded");
    FLIP_CHECK(hipSetDevice(ctx->device));
    return ensure_particle_capacity(ctx, capacity);
}

extern "C" int flip_particles_upload(flip_ctx *ctx, int64_t n, const double *pos, const double *vel, const float *mass) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(n >= 0 && n < ((int64_t)1 << 30), "flip_particles_upload: particle count out of range (fewer than 2^30 per context)");
    FLIP_REQUIRE(n == 0 || (pos && vel && mass), "flip_particles_upload: null array");
    FLIP_CHECK(hipSetDevice(ctx->device));
    if (int rc = ensure_particle_capacity(ctx, n)) return rc;
    ctx->n = n;
    ctx->cur = 0;
    ctx->binned = false;
    ctx->slots_valid = false;
    ctx->prec_valid = false;
    ctx->density_valid = false;
    if (n == 0) return 0;
    FLIP_CHECK(hipMemcpyAsync(ctx->stage, pos, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    FLIP_CHECK(hipMemcpyAsync(ctx->stage + 3 * ctx->cap, vel, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    FLIP_CHECK(hipMemcpyAsync(ctx->mass[0], mass, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    double **q = ctx->p[0];
    k_unpack_particles<<<flip_grid_blocks(n), FLIP_BLOCK, 0, ctx->stream>>>(
        n, ctx->stage, ctx->stage + 3 * ctx->cap, q[0], q[1], q[2], q[3], q[4], q[5], ctx->pid[0]);
    FLIP_LAUNCH_CHECK();
    FLIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int flip_particles_download(flip_ctx *ctx, double *pos, double *vel) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_CHECK(hipSetDevice(ctx->device));
    const int64_t n = ctx->n;
    if (n == 0) return 0;
    double **q = ctx->p[ctx->cur];
    const int32_t *pid = ctx->pid[ctx->cur];
    if (pos) {
        k_pack_particles<<<flip_grid_blocks(n), FLIP_BLOCK, 0, ctx->stream>>>(n, q[0], q[1], q[2], pid, ctx->stage);
        FLIP_LAUNCH_CHECK();
        FLIP_CHECK(hipMemcpyAsync(pos, ctx->stage, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (vel) {
        k_pack_particles<<<flip_grid_blocks(n), FLIP_BLOCK, 0, ctx->stream>>>(n, q[3], q[4], q[5], pid,
                                                                            ctx->stage + 3 * ctx->cap);
        FLIP_LAUNCH_CHECK();
        FLIP_CHECK(hipMemcpyAsync(vel, ctx->stage + 3 * ctx->cap, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->stream));
    }
    FLIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

__global__ void k_pack_mass(long long n, const float *__restrict__ m, const int32_t *__restrict__ pid, float *__restrict__ out) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) out[pid[s]] = m[s];
}
extern "C" int flip_particles_download_mass(flip_ctx *ctx, float *mass) {
    FLIP_REQUIRE(ctx && mass, "flip_particles_download_mass: null argument");
    FLIP_CHECK(hipSetDevice(ctx->device));
    const int64_t n = ctx->n;
    if (n == 0) return 0;
    float *tmp = reinterpret_cast<float *>(ctx->stage);
    k_pack_mass<<<flip_grid_blocks(n), FLIP_BLOCK, 0, ctx->stream>>>(n, ctx->mass[ctx->cur], ctx->pid[ctx->cur], tmp);
    FLIP_LAUNCH_CHECK();
    FLIP_CHECK(hipMemcpyAsync(mass, tmp, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    FLIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int64_t flip_particles_count(const flip_ctx *ctx) { return ctx ? ctx->n : -1; }

// ---------------------------------------------------------------------------------------------
// A second stream inside a substep.  Every entry point enqueues on ctx->stream; between flip_side_select(ctx, 1) and
// flip_side_select(ctx, 0) that is the side stream.  The caller puts only stages there that neither read what the main
// stream's stages write nor write what they read (include/flip_hip.h lists the two chains of a substep that qualify).

// A process has few hardware queues (four by default), the runtime deals its streams out to them, and two streams on
// one queue run in that queue's order: a new stream may well share the queue of the context's own stream and overlap
// nothing.  Which queue a stream got cannot be asked, so it is measured: a kernel that spins for a fixed time runs on
// the context's stream and on a candidate; if the two take one kernel's time the candidate is kept, else the next is
// tried.  The last candidate is kept whatever it measures (the side stream is then merely of no use).
__global__ void k_spin(long long ticks) {
    const long long t0 = wall_clock64();
    // bounded in iterations as well, so that the kernel ends whatever the clock does (some tens of milliseconds at most)
    for (int it = 0; it < 200000 && wall_clock64() - t0 < ticks; ++it) __builtin_amdgcn_s_sleep(8);
}
static int pick_side_stream(flip_ctx *ctx) {
    int rate_khz = 100000;   // wall_clock64 ticks per millisecond
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, ctx->device);
    const long long ticks = (long long)rate_khz / 5;   // 0.2 ms
    hipEvent_t ev[3] = {};
    for (hipEvent_t &e : ev) FLIP_CHECK(hipEventCreate(&e));
    hipStream_t keep = nullptr;
    std::vector<hipStream_t> rejected;
    int rc = 0;
    for (int attempt = 0; attempt < 6 && !keep && !rc; ++attempt) {
        hipStream_t cand = nullptr;
        if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) { rc = flip_fail(ctx, "hipStreamCreateWithFlags of the side stream", __FILE__, __LINE__); break; }
        bool ok = hipEventRecord(ev[0], ctx->stream) == hipSuccess && hipStreamWaitEvent(cand, ev[0], 0) == hipSuccess;
        if (ok) {
            k_spin<<<1, 64, 0, ctx->stream>>>(ticks);
            k_spin<<<1, 64, 0, cand>>>(ticks);
            ok = hipGetLastError() == hipSuccess && hipEventRecord(ev[1], ctx->stream) == hipSuccess && hipEventRecord(ev[2], cand) == hipSuccess
                 && hipEventSynchronize(ev[1]) == hipSuccess && hipEventSynchronize(ev[2]) == hipSuccess;
        }
        float ms_main = 0.f, ms_cand = 0.f;
        ok = ok && hipEventElapsedTime(&ms_main, ev[0], ev[1]) == hipSuccess && hipEventElapsedTime(&ms_cand, ev[0], ev[2]) == hipSuccess;
        if (!ok) { (void)hipStreamDestroy(cand); rc = flip_fail(ctx, "probing the side stream", __FILE__, __LINE__); break; }
        const float both = ms_main > ms_cand ? ms_main : ms_cand;
        if (both < 0.33f || attempt == 5) keep = cand;   // side by side: about 0.2 ms; one after the other: 0.4 ms
        else rejected.push_back(cand);                   // kept alive until the choice is made, so that the next one lands elsewhere
    }
    for (hipStream_t s : rejected) (void)hipStreamDestroy(s);
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    if (rc) return rc;
    ctx->side_stream = keep;
    return 0;
}

extern "C" int flip_set_side_stream(flip_ctx *ctx, int enabled) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(!ctx->side_forked && !ctx->side_selected, "flip_set_side_stream: between flip_side_fork and flip_side_join");
    ctx->side_enabled = enabled != 0;
    return 0;
}
extern "C" int flip_side_fork(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(!ctx->side_selected, "flip_side_fork: the side stream is selected");
    FLIP_REQUIRE(!ctx->pair_on, "flip_side_fork: inside flip_pair_begin .. flip_pair_end");
    if (!ctx->side_enabled) return 0;
    FLIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->side_stream) {
        if (int rc = pick_side_stream(ctx)) return rc;
        for (hipEvent_t &e : ctx->side_event) FLIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (ctx->side_forked) { // a second fork before the join: the side work so far also precedes what follows on the side
        return 0;
    }
    FLIP_CHECK(hipEventRecord(ctx->side_event[0], ctx->stream));
    FLIP_CHECK(hipStreamWaitEvent(ctx->side_stream, ctx->side_event[0], 0));
    ctx->side_forked = true;
    return 0;
}
extern "C" int flip_side_select(flip_ctx *ctx, int side) {
    FLIP_REQUIRE(ctx, "null context");
    if (!ctx->side_enabled) return 0;
    if (side) {
        FLIP_REQUIRE(ctx->side_forked, "flip_side_select: call flip_side_fork first");
        if (!ctx->side_selected) {
            ctx->side_main_stream = ctx->stream;
            ctx->stream = ctx->side_stream;
            ctx->side_selected = true;
        }
    } else if (ctx->side_selected) {
        ctx->stream = ctx->side_main_stream;
        ctx->side_selected = false;
    }
    return 0;
}
extern "C" int flip_side_join(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx, "null context");
    if (ctx->side_selected) {
        ctx->stream = ctx->side_main_stream;
        ctx->side_selected = false;
    }
    if (!ctx->side_forked) return 0;
    ctx->side_forked = false;
    FLIP_CHECK(hipSetDevice(ctx->device));
    FLIP_CHECK(hipEventRecord(ctx->side_event[1], ctx->side_stream));
    FLIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->side_event[1], 0));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Exclusive scan of int32 in place, reduce then scan: the sum of every tile of 4096 items (one read of the data), the
// scan of those sums (one workgroup for up to 4096 tiles, the same three steps again above that), then every tile scans
// itself starting from its offset (one read, one write).  A wave owns 1024 consecutive items of a tile and reads them in
// four rounds of one int4 per lane, 1 KB per instruction.  Deterministic (integers).

#define SCAN_ROUNDS 4
#define SCAN_TILE (FLIP_BLOCK * 4 * SCAN_ROUNDS)

__device__ __forceinline__ int4 d_scan_load(const int32_t *__restrict__ data, long long at, long long n, bool vec) {
    if (vec && at + 4 <= n) return *reinterpret_cast<const int4 *>(data + at);
    int4 v = make_int4(0, 0, 0, 0);
    if (at < n) v.x = data[at];
    if (at + 1 < n) v.y = data[at + 1];
    if (at + 2 < n) v.z = data[at + 2];
    if (at + 3 < n) v.w = data[at + 3];
    return v;
}
__device__ __forceinline__ long long d_scan_at(int q) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (long long)blockIdx.x * SCAN_TILE + (long long)wave * (SCAN_TILE / 4) + q * 256 + 4 * lane;
}
__global__ void k_scan_sums(const int32_t *__restrict__ data, long long n, int32_t *__restrict__ sums, int vec) {
    __shared__ int32_t wave_tot[4];
    int32_t s = 0;
#pragma unroll
    for (int q = 0; q < SCAN_ROUNDS; ++q) {
        const int4 v = d_scan_load(data, d_scan_at(q), n, vec != 0);
        s += v.x + v.y + v.z + v.w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, FLIP_WAVE);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}
// offsets == nullptr: the data are one tile
__global__ void k_scan_tile(int32_t *data, long long n, const int32_t *__restrict__ offsets, int vec) {
    __shared__ int32_t wave_tot[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int4 v[SCAN_ROUNDS];
    int32_t excl[SCAN_ROUNDS];
    int32_t carry = 0;
#pragma unroll
    for (int q = 0; q < SCAN_ROUNDS; ++q) v[q] = d_scan_load(data, d_scan_at(q), n, vec != 0);
#pragma unroll
    for (int q = 0; q < SCAN_ROUNDS; ++q) {
        const int32_t s = v[q].x + v[q].y + v[q].z + v[q].w;
        int32_t incl = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t t = __shfl_up(incl, off, FLIP_WAVE);
            if (lane >= off) incl += t;
        }
        excl[q] = carry + incl - s;
        carry += __shfl(incl, 63, FLIP_WAVE);
    }
    if (lane == 0) wave_tot[wave] = carry;
    __syncthreads();
    int32_t base = offsets ? offsets[blockIdx.x] : 0;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
#pragma unroll
    for (int q = 0; q < SCAN_ROUNDS; ++q) {
        const long long at = d_scan_at(q);
        int4 o;
        o.x = base + excl[q];
        o.y = o.x + v[q].x;
        o.z = o.y + v[q].y;
        o.w = o.z + v[q].z;
        if (vec && at + 4 <= n) *reinterpret_cast<int4 *>(data + at) = o;
        else {
            if (at < n) data[at] = o.x;
            if (at + 1 < n) data[at + 1] = o.y;
            if (at + 2 < n) data[at + 2] = o.z;
            if (at + 3 < n) data[at + 3] = o.w;
        }
    }
}

static int scan_rec(flip_ctx *ctx, int32_t *data, int64_t n, int32_t *tmp, int64_t tmp_len) {
    const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    const int vec = ((uintptr_t)data & 15) == 0 ? 1 : 0;
    if (nb <= 1) {
        k_scan_tile<<<1, FLIP_BLOCK, 0, ctx->stream>>>(data, n, nullptr, vec);
        FLIP_LAUNCH_CHECK();
        return 0;
    }
    const int64_t nb4 = (nb + 3) & ~(int64_t)3;   // the next level's sums start on a 16-byte boundary
    FLIP_REQUIRE(nb4 <= tmp_len, "scan scratch too small");
    k_scan_sums<<<(int)nb, FLIP_BLOCK, 0, ctx->stream>>>(data, n, tmp, vec);
    FLIP_LAUNCH_CHECK();
    if (int rc = scan_rec(ctx, tmp, nb, tmp + nb4, tmp_len - nb4)) return rc;
    k_scan_tile<<<(int)nb, FLIP_BLOCK, 0, ctx->stream>>>(data, n, tmp, vec);
    FLIP_LAUNCH_CHECK();
    return 0;
}

__global__ void k_copy_i32(int32_t *__restrict__ dst, const int32_t *__restrict__ src) { dst[0] = src[0]; }
int flip_copy_i32(flip_ctx *ctx, int32_t *dst, const int32_t *src, hipStream_t st) {
    k_copy_i32<<<1, 1, 0, st>>>(dst, src);
    FLIP_LAUNCH_CHECK();
    return 0;
}

int flip_exclusive_scan(flip_ctx *ctx, int32_t *data, int64_t n) {
    // scratch for the sums of every level
    int64_t need = 8;
    for (int64_t m = (n + SCAN_TILE - 1) / SCAN_TILE; m > 1; m = (m + SCAN_TILE - 1) / SCAN_TILE) need += (m + 3) & ~(int64_t)3;
    if (need > ctx->scan_tmp_len) {
        dfree(ctx->scan_tmp);
        hipError_t e = dalloc(&ctx->scan_tmp, need);
        if (e != hipSuccess) return flip_fail_hip(ctx, e, "hipMalloc of the scan scratch", __FILE__, __LINE__);
        ctx->scan_tmp_len = need;
    }
    return scan_rec(ctx, data, n, ctx->scan_tmp, ctx->scan_tmp_len);
}

// The scan on a caller's array (parity hook: the binning, the labels, the row lists and every multigrid level go through
// flip_exclusive_scan; this runs it on n values of the host array, starting `offset` items into a device buffer so that
// the path for data that are not 16-byte aligned is reachable).
extern "C" int flip_scan_i32(flip_ctx *ctx, int32_t *host_values, int64_t n, int32_t offset) {
    FLIP_REQUIRE(ctx && (host_values || n == 0), "flip_scan_i32: null argument");
    FLIP_REQUIRE(n >= 0 && offset >= 0 && offset < 4, "flip_scan_i32: bad size or offset");
    FLIP_CHECK(hipSetDevice(ctx->device));
    if (n == 0) return 0;
    int32_t *buf = nullptr;
    FLIP_CHECK(hipMalloc((void **)&buf, (size_t)(n + 4) * 4));
    int rc = 0;
    if (hipMemcpyAsync(buf + offset, host_values, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = flip_fail(ctx, "flip_scan_i32: copy to the device", __FILE__, __LINE__);
    if (!rc) rc = flip_exclusive_scan(ctx, buf + offset, n);
    if (!rc && hipMemcpyAsync(host_values, buf + offset, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc = flip_fail(ctx, "flip_scan_i32: copy from the device", __FILE__, __LINE__);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) rc = flip_fail(ctx, "flip_scan_i32: synchronise", __FILE__, __LINE__);
    (void)hipFree(buf);
    return rc;
}
