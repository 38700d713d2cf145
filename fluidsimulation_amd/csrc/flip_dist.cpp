// flip_dist.cpp -- libflip_dist.so: the z-slab driver of include/flip_dist.h.  Host code only: it calls libflip_hip.so
// through its C ABI and RCCL through the nccl API, everything enqueued on the context's stream.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/flip_dist.h"

namespace {
std::string g_create_error;

// Two communicators over the same ranks: the two solves of a substep run side by side on two streams, and operations on
// one communicator are serialised whatever stream they are enqueued on.  `channel` says which one the next call uses.
struct RcclState {
    ncclComm_t comm[2] = { nullptr, nullptr };
    int channel = 0;
    int rank = 0, world = 1;
};

int rccl_all_reduce(void *user, double *buf, int64_t count, int op, void *stream) {
    RcclState *s = static_cast<RcclState *>(user);
    const ncclResult_t r = ncclAllReduce(buf, buf, (size_t)count, ncclDouble, op == FLIP_DIST_MAX ? ncclMax : ncclSum, s->comm[s->channel], (hipStream_t)stream);
    return r == ncclSuccess ? 0 : (int)r;
}
int rccl_exchange(void *user, const void *send_lo, int64_t send_lo_bytes, void *recv_lo, int64_t recv_lo_bytes, const void *send_hi,
                  int64_t send_hi_bytes, void *recv_hi, int64_t recv_hi_bytes, void *stream) {
    RcclState *s = static_cast<RcclState *>(user);
    hipStream_t st = (hipStream_t)stream;
    ncclComm_t comm = s->comm[s->channel];
    ncclResult_t r = ncclGroupStart();
    // grouped: the four transfers of a face pair progress together, no ordering between ranks can deadlock them
    if (r == ncclSuccess && s->rank > 0 && send_lo) r = ncclSend(send_lo, (size_t)send_lo_bytes, ncclChar, s->rank - 1, comm, st);
    if (r == ncclSuccess && s->rank > 0 && recv_lo) r = ncclRecv(recv_lo, (size_t)recv_lo_bytes, ncclChar, s->rank - 1, comm, st);
    if (r == ncclSuccess && s->rank < s->world - 1 && send_hi) r = ncclSend(send_hi, (size_t)send_hi_bytes, ncclChar, s->rank + 1, comm, st);
    if (r == ncclSuccess && s->rank < s->world - 1 && recv_hi) r = ncclRecv(recv_hi, (size_t)recv_hi_bytes, ncclChar, s->rank + 1, comm, st);
    const ncclResult_t e = ncclGroupEnd();
    if (r == ncclSuccess) r = e;
    return r == ncclSuccess ? 0 : (int)r;
}
} // namespace

struct flip_dist {
    flip_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    bool slab = false;           // the context is in slab mode (world > 1 or forced)
    bool pair_solves = true;     // the two slab solves of a substep side by side (FLIP_DIST_SEQUENTIAL_SOLVES: one after the other)
    int32_t ghost = 0, overlap = 0;
    int32_t k0 = 0, k1 = 0;      // owned planes in the context's own plane numbers
    int32_t g_lo = 0, g_hi = 0;  // ghost planes of this box toward the lower / upper neighbour
    int32_t g_of_lower = 0, g_of_upper = 0; // ghost planes of the neighbours' boxes toward this rank
    int64_t plane = 0;           // cells per z plane
    flip_dist_transport tr{};
    RcclState rccl;              // used when the transport is RCCL
    bool own_comm = false;
    hipStream_t stream = nullptr;
    int device = 0;
    // particle messages: 64-byte records
    int64_t capacity = 0;
    void *send_lo = nullptr, *send_hi = nullptr, *recv_lo = nullptr, *recv_hi = nullptr;
    int64_t *counts_dev = nullptr;   // [4] send_lo, send_hi, recv_lo, recv_hi
    double *small_dev = nullptr;     // [16] staging of flip_dist_all_reduce
    int64_t last_sent[2] = { 0, 0 }, last_received[2] = { 0, 0 };
    std::string err;
};

namespace {
int fail(flip_dist *d, const std::string &what) {
    if (d) d->err = what;
    else g_create_error = what;
    return 1;
}
int fail_ctx(flip_dist *d, const char *call) {
    return fail(d, std::string(call) + ": " + flip_last_error(d->ctx));
}
static const bool g_dist_trace = getenv("FLIP_DIST_TRACE") != nullptr;   // every library call named on stderr before it is made
static const bool g_dist_trace_sync = g_dist_trace && atoi(getenv("FLIP_DIST_TRACE")) > 1;   // ... and waited for
#define DIST_FLIP(call)                                                         \
    do {                                                                        \
        if (g_dist_trace) { fprintf(stderr, "flip_dist: %s\n", #call); fflush(stderr); } \
        if ((call) != 0) return fail_ctx(d, #call);                             \
        if (g_dist_trace_sync) (void)hipDeviceSynchronize();                    \
    } while (0)
#define DIST_HIP(call)                                                                                   \
    do {                                                                                                 \
        const hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return fail(d, std::string(#call) + ": " + hipGetErrorString(e_));         \
    } while (0)
#define DIST_TR(call)                                                                                    \
    do {                                                                                                 \
        const int e_ = (call);                                                                           \
        if (e_ != 0) return fail(d, std::string(#call) + ": transport error " + std::to_string(e_));     \
    } while (0)

struct SolveBuffers {
    double *sums = nullptr, *z = nullptr, *x = nullptr, *r = nullptr, *rlo = nullptr, *rhi = nullptr;
};
int solve_buffers(flip_dist *d, SolveBuffers &b) {
    void *p = nullptr;
    int64_t n = 0;
    DIST_FLIP(flip_pcg_dist_buffer(d->ctx, 0, &p, &n)); b.sums = (double *)p;
    DIST_FLIP(flip_pcg_dist_buffer(d->ctx, 1, &p, &n)); b.z = (double *)p;
    DIST_FLIP(flip_pcg_dist_buffer(d->ctx, 2, &p, &n)); b.x = (double *)p;
    DIST_FLIP(flip_pcg_dist_buffer(d->ctx, 3, &p, &n)); b.r = (double *)p;
    if (d->overlap > 0) {
        DIST_FLIP(flip_pcg_dist_buffer(d->ctx, 4, &p, &n)); b.rlo = (double *)p;
        DIST_FLIP(flip_pcg_dist_buffer(d->ctx, 5, &p, &n)); b.rhi = (double *)p;
    }
    return 0;
}
// planes [a, a + n) of a box vector toward / from the neighbours; a side without a neighbour is skipped
int exchange_planes(flip_dist *d, const double *send_lo, int64_t n_send_lo, double *recv_lo, int64_t n_recv_lo, const double *send_hi,
                    int64_t n_send_hi, double *recv_hi, int64_t n_recv_hi) {
    const bool lo = d->rank > 0, hi = d->rank < d->world - 1;
    if (!lo && !hi) return 0;
    const int64_t pb = d->plane * (int64_t)sizeof(double);
    DIST_TR(d->tr.exchange(d->tr.user, lo ? send_lo : nullptr, lo ? n_send_lo * pb : 0, lo ? recv_lo : nullptr, lo ? n_recv_lo * pb : 0,
                           hi ? send_hi : nullptr, hi ? n_send_hi * pb : 0, hi ? recv_hi : nullptr, hi ? n_recv_hi * pb : 0, d->stream));
    return 0;
}

// z = M^-1 r and the global r.r, r.z: the phases of flip_hip.h around the preconditioner
int precondition_and_sums(flip_dist *d, const SolveBuffers &b) {
    const int64_t P = d->plane, k0 = d->k0, k1 = d->k1, m = d->overlap;
    if (m > 0) {
        if (exchange_planes(d, b.r + k0 * P, m, b.r + (k0 - m) * P, m, b.r + (k1 - m) * P, m, b.r + k1 * P, m)) return 1;
        DIST_FLIP(flip_pcg_dist_precondition(d->ctx));
        if (exchange_planes(d, b.z + (k0 - m) * P, m + 1, b.rlo, m + 1, b.z + (k1 - 1) * P, m + 1, b.rhi, m + 1)) return 1;
        DIST_FLIP(flip_pcg_dist_combine(d->ctx));
    }
    DIST_TR(d->tr.all_reduce(d->tr.user, b.sums, 16, FLIP_DIST_SUM, d->stream));
    if (m == 0 && exchange_planes(d, b.z + k0 * P, 1, b.z + (k0 - 1) * P, 1, b.z + (k1 - 1) * P, 1, b.z + k1 * P, 1)) return 1;
    return 0;
}

// One slab solve as the driver sees it: the buffers of the system it works on, the stream that system's kernels are
// enqueued on and the communicator its collectives use.  `select` makes that system the context's active one (only
// needed when two solves are in progress, flip_pair_select).
struct SlabSolve {
    int slot = -1;            // -1: the context's only system; 0 / 1: a system of a pair
    SolveBuffers b;
    int32_t done = 0, it = 0;
};
int slab_select(flip_dist *d, SlabSolve &s) {
    if (s.slot >= 0) {
        DIST_FLIP(flip_pair_select(d->ctx, s.slot));
        void *st = nullptr;
        DIST_FLIP(flip_get_stream(d->ctx, &st, nullptr));
        d->stream = (hipStream_t)st;
        d->rccl.channel = s.slot;
    }
    return 0;
}
int slab_begin(flip_dist *d, SlabSolve &s, double tol, int32_t max_iter, int precond) {
    if (slab_select(d, s)) return 1;
    if (solve_buffers(d, s.b)) return 1;
    DIST_FLIP(flip_pcg_dist_begin(d->ctx, tol, max_iter, precond));
    return precondition_and_sums(d, s.b);
}
int slab_iteration(flip_dist *d, SlabSolve &s) {
    if (slab_select(d, s)) return 1;
    DIST_FLIP(flip_pcg_dist_direction(d->ctx));
    DIST_TR(d->tr.all_reduce(d->tr.user, s.b.sums, 16, FLIP_DIST_SUM, d->stream));
    DIST_FLIP(flip_pcg_dist_update(d->ctx));
    return precondition_and_sums(d, s.b);
}
int slab_finish(flip_dist *d, SlabSolve &s, flip_pcg_info *info) {
    if (slab_select(d, s)) return 1;
    // every ghost plane of the solution from the neighbour that owns it
    const int64_t P = d->plane, k0 = d->k0, k1 = d->k1;
    if (exchange_planes(d, s.b.x + k0 * P, d->g_of_lower, s.b.x + (k0 - d->g_lo) * P, d->g_lo, s.b.x + (k1 - d->g_of_upper) * P, d->g_of_upper,
                        s.b.x + k1 * P, d->g_hi))
        return 1;
    DIST_FLIP(flip_pcg_dist_finish(d->ctx, info));
    return 0;
}

int slab_cg(flip_dist *d, double tol, int32_t max_iter, int precond, flip_pcg_info *info) {
    SlabSolve s;
    if (slab_begin(d, s, tol, max_iter, precond)) return 1;
    while (!s.done) {
        // the stop flag is a function of all-reduced values only: every rank leaves after the same batch
        for (int q = 0; q < 8; ++q)
            if (slab_iteration(d, s)) return 1;
        DIST_FLIP(flip_pcg_dist_status(d->ctx, &s.done, &s.it));
    }
    return slab_finish(d, s, info);
}

// The rigid-body agreement of flip_dist_solve (see there) for the active system.
int agree_no_dependent_coupling(flip_dist *d) {
    int32_t coupled = 0;
    DIST_FLIP(flip_system_dependent_coupled(d->ctx, &coupled));
    double any = coupled > 0 ? 1.0 : 0.0;
    if (flip_dist_all_reduce(d, &any, 1, FLIP_DIST_MAX)) return 1;
    if (any > 0.0)
        return fail(d, "flip_dist_solve: the rigid body touches dependent rows (cells with phi_fluid >= 0 that carry a pressure row) on some "
                       "rank; the slab solver needs the eliminated system, which such rows rule out");
    return 0;
}

int create_common(flip_ctx *ctx, int rank, int world, const int32_t *starts, int32_t ghost, int32_t overlap, int64_t particle_capacity,
                  int flags, flip_dist **out) {
    flip_dist *d = nullptr; // the macros record into the creation error
    if (!ctx || !out || !starts) return fail(nullptr, "flip_dist_create: null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(nullptr, "flip_dist_create: rank out of range");
    if (starts[0] != 0) return fail(nullptr, "flip_dist_create: starts[0] must be 0");
    for (int r = 0; r < world; ++r)
        if (starts[r + 1] <= starts[r]) return fail(nullptr, "flip_dist_create: starts must ascend");
    if (world > 1 && ghost < 1) return fail(nullptr, "flip_dist_create: ghost planes are needed between slabs");
    for (int r = 0; r < world && world > 1; ++r)
        if (starts[r + 1] - starts[r] < ghost) return fail(nullptr, "flip_dist_create: a slab must be at least as thick as the ghost zone");
    if (overlap < 0 || (overlap > 0 && overlap >= ghost && world > 1)) return fail(nullptr, "flip_dist_create: the overlap must be smaller than the ghost zone");
    flip_grid_desc g;
    if (flip_get_desc(ctx, &g) != 0) return fail(nullptr, "flip_dist_create: flip_get_desc failed");
    const int32_t nzg = starts[world];
    const int32_t o0 = starts[rank], o1 = starts[rank + 1];
    const int32_t e0 = o0 - ghost > 0 ? o0 - ghost : 0, e1 = o1 + ghost < nzg ? o1 + ghost : nzg;
    if (world > 1 && g.nz != e1 - e0) return fail(nullptr, "flip_dist_create: the context's box is not this rank's extended slab");
    d = new flip_dist();
    d->ctx = ctx; d->rank = rank; d->world = world; d->ghost = ghost;
    d->slab = world > 1 || (flags & FLIP_DIST_FORCE_SLAB);
    d->pair_solves = !(flags & FLIP_DIST_SEQUENTIAL_SOLVES);
    d->overlap = (world > 1) ? overlap : 0;
    d->plane = (int64_t)g.nx * g.ny;
    if (world > 1) {
        d->k0 = o0 - e0; d->k1 = o1 - e0;
        d->g_lo = o0 - e0; d->g_hi = e1 - o1;
        if (rank > 0) d->g_of_lower = (starts[rank] + ghost < nzg ? starts[rank] + ghost : nzg) - starts[rank];
        if (rank < world - 1) d->g_of_upper = starts[rank + 1] - (starts[rank + 1] - ghost > 0 ? starts[rank + 1] - ghost : 0);
    } else {
        d->k0 = 0; d->k1 = g.nz;
    }
    void *st = nullptr;
    if (flip_get_stream(ctx, &st, &d->device) != 0) { delete d; return fail(nullptr, "flip_dist_create: flip_get_stream failed"); }
    d->stream = (hipStream_t)st;
    hipError_t e = hipSetDevice(d->device);
    d->capacity = particle_capacity > 1 ? particle_capacity : 1;
    if (world > 1) {
        void **bufs[4] = { &d->send_lo, &d->send_hi, &d->recv_lo, &d->recv_hi };
        for (void **p : bufs)
            if (e == hipSuccess) e = hipMalloc(p, (size_t)d->capacity * 64);
        if (e == hipSuccess) e = hipMalloc((void **)&d->counts_dev, 4 * sizeof(int64_t));
    }
    if (e == hipSuccess) e = hipMalloc((void **)&d->small_dev, 16 * sizeof(double));
    if (e != hipSuccess) {
        const std::string msg = std::string("flip_dist_create: ") + hipGetErrorString(e);
        flip_dist_destroy(d);
        return fail(nullptr, msg);
    }
    if (d->slab) {
        if (flip_set_slab(ctx, d->k0, d->k1) != 0 || (d->overlap > 0 && flip_set_slab_overlap(ctx, d->overlap) != 0)) {
            const std::string msg = std::string("flip_dist_create: ") + flip_last_error(ctx);
            flip_dist_destroy(d);
            return fail(nullptr, msg);
        }
    }
    (void)flags;
    *out = d;
    return 0;
}
} // namespace

extern "C" int flip_dist_unique_id(void *id_out) {
    if (!id_out) return fail(nullptr, "flip_dist_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == FLIP_DIST_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    memcpy(id_out, &id, sizeof id);
    return 0;
}

extern "C" int flip_dist_create(flip_ctx *ctx, int rank, int world, const void *unique_id, const int32_t *starts, int32_t ghost, int32_t overlap,
                                int64_t particle_capacity, int flags, flip_dist **out) {
    if (!unique_id) return fail(nullptr, "flip_dist_create: null unique id");
    if (int rc = create_common(ctx, rank, world, starts, ghost, overlap, particle_capacity, flags, out)) return rc;
    flip_dist *d = *out;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    const ncclResult_t r = ncclCommInitRank(&d->rccl.comm[0], world, id, rank);
    if (r != ncclSuccess) {
        const std::string msg = std::string("ncclCommInitRank: ") + ncclGetErrorString(r);
        flip_dist_destroy(d);
        *out = nullptr;
        return fail(nullptr, msg);
    }
    d->own_comm = true;
    // the second communicator, for the solve that runs beside the first one (same ranks, same order)
    if (ncclCommSplit(d->rccl.comm[0], 0, rank, &d->rccl.comm[1], nullptr) != ncclSuccess || !d->rccl.comm[1]) {
        d->rccl.comm[1] = d->rccl.comm[0];   // one communicator serves both: its operations then run one after the other
    }
    d->rccl.rank = rank; d->rccl.world = world;
    d->tr.user = &d->rccl;
    d->tr.all_reduce = rccl_all_reduce;
    d->tr.exchange = rccl_exchange;
    return 0;
}

extern "C" int flip_dist_create_with_transport(flip_ctx *ctx, int rank, int world, const flip_dist_transport *transport, const int32_t *starts,
                                               int32_t ghost, int32_t overlap, int64_t particle_capacity, int flags, flip_dist **out) {
    if (!transport || !transport->all_reduce || !transport->exchange) return fail(nullptr, "flip_dist_create_with_transport: incomplete transport");
    if (int rc = create_common(ctx, rank, world, starts, ghost, overlap, particle_capacity, flags, out)) return rc;
    (*out)->tr = *transport;
    return 0;
}

extern "C" void flip_dist_destroy(flip_dist *d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->stream || d->ctx) (void)hipStreamSynchronize(d->stream);
    if (d->own_comm) {
        if (d->rccl.comm[1] && d->rccl.comm[1] != d->rccl.comm[0]) (void)ncclCommDestroy(d->rccl.comm[1]);
        if (d->rccl.comm[0]) (void)ncclCommDestroy(d->rccl.comm[0]);
    }
    void *bufs[6] = { d->send_lo, d->send_hi, d->recv_lo, d->recv_hi, d->counts_dev, d->small_dev };
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    delete d;
}

extern "C" const char *flip_dist_last_error(const flip_dist *d) { return d ? d->err.c_str() : g_create_error.c_str(); }

extern "C" int flip_dist_all_reduce(flip_dist *d, double *values, int32_t count, int op) {
    if (!d || !values) return fail(d, "flip_dist_all_reduce: null argument");
    if (count < 0 || count > 16) return fail(d, "flip_dist_all_reduce: at most 16 values");
    if (d->world == 1 || count == 0) return 0;
    DIST_HIP(hipSetDevice(d->device));
    DIST_HIP(hipMemcpyAsync(d->small_dev, values, (size_t)count * sizeof(double), hipMemcpyHostToDevice, d->stream));
    DIST_TR(d->tr.all_reduce(d->tr.user, d->small_dev, count, op, d->stream));
    DIST_HIP(hipMemcpyAsync(values, d->small_dev, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, d->stream));
    DIST_HIP(hipStreamSynchronize(d->stream));
    return 0;
}

extern "C" int flip_dist_solve(flip_dist *d, int kind, float dt, double tol, int32_t max_iter, int precond, flip_pcg_info *info) {
    if (!d || !info) return fail(d, "flip_dist_solve: null argument");
    DIST_FLIP(flip_build_system(d->ctx, kind, dt));
    if (!d->slab) {
        DIST_FLIP(flip_pcg_solve(d->ctx, tol, max_iter, precond, info));
        return 0;
    }
    // The slab solver works on the eliminated system.  A rigid-body row that is also a dependent row (a free-surface cell
    // next to the body) rules that out, and whether one exists is a property of all slabs together: agree before the
    // first collective of the solve, so that every rank takes the same way out.
    if (agree_no_dependent_coupling(d)) return 1;
    return slab_cg(d, tol, max_iter, precond, info);
}

extern "C" int flip_dist_solve_pair(flip_dist *d, float dt, double tol, int32_t max_iter, int precond, flip_pcg_info *info_divergence,
                                    flip_pcg_info *info_density) {
    if (!d || !info_divergence || !info_density) return fail(d, "flip_dist_solve_pair: null argument");
    if (!d->slab) {
        DIST_FLIP(flip_solve_pair(d->ctx, dt, tol, max_iter, precond, info_divergence, info_density));
        return 0;
    }
    // Both systems are assembled first (the density system needs nothing the divergence solve produces), each into a
    // system set and on a stream of its own; then the two slab solves advance in turn, an iteration of one, an iteration
    // of the other: nothing orders the two streams against each other, so the all-reduces and plane exchanges of one
    // solve travel while the kernels of the other run.  Every rank issues the same calls in the same order.
    hipStream_t main_stream = d->stream;
    DIST_FLIP(flip_pair_begin(d->ctx));
    SlabSolve s[2];
    s[0].slot = 0; s[1].slot = 1;
    int rc = 0;
    auto leave = [&](int code) {
        const std::string text = d->err;
        (void)flip_pair_end(d->ctx);
        d->stream = main_stream;
        d->rccl.channel = 0;
        if (code) d->err = text;
        return code;
    };
    // slot 0: the divergence system, which stays the active one afterwards; slot 1: the density system
    if (slab_select(d, s[1])) return leave(1);
    if (flip_build_system(d->ctx, FLIP_SOLVE_DENSITY, dt) != 0) return leave(fail_ctx(d, "flip_build_system"));
    if (slab_select(d, s[0])) return leave(1);
    if (flip_build_system(d->ctx, FLIP_SOLVE_DIVERGENCE, dt) != 0) return leave(fail_ctx(d, "flip_build_system"));
    if (agree_no_dependent_coupling(d)) return leave(1);
    for (int q = 1; q >= 0 && !rc; --q) rc = slab_begin(d, s[q], tol, max_iter, precond);
    if (rc) return leave(1);
    while (!s[0].done || !s[1].done) {
        for (int b = 0; b < 8 && !rc; ++b)
            for (int q = 0; q < 2 && !rc; ++q)
                if (!s[q].done) rc = slab_iteration(d, s[q]);
        for (int q = 0; q < 2 && !rc; ++q)
            if (!s[q].done) {
                rc = slab_select(d, s[q]);
                if (!rc && flip_pcg_dist_status(d->ctx, &s[q].done, &s[q].it) != 0) rc = fail_ctx(d, "flip_pcg_dist_status");
            }
        if (rc) return leave(1);
    }
    if (slab_finish(d, s[1], info_density)) return leave(1);
    if (slab_finish(d, s[0], info_divergence)) return leave(1);
    return leave(0);
}

extern "C" int flip_dist_exchange_particles(flip_dist *d, int64_t sent[2], int64_t received[2]) {
    if (!d) return fail(d, "flip_dist_exchange_particles: null handle");
    if (sent) sent[0] = sent[1] = 0;
    if (received) received[0] = received[1] = 0;
    if (d->world == 1) return 0;
    DIST_HIP(hipSetDevice(d->device));
    const bool lo = d->rank > 0, hi = d->rank < d->world - 1;
    const int32_t big = 1 << 30;
    int64_t cnt[4] = { 0, 0, 0, 0 };
    // everything authoritative that now lies inside the neighbour's extended slab: migrants and its ghost copies
    if (lo) DIST_FLIP(flip_particles_pack(d->ctx, -big, d->k0 + d->g_of_lower, d->send_lo, d->capacity, &cnt[0]));
    if (hi) DIST_FLIP(flip_particles_pack(d->ctx, d->k1 - d->g_of_upper, big, d->send_hi, d->capacity, &cnt[1]));
    DIST_HIP(hipMemcpyAsync(d->counts_dev, cnt, sizeof cnt, hipMemcpyHostToDevice, d->stream));
    DIST_TR(d->tr.exchange(d->tr.user, lo ? d->counts_dev + 0 : nullptr, lo ? 8 : 0, lo ? d->counts_dev + 2 : nullptr, lo ? 8 : 0,
                           hi ? d->counts_dev + 1 : nullptr, hi ? 8 : 0, hi ? d->counts_dev + 3 : nullptr, hi ? 8 : 0, d->stream));
    DIST_HIP(hipMemcpyAsync(cnt, d->counts_dev, sizeof cnt, hipMemcpyDeviceToHost, d->stream));
    DIST_HIP(hipStreamSynchronize(d->stream));
    const int64_t m_lo = lo ? cnt[2] : 0, m_hi = hi ? cnt[3] : 0;
    if (m_lo > d->capacity || m_hi > d->capacity) return fail(d, "flip_dist_exchange_particles: a particle message exceeds the capacity given to flip_dist_create");
    // at least one record travels either way, so that no message is empty
    const int64_t rb = 64;
    DIST_TR(d->tr.exchange(d->tr.user, lo ? d->send_lo : nullptr, lo ? (cnt[0] > 1 ? cnt[0] : 1) * rb : 0, lo ? d->recv_lo : nullptr,
                           lo ? (m_lo > 1 ? m_lo : 1) * rb : 0, hi ? d->send_hi : nullptr, hi ? (cnt[1] > 1 ? cnt[1] : 1) * rb : 0,
                           hi ? d->recv_hi : nullptr, hi ? (m_hi > 1 ? m_hi : 1) * rb : 0, d->stream));
    DIST_FLIP(flip_particles_exchange_apply(d->ctx, d->recv_lo, m_lo, d->recv_hi, m_hi));
    d->last_sent[0] = cnt[0]; d->last_sent[1] = cnt[1];
    d->last_received[0] = m_lo; d->last_received[1] = m_hi;
    if (sent) { sent[0] = cnt[0]; sent[1] = cnt[1]; }
    if (received) { received[0] = m_lo; received[1] = m_hi; }
    return 0;
}

extern "C" int flip_dist_last_exchange(const flip_dist *d, int64_t sent[2], int64_t received[2]) {
    if (!d) return 1;
    if (sent) { sent[0] = d->last_sent[0]; sent[1] = d->last_sent[1]; }
    if (received) { received[0] = d->last_received[0]; received[1] = d->last_received[1]; }
    return 0;
}

extern "C" int flip_dist_substep(flip_dist *d, float dt, const flip_dist_step_options *o, flip_pcg_info info[2]) {
    if (!d || !o || !info) return fail(d, "flip_dist_substep: null argument");
    flip_ctx *c = d->ctx;
    memset(info, 0, 2 * sizeof(flip_pcg_info));
    const bool body = o->rigid_body != 0;
    // Stages that do not depend on each other go to the context's two streams (flip_side_fork / select / join); a failed
    // call leaves through the guard, which puts the context back on its own stream.
    struct SideGuard { flip_ctx *c; ~SideGuard() { (void)flip_side_join(c); } } side_guard{ c };
    DIST_FLIP(flip_bin_particles(c, nullptr));
    if (o->levelset_radius > 0.f) DIST_FLIP(flip_fluid_levelset_from_particles(c, o->levelset_radius, 3.0f)); // :324
    DIST_FLIP(flip_side_fork(c));
    DIST_FLIP(flip_side_select(c, 1));
    DIST_FLIP(flip_index_cells(c, nullptr));                  // :486  (reads the face fractions of the previous substep)
    DIST_FLIP(flip_side_select(c, 0));
    DIST_FLIP(flip_p2g(c));                                   // :337
    DIST_FLIP(flip_mark_valid_from_active(c));                // :343-345
    DIST_FLIP(flip_extrapolate(c, 2));                        // :349
    if (o->constrain) {
        DIST_FLIP(flip_constrain_velocity(c, 0));             // :357
        if (body) DIST_FLIP(flip_constrain_velocity(c, 1));   // :358-361
    }
    DIST_FLIP(flip_side_select(c, 1));
    DIST_FLIP(flip_face_fractions(c));                        // :488  (writes the other buffer: the calls above keep the old values)
    DIST_FLIP(flip_side_select(c, 0));
    DIST_FLIP(flip_update_prev(c));                           // :367
    DIST_FLIP(flip_add_forces(c, dt));                        // :373
    DIST_FLIP(flip_density(c, nullptr));                      // :491
    DIST_FLIP(flip_side_join(c));
    if (o->density_projection && (!d->slab || d->pair_solves)) {
        // both solves side by side (:505 and :521): on one context flip_solve_pair, over slabs the two slab solves in turn
        if (flip_dist_solve_pair(d, dt, o->tol, o->max_iter, o->precond, &info[0], &info[1])) return 1;
        DIST_FLIP(flip_side_fork(c));
        DIST_FLIP(flip_side_select(c, 1));
        DIST_FLIP(flip_select_system(c, FLIP_SOLVE_DENSITY));
        DIST_FLIP(flip_project_displacement(c, dt));          // :523
        DIST_FLIP(flip_select_system(c, FLIP_SOLVE_DIVERGENCE));
        DIST_FLIP(flip_side_select(c, 0));
        DIST_FLIP(flip_project_velocity(c, dt));              // :507
        if (body && d->slab) {
            // V += dt Minv (J p), J p summed over every slab's rows (:1095-1100)
            double jp[6];
            DIST_FLIP(flip_solid_coupling_partial(c, jp));
            if (flip_dist_all_reduce(d, jp, 6, FLIP_DIST_SUM)) return 1;
            DIST_FLIP(flip_solid_velocity_apply(c, dt, jp));
        }
    } else {
        if (flip_dist_solve(d, FLIP_SOLVE_DIVERGENCE, dt, o->tol, o->max_iter, o->precond, &info[0])) return 1;   // :505
        DIST_FLIP(flip_project_velocity(c, dt));              // :507
        if (body && d->slab) {
            // V += dt Minv (J p), J p summed over every slab's rows (:1095-1100)
            double jp[6];
            DIST_FLIP(flip_solid_coupling_partial(c, jp));
            if (flip_dist_all_reduce(d, jp, 6, FLIP_DIST_SUM)) return 1;
            DIST_FLIP(flip_solid_velocity_apply(c, dt, jp));
        }
        if (o->density_projection) {
            if (flip_dist_solve(d, FLIP_SOLVE_DENSITY, dt, o->tol, o->max_iter, o->precond, &info[1])) return 1;  // :521
            DIST_FLIP(flip_project_displacement(c, dt));      // :523
        }
    }
    DIST_FLIP(flip_extrapolate(c, 2));                        // :398
    if (o->constrain) {
        DIST_FLIP(flip_constrain_velocity(c, 0));             // :404
        if (body) DIST_FLIP(flip_constrain_velocity(c, 1));   // :405-408
    }
    DIST_FLIP(flip_update_diff(c));                           // :416
    DIST_FLIP(flip_side_join(c));
    // :420 transfer_from_grid_to_particles, :426 advectParticles, :429 apply_particle_displacements: one pass, same bits
    DIST_FLIP(flip_update_particles(c, o->flip_pic_ratio, dt, o->advect, o->density_projection));
    return flip_dist_exchange_particles(d, nullptr, nullptr);
}
