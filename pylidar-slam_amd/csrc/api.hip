// C ABI of libicp_mi355x.so: context management, staging, and the orchestration of the kernels in the other
// translation units.  Signatures and the reference interfaces they replace: include/icp_mi355x.h.
#include <math.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "icp_internal.h"
#include "map_move_device.h"

using namespace icp;

namespace icp {

hipError_t DeviceBuffer::reserve(size_t need, bool keep, hipStream_t stream) {
    if (need <= bytes && ptr) return hipSuccess;
    size_t cap = bytes ? bytes : 256;
    while (cap < need) cap = cap + cap / 2 + 256;
    void* np = nullptr;
    hipError_t e = hipMalloc(&np, cap);
    if (e != hipSuccess) return e;
    if (ptr) {
        if (keep && bytes) {
            e = hipMemcpyAsync(np, ptr, bytes, hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) return e;
        }
        // hipFree synchronises the device, so pending work on the old block has drained
        (void)hipFree(ptr);
    }
    ptr = np;
    bytes = cap;
    return hipSuccess;
}

void DeviceBuffer::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
}

int prof_begin(icp_ctx* ctx, int kind, int iter) {
    Profile& p = ctx->prof;
    if (!p.enabled || !p.sample_now || !((p.mask >> kind) & 1)) return -1;
    if (kind == 0 && p.rotate && iter >= 0 && iter != p.rotate_index) return -1;
    const int ev = (int)p.pending.size();
    if (ev >= (int)p.pool.size()) {
        hipEvent_t a, b;
        // (timing events only: no system-scope fence when they are recorded — the cache write-back and invalidation of the
        // default flavour is charged to the work behind the event, hip_runtime_api.h; they are read after the registration's
        // own result event has been waited for)
        if (hipEventCreateWithFlags(&a, hipEventDisableSystemFence) != hipSuccess ||
            hipEventCreateWithFlags(&b, hipEventDisableSystemFence) != hipSuccess)
            return -1;
        p.pool.push_back({a, b});
    }
    (void)hipEventRecord(p.pool[ev].first, ctx->stream);
    p.pending.push_back({kind, ev, iter});
    return ev;
}

void prof_end(icp_ctx* ctx, int token) {
    if (token < 0) return;
    (void)hipEventRecord(ctx->prof.pool[token].second, ctx->stream);
}

static void prof_collect(icp_ctx* ctx) {
    Profile& p = ctx->prof;
    for (auto& r : p.pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.pool[r.ev].first, p.pool[r.ev].second) == hipSuccess) {
            p.ms[r.kind] += ms;
            p.launches[r.kind] += 1;
            if (r.kind == 0 && r.iter >= 0) {
                const int slot = r.iter < Profile::ITER_SLOTS ? r.iter : Profile::ITER_SLOTS - 1;
                p.ms_iter[slot] += ms;
                p.launches_iter[slot] += 1;
            }
        }
    }
    p.pending.clear();
}

// ---- small kernels owned by the API layer ---------------------------------------------------------------------------
__global__ void k_state_init(RegState* st, Pose16 init, int keep_pose, unsigned long long* box, unsigned gen,
                             float* hist) {
    if (threadIdx.x >= 64 || blockIdx.x != 0) return;
    // the state, generation `gen` of the pose mailbox (the initial guess) and entry 0 of the pose history
    state_init_wave(st, init.m, keep_pose, box, gen, hist, (int)threadIdx.x);
}

// what an event pair adds: icp_profile_event_floor brackets this kernel like a real launch — it spins for `ticks` of the
// 100 MHz wall clock (ticks = 0: returns at once), so its run time is known from the inside
__global__ void k_event_floor(long long ticks) {
    if (ticks <= 0 || threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}

__global__ void k_flag_not_nan(const float* __restrict__ xyz, long long n, int skip_null, int* __restrict__ flags) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    bool ok = (x == x && y == y && z == z);
    if (skip_null && x == 0.f && y == 0.f && z == 0.f) ok = false;
    flags[i] = ok ? 1 : 0;
}

// planar [3,H*W] vertex map -> interleaved points + "norm > thr and not NaN" flags (local_map.py:320-327)
__global__ void k_vmap_points(const float* __restrict__ vmap, int npix, float thr, float* __restrict__ pts,
                              int* __restrict__ flags) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const float x = vmap[p], y = vmap[npix + p], z = vmap[2 * npix + p];
    pts[3 * p] = x;
    pts[3 * p + 1] = y;
    pts[3 * p + 2] = z;
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
    flags[p] = (nrm > thr) ? 1 : 0;  // NaN compares false
}

}  // namespace icp

static constexpr size_t DBG_ITER_BYTES = 64 + 24 * 1024 * 4 * sizeof(long long);
static constexpr size_t DBG_BYTES = DBG_ITER_BYTES + 8192 * 4 * sizeof(long long);  // dev statistics ("search_stats"): + the normal kernel's blocks

// ---- helpers --------------------------------------------------------------------------------------------------------
// Every entry point runs on the context's device and leaves the calling thread's current device as it found it (a
// process may hold contexts on several GPUs, and torch shares the thread's current device with us).
// ... and — `join` — orders itself behind a map update still running on the context's map stream ("overlap_map_update":
// the re-expression, grid rebuild and normal estimation behind a registration run on a stream of their own, beside the next
// frame's preprocessing on the caller's stream).  Entry points that touch neither the map nor its search structure nor the
// registration state (projection, grid sample, de-skew, compaction of targets, the wait for a pose) do not join.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const icp_ctx* ctx, bool join = true) : DeviceGuard(ctx ? ctx->cfg.device : -1) {
        if (join && ctx && ctx->map_stream_busy) {
            icp_ctx* c = const_cast<icp_ctx*>(ctx);
            (void)hipStreamWaitEvent(c->stream, c->map_done_event, 0);
            c->map_stream_busy = false;
        }
    }
    explicit DeviceGuard(int device) {
        if (device < 0) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

// Contexts of this process with a registration in flight, per device: the lead launches (and the resident tail) make the
// workgroups of a launch wait for one another, which is only worth it — and only safe from a 50 ms bail-out — while no
// other launch competes for the CUs.  A context that finds another one registering on its device enqueues per-iteration
// launches with a solving launch each for that registration (contexts of OTHER processes cannot be
// seen: there the timed-out hand-off is recovered and the context keeps to plain launches, recover_handoff).
static std::atomic<int> g_registering[64];

static void registering_enter(icp_ctx* ctx) {
    if (ctx->counted_registering) return;
    const int d = ctx->cfg.device >= 0 && ctx->cfg.device < 64 ? ctx->cfg.device : 0;
    g_registering[d].fetch_add(1);
    ctx->counted_registering = true;
}
static void registering_leave(icp_ctx* ctx) {
    if (!ctx->counted_registering) return;
    const int d = ctx->cfg.device >= 0 && ctx->cfg.device < 64 ? ctx->cfg.device : 0;
    g_registering[d].fetch_sub(1);
    ctx->counted_registering = false;
}
static bool device_shared_with_another_registration(const icp_ctx* ctx) {
    const int d = ctx->cfg.device >= 0 && ctx->cfg.device < 64 ? ctx->cfg.device : 0;
    return g_registering[d].load() > (ctx->counted_registering ? 1 : 0);
}

// An entry point of the registration that leaves on an error must not leave its context counted among the registering ones
// (every other context of the device then ran without lead launches for the life of the process): on scope exit,
// a context with no registration in progress and no result pending leaves the count.
struct RegisteringGuard {
    icp_ctx* ctx;
    explicit RegisteringGuard(icp_ctx* c) : ctx(c) {}
    ~RegisteringGuard() {
        if (ctx && !ctx->in_registration && ctx->r_count == 0) registering_leave(ctx);
    }
};

static int fail(icp_ctx* ctx, int code, const char* msg) {
    if (ctx) ctx->error = msg;
    return code;
}

// returns a device pointer for `n_bytes` of caller data (staging a host buffer through `stage`)
static int import_buffer(icp_ctx* ctx, const void* src, size_t n_bytes, int mem, DeviceBuffer& stage,
                         const void** dev_out) {
    if (n_bytes == 0) {
        *dev_out = nullptr;
        return ICP_OK;
    }
    if (!src) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "null input pointer");
    if (mem == ICP_MEM_DEVICE) {
        *dev_out = src;
        return ICP_OK;
    }
    ICP_HIP(ctx, stage.reserve(n_bytes));
    ICP_HIP(ctx, hipMemcpyAsync(stage.ptr, src, n_bytes, hipMemcpyHostToDevice, ctx->stream));
    *dev_out = stage.ptr;
    return ICP_OK;
}

// device scratch or the caller's device pointer, depending on where the output lives
static int export_target(icp_ctx* ctx, void* dst, size_t n_bytes, int out_mem, DeviceBuffer& stage, void** dev_out) {
    if (!dst || n_bytes == 0) {
        *dev_out = nullptr;
        return ICP_OK;
    }
    if (out_mem == ICP_MEM_DEVICE) {
        *dev_out = dst;
        return ICP_OK;
    }
    ICP_HIP(ctx, stage.reserve(n_bytes));
    *dev_out = stage.ptr;
    return ICP_OK;
}

static int export_finish(icp_ctx* ctx, void* dst, const void* dev, size_t n_bytes, int out_mem) {
    if (!dst || n_bytes == 0 || out_mem == ICP_MEM_DEVICE) return ICP_OK;
    ICP_HIP(ctx, hipMemcpyAsync(dst, dev, n_bytes, hipMemcpyDeviceToHost, ctx->stream));
    return ICP_OK;
}

static int ensure_state(icp_ctx* ctx) {
    static_assert(sizeof(RegState) <= STATE_BLOCK, "RegState outgrew its slot");
    const int cap = ctx->cfg.max_num_alignments > 1 ? ctx->cfg.max_num_alignments : 1;
    if (cap > ctx->hist_cap || !ctx->state.ptr) {
        const int newcap = cap > ctx->hist_cap ? cap : ctx->hist_cap;
        ICP_HIP(ctx, ctx->state.reserve(STATE_BLOCK + (size_t)newcap * (sizeof(double) + 6 * sizeof(float))));
        ctx->hist_cap = newcap;
        ctx->have_device_pose = false;  // a fresh allocation holds no registration result
        ctx->stats_pending = false;     // ... and no grid statistics
        ICP_HIP(ctx, hipMemsetAsync(ctx->state.ptr, 0, STATE_BLOCK, ctx->stream));
        ctx->loss_hist = (double*)(ctx->state.as<char>() + STATE_BLOCK);
        ctx->dx_hist = (float*)(ctx->state.as<char>() + STATE_BLOCK + (size_t)newcap * sizeof(double));
        ICP_HIP(ctx, ctx->pose_hist_buf.reserve((size_t)(newcap + 1) * 12 * sizeof(float)));
        ctx->pose_hist = ctx->pose_hist_buf.as<float>();
    }
    ICP_HIP(ctx, ctx->neq_own.reserve(NEQ * sizeof(double)));
    if (!ctx->neq) ctx->neq = ctx->neq_own.as<double>();
    ICP_HIP(ctx, ctx->counter.reserve(64));
    if (!ctx->posebox.ptr) {  // pose mailbox (2 parities) + ticket counter of the lead launches
        ICP_HIP(ctx, ctx->posebox.reserve(BOX_BYTES));
        ICP_HIP(ctx, hipMemsetAsync(ctx->posebox.ptr, 0, ctx->posebox.bytes, ctx->stream));
        ctx->box_gen = 0;
    }
    return ICP_OK;
}

static Pose16 pose_or_identity(const float* init_pose) {
    Pose16 p;
    if (init_pose) {
        memcpy(p.m, init_pose, sizeof(p.m));
    } else {
        memset(p.m, 0, sizeof(p.m));
        p.m[0] = p.m[5] = p.m[10] = p.m[15] = 1.f;
    }
    return p;
}

// targets packed and the state initialised by one launch (a launch of its own only when there are no targets)
static int prepare_targets_and_state(icp_ctx* ctx, int64_t n, const float* init_pose, bool keep_pose = false,
                                     PackDesc* defer = nullptr) {
    const Pose16 p = pose_or_identity(init_pose);
    if (defer) memset(defer, 0, sizeof(*defer));
    if (n > 0) return prepare_targets(ctx, ctx->tgt_ptr, n, &p, keep_pose, defer);
    hipLaunchKernelGGL(k_state_init, dim3(1), dim3(64), 0, ctx->stream, reg_state(ctx), p, keep_pose ? 1 : 0,
                       pose_box(ctx), next_box_generation(ctx), ctx->pose_hist);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

static int init_state(icp_ctx* ctx, const float* init_pose, bool keep_pose = false) {
    const Pose16 p = pose_or_identity(init_pose);
    hipLaunchKernelGGL(k_state_init, dim3(1), dim3(64), 0, ctx->stream, reg_state(ctx), p, keep_pose ? 1 : 0,
                       pose_box(ctx), next_box_generation(ctx), ctx->pose_hist);
    ICP_HIP(ctx, hipGetLastError());
    return ICP_OK;
}

static void exchange_release(icp_ctx* ctx);
static int continue_launch(icp_ctx* ctx, int count);

// Normals of the whole map at once (one dense launch, then the fused iteration kernel) or lazily for the map points the
// scan touches (local_map.py:397-422; three more launches per iteration, each a latency-bound chain when the scan is
// small)?  Same values either way.  Eager costs ~1 us per 1000 map points; measured, it wins whenever the map is at most
// twice the scan, and for any scan while the map stays below ~10^6 points (a 6 000-point grid sample against 180 000
// map points: 0.19 ms eager vs 0.45 ms for four lazy iterations; 200 000 points against 10^6: 2.8 vs 4.7 ms per frame of
// twenty iterations).
// (wants_eager_normals: icp_internal.h — the grid build asks the same question before it builds the neighbourhood lists)
static int enqueue_iterations(icp_ctx* ctx, bool poll_allowed, int first = 0, int count = -1);

// ---- lifecycle ------------------------------------------------------------------------------------------------------
extern "C" {

const char* icp_version(void) { return "icp_mi355x 0.1.0 (gfx950)"; }

void icp_default_config(icp_config* cfg) {
    if (!cfg) return;
    memset(cfg, 0, sizeof(*cfg));
    cfg->height = 64;
    cfg->width = 1024;
    cfg->up_fov = 3.0f;
    cfg->down_fov = -24.0f;
    cfg->max_num_alignments = 100;       // icp_odometry.py:37
    cfg->threshold_delta_pose = 1.0e-4f; // :48
    cfg->scheme = ICP_SCHEME_LEAST_SQUARE;  // alignment.py:77 (ConfigStore default: plain least squares)
    cfg->sigma = 0.5f;
    cfg->local_map_size = 20;            // local_map.py:249
    cfg->num_neighbors_normals = 10;     // :250
    cfg->cell_size = 0.0f;  // <= 0: auto-tuned from the measured map occupancy
    cfg->max_rings = 2;  // fine rings; beyond that the coarse level (4x cells, 6 rings) takes over (3 measured slower)
    cfg->device = 0;
    cfg->poll_every = 4;
}

int icp_create(const icp_config* cfg, icp_ctx** out) {
    if (!cfg || !out) return ICP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || cfg->device >= count) return ICP_ERR_NO_DEVICE;
    if (cfg->height <= 0 || cfg->width <= 0 || cfg->max_rings < 1 ||
        cfg->num_neighbors_normals < 1 || cfg->num_neighbors_normals > 64 || cfg->local_map_size < 1 ||
        cfg->max_num_alignments < 1)
        return ICP_ERR_INVALID_ARGUMENT;
    DeviceGuard device_guard(cfg->device);
    int current = -1;
    if (hipGetDevice(&current) != hipSuccess || current != cfg->device) return ICP_ERR_HIP;
    icp_ctx* ctx = new icp_ctx();
    ctx->cfg = *cfg;
    ctx->cell_h = cfg->cell_size > 0.f ? cfg->cell_size : 0.5f;
    int rc = ensure_state(ctx);
    if (rc == ICP_OK) rc = init_state(ctx, nullptr);
    if (rc != ICP_OK) {
        delete ctx;
        return rc;
    }
    *out = ctx;
    return ICP_OK;
}

void icp_destroy(icp_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard device_guard(ctx);
    registering_leave(ctx);
    (void)hipDeviceSynchronize();
    DeviceBuffer* bufs[] = {&ctx->map_xyz[0], &ctx->map_xyz[1], &ctx->table,   &ctx->sorted_pts, &ctx->normals,
                            &ctx->nflag,      &ctx->slot_of,    &ctx->rank_of, &ctx->scan_tmp,   &ctx->worklist,
                            &ctx->targets,    &ctx->nn_pos,     &ctx->partials, &ctx->state,
                            &ctx->neq_own,    &ctx->zbuf,    &ctx->stage_in,   &ctx->stage_out,
                            &ctx->stage_out2, &ctx->flags,      &ctx->scan_a,  &ctx->scan_b,     &ctx->sort_tmp,
                            &ctx->keys_a,     &ctx->keys_b,     &ctx->vals_a,  &ctx->vals_b,     &ctx->counter,
                            &ctx->tgt4,       &ctx->slot_of_cell,
                            &ctx->rows,       &ctx->row_of_pos, &ctx->cslot_of,
                            &ctx->csorted,    &ctx->crank_of,  &ctx->pos_of_orig, &ctx->dbg_counts, &ctx->nn_cache,  &ctx->pm_v,
                            &ctx->pm_n,       &ctx->pm_mv,      &ctx->pm_mn,      &ctx->pm_z,      &ctx->pm_tmp,
                            &ctx->vox_out,    &ctx->seed_orig,  &ctx->scan_desc,  &ctx->posebox,   &ctx->pose_hist_buf,
                            &ctx->hood,       &ctx->normals_carry, &ctx->tail_rows, &ctx->normals_tail, &ctx->nn_rec,
                            &ctx->cell_list[0][0], &ctx->cell_list[0][1], &ctx->cell_list[1][0], &ctx->cell_list[1][1], &ctx->cell_counts};
    for (DeviceBuffer* b : bufs) b->release();
    for (auto& r : ctx->rslot) {
        if (r.host) (void)hipHostFree(r.host);
        if (r.event) (void)hipEventDestroy(r.event);
    }
    if (ctx->switch_event) (void)hipEventDestroy(ctx->switch_event);
    if (ctx->map_done_event) (void)hipEventDestroy(ctx->map_done_event);
    if (ctx->map_start_event) (void)hipEventDestroy(ctx->map_start_event);
    if (ctx->map_stream) (void)hipStreamDestroy(ctx->map_stream);
    ctx->staged_xyz.release();
    if (ctx->staged_count_host) (void)hipHostFree(ctx->staged_count_host);
    if (ctx->staged_event) (void)hipEventDestroy(ctx->staged_event);
    exchange_release(ctx);
    ctx->x_seq.release();
    for (auto& e : ctx->prof.pool) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    delete ctx;
}

const char* icp_last_error(const icp_ctx* ctx) { return ctx ? ctx->error.c_str() : "null context"; }

int icp_set_stream(icp_ctx* ctx, void* hip_stream) {
    DeviceGuard device_guard(ctx, false);  // (callers re-announce their stream before every call: no join for the same stream)
    if (!ctx) return ICP_ERR_INVALID_ARGUMENT;
    hipStream_t next = (hipStream_t)hip_stream;
    // (the same stream re-announced: nothing to order — but iterations a chunked launch holds back still go first, as the
    // header promises for every entry point behind which the caller may enqueue work of its own)
    if (next == ctx->stream && ctx->launch_remaining <= 0) return ICP_OK;
    { DeviceGuard join_map_stream(ctx); }  // a new stream: the old one first takes in the map stream, the new one follows it
    {   // iterations a chunked launch still holds back run against the map / configuration they were launched with
        const int rc_held = continue_launch(ctx, -1);
        if (rc_held) return rc_held;
    }
    if (next != ctx->stream) {
        // work already enqueued on the previous stream (map builds, a registration in flight) must precede what follows
        if (!ctx->switch_event) ICP_HIP(ctx, hipEventCreateWithFlags(&ctx->switch_event, hipEventDisableTiming));
        ICP_HIP(ctx, hipEventRecord(ctx->switch_event, ctx->stream));
        ICP_HIP(ctx, hipStreamWaitEvent(next, ctx->switch_event, 0));
        ctx->stream = next;
    }
    return ICP_OK;
}

int icp_set_option(icp_ctx* ctx, const char* name, double value) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !name) return ICP_ERR_INVALID_ARGUMENT;
    {   // iterations a chunked launch still holds back run against the map / configuration they were launched with
        const int rc_held = continue_launch(ctx, -1);
        if (rc_held) return rc_held;
    }
    if (ctx->in_registration) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "registration in progress");
    const std::string k(name);
    const int iv = (int)value;
    if (k == "nn_cache") ctx->use_nn_cache = iv < 0 ? 0 : (iv > 2 ? 2 : iv);
    else if (k == "hit_records") ctx->hit_records = iv != 0 ? 1 : 0;
    else if (k == "late_from") ctx->late_from = iv < 0 ? -1 : (int)iv;
    else if (k == "late_waves") ctx->late_waves = iv >= 8 ? 8 : 6;
    else if (k == "fuse_iteration") ctx->fuse_iteration = iv != 0;
    else if (k == "iterate_dense") ctx->iterate_dense = iv != 0;
    else if (k == "narrow_from") ctx->narrow_from = (int)iv;
    else if (k == "profile_every") ctx->prof.every = iv < 1 ? 1 : (int)iv;
    else if (k == "profile_rotate") ctx->prof.rotate = iv != 0;
    else if (k == "wave_misses") ctx->wave_misses = iv < 0 ? 0 : (int)iv;
    else if (k == "wave_misses_dense") ctx->wave_misses_dense = iv < 0 ? 0 : (int)iv;
    else if (k == "frame_seed") { ctx->frame_seed = iv != 0; ctx->seed_n = 0; }
    else if (k == "knn_rings") ctx->knn_rings = iv;
    else if (k == "knn_lanes") ctx->knn_lanes = iv == 2 ? 2 : 4;
    else if (k == "scan_poll_limit") ctx->scan_poll_limit = iv < 0 ? 0 : iv;
    else if (k == "exchange_timeout_ms") ctx->exchange_timeout_ms = value > 1.0 ? value : 1.0;
    else if (k == "lead_solve") { ctx->lead_solve = value != 0.0 ? 1 : 0; if (ctx->lead_solve) ctx->tail_disabled = ctx->handoff_disabled = false; }
    else if (k == "resident_tail") { ctx->resident_tail = iv < 0 ? 0 : (int)iv; ctx->tail_disabled = false; }
    else if (k == "resident_tail_max_blocks") ctx->resident_tail_max_blocks = iv < 0 ? 0 : (int)iv;
    else if (k == "ball_search") ctx->ball_search = value != 0.0 ? 1 : 0;
    else if (k == "wide_until") ctx->wide_until = iv < 0 ? 0 : (int)iv;
    else if (k == "far_lanes") ctx->far_lanes = iv >= 16 ? 16 : 0;
    else if (k == "far_min") ctx->far_min = iv < 0 ? 0 : (int)iv;
    else if (k == "far_max") ctx->far_max = iv < 0 ? 0 : (iv > 512 ? 512 : (int)iv);
    else if (k == "insert_by_cell") ctx->insert_by_cell = iv != 0 ? 1 : 0;
    else if (k == "cell_lists") ctx->cell_lists = iv != 0 ? 1 : 0;
    else if (k == "normals_tail_stream") ctx->normals_tail_stream = iv != 0 ? 1 : 0;
    else if (k == "normals_list") ctx->normals_list = iv != 0 ? 1 : 0;
    else if (k == "ball_lanes") ctx->ball_lanes = iv >= 8 ? 8 : (iv >= 4 ? 4 : (iv >= 2 ? 2 : 1));
    else if (k == "ball_empty") ctx->ball_empty = iv != 0 ? 1 : 0;
    else if (k == "ball_max") ctx->ball_max = iv < 4 ? 4 : (iv > 256 ? 256 : (int)iv);
    else if (k == "lead_timeout_ms") ctx->lead_timeout_ms = value > 1.0e-5 ? value : 1.0e-5;  // (>= one tick of the 100 MHz clock: tests go there)
    else if (k == "chunked_launch") ctx->chunked_launch = value != 0.0 ? 1 : 0;
    else if (k == "flat_rows") ctx->flat_rows = iv < 0 ? 0 : (iv > 2 ? 2 : (int)iv);
    else if (k == "xcd_sectors") ctx->xcd_sectors = value != 0.0 ? 1 : 0;
    else if (k == "lead_after_dense") ctx->lead_after_dense = value != 0.0 ? 1 : 0;
    else if (k == "refresh_margin") ctx->refresh_margin = value > 0.0 ? (float)value : 0.f;
    else if (k == "refresh_at") ctx->refresh_at = iv;
    else if (k == "prune_guard") ctx->prune_guard = value > 0.0 ? (float)value : 0.f;
    else if (k == "hoods") ctx->hoods = iv < 0 ? 0 : (iv > 2 ? 2 : (int)iv);
    else if (k == "lazy_fused") ctx->lazy_fused = iv < 0 ? 0 : (iv > 2 ? 2 : (int)iv);
    else if (k == "lazy_fused_ratio") ctx->lazy_fused_ratio = value > 0.0 ? value : 4.0;
    else if (k == "carry_normals") ctx->carry_normals = value != 0.0 ? 1 : 0;
    else if (k == "overlap_map_update") ctx->overlap_map_update = value != 0.0 ? 1 : 0;
    else if (k == "eager_normals_limit") ctx->eager_normals_limit = value > 0.0 ? (long long)value : 0;
    else if (k == "target_occupancy") ctx->target_occupancy = value > 0.1 ? value : 16.0;
    else if (k == "search_stats") {
        // 1: path counters + phase stamps, 2: stamps only (no atomics); 3 / 4: as 1 / 2, plus the per-workgroup dump
        ctx->search_stats_blocks = iv == 3 || iv == 4;
        ctx->search_stats = iv == 3 ? 1 : (iv == 4 ? 2 : (int)iv);
        if (ctx->search_stats) {  // 16 path counters + 4 phase timestamps per workgroup and iteration
            ICP_HIP(ctx, ctx->dbg_counts.reserve(DBG_BYTES));
            ICP_HIP(ctx, hipMemsetAsync(ctx->dbg_counts.ptr, 0, DBG_BYTES, ctx->stream));
        }
    } else {
        return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "unknown option");
    }
    return ICP_OK;
}

int icp_set_cost(icp_ctx* ctx, int32_t cost) {
    DeviceGuard device_guard(ctx);
    if (!ctx) return ICP_ERR_INVALID_ARGUMENT;
    {   // iterations a chunked launch still holds back run against the map / configuration they were launched with
        const int rc_held = continue_launch(ctx, -1);
        if (rc_held) return rc_held;
    }
    if (cost != ICP_COST_POINT_TO_PLANE && cost != ICP_COST_POINT_TO_POINT)
        return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "unknown alignment mode");
    if (ctx->in_registration) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "registration in progress");
    ctx->cost = cost;
    return ICP_OK;
}

int icp_synchronize(icp_ctx* ctx) {
    DeviceGuard device_guard(ctx);
    if (!ctx) return ICP_ERR_INVALID_ARGUMENT;
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (the guard above has ordered the stream behind the map stream)
    return ICP_OK;
}

int icp_set_alignment(icp_ctx* ctx, int32_t scheme, float sigma, int32_t max_num_alignments,
                      float threshold_delta_pose) {
    DeviceGuard device_guard(ctx);
    if (!ctx) return ICP_ERR_INVALID_ARGUMENT;
    {   // iterations a chunked launch still holds back run against the map / configuration they were launched with
        const int rc_held = continue_launch(ctx, -1);
        if (rc_held) return rc_held;
    }
    if (scheme < 0 || scheme > ICP_SCHEME_CAUCHY || max_num_alignments < 1)
        return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "bad alignment parameters");
    ctx->cfg.scheme = scheme;
    ctx->cfg.sigma = sigma;
    ctx->cfg.max_num_alignments = max_num_alignments;
    ctx->cfg.threshold_delta_pose = threshold_delta_pose;
    return ensure_state(ctx);
}

// ---- projection -----------------------------------------------------------------------------------------------------
int icp_project(icp_ctx* ctx, const float* xyz, int64_t n, int mem, float* vmap_out, int32_t* index_out, int out_mem) {
    DeviceGuard device_guard(ctx, false);  // (beside a map update on its own stream)
    if (!ctx || n < 0) return ICP_ERR_INVALID_ARGUMENT;
    const size_t npix = (size_t)ctx->cfg.height * ctx->cfg.width;
    const void* in;
    int rc = import_buffer(ctx, xyz, (size_t)n * 12, mem, ctx->stage_in, &in);
    if (rc) return rc;
    void *vdev, *idev;
    if ((rc = export_target(ctx, vmap_out, npix * 12, out_mem, ctx->stage_out, &vdev))) return rc;
    if ((rc = export_target(ctx, index_out, npix * 4, out_mem, ctx->stage_out2, &idev))) return rc;
    if ((rc = project_device(ctx, (const float*)in, n, (float*)vdev, (int32_t*)idev))) return rc;
    if ((rc = export_finish(ctx, vmap_out, vdev, npix * 12, out_mem))) return rc;
    if ((rc = export_finish(ctx, index_out, idev, npix * 4, out_mem))) return rc;
    if (out_mem == ICP_MEM_HOST || mem == ICP_MEM_HOST) ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

int icp_project_rows(icp_ctx* ctx, const float* xyz, int64_t n, float* vmap_out, float* rows_out) {
    DeviceGuard device_guard(ctx, false);  // (beside a map update on its own stream)
    if (!ctx || n < 0 || (n > 0 && !xyz) || !vmap_out || !rows_out) return ICP_ERR_INVALID_ARGUMENT;
    return project_device(ctx, xyz, n, vmap_out, nullptr, false, rows_out);
}

int icp_project_pixels(icp_ctx* ctx, const float* xyz, int64_t n, int mem, float* rows_out, float* cols_out,
                       int out_mem) {
    DeviceGuard device_guard(ctx, false);  // (beside a map update on its own stream)
    if (!ctx || n < 0 || !rows_out || !cols_out) return ICP_ERR_INVALID_ARGUMENT;
    const void* in;
    int rc = import_buffer(ctx, xyz, (size_t)n * 12, mem, ctx->stage_in, &in);
    if (rc) return rc;
    void *rdev, *cdev;
    if ((rc = export_target(ctx, rows_out, (size_t)n * 4, out_mem, ctx->stage_out, &rdev))) return rc;
    if ((rc = export_target(ctx, cols_out, (size_t)n * 4, out_mem, ctx->stage_out2, &cdev))) return rc;
    if ((rc = project_pixels_device(ctx, (const float*)in, n, (float*)rdev, (float*)cdev))) return rc;
    if ((rc = export_finish(ctx, rows_out, rdev, (size_t)n * 4, out_mem))) return rc;
    if ((rc = export_finish(ctx, cols_out, cdev, (size_t)n * 4, out_mem))) return rc;
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

int icp_kitti_correct_scan(icp_ctx* ctx, const float* scan, int64_t n, int stride, int mem, double* xyz_out,
                           int out_mem) {
    DeviceGuard device_guard(ctx, false);  // (beside a map update on its own stream)
    if (!ctx || n < 0 || stride < 3 || (n > 0 && (!scan || !xyz_out))) return ICP_ERR_INVALID_ARGUMENT;
    const void* in;
    int rc = import_buffer(ctx, scan, (size_t)n * stride * 4, mem, ctx->stage_in, &in);
    if (rc) return rc;
    void* odev;
    if ((rc = export_target(ctx, xyz_out, (size_t)n * 24, out_mem, ctx->stage_out, &odev))) return rc;
    if ((rc = kitti_correct_device(ctx, (const float*)in, n, stride, (double*)odev))) return rc;
    if ((rc = export_finish(ctx, xyz_out, odev, (size_t)n * 24, out_mem))) return rc;
    if (out_mem == ICP_MEM_HOST || mem == ICP_MEM_HOST) ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

// ---- grid sampling --------------------------------------------------------------------------------------------------
int icp_voxel_hash(icp_ctx* ctx, const float* xyz, int64_t n, int mem, double voxel_size, int64_t* voxels_out,
                   int64_t* hashes_out, int out_mem) {
    DeviceGuard device_guard(ctx, false);  // (beside a map update on its own stream)
    if (!ctx || n < 0 || !(voxel_size > 0)) return ICP_ERR_INVALID_ARGUMENT;
    const void* in;
    int rc = import_buffer(ctx, xyz, (size_t)n * 12, mem, ctx->stage_in, &in);
    if (rc) return rc;
    void *vdev, *hdev;
    if ((rc = export_target(ctx, voxels_out, (size_t)n * 24, out_mem, ctx->stage_out, &vdev))) return rc;
    if ((rc = export_target(ctx, hashes_out, (size_t)n * 8, out_mem, ctx->stage_out2, &hdev))) return rc;
    if ((rc = voxel_hash_device(ctx, (const float*)in, n, voxel_size, (long long*)vdev, (long long*)hdev))) return rc;
    if ((rc = export_finish(ctx, voxels_out, vdev, (size_t)n * 24, out_mem))) return rc;
    if ((rc = export_finish(ctx, hashes_out, hdev, (size_t)n * 8, out_mem))) return rc;
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

int icp_grid_sample(icp_ctx* ctx, const float* xyz, int64_t n, int mem, double voxel_size, int64_t* indices_out,
                    float* points_out, int64_t* count_out, int out_mem) {
    DeviceGuard device_guard(ctx, false);  // (beside a map update on its own stream)
    if (!ctx || n < 0 || !(voxel_size > 0) || !count_out) return ICP_ERR_INVALID_ARGUMENT;
    int rc = ensure_state(ctx);
    if (rc) return rc;
    const void* in;
    if ((rc = import_buffer(ctx, xyz, (size_t)n * 12, mem, ctx->stage_in, &in))) return rc;
    void *idev, *pdev;
    if ((rc = export_target(ctx, indices_out, (size_t)n * 8, out_mem, ctx->stage_out, &idev))) return rc;
    if ((rc = export_target(ctx, points_out, (size_t)n * 12, out_mem, ctx->stage_out2, &pdev))) return rc;
    int* count_dev = ctx->counter.as<int>();
    int count = 0;  // (read back inside: the sort of the distinct voxels is sized by it)
    if ((rc = grid_sample_device(ctx, (const float*)in, n, voxel_size, (long long*)idev, (float*)pdev, count_dev, &count)))
        return rc;
    *count_out = count;
    if ((rc = export_finish(ctx, indices_out, idev, (size_t)count * 8, out_mem))) return rc;
    if ((rc = export_finish(ctx, points_out, pdev, (size_t)count * 12, out_mem))) return rc;
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

int icp_voxel_statistics(icp_ctx* ctx, const float* xyz, int64_t n, int mem, double voxel_size, int64_t* voxels_out,
                         int64_t* hashes_out, int64_t* voxel_ids_out, int64_t* num_voxels_out, int64_t* sizes_out,
                         float* means_out, float* covs_out, int out_mem) {
    DeviceGuard device_guard(ctx, false);  // (beside a map update on its own stream)
    if (!ctx || n < 0 || !(voxel_size > 0) || !num_voxels_out || !voxel_ids_out) return ICP_ERR_INVALID_ARGUMENT;
    const bool stats = sizes_out || means_out || covs_out;
    if (stats && !(sizes_out && means_out && covs_out)) return ICP_ERR_INVALID_ARGUMENT;
    int rc = ensure_state(ctx);
    if (rc) return rc;
    *num_voxels_out = 0;
    if (n == 0) return ICP_OK;
    const void* in;
    if ((rc = import_buffer(ctx, xyz, (size_t)n * 12, mem, ctx->stage_in, &in))) return rc;
    // device side of every output: the caller's buffers (device) or one staging block (host)
    const size_t off_vox = 0, off_hash = off_vox + (size_t)n * 24, off_ids = off_hash + (size_t)n * 8,
                 off_sizes = off_ids + (size_t)n * 8, off_means = off_sizes + (size_t)n * 8,
                 off_covs = off_means + (size_t)n * 12, total = off_covs + (size_t)n * 36;
    char* base = nullptr;
    if (out_mem != ICP_MEM_DEVICE) {
        ICP_HIP(ctx, ctx->vox_out.reserve(total));
        base = ctx->vox_out.as<char>();
    }
    auto dev = [&](void* user, size_t off) -> void* {
        if (!user) return nullptr;
        return out_mem == ICP_MEM_DEVICE ? user : (void*)(base + off);
    };
    long long* d_vox = (long long*)dev(voxels_out, off_vox);
    long long* d_hash = (long long*)dev(hashes_out, off_hash);
    long long* d_ids = (long long*)dev(voxel_ids_out, off_ids);
    long long* d_sizes = (long long*)dev(sizes_out, off_sizes);
    float* d_means = (float*)dev(means_out, off_means);
    float* d_covs = (float*)dev(covs_out, off_covs);
    int* count_dev = ctx->counter.as<int>();
    if ((rc = voxel_statistics_device(ctx, (const float*)in, n, voxel_size, d_vox, d_hash, d_ids, d_sizes, d_means,
                                      d_covs, count_dev)))
        return rc;
    int count = 0;
    ICP_HIP(ctx, hipMemcpyAsync(&count, count_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *num_voxels_out = count;
    if ((rc = export_finish(ctx, voxels_out, d_vox, (size_t)n * 24, out_mem))) return rc;
    if ((rc = export_finish(ctx, hashes_out, d_hash, (size_t)n * 8, out_mem))) return rc;
    if ((rc = export_finish(ctx, voxel_ids_out, d_ids, (size_t)n * 8, out_mem))) return rc;
    if ((rc = export_finish(ctx, sizes_out, d_sizes, (size_t)count * 8, out_mem))) return rc;
    if ((rc = export_finish(ctx, means_out, d_means, (size_t)count * 12, out_mem))) return rc;
    if ((rc = export_finish(ctx, covs_out, d_covs, (size_t)count * 36, out_mem))) return rc;
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

// the device-resident pipeline's grid sample: nothing comes back to the host (see include/icp_mi355x.h)
static int grid_sample_padded(icp_ctx* ctx, const void* xyz, int64_t n, double voxel_size, int64_t* indices_out,
                              void* points_out, int32_t* count_out, size_t elem) {
    if (!ctx || n < 0 || !(voxel_size > 0) || (n > 0 && (!xyz || !points_out))) return ICP_ERR_INVALID_ARGUMENT;
    if (ctx->in_registration) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "registration in progress");
    int rc = ensure_state(ctx);
    if (rc) return rc;
    int* count_dev = count_out ? (int*)count_out : ctx->counter.as<int>();
    if (n == 0) {
        ICP_HIP(ctx, hipMemsetAsync(count_dev, 0, sizeof(int), ctx->stream));
        return ICP_OK;
    }
    // rows behind the V samples: NaN points (0xFFFFFFFF / 0xFFFFFFFFFFFFFFFF are NaNs), index -1 — filled by the first launch
    // of the sample
    int unused = 0;
    if (elem == 4)
        return grid_sample_device(ctx, (const float*)xyz, n, voxel_size, (long long*)indices_out, (float*)points_out, count_dev,
                                  &unused, true);
    return grid_sample_f64_device(ctx, (const double*)xyz, n, voxel_size, (long long*)indices_out, (double*)points_out,
                                  count_dev, &unused, true);
}

int icp_grid_sample_padded(icp_ctx* ctx, const float* xyz, int64_t n, double voxel_size, int64_t* indices_out,
                           float* points_out, int32_t* count_out) {
    DeviceGuard device_guard(ctx, false);  // (beside a map update on its own stream)
    return grid_sample_padded(ctx, xyz, n, voxel_size, indices_out, points_out, count_out, sizeof(float));
}

int icp_grid_sample_padded_f64(icp_ctx* ctx, const double* xyz, int64_t n, double voxel_size, int64_t* indices_out,
                               double* points_out, int32_t* count_out) {
    DeviceGuard device_guard(ctx, false);  // (beside a map update on its own stream)
    return grid_sample_padded(ctx, xyz, n, voxel_size, indices_out, points_out, count_out, sizeof(double));
}

int icp_grid_sample_f64(icp_ctx* ctx, const double* xyz, int64_t n, int mem, double voxel_size, int64_t* indices_out,
                        double* points_out, int64_t* count_out, int out_mem) {
    DeviceGuard device_guard(ctx, false);  // (beside a map update on its own stream)
    if (!ctx || n < 0 || !(voxel_size > 0) || !count_out) return ICP_ERR_INVALID_ARGUMENT;
    int rc = ensure_state(ctx);
    if (rc) return rc;
    const void* in;
    if ((rc = import_buffer(ctx, xyz, (size_t)n * 24, mem, ctx->stage_in, &in))) return rc;
    void *idev, *pdev;
    if ((rc = export_target(ctx, indices_out, (size_t)n * 8, out_mem, ctx->stage_out, &idev))) return rc;
    if ((rc = export_target(ctx, points_out, (size_t)n * 24, out_mem, ctx->stage_out2, &pdev))) return rc;
    int* count_dev = ctx->counter.as<int>();
    int count = 0;
    if ((rc = grid_sample_f64_device(ctx, (const double*)in, n, voxel_size, (long long*)idev, (double*)pdev, count_dev,
                                     &count)))
        return rc;
    *count_out = count;
    if ((rc = export_finish(ctx, indices_out, idev, (size_t)count * 8, out_mem))) return rc;
    if ((rc = export_finish(ctx, points_out, pdev, (size_t)count * 24, out_mem))) return rc;
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

// ---- de-skew --------------------------------------------------------------------------------------------------------
int icp_distort(icp_ctx* ctx, const float* xyz, const double* timestamps, int64_t n, int mem, const double rel_pose[16],
                double* xyz_out, int out_mem) {
    DeviceGuard device_guard(ctx, false);  // (beside a map update on its own stream)
    if (!ctx || n < 0 || !rel_pose || (n > 0 && (!xyz || !timestamps || !xyz_out))) return ICP_ERR_INVALID_ARGUMENT;
    const void *in, *ts;
    int rc = import_buffer(ctx, xyz, (size_t)n * 12, mem, ctx->stage_in, &in);
    if (rc) return rc;
    if ((rc = import_buffer(ctx, timestamps, (size_t)n * 8, mem, ctx->stage_out2, &ts))) return rc;
    void* odev;
    if ((rc = export_target(ctx, xyz_out, (size_t)n * 24, out_mem, ctx->stage_out, &odev))) return rc;
    if ((rc = distort_device(ctx, (const float*)in, (const double*)ts, n, rel_pose, (double*)odev))) return rc;
    if ((rc = export_finish(ctx, xyz_out, odev, (size_t)n * 24, out_mem))) return rc;
    if (out_mem == ICP_MEM_HOST || mem == ICP_MEM_HOST) ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

// ---- local map ------------------------------------------------------------------------------------------------------
int icp_map_init(icp_ctx* ctx) {
    DeviceGuard device_guard(ctx);
    if (!ctx) return ICP_ERR_INVALID_ARGUMENT;
    {   // iterations a chunked launch still holds back run against the map / configuration they were launched with
        const int rc_held = continue_launch(ctx, -1);
        if (rc_held) return rc_held;
    }
    ctx->map_m = 0;
    ctx->cloud_sizes.clear();
    ctx->grid_valid = false;
    ctx->move_job = MapMoveJob();
    return ICP_OK;
}

int icp_map_set(icp_ctx* ctx, const float* xyz, int64_t m, int mem) {
    DeviceGuard device_guard(ctx);
    if (!ctx || m < 0 || (m > 0 && !xyz)) return ICP_ERR_INVALID_ARGUMENT;
    {   // iterations a chunked launch still holds back run against the map / configuration they were launched with
        const int rc_held = continue_launch(ctx, -1);
        if (rc_held) return rc_held;
    }
    icp_map_init(ctx);  // set_map_pointcloud() calls init(): `_local_map_num_elements` stays empty (local_map.py:294)
    DeviceBuffer& dst = ctx->map_xyz[ctx->map_cur];
    ICP_HIP(ctx, dst.reserve((size_t)(m > 0 ? m : 1) * 12));
    if (m > 0)
        ICP_HIP(ctx, hipMemcpyAsync(dst.ptr, xyz, (size_t)m * 12,
                                    mem == ICP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                    ctx->stream));
    ctx->map_m = m;
    int rc = stash_frame_seeds(ctx, 0, false);  // a new map: the old neighbours mean nothing
    ctx->order_job = false;  // (a new map: the previous grid says nothing about it)
    if (!rc) rc = build_grid(ctx);
    if (rc) return rc;
    if (mem == ICP_MEM_HOST) ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

// shared tail of the two update flavours: `new_dev` [n,3] device rows with flags, or nothing
// `known_count` >= 0: `new_dev` holds exactly that many valid rows, already in order (a staged cloud) — copied, not compacted,
// and nothing is read back
static int map_update_body(icp_ctx* ctx, const float rel_pose[16], const float* new_dev, const int* flags_dev,
                           int64_t n, bool has_cloud, int64_t* inserted_out, int64_t known_count,
                           GridBuildDesc* defer = nullptr);

// The update on the context's MAP STREAM where nothing it does needs the context's scratch buffers (a pose-only update, or
// a cloud whose valid rows have been compacted and counted already: the staged insertion): ordered behind everything the
// caller's stream holds so far, ahead of the next entry point that touches the map (DeviceGuard's join) — and beside what
// touches neither: the upload, grid sample and projection of the NEXT frame.  Measured on the published configuration:
// the GPU was busy end to end with ~250 us of map update in front of ~110 us of preprocessing per frame.
static int map_update_impl(icp_ctx* ctx, const float rel_pose[16], const float* new_dev, const int* flags_dev,
                           int64_t n, bool has_cloud, int64_t* inserted_out, int64_t known_count = -1) {
    int rc = ensure_state(ctx);
    if (rc) return rc;
    const bool scratch_free = !has_cloud || known_count >= 0;
    const bool first_cloud = ctx->map_m == 0 && ctx->cloud_sizes.empty() && !ctx->grid_valid;
    if (!ctx->overlap_map_update || !scratch_free || first_cloud || ctx->exchange_on || ctx->prof.enabled || ctx->search_stats)
        return map_update_body(ctx, rel_pose, new_dev, flags_dev, n, has_cloud, inserted_out, known_count);
    if (!ctx->map_stream) {
        ICP_HIP(ctx, hipStreamCreateWithFlags(&ctx->map_stream, hipStreamNonBlocking));
        ICP_HIP(ctx, hipEventCreateWithFlags(&ctx->map_done_event, hipEventDisableTiming));
        ICP_HIP(ctx, hipEventCreateWithFlags(&ctx->map_start_event, hipEventDisableTiming));
    }
    ICP_HIP(ctx, hipEventRecord(ctx->map_start_event, ctx->stream));
    ICP_HIP(ctx, hipStreamWaitEvent(ctx->map_stream, ctx->map_start_event, 0));
    hipStream_t caller = ctx->stream;
    ctx->stream = ctx->map_stream;
    rc = map_update_body(ctx, rel_pose, new_dev, flags_dev, n, has_cloud, inserted_out, known_count);
    ctx->stream = caller;
    ICP_HIP(ctx, hipEventRecord(ctx->map_done_event, ctx->map_stream));
    ctx->map_stream_busy = true;
    return rc;
}

// defer: the launches of the grid build are left to the caller (*defer receives their arguments: icp_batch_map_update runs them
// for B maps at once), and so is what follows them — build_grid_finish and the eager normals (map_update_finish)
static int map_update_finish(icp_ctx* ctx) {
    // the next registration will want every normal at once (same rule as register_begin, with the size of the scan just
    // registered standing in for the next one): estimate them NOW, behind the rebuild, so that the GPU works through the
    // caller's preparation of the next frame (host staging, upload) instead of starting on them when that frame arrives
    if (ctx->have_device_pose && ctx->tgt_n > 0 && wants_eager_normals(ctx, ctx->tgt_n)) return launch_normals_all(ctx, true);
    return ICP_OK;
}

static int map_update_body(icp_ctx* ctx, const float rel_pose[16], const float* new_dev, const int* flags_dev,
                           int64_t n, bool has_cloud, int64_t* inserted_out, int64_t known_count, GridBuildDesc* defer) {
    int rc = ICP_OK;
    ctx->move_job = MapMoveJob();  // (a job left behind by an update that failed half-way)
    ctx->order_job = false;
    ctx->carry_job = false;
    int64_t inserted = 0;
    int64_t evicted = 0;
    if (ctx->map_m == 0 && ctx->cloud_sizes.empty() && !ctx->grid_valid) {
        // first cloud: the map becomes the cloud, the pose is ignored (local_map.py:334-337)
        DeviceBuffer& dst = ctx->map_xyz[ctx->map_cur];
        ICP_HIP(ctx, dst.reserve((size_t)(n > 0 ? n : 1) * 12));
        int* count_dev = ctx->counter.as<int>();
        if (has_cloud && known_count >= 0) {
            if (known_count > 0)
                ICP_HIP(ctx, hipMemcpyAsync(dst.ptr, new_dev, (size_t)known_count * 12, hipMemcpyDeviceToDevice, ctx->stream));
            inserted = known_count;
        } else if (has_cloud) {
            if ((rc = compact_rows(ctx, new_dev, flags_dev, n, 3, dst.as<float>(), count_dev))) return rc;
            int c = 0;
            ICP_HIP(ctx, hipMemcpyAsync(&c, count_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
            inserted = c;
        }
        ctx->map_m = inserted;
        ctx->cloud_sizes.push_back(inserted);
    } else {
        float inv[16];
        if (rel_pose && !invert4(rel_pose, inv)) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "singular relative pose");
        // eviction is decided by the number of clouds only (local_map.py:356-360)
        int64_t evict = 0;
        const size_t clouds_after = ctx->cloud_sizes.size() + (has_cloud ? 1 : 0);
        if ((int64_t)clouds_after > ctx->cfg.local_map_size && !ctx->cloud_sizes.empty()) evict = ctx->cloud_sizes[0];
        if (evict > ctx->map_m) evict = ctx->map_m;
        evicted = evict;
        const int64_t keep = ctx->map_m - evict;
        const int next = ctx->map_cur ^ 1;
        DeviceBuffer& dst = ctx->map_xyz[next];
        ICP_HIP(ctx, dst.reserve((size_t)(keep + (has_cloud ? n : 0) + 1) * 12));
        const float* src = ctx->map_xyz[ctx->map_cur].as<float>() + 3 * evict;
        // a pose-only update (nothing inserted, nothing evicted: every point keeps its neighbours) of a map whose grid holds
        // estimated normals: they are rotated with the points instead of being cleared (option "carry_normals")
        ctx->carry_job = ctx->carry_normals && !has_cloud && evict == 0 && keep > 0 && ctx->grid_valid &&
                         ctx->cost == ICP_COST_POINT_TO_PLANE;
        ctx->carry_m = keep;
        ctx->order_job = ctx->grid_valid && keep > 0;
        ctx->order_old_m = ctx->map_m;
        ctx->order_evicted = evict;
        ctx->order_kept = keep;
        if (keep > 0) {  // the re-expression rides in the first launch of the grid build below
            MapMoveJob& job = ctx->move_job;
            job.in = src;
            job.out = dst.as<float>();
            job.m = keep;
            memset(job.rel.m, 0, sizeof(job.rel.m));
            if (rel_pose) memcpy(job.rel.m, rel_pose, sizeof(job.rel.m));
            job.st = rel_pose ? (const RegState*)nullptr : (const RegState*)reg_state(ctx);
            // (a pose-only update by the device-resident pose behind a registration the host has not collected yet: should
            // that registration turn out to have been cut short by a timed-out hand-off, the device skips the move and
            // icp_register_end repeats this update once the loop has been finished — recover_handoff)
            if (!rel_pose && !has_cloud && ctx->result_pending()) ctx->update_behind_registration = true;
        }
        if (has_cloud && known_count >= 0) {
            if (known_count > 0)
                ICP_HIP(ctx, hipMemcpyAsync(dst.as<float>() + 3 * keep, new_dev, (size_t)known_count * 12,
                                            hipMemcpyDeviceToDevice, ctx->stream));
            inserted = known_count;
            ctx->cloud_sizes.push_back(inserted);
        } else if (has_cloud) {
            int* count_dev = ctx->counter.as<int>();
            if ((rc = compact_rows(ctx, new_dev, flags_dev, n, 3, dst.as<float>() + 3 * keep, count_dev))) return rc;
            int c = 0;
            ICP_HIP(ctx, hipMemcpyAsync(&c, count_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
            inserted = c;
            ctx->cloud_sizes.push_back(inserted);
        }
        if ((int64_t)ctx->cloud_sizes.size() > ctx->cfg.local_map_size) ctx->cloud_sizes.erase(ctx->cloud_sizes.begin());
        ctx->map_cur = next;
        ctx->map_m = keep + inserted;
        ICP_HIP(ctx, hipGetLastError());
    }
    if (inserted_out) *inserted_out = inserted;
    if ((rc = stash_frame_seeds(ctx, evicted, true))) return rc;  // kept points keep their order: index - evicted
    if (defer) memset(defer, 0, sizeof(*defer));  // (an empty map builds nothing: a descriptor of zero workgroups)
    if ((rc = build_grid(ctx, defer))) return rc;
    return defer ? ICP_OK : map_update_finish(ctx);
}

int icp_map_update(icp_ctx* ctx, const float rel_pose[16], const float* new_xyz, int64_t n, int mem, int row_mode,
                   int64_t* inserted_out) {
    DeviceGuard device_guard(ctx);
    if (!ctx || n < 0) return ICP_ERR_INVALID_ARGUMENT;
    if (!rel_pose && !ctx->have_device_pose)
        return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "rel_pose = NULL needs a previous registration on this context");
    const bool has_cloud = new_xyz != nullptr;
    // a registration that stops on an error must leave the window untouched (the reference raises before it touches the
    // map, icp_odometry.py:286): while its status is still unknown to the host, only the pose-only update — which the
    // device skips on an error — may be enqueued behind it
    if (!rel_pose && has_cloud && ctx->result_pending())
        return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "rel_pose = NULL with a new cloud: collect the pending registration "
                                                   "(icp_register_end) first");
    {   // iterations a chunked launch holds back go first: the pose-only update reads the END of that registration, and
        // ANY update rebuilds the grid the held-back iterations search (their NN cache and pose history belong to it)
        const int rc0 = continue_launch(ctx, -1);
        if (rc0) return rc0;
    }
    const void* in = nullptr;
    int rc;
    if (has_cloud) {
        if ((rc = import_buffer(ctx, new_xyz, (size_t)n * 12, mem, ctx->stage_in, &in))) return rc;
        ICP_HIP(ctx, ctx->flags.reserve((size_t)(n > 0 ? n : 1) * 4));
        if (n > 0)
            hipLaunchKernelGGL(k_flag_not_nan, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const float*)in, (long long)n, row_mode == ICP_TARGETS_SKIP_NULL ? 1 : 0,
                               ctx->flags.as<int>());
    }
    return map_update_impl(ctx, rel_pose, (const float*)in, ctx->flags.as<int>(), n, has_cloud, inserted_out);
}

int icp_map_stage_cloud(icp_ctx* ctx, const float* xyz, int64_t n, int mem, int row_mode) {
    DeviceGuard device_guard(ctx);
    if (!ctx || n < 0 || (n > 0 && !xyz)) return ICP_ERR_INVALID_ARGUMENT;
    if (ctx->in_registration) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "registration in progress");
    {   // (as every entry point that enqueues work: iterations a chunked launch holds back go first — stream order = call order)
        const int rc_held = continue_launch(ctx, -1);
        if (rc_held) return rc_held;
    }
    int rc = ensure_state(ctx);
    if (rc) return rc;
    if (!ctx->staged_count_host) ICP_HIP(ctx, hipHostMalloc((void**)&ctx->staged_count_host, sizeof(int), hipHostMallocDefault));
    if (!ctx->staged_event) ICP_HIP(ctx, hipEventCreateWithFlags(&ctx->staged_event, hipEventDisableTiming));
    // an earlier staged cloud that was never consumed (a pose-only frame): its count copy may still be in flight — it must
    // not land on the pinned word after the host has rewritten it
    if (ctx->staged_rows >= 0) ICP_HIP(ctx, hipEventSynchronize(ctx->staged_event));
    ctx->staged_rows = -1;
    *ctx->staged_count_host = 0;
    if (n > 0) {
        const void* in = nullptr;
        if ((rc = import_buffer(ctx, xyz, (size_t)n * 12, mem, ctx->stage_in, &in))) return rc;
        ICP_HIP(ctx, ctx->staged_xyz.reserve((size_t)n * 12));
        // two launches; the second one stores the count straight into the pinned word (mapped into the device)
        int* count_mapped = nullptr;
        ICP_HIP(ctx, hipHostGetDevicePointer((void**)&count_mapped, ctx->staged_count_host, 0));
        if ((rc = compact_valid_rows(ctx, (const float*)in, n, row_mode == ICP_TARGETS_SKIP_NULL,
                                     ctx->staged_xyz.as<float>(), nullptr, -1, count_mapped)))
            return rc;
        if (mem == ICP_MEM_HOST) ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's buffer is free again
    }
    ICP_HIP(ctx, hipEventRecord(ctx->staged_event, ctx->stream));
    ctx->staged_rows = n;
    return ICP_OK;
}

int icp_map_update_staged(icp_ctx* ctx, const float rel_pose[16], int64_t* inserted_out) {
    DeviceGuard device_guard(ctx);
    if (!ctx) return ICP_ERR_INVALID_ARGUMENT;
    if (ctx->staged_rows < 0) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "no staged cloud (icp_map_stage_cloud)");
    if (!rel_pose && !ctx->have_device_pose)
        return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "rel_pose = NULL needs a previous registration on this context");
    if (!rel_pose && ctx->result_pending())
        return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "rel_pose = NULL with a new cloud: collect the pending registration "
                                                   "(icp_register_end) first");
    {   // (as icp_map_update: iterations a chunked launch holds back go first)
        const int rc0 = continue_launch(ctx, -1);
        if (rc0) return rc0;
    }
    ICP_HIP(ctx, hipEventSynchronize(ctx->staged_event));  // long past when a registration was collected in between
    const int64_t count = *ctx->staged_count_host;
    ctx->staged_rows = -1;  // consumed
    return map_update_impl(ctx, rel_pose, ctx->staged_xyz.as<float>(), nullptr, count, true, inserted_out, count);
}

int icp_compact_targets(icp_ctx* ctx, const float* xyz, int64_t n, int target_mode, float* out, int64_t cap) {
    DeviceGuard device_guard(ctx, false);  // (beside a map update on its own stream)
    if (!ctx || n < 0 || cap < 0 || (n > 0 && !xyz) || (cap > 0 && !out)) return ICP_ERR_INVALID_ARGUMENT;
    int rc = ensure_state(ctx);
    if (rc) return rc;
    if (cap > 0) ICP_HIP(ctx, hipMemsetAsync(out, 0, (size_t)cap * 12, ctx->stream));
    if (n == 0 || cap == 0) return ICP_OK;
    return compact_valid_rows(ctx, xyz, n, target_mode == ICP_TARGETS_SKIP_NULL, out, ctx->counter.as<int>(), cap, nullptr);
}

int icp_map_update_vertex_map(icp_ctx* ctx, const float rel_pose[16], const float* vmap, int mem,
                              int64_t* inserted_out) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !rel_pose || !vmap) return ICP_ERR_INVALID_ARGUMENT;
    {   // iterations a chunked launch still holds back run against the map / configuration they were launched with
        const int rc_held = continue_launch(ctx, -1);
        if (rc_held) return rc_held;
    }
    const int npix = ctx->cfg.height * ctx->cfg.width;
    const void* in;
    int rc = import_buffer(ctx, vmap, (size_t)npix * 12, mem, ctx->stage_in, &in);
    if (rc) return rc;
    ICP_HIP(ctx, ctx->stage_out.reserve((size_t)npix * 12));
    ICP_HIP(ctx, ctx->flags.reserve((size_t)npix * 4));
    hipLaunchKernelGGL(k_vmap_points, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, (const float*)in, npix, 0.01f,
                       ctx->stage_out.as<float>(), ctx->flags.as<int>());
    return map_update_impl(ctx, rel_pose, ctx->stage_out.as<float>(), ctx->flags.as<int>(), npix, true, inserted_out);
}

int64_t icp_map_size(const icp_ctx* ctx) { return ctx ? ctx->map_m : 0; }
int icp_handoff_fallbacks(const icp_ctx* ctx) { return ctx ? ctx->handoff_fallbacks : 0; }
int icp_map_num_clouds(const icp_ctx* ctx) { return ctx ? (int)ctx->cloud_sizes.size() : 0; }

int icp_map_get(icp_ctx* ctx, float* xyz_out, int out_mem) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !xyz_out) return ICP_ERR_INVALID_ARGUMENT;
    if (ctx->map_m == 0) return ICP_OK;
    ICP_HIP(ctx, hipMemcpyAsync(xyz_out, ctx->map_xyz[ctx->map_cur].ptr, (size_t)ctx->map_m * 12,
                                out_mem == ICP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                                ctx->stream));
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

int icp_nearest_neighbor_search(icp_ctx* ctx, const float* xyz, int64_t n, int mem, float* neighbor_points_out,
                                float* neighbor_normals_out, int32_t* neighbor_index_out, int out_mem) {
    DeviceGuard device_guard(ctx);
    if (!ctx || n < 0) return ICP_ERR_INVALID_ARGUMENT;
    if (ctx->map_m <= 0 || !ctx->grid_valid) return fail(ctx, ICP_ERR_EMPTY_MAP, "the local map is empty");
    if (ctx->in_registration || ctx->result_pending())
        return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "registration in progress");
    int rc = ensure_state(ctx);
    if (rc) return rc;
    const void* in;
    if ((rc = import_buffer(ctx, xyz, (size_t)n * 12, mem, ctx->targets, &in))) return rc;
    ctx->tgt_ptr = (const float*)in;
    ctx->tgt_n = n;
    ctx->tgt_mode = ICP_TARGETS_ALL;
    ctx->have_device_pose = false;  // the search re-initialises the device state: it no longer holds a registration
    ICP_HIP(ctx, ctx->nn_pos.reserve((size_t)(n > 0 ? n : 1) * 4));
    if ((rc = prepare_targets_and_state(ctx, n, nullptr))) return rc;
    if ((rc = launch_search_raw(ctx))) return rc;
    if (neighbor_normals_out && (rc = launch_normals(ctx))) return rc;
    void *pdev, *ndev, *idev;
    DeviceBuffer& s3 = ctx->flags;  // third staging area
    if ((rc = export_target(ctx, neighbor_points_out, (size_t)n * 12, out_mem, ctx->stage_out, &pdev))) return rc;
    if ((rc = export_target(ctx, neighbor_normals_out, (size_t)n * 12, out_mem, ctx->stage_out2, &ndev))) return rc;
    if ((rc = export_target(ctx, neighbor_index_out, (size_t)n * 4, out_mem, s3, &idev))) return rc;
    if ((rc = launch_gather_neighbors(ctx, n, (float*)pdev, (float*)ndev, (int32_t*)idev))) return rc;
    if ((rc = export_finish(ctx, neighbor_points_out, pdev, (size_t)n * 12, out_mem))) return rc;
    if ((rc = export_finish(ctx, neighbor_normals_out, ndev, (size_t)n * 12, out_mem))) return rc;
    if ((rc = export_finish(ctx, neighbor_index_out, idev, (size_t)n * 4, out_mem))) return rc;
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

int icp_last_neighbors(icp_ctx* ctx, int32_t* neighbor_index_out, float pose_out[12], int out_mem) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !neighbor_index_out) return ICP_ERR_INVALID_ARGUMENT;
    if (ctx->in_registration || ctx->result_pending())
        return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "registration in progress");
    // the cache must describe the targets of the last registration against the grid that is still current
    // (`cache_fresh`: a fused launch of the LAST registration wrote the entries — an unfused registration leaves the cache of
    // an earlier one behind)
    if (!ctx->cache_fresh || ctx->cache_n <= 0 || ctx->cache_n != ctx->tgt_n || ctx->cache_gen != ctx->grid_gen || !ctx->grid_valid ||
        ctx->iter_in_registration <= 0 || ctx->last_iterations <= 0 || !ctx->pose_hist)
        return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "no fused registration against the current map to read neighbours from");
    // the last iteration that ran: RegState.iter - 1 (a loop stopped early by its threshold enqueues no further search)
    const int it = (ctx->last_iterations < ctx->iter_in_registration ? ctx->last_iterations : ctx->iter_in_registration) - 1;
    if (it >= ctx->hist_cap) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "pose history too short");
    const int64_t n = ctx->tgt_n;
    void* idev;
    int rc;
    if ((rc = export_target(ctx, neighbor_index_out, (size_t)n * 4, out_mem, ctx->flags, &idev))) return rc;
    if ((rc = launch_last_neighbors(ctx, it, (int*)idev))) return rc;
    if ((rc = export_finish(ctx, neighbor_index_out, idev, (size_t)n * 4, out_mem))) return rc;
    if (pose_out)
        ICP_HIP(ctx, hipMemcpyAsync(pose_out, ctx->pose_hist + (size_t)it * 12, 12 * sizeof(float), hipMemcpyDeviceToHost,
                                    ctx->stream));
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

// ---- projective local map -------------------------------------------------------------------------------------------
int icp_compute_normal_map(icp_ctx* ctx, const float* vmap, int mem, int kernel_size, float* nmap_out, int out_mem) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !vmap || !nmap_out || kernel_size < 1 || kernel_size > 15 || !(kernel_size & 1))
        return ICP_ERR_INVALID_ARGUMENT;
    const size_t bytes = (size_t)ctx->cfg.height * ctx->cfg.width * 12;
    const void* in;
    int rc = import_buffer(ctx, vmap, bytes, mem, ctx->stage_in, &in);
    if (rc) return rc;
    void* odev;
    if ((rc = export_target(ctx, nmap_out, bytes, out_mem, ctx->stage_out, &odev))) return rc;
    if ((rc = normal_map_device(ctx, (const float*)in, kernel_size, (float*)odev))) return rc;
    if ((rc = export_finish(ctx, nmap_out, odev, bytes, out_mem))) return rc;
    if (out_mem == ICP_MEM_HOST || mem == ICP_MEM_HOST) ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

int icp_compute_neighbors(icp_ctx* ctx, const float* tgt_vmap, const float* ref_vmaps, const float* ref_fields,
                          int k_maps, int c_fields, int mem, float* neighbors_out, float* fields_out, int out_mem) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !tgt_vmap || !ref_vmaps || !neighbors_out || k_maps < 1 || c_fields < 0) return ICP_ERR_INVALID_ARGUMENT;
    const size_t px = (size_t)ctx->cfg.height * ctx->cfg.width;
    const void *t, *r, *f = nullptr;
    int rc = import_buffer(ctx, tgt_vmap, px * 12, mem, ctx->stage_in, &t);
    if (rc) return rc;
    if ((rc = import_buffer(ctx, ref_vmaps, px * 12 * k_maps, mem, ctx->targets, &r))) return rc;
    if (ref_fields && c_fields > 0 &&
        (rc = import_buffer(ctx, ref_fields, px * 4 * k_maps * c_fields, mem, ctx->pm_tmp, &f)))
        return rc;
    void *nb, *fo = nullptr;
    if ((rc = export_target(ctx, neighbors_out, px * 12, out_mem, ctx->stage_out, &nb))) return rc;
    if (f && fields_out && (rc = export_target(ctx, fields_out, px * 4 * c_fields, out_mem, ctx->stage_out2, &fo)))
        return rc;
    if ((rc = neighbors_device(ctx, (const float*)t, (const float*)r, (const float*)f, k_maps, c_fields, (float*)nb,
                               (float*)fo)))
        return rc;
    if ((rc = export_finish(ctx, neighbors_out, nb, px * 12, out_mem))) return rc;
    if (fo && (rc = export_finish(ctx, fields_out, fo, px * 4 * c_fields, out_mem))) return rc;
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

int icp_pmap_init(icp_ctx* ctx) {
    DeviceGuard device_guard(ctx);
    if (!ctx) return ICP_ERR_INVALID_ARGUMENT;
    ctx->pm_slots.clear();
    ctx->pm_poses.clear();
    return ICP_OK;
}

int icp_pmap_num_maps(const icp_ctx* ctx) { return ctx ? (int)ctx->pm_slots.size() : 0; }

int icp_pmap_update(icp_ctx* ctx, const float rel_pose[16], const float* vmap, int mem, int normals_kernel_size) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !rel_pose) return ICP_ERR_INVALID_ARGUMENT;
    if (vmap && (normals_kernel_size < 1 || normals_kernel_size > 15 || !(normals_kernel_size & 1)))
        return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "normals_kernel_size must be odd, 1..15");
    int rc = ensure_state(ctx);
    if (rc) return rc;
    const size_t bytes = (size_t)ctx->cfg.height * ctx->cfg.width * 12;
    const void* in = nullptr;
    if (vmap) {
        if ((rc = import_buffer(ctx, vmap, bytes, mem, ctx->stage_in, &in))) return rc;
        ICP_HIP(ctx, ctx->pm_tmp.reserve(bytes));
        if ((rc = normal_map_device(ctx, (const float*)in, normals_kernel_size, ctx->pm_tmp.as<float>()))) return rc;
    }
    const int cap = ctx->cfg.local_map_size + 1;
    auto free_slot = [&]() {
        for (int s = 0; s < cap; ++s) {
            bool used = false;
            for (int u : ctx->pm_slots) used |= (u == s);
            if (!used) return s;
        }
        return -1;
    };
    if (ctx->pm_slots.empty()) {
        if (!vmap) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "the first update needs a vertex map");
        icp_ctx::PmPose p;
        memcpy(p.m, rel_pose, sizeof(p.m));  // `_local_map_poses = relative_pose` (local_map.py:147)
        if ((rc = pmap_store_slot(ctx, 0, (const float*)in, ctx->pm_tmp.as<float>()))) return rc;
        ctx->pm_slots.push_back(0);
        ctx->pm_poses.push_back(p);
    } else {
        float inv[16];
        if (!invert4(rel_pose, inv)) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "singular relative pose");
        for (auto& p : ctx->pm_poses) {  // old_poses = relative_pose.inverse() @ poses (:149)
            float out[16];
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c) {
                    double a = 0.0;
                    for (int k = 0; k < 4; ++k) a += (double)inv[4 * r + k] * (double)p.m[4 * k + c];
                    out[4 * r + c] = (float)a;
                }
            memcpy(p.m, out, sizeof(out));
        }
        if (vmap) {
            const int slot = free_slot();
            if (slot < 0) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "no free projective-map slot");
            if ((rc = pmap_store_slot(ctx, slot, (const float*)in, ctx->pm_tmp.as<float>()))) return rc;
            icp_ctx::PmPose eye;
            memset(eye.m, 0, sizeof(eye.m));
            eye.m[0] = eye.m[5] = eye.m[10] = eye.m[15] = 1.f;
            ctx->pm_slots.push_back(slot);
            ctx->pm_poses.push_back(eye);
        }
        if ((int)ctx->pm_poses.size() > ctx->cfg.local_map_size) {  // :166-171
            ctx->pm_slots.erase(ctx->pm_slots.begin());
            ctx->pm_poses.erase(ctx->pm_poses.begin());
        }
    }
    if ((rc = pmap_build(ctx))) return rc;
    if (vmap && mem == ICP_MEM_HOST) ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

int icp_pmap_get_model(icp_ctx* ctx, float* model_v4_out, float* model_n4_out, int out_mem) {
    DeviceGuard device_guard(ctx);
    if (!ctx) return ICP_ERR_INVALID_ARGUMENT;
    const size_t bytes = (size_t)ctx->pm_slots.size() * ctx->cfg.height * ctx->cfg.width * 16;
    if (bytes == 0) return ICP_OK;
    const hipMemcpyKind kind = out_mem == ICP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (model_v4_out) ICP_HIP(ctx, hipMemcpyAsync(model_v4_out, ctx->pm_mv.ptr, bytes, kind, ctx->stream));
    if (model_n4_out) ICP_HIP(ctx, hipMemcpyAsync(model_n4_out, ctx->pm_mn.ptr, bytes, kind, ctx->stream));
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

int icp_pmap_nearest_neighbor_search(icp_ctx* ctx, const float* xyz, int64_t n, int mem, float* rows9_out,
                                     int64_t* count_out, int out_mem) {
    DeviceGuard device_guard(ctx);
    if (!ctx || n < 0 || !rows9_out || !count_out) return ICP_ERR_INVALID_ARGUMENT;
    if (ctx->pm_slots.empty()) return fail(ctx, ICP_ERR_EMPTY_MAP, "the local map is empty");
    int rc = ensure_state(ctx);
    if (rc) return rc;
    const size_t npix = (size_t)ctx->cfg.height * ctx->cfg.width;
    const void* in;
    if ((rc = import_buffer(ctx, xyz, (size_t)n * 12, mem, ctx->targets, &in))) return rc;
    ICP_HIP(ctx, ctx->pm_tmp.reserve(npix * 36));
    ICP_HIP(ctx, ctx->flags.reserve(npix * 4));
    if ((rc = pmap_associate(ctx, (const float*)in, n, ctx->pm_tmp.as<float>(), ctx->flags.as<int>()))) return rc;
    void* odev;
    if ((rc = export_target(ctx, rows9_out, npix * 36, out_mem, ctx->stage_out, &odev))) return rc;
    int* count_dev = ctx->counter.as<int>();
    if ((rc = compact_rows(ctx, ctx->pm_tmp.as<float>(), ctx->flags.as<int>(), (int64_t)npix, 9, (float*)odev, count_dev)))
        return rc;
    int count = 0;
    ICP_HIP(ctx, hipMemcpyAsync(&count, count_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *count_out = count;
    if ((rc = export_finish(ctx, rows9_out, odev, (size_t)count * 36, out_mem))) return rc;
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

int icp_pmap_register(icp_ctx* ctx, const float* xyz, int64_t n, int mem, int target_mode, const float init_pose[16],
                      icp_register_result* result, double* loss_per_iter_out, float* dx_per_iter_out) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !result || n < 0 || (n > 0 && !xyz)) return ICP_ERR_INVALID_ARGUMENT;
    {   // iterations a chunked launch still holds back run against the map / configuration they were launched with
        const int rc_held = continue_launch(ctx, -1);
        if (rc_held) return rc_held;
    }
    if (ctx->pm_slots.empty()) return fail(ctx, ICP_ERR_EMPTY_MAP, "the local map is empty");
    int rc = ensure_state(ctx);
    if (rc) return rc;
    const void* in;
    if ((rc = import_buffer(ctx, xyz, (size_t)n * 12, mem, ctx->targets, &in))) return rc;
    ctx->tgt_ptr = (const float*)in;
    ctx->tgt_n = n;
    ctx->tgt_mode = target_mode;
    if ((rc = prepare_targets_and_state(ctx, n, init_pose))) return rc;
    ctx->have_device_pose = false;  // icp_map_update(rel_pose = NULL) follows a registration against the kd-tree style map only
    ctx->in_registration = true;
    const int iters = ctx->cfg.max_num_alignments;
    const int poll = ctx->cfg.threshold_delta_pose > 0.f ? ctx->cfg.poll_every : 0;
    for (int it = 0; it < iters; ++it) {
        int blocks = 0;
        rc = pmap_iterate(ctx, &blocks);
        if (!rc) rc = launch_sum_solve(ctx, blocks);
        if (rc) {
            ctx->in_registration = false;
            return rc;
        }
        if (poll > 0 && (it + 1) % poll == 0 && it + 1 < iters) {
            int done = 0;
            ICP_HIP(ctx, hipMemcpyAsync(&done, &reg_state(ctx)->done, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (done) break;
        }
    }
    return icp_register_end(ctx, result, loss_per_iter_out, dx_per_iter_out);
}

// ---- alignment on given correspondences -----------------------------------------------------------------------------
static int finish_align(icp_ctx* ctx, float params_out[6], float pose_out[16], double* loss_out, double* normal_eq_out);

int icp_align_point_to_plane(icp_ctx* ctx, const float* ref_points, const float* tgt_points, const float* ref_normals,
                             int64_t n, int mem, float dx_out[6], float pose_out[16], double* loss_out,
                             double* normal_eq_out, float* residuals_out) {
    DeviceGuard device_guard(ctx);
    if (!ctx || n <= 0 || !ref_points || !tgt_points || !ref_normals) return ICP_ERR_INVALID_ARGUMENT;
    int rc = ensure_state(ctx);
    if (rc) return rc;
    const void *r, *t, *nr;
    void* res_dev;
    if ((rc = import_buffer(ctx, ref_points, (size_t)n * 12, mem, ctx->stage_in, &r))) return rc;
    if ((rc = import_buffer(ctx, tgt_points, (size_t)n * 12, mem, ctx->targets, &t))) return rc;
    if ((rc = import_buffer(ctx, ref_normals, (size_t)n * 12, mem, ctx->stage_out2, &nr))) return rc;
    if ((rc = export_target(ctx, residuals_out, (size_t)n * 4, mem, ctx->flags, &res_dev))) return rc;
    if ((rc = launch_align_given(ctx, (const float*)r, (const float*)t, (const float*)nr, n, (float*)res_dev))) return rc;
    if ((rc = export_finish(ctx, residuals_out, res_dev, (size_t)n * 4, mem))) return rc;
    return finish_align(ctx, dx_out, pose_out, loss_out, normal_eq_out);
}

static int finish_align(icp_ctx* ctx, float params_out[6], float pose_out[16], double* loss_out, double* normal_eq_out) {
    char host[144];
    double neq[NEQ];
    ICP_HIP(ctx, hipMemcpyAsync(host, ctx->stage_out.ptr, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
    ICP_HIP(ctx, hipMemcpyAsync(neq, ctx->neq, sizeof(neq), hipMemcpyDeviceToHost, ctx->stream));
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const float* f = (const float*)host;
    if (params_out) memcpy(params_out, f, 6 * sizeof(float));
    if (pose_out) memcpy(pose_out, f + 6, 16 * sizeof(float));
    if (loss_out) memcpy(loss_out, host + 128, sizeof(double));
    if (normal_eq_out) memcpy(normal_eq_out, neq, sizeof(neq));
    int status;
    memcpy(&status, host + 136, sizeof(int));
    if (status == ICP_ERR_INVALID_JACOBIAN)
        return fail(ctx, ICP_ERR_INVALID_JACOBIAN, "Invalid Jacobian in Gauss Newton minimization");
    return status;
}

int icp_align_point_to_point(icp_ctx* ctx, const float* ref_points, const float* tgt_points, int64_t n, int mem,
                             const float x0[6], float params_out[6], float pose_out[16], double* loss_out,
                             double* normal_eq_out, float* residuals_out) {
    DeviceGuard device_guard(ctx);
    if (!ctx || n <= 0 || !ref_points || !tgt_points) return ICP_ERR_INVALID_ARGUMENT;
    int rc = ensure_state(ctx);
    if (rc) return rc;
    const void *r, *t;
    void* res_dev;
    if ((rc = import_buffer(ctx, ref_points, (size_t)n * 12, mem, ctx->stage_in, &r))) return rc;
    if ((rc = import_buffer(ctx, tgt_points, (size_t)n * 12, mem, ctx->targets, &t))) return rc;
    if ((rc = export_target(ctx, residuals_out, (size_t)n * 4, mem, ctx->flags, &res_dev))) return rc;
    if ((rc = launch_align_p2p(ctx, (const float*)r, (const float*)t, n, x0, (float*)res_dev))) return rc;
    if ((rc = export_finish(ctx, residuals_out, res_dev, (size_t)n * 4, mem))) return rc;
    return finish_align(ctx, params_out, pose_out, loss_out, normal_eq_out);
}

// 3x3 SVD A = U diag(s) V^T in f64 by one-sided Jacobi, singular values sorted descending (LAPACK's order, which the
// reflection fix of weighted_procrustes relies on); a vanishing direction is completed by a cross product
static void svd3(const double A[9], double U[9], double sv[3], double V[9]) {
    double B[9];
    memcpy(B, A, sizeof(B));
    for (int k = 0; k < 9; ++k) V[k] = (k % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, b = 0, c = 0;
                for (int i = 0; i < 3; ++i) {
                    a += B[3 * i + p] * B[3 * i + p];
                    b += B[3 * i + q] * B[3 * i + q];
                    c += B[3 * i + p] * B[3 * i + q];
                }
                if (fabs(c) <= 1e-300 || fabs(c) <= 1e-17 * sqrt(a * b)) continue;
                off += fabs(c);
                const double zeta = (b - a) / (2.0 * c);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < 3; ++i) {
                    const double bp = B[3 * i + p], bq = B[3 * i + q];
                    B[3 * i + p] = cs * bp - sn * bq;
                    B[3 * i + q] = sn * bp + cs * bq;
                    const double vp = V[3 * i + p], vq = V[3 * i + q];
                    V[3 * i + p] = cs * vp - sn * vq;
                    V[3 * i + q] = sn * vp + cs * vq;
                }
            }
        if (off == 0.0) break;
    }
    int order[3] = {0, 1, 2};
    double nrm[3];
    for (int j = 0; j < 3; ++j) nrm[j] = sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (nrm[order[b]] > nrm[order[a]]) std::swap(order[a], order[b]);
    double Vs[9];
    for (int j = 0; j < 3; ++j) {
        const int c = order[j];
        sv[j] = nrm[c];
        for (int i = 0; i < 3; ++i) {
            Vs[3 * i + j] = V[3 * i + c];
            U[3 * i + j] = nrm[c] > 0.0 ? B[3 * i + c] / nrm[c] : 0.0;
        }
    }
    memcpy(V, Vs, sizeof(Vs));
    const double tiny = 1e-14 * (sv[0] > 0.0 ? sv[0] : 1.0);
    if (sv[1] <= tiny) {  // rank <= 1: any unit vector orthogonal to u0
        const double ax = fabs(U[0]), ay = fabs(U[3]), az = fabs(U[6]);
        double e[3] = {0, 0, 0};
        e[(ax <= ay && ax <= az) ? 0 : (ay <= az ? 1 : 2)] = 1.0;
        if (sv[0] <= 0.0) { U[0] = 1.0; U[3] = U[6] = 0.0; e[0] = 0.0; e[1] = 1.0; e[2] = 0.0; }
        const double d = e[0] * U[0] + e[1] * U[3] + e[2] * U[6];
        double w[3] = {e[0] - d * U[0], e[1] - d * U[3], e[2] - d * U[6]};
        const double wn = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        U[1] = w[0] / wn; U[4] = w[1] / wn; U[7] = w[2] / wn;
    }
    if (sv[2] <= tiny) {  // u2 = u0 x u1
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
}

static double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

int icp_weighted_procrustes(icp_ctx* ctx, const float* tgt_points, const float* ref_points, const float* weights,
                            int64_t n, int mem, double pose_out[16]) {
    DeviceGuard device_guard(ctx);
    if (!ctx || n <= 0 || !ref_points || !tgt_points || !pose_out) return ICP_ERR_INVALID_ARGUMENT;
    int rc = ensure_state(ctx);
    if (rc) return rc;
    const void *r, *t, *w = nullptr;
    if ((rc = import_buffer(ctx, ref_points, (size_t)n * 12, mem, ctx->stage_in, &r))) return rc;
    if ((rc = import_buffer(ctx, tgt_points, (size_t)n * 12, mem, ctx->targets, &t))) return rc;
    if (weights && (rc = import_buffer(ctx, weights, (size_t)n * 4, mem, ctx->stage_out2, &w))) return rc;
    double sums[NEQ];
    if ((rc = launch_procrustes_pass(ctx, (const float*)t, (const float*)r, (const float*)w, n, nullptr, nullptr, sums)))
        return rc;
    if (!(sums[0] != 0.0)) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "the weights sum to zero");
    float mu_t[3], mu_r[3];  // float32 means, as the reference forms them in the dtype of the points
    for (int a = 0; a < 3; ++a) {
        mu_t[a] = (float)(sums[1 + a] / sums[0]);
        mu_r[a] = (float)(sums[4 + a] / sums[0]);
    }
    double cov[NEQ];
    if ((rc = launch_procrustes_pass(ctx, (const float*)t, (const float*)r, nullptr, n, mu_t, mu_r, cov))) return rc;
    double U[9], sv[3], V[9];
    svd3(cov, U, sv, V);
    const double sgn = det3(U) * det3(V) < 0.0 ? -1.0 : 1.0;  // S[-1,-1] = -1 (:49-51)
    double R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + sgn * U[3 * i + 2] * V[3 * j + 2];
    for (int k = 0; k < 16; ++k) pose_out[k] = (k % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) pose_out[4 * i + j] = R[3 * i + j];
        pose_out[4 * i + 3] = (double)mu_r[i] - (R[3 * i] * mu_t[0] + R[3 * i + 1] * mu_t[1] + R[3 * i + 2] * mu_t[2]);
    }
    return ICP_OK;
}

// ---- registration ---------------------------------------------------------------------------------------------------
// defer_pack: the launch that packs the targets and initialises the state is left to the caller (a batched registration runs
// it for all its members at once)
static int register_begin(icp_ctx* ctx, const float* xyz, int64_t n, int mem, int target_mode,
                          const float init_pose[16], bool from_last, PackDesc* defer_pack = nullptr) {
    if (!ctx || n < 0 || (n > 0 && !xyz)) return ICP_ERR_INVALID_ARGUMENT;
    if (ctx->map_m <= 0 || !ctx->grid_valid) return fail(ctx, ICP_ERR_EMPTY_MAP, "the local map is empty");
    if (ctx->in_registration) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "registration in progress");
    if (from_last && !ctx->have_device_pose)
        return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "no previous registration on this context to start from");
    int rc = ensure_state(ctx);
    if (rc) return rc;
    const void* in;
    if ((rc = import_buffer(ctx, xyz, (size_t)n * 12, mem, ctx->targets, &in))) return rc;
    ctx->tgt_ptr = (const float*)in;
    ctx->tgt_n = n;
    ctx->tgt_mode = target_mode;
    ICP_HIP(ctx, ctx->nn_pos.reserve((size_t)(n > 0 ? n : 1) * 4));
    if ((rc = prepare_targets_and_state(ctx, n, init_pose, from_last, defer_pack))) return rc;
    ctx->have_device_pose = true;
    // event pairs around the kernels of every `every`-th registration only: a pair costs ~2 us of stream time
    ctx->prof.sample_now = ctx->prof.every <= 1 || (ctx->prof.registrations % ctx->prof.every) == 0;
    ctx->prof.rotate_index = (int)((ctx->prof.registrations / (ctx->prof.every > 1 ? ctx->prof.every : 1)) %
                                   (ctx->cfg.max_num_alignments > 0 ? ctx->cfg.max_num_alignments : 1));
    ctx->prof.registrations += 1;
    // normals: lazily for the map points the scan touches (local_map.py:397-422) when the map is much larger than the
    // scan, all at once otherwise (same values; one dense launch instead of a sparse one per iteration)
    if (!ctx->normals_ready && wants_eager_normals(ctx, n) && (rc = launch_normals_all(ctx))) return rc;
    // ... or on demand inside the fused iteration kernel, where the scan touches a small part of the map ("lazy_fused")
    ctx->lazy_now = !ctx->normals_ready && wants_lazy_fused(ctx, n);
    registering_enter(ctx);
    // lead launches or not: decided ONCE per registration (another context of the process starting or ending a
    // registration between two chunks of this one flipped the answer — a chunk without a lead publishes no pose, the next
    // lead chunk then took a stale one from the mailbox)
    ctx->lead_latched = ctx->lead_solve && !ctx->handoff_disabled && !device_shared_with_another_registration(ctx);
    ctx->in_registration = true;
    ctx->iter_in_registration = 0;
    ctx->cache_fresh = false;
    ctx->searches_in_registration = 0;
    return ICP_OK;
}

int icp_register_begin(icp_ctx* ctx, const float* xyz, int64_t n, int mem, int target_mode,
                       const float init_pose[16]) {
    DeviceGuard device_guard(ctx);
    if (ctx && ctx->result_pending()) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "collect the pending result first");
    return register_begin(ctx, xyz, n, mem, target_mode, init_pose, false);
}

// the fused search + rows kernel needs every normal it may touch: the eager schedule
static bool fused_path(const icp_ctx* ctx) {
    return ctx->cost == ICP_COST_POINT_TO_PLANE && (ctx->normals_ready || ctx->lazy_now) && ctx->fuse_iteration;
}

int icp_iteration_accumulate(icp_ctx* ctx) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !ctx->in_registration) return ICP_ERR_INVALID_ARGUMENT;
    int rc;
    if (fused_path(ctx)) {
        int rows = 0, quad = 1;
        if ((rc = launch_iterate_fused(ctx, &rows, &quad))) return rc;
        return launch_sum_partials(ctx, rows, quad);
    }
    if ((rc = launch_search(ctx))) return rc;
    ctx->iter_in_registration += 1;  // (the fused launch counts itself)
    if (ctx->cost == ICP_COST_POINT_TO_POINT) return launch_reduce_p2p(ctx, false);
    if ((rc = launch_normals(ctx))) return rc;
    return launch_reduce(ctx);
}

int icp_iteration_solve(icp_ctx* ctx) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !ctx->in_registration) return ICP_ERR_INVALID_ARGUMENT;
    return launch_solve(ctx);
}

// layout of the pinned result block: the device state allocation verbatim (RegState | pad | loss[hist_cap] |
// dx[6 * hist_cap])
static size_t state_bytes(const icp_ctx* ctx) {
    return STATE_BLOCK + (size_t)ctx->hist_cap * (sizeof(double) + 6 * sizeof(float));
}

// the pinned slot the result of the registration being enqueued will land in (refresh = true: the newest pending one again)
static int result_slot(icp_ctx* ctx, bool refresh, icp_ctx::ResultSlot** out) {
    if (!refresh && ctx->r_count >= 2) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "two results are already pending");
    icp_ctx::ResultSlot& r = ctx->rslot[(ctx->r_head + ctx->r_count - (refresh ? 1 : 0)) & 1];
    const size_t need = state_bytes(ctx);
    if (need > r.bytes) {
        if (r.host) (void)hipHostFree(r.host);
        r.host = nullptr;
        r.bytes = 0;
        ICP_HIP(ctx, hipHostMalloc(&r.host, need, hipHostMallocDefault));
        r.bytes = need;
    }
    if (!r.event) ICP_HIP(ctx, hipEventCreateWithFlags(&r.event, hipEventDisableTiming));
    *out = &r;
    return ICP_OK;
}

// call before enqueue_iterations: the last solving launch may then write the result block into the slot itself
static int result_fold_begin(icp_ctx* ctx, bool refresh) {
    ctx->result_fold_to = nullptr;
    ctx->result_folded = false;
    icp_ctx::ResultSlot* r = nullptr;
    const int rc = result_slot(ctx, refresh, &r);
    if (rc) return rc;
    void* mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, r->host, 0) == hipSuccess && mapped) {
        ctx->result_fold_to = (char*)mapped;
        ctx->result_fold_bytes = state_bytes(ctx);
    }
    return ICP_OK;
}

// refresh = true: the newest pending slot is copied again (a further chunk of its registration has been enqueued)
static int enqueue_result_copy(icp_ctx* ctx, bool refresh = false, hipEvent_t shared_event = nullptr) {
    icp_ctx::ResultSlot* rp = nullptr;
    const int rc_slot = result_slot(ctx, refresh, &rp);
    if (rc_slot) return rc_slot;
    icp_ctx::ResultSlot& r = *rp;
    const size_t sb = state_bytes(ctx);
    char* h = (char*)r.host;
    // state + histories (unless the last solving launch has written them there already)
    if (!ctx->result_folded) ICP_HIP(ctx, hipMemcpyAsync(h, ctx->state.ptr, sb, hipMemcpyDeviceToHost, ctx->stream));
    ctx->result_folded = false;
    ctx->result_fold_to = nullptr;
    // (a batched registration: ONE event behind the results of all its members, recorded by the batch — icp_batch_*)
    r.wait = shared_event ? shared_event : r.event;
    if (!shared_event) ICP_HIP(ctx, hipEventRecord(r.event, ctx->stream));
    if (refresh) return ICP_OK;
    // the grid statistics about to be read back belong to the current build; a later build starts a new pending set
    r.stats = ctx->stats_pending;
    r.stats_m = ctx->stats_m_pending;
    r.stats_h = ctx->stats_h_pending;
    ctx->stats_pending = false;
    r.eager_normals = ctx->normals_eager_count;
    ctx->normals_eager_count = 0;
    ctx->r_count += 1;
    return ICP_OK;
}

// enqueues the iterations a chunked launch has held back (all of them, or the next `count`) and copies the result again
static int continue_launch(icp_ctx* ctx, int count) {
    if (ctx->batch_hold)  // (a batched registration of this context is enqueued in chunks: its batch owns what is held back)
        return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "a batched registration of this context still holds iterations back: "
                                                   "icp_batch_register_end (or icp_batch_map_update) first");
    if (ctx->launch_remaining <= 0) return ICP_OK;
    if (count < 0 || count > ctx->launch_remaining) count = ctx->launch_remaining;
    ctx->in_registration = true;
    int rc = result_fold_begin(ctx, true);
    if (!rc) rc = enqueue_iterations(ctx, false, ctx->launch_enqueued, count);
    ctx->in_registration = false;
    ctx->launch_enqueued += count;
    ctx->launch_remaining -= count;
    if (!rc) rc = enqueue_result_copy(ctx, true);
    ctx->result_fold_to = nullptr;
    ctx->result_folded = false;
    return rc;
}

// A hand-off inside a lead launch / the resident tail ran out of its wall-clock budget (RegState.handoff_timeouts: a
// workgroup the others wait for is not running — the GPU is shared with foreign work).  The launches behind it solved
// nothing more (lead_solve / k_sum_solve stop at the counter), so the state holds the last complete iteration; a pose-only
// map update enqueued behind the registration moved nothing (map_move_prepare).  Here: the context stops using hand-offs
// (per-iteration launches with a solving launch each: "lead_solve" 0), the remaining iterations run on those — the same
// iterations on the same map, hence the same bits — and the skipped map update is repeated with the final pose.
// `block` receives the state + histories (the layout of the pinned result block).
static int recover_handoff(icp_ctx* ctx, RegState& st, std::vector<char>& block) {
    { DeviceGuard join_map_stream(ctx); }  // (a map update on its own stream: everything below follows it)
    ctx->tail_disabled = true;
    ctx->handoff_disabled = true;  // (the user's "lead_solve" stays as set; setting it again re-arms the hand-offs)
    ctx->lead_latched = false;
    ctx->handoff_fallbacks += 1;
    ctx->launch_remaining = 0;
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ICP_HIP(ctx, hipMemcpy(&st, ctx->state.ptr, sizeof(st), hipMemcpyDeviceToHost));
    ICP_HIP(ctx, hipMemsetAsync(&reg_state(ctx)->handoff_timeouts, 0, sizeof(int), ctx->stream));
    int rc = ICP_OK;
    if (!st.done && st.status == ICP_OK && st.iter < ctx->cfg.max_num_alignments) {
        ctx->in_registration = true;
        ctx->iter_in_registration = st.iter;
        ctx->cache_fresh = false;  // (a map update in between has rebuilt the grid: the first launch searches everything)
        ctx->searches_in_registration = 0;
        if (!ctx->normals_ready && wants_eager_normals(ctx, ctx->tgt_n) && (rc = launch_normals_all(ctx))) return rc;
        rc = enqueue_iterations(ctx, false, st.iter, -1);
        ctx->in_registration = false;
        if (rc) return rc;
    }
    if (ctx->update_behind_registration) {  // the pose-only update the device skipped, now with the final pose
        ctx->update_behind_registration = false;
        if ((rc = map_update_impl(ctx, nullptr, nullptr, nullptr, 0, false, nullptr))) return rc;
    }
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    block.resize(state_bytes(ctx));
    ICP_HIP(ctx, hipMemcpy(block.data(), ctx->state.ptr, block.size(), hipMemcpyDeviceToHost));
    memcpy(&st, block.data(), sizeof(st));
    return ICP_OK;
}

int icp_register_end(icp_ctx* ctx, icp_register_result* result, double* loss_per_iter_out, float* dx_per_iter_out) {
    DeviceGuard device_guard(ctx, false);  // (the pose arrives while the map update runs)
    if (!ctx || !result || (!ctx->in_registration && !ctx->result_pending())) return ICP_ERR_INVALID_ARGUMENT;
    if (ctx->batch_hold)
        return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "a batched registration of this context still holds iterations back: "
                                                   "collect it with icp_batch_register_end");
    RegisteringGuard leave_on_exit(ctx);
    RegState st;
    const bool async = ctx->result_pending();
    bool had_stats;
    int64_t stats_m, eager;
    float stats_h;
    const char* pinned = nullptr;
    if (async) {  // copies were enqueued right behind the last iteration: wait for those only (the OLDEST result)
        icp_ctx::ResultSlot& r = ctx->rslot[ctx->r_head];
        ICP_HIP(ctx, hipEventSynchronize(r.wait));
        pinned = (const char*)r.host;
        memcpy(&st, pinned, sizeof(st));
        // a chunked launch (only ever the newest pending registration, here also the oldest): still running and
        // iterations held back -> the next chunk, then look again
        while (ctx->r_count == 1 && ctx->launch_remaining > 0 && !st.done && st.status == ICP_OK) {
            const int rc2 = continue_launch(ctx, 4);
            if (rc2) return rc2;
            ICP_HIP(ctx, hipEventSynchronize(r.wait));
            memcpy(&st, pinned, sizeof(st));
        }
        if (ctx->r_count == 1) ctx->launch_remaining = 0;  // (finished early: the rest is never enqueued)
        ctx->r_head ^= 1;
        ctx->r_count -= 1;
        had_stats = r.stats;
        stats_m = r.stats_m;
        stats_h = r.stats_h;
        eager = r.eager_normals;
    } else {
        ctx->in_registration = false;
        had_stats = ctx->stats_pending;
        stats_m = ctx->stats_m_pending;
        stats_h = ctx->stats_h_pending;
        ctx->stats_pending = false;
        eager = ctx->normals_eager_count;
        ctx->normals_eager_count = 0;
        ICP_HIP(ctx, hipMemcpyAsync(&st, ctx->state.ptr, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
        ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    std::vector<char> recovered;
    if (st.handoff_timeouts > 0 && ctx->r_count == 0) {  // (with a newer registration already enqueued behind it: the error below)
        const int rc_rec = recover_handoff(ctx, st, recovered);
        if (rc_rec) return rc_rec;
        if (async) pinned = recovered.data();
    }
    if (ctx->r_count == 0) ctx->update_behind_registration = false;  // (the guard above leaves the count of registering contexts)
    if (had_stats && st.grid_cells > 0) {
        ctx->occupied_cells = st.grid_cells;
        ctx->stats_m = stats_m;
        ctx->stats_h = stats_h;
    }
    if (ctx->search_stats) {
        std::vector<char> raw(DBG_BYTES);
        if (hipMemcpy(raw.data(), ctx->dbg_counts.ptr, DBG_BYTES, hipMemcpyDeviceToHost) == hipSuccess) {
            const int* c = (const int*)raw.data();
            fprintf(stderr, "[icp stats] N=%lld M=%lld h=%.3f iters=%d ring1=%d need_ring2=%d need_ring3=%d fine_failed=%d exhaustive=%d own_empty=%d misses=%d by-iteration(0-2,3-5,..)=%d,%d,%d,%d,%d,%d,%d knn: beyond ring 1 %d, beyond the fine rings %d\n",
                    (long long)ctx->tgt_n, (long long)ctx->map_m, ctx->cell_h, st.iter, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[8], c[9], c[10], c[11], c[12], c[13], c[14], c[7], c[15]);
            // phase timestamps per iteration launch (100 MHz wall clock -> us)
            const long long* tall = (const long long*)(c + 16);
            const int nb = (int)((ctx->tgt_n * 4 + 511) / 512);
            for (int it = 0; it < st.iter && it < 24 && nb > 0 && nb <= 1024; ++it) {
                const long long* t = tall + 4 * (size_t)it * 1024;
                if (t[0] == 0) continue;
                const long long MASK = (1ll << 48) - 1;
                long long first = t[0], last_start = t[0], last_end = t[3];
                double a = 0, b = 0, r = 0, amax = 0, bmax = 0, rmax = 0;
                int bmax_miss = 0, with_miss = 0, total_miss = 0, b_over2 = 0, b_over5 = 0;
                int seen = 0;
                first &= MASK, last_start &= MASK, last_end &= MASK;
                // (search_stats = 1: the top 16 bits of stamps 0 / 2 / 3 carry the workgroup's queries that went to the
                // coarse level / beyond ring 1 / found their own cell empty)
                struct Blk { double b; int miss, ring2, empty, coarse; };
                std::vector<Blk> blks;
                for (int i = 0; i < nb; ++i) {
                    const long long* q = t + 4 * i;
                    if (q[0] == 0) continue;  // the 512-queries-per-block shape launches a quarter of the blocks
                    if ((q[3] & MASK) == 0) continue;  // (slot 1023 of a lead launch: the LEAD's three stamps, no end stamp — not a workgroup of the iteration)
                    ++seen;
                    const long long t0 = q[0] & MASK, t1 = q[1] & MASK, t2 = q[2] & MASK, t3 = q[3] & MASK;
                    const int miss = (int)(q[1] >> 48);
                    if (t0 < first) first = t0;
                    if (t0 > last_start) last_start = t0;
                    if (t3 > last_end) last_end = t3;
                    const double da = (t1 - t0) * 0.01, db = (t2 - t1) * 0.01, dr = (t3 - t2) * 0.01;
                    blks.push_back(Blk{db, miss, (int)(q[2] >> 48), (int)(q[3] >> 48), (int)(q[0] >> 48)});
                    a += da; b += db; r += dr;
                    if (da > amax) amax = da;
                    if (db > bmax) { bmax = db; bmax_miss = miss; }
                    if (dr > rmax) rmax = dr;
                    with_miss += miss > 0;
                    total_miss += miss;
                    b_over2 += db > 2.0;
                    b_over5 += db > 5.0;
                }
                if (seen < 1023 && t[4 * 1023] != 0) {  // (the lead's stamps carry no counters)  // the lead workgroup of a lead launch left its stamps in slot 1023
                    const long long* q = t + 4 * 1023;
                    fprintf(stderr, "[icp lead] it %2d: starts %.2f us after the first workgroup; rows summed after %.2f, solved "
                                    "and published after %.2f more\n",
                            it, (q[0] - first) * 0.01, (q[1] - q[0]) * 0.01, (q[2] - q[1]) * 0.01);
                }
                if (it < 4 && ctx->search_stats_blocks) {  // dev ("search_stats" 3 | 4): the search phase of every workgroup, by logical block
                    fprintf(stderr, "[icp blocks] it %d:", it);
                    for (int i = 0; i < nb; ++i) {
                        const long long* q = t + 4 * i;
                        if (q[0] == 0) continue;
                        fprintf(stderr, " %d:%lld:%lld:%lld:%d:%d:%d:%d", i, ((q[1] & MASK) - (q[0] & MASK)),
                                ((q[2] & MASK) - (q[1] & MASK)), ((q[3] & MASK) - (q[2] & MASK)), (int)(q[1] >> 48),
                                (int)(q[2] >> 48), (int)(q[3] >> 48), (int)(q[0] >> 48));
                    }
                    fprintf(stderr, "\n");
                }
                fprintf(stderr, "[icp phases] it %2d: start skew %.2f, span %.2f us; A mean %.2f max %.2f; B mean %.2f max %.2f (that block: %d misses), blocks with B > 2 us: %d, > 5 us: %d; misses %d in %d blocks; reduce mean %.2f max %.2f\n",
                        it, (last_start - first) * 0.01, (last_end - first) * 0.01, a / seen, amax, b / seen, bmax, bmax_miss,
                        b_over2, b_over5, total_miss, with_miss, r / seen, rmax);
                if (ctx->search_stats == 1 && it < 8 && blks.size() >= 16) {
                    // what the slow workgroups have that the others do not: deciles of the search phase
                    std::sort(blks.begin(), blks.end(), [](const Blk& x, const Blk& y) { return x.b < y.b; });
                    const size_t nbk = blks.size();
                    fprintf(stderr, "[icp deciles] it %2d (B us: misses / beyond ring 1 / own cell empty / coarse, means):", it);
                    for (int d = 0; d < 10; ++d) {
                        const size_t lo = nbk * d / 10, hi = nbk * (d + 1) / 10;
                        double sb = 0, sm = 0, s2 = 0, se = 0, sc = 0;
                        for (size_t i = lo; i < hi; ++i) {
                            sb += blks[i].b; sm += blks[i].miss; s2 += blks[i].ring2; se += blks[i].empty; sc += blks[i].coarse;
                        }
                        const double k = (double)(hi - lo);
                        fprintf(stderr, " %.1f: %.0f/%.1f/%.1f/%.1f", sb / k, sm / k, s2 / k, se / k, sc / k);
                    }
                    fprintf(stderr, "\n");
                }
            }
            {   // eager normal estimation: 4 stamps per block (start, ring 1 done, stragglers done, eigen done)
                const long long* t = (const long long*)(raw.data() + DBG_ITER_BYTES);
                const int nb = (int)((ctx->map_m + 63) / 64);
                if (nb > 0 && nb <= 8192 && t[0] != 0) {
                    long long first = t[0], last_end = t[3];
                    double a = 0, b = 0, e = 0, amax = 0, bmax = 0;
                    int hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (int i = 0; i < nb; ++i) {
                        const long long* q = t + 4 * i;
                        if (q[0] < first) first = q[0];
                        if (q[3] > last_end) last_end = q[3];
                        const double da = (q[1] - q[0]) * 0.01, db = (q[2] - q[1]) * 0.01, de = (q[3] - q[2]) * 0.01;
                        a += da; b += db; e += de;
                        if (da > amax) amax = da;
                        if (db > bmax) bmax = db;
                        int bin = (int)((q[3] - q[0]) * 0.01 / 10.0);
                        hist[bin > 7 ? 7 : bin]++;
                    }
                    long long last_start = first;
                    for (int i = 0; i < nb; ++i) if (t[4 * i] > last_start) last_start = t[4 * i];
                    fprintf(stderr, "[icp normals] blocks=%d span %.1f us (last block starts at %.1f); ring 1 mean %.1f max %.1f; stragglers mean %.1f max %.1f; eigen mean %.1f; block duration histogram (10 us bins): %d %d %d %d %d %d %d %d\n",
                            nb, (last_end - first) * 0.01, (last_start - first) * 0.01, a / nb, amax, b / nb, bmax, e / nb,
                            hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
                }
            }
            (void)hipMemset(ctx->dbg_counts.ptr, 0, DBG_BYTES);
        }
    }
    st.normals_computed += eager;
    memcpy(result->pose, st.pose, sizeof(st.pose));
    memcpy(result->params, st.params, sizeof(st.params));
    result->iterations = st.iter;
    ctx->last_iterations = st.iter;
    ctx->last_valid_targets = st.n_targets;
    result->converged = st.converged;
    result->status = st.status;
    result->num_targets = st.n_targets;
    result->normals_computed = st.normals_computed;
    const int k = st.iter < ctx->hist_cap ? st.iter : ctx->hist_cap;
    if (async) {
        const char* lh = pinned + STATE_BLOCK;
        if (k > 0 && loss_per_iter_out) memcpy(loss_per_iter_out, lh, (size_t)k * sizeof(double));
        if (k > 0 && dx_per_iter_out)
            memcpy(dx_per_iter_out, lh + (size_t)ctx->hist_cap * sizeof(double), (size_t)k * 6 * sizeof(float));
    } else {
        if (k > 0 && loss_per_iter_out)
            ICP_HIP(ctx, hipMemcpy(loss_per_iter_out, ctx->loss_hist, (size_t)k * sizeof(double), hipMemcpyDeviceToHost));
        if (k > 0 && dx_per_iter_out)
            ICP_HIP(ctx, hipMemcpy(dx_per_iter_out, ctx->dx_hist, (size_t)k * 6 * sizeof(float), hipMemcpyDeviceToHost));
    }
    if (ctx->prof.enabled) prof_collect(ctx);
    if (st.handoff_timeouts > 0)
        return fail(ctx, ICP_ERR_HIP, "internal: the pose hand-off inside a fused iteration launch timed out");
    if (st.status == ICP_ERR_INVALID_JACOBIAN)
        return fail(ctx, ICP_ERR_INVALID_JACOBIAN, "Invalid Jacobian in Gauss Newton minimization");
    if (st.status == ICP_ERR_EXCHANGE) {
        // the ranks' sequence counters and inbox tags may have diverged (a rank that timed out did not advance, one that
        // received everything did): the exchange is unusable until it is created and connected again
        ctx->exchange_on = false;
        return fail(ctx, ICP_ERR_EXCHANGE, "multi-GPU exchange: a peer did not deliver its normal equations in time; "
                                           "the context left exchange mode (icp_exchange_create / _connect again)");
    }
    return st.status;
}

// enqueues iterations [first, first + count) of the registration in progress (count < 0: all of them)
static int enqueue_iterations(icp_ctx* ctx, bool poll_allowed, int first, int count) {
    int rc = ICP_OK;
    const int iters = count < 0 ? ctx->cfg.max_num_alignments
                                : (first + count < ctx->cfg.max_num_alignments ? first + count : ctx->cfg.max_num_alignments);
    // the loop never converges early when the threshold is <= 0 (forced iteration count): no point polling
    const int poll = (poll_allowed && ctx->cfg.threshold_delta_pose > 0.f) ? ctx->cfg.poll_every : 0;
    // lead launches: the solve of iteration k rides in the head of launch k + 1 (LeadArgs, icp_internal.h); one summing /
    // solving launch remains, behind the last iteration.  Not with the host polling in between (it reads the RegState,
    // which would lag one iteration) and not with the in-library exchange (its solve waits for the peers)
    const bool lead = ctx->lead_latched && fused_path(ctx) && !ctx->exchange_on && poll == 0;
    int prev_rows = 0, prev_quad = 1;  // rows a lead launch still has to solve
    for (int it = first; it < iters; ++it) {
        if (lead && fused_tail_possible(ctx, prev_rows, iters - it)) {
            // the resident tail: every remaining iteration, and the solve behind the last one, in ONE launch
            int rows = 0, quad = 1;
            rc = launch_iterate_fused(ctx, &rows, &quad, true, prev_rows, prev_quad, iters - it);
            if (rc) {
                ctx->in_registration = false;
                return rc;
            }
            break;
        }
        if (lead) {
            // every launch takes its pose from the mailbox; the NARROW ones (late iterations: one workgroup more fits beside
            // the others) also solve the iteration before them, the others follow a summing / solving launch as before
            int rows = 0, quad = 1;
            rc = launch_iterate_fused(ctx, &rows, &quad, true, prev_rows, prev_quad);
            prev_rows = rows;
            prev_quad = quad;
            // (rows of a dense launch: four times as many, a quarter of a microsecond each for ONE lead workgroup — with
            // "lead_after_dense" = 0 they keep their own summing launch and the lead chain starts one iteration later)
            if (!rc && (it + 1 == iters || !next_fused_launch_is_narrow(ctx) || (quad && !ctx->lead_after_dense))) {
                // the rows of this launch: the parity it has just written
                rc = launch_sum_solve(ctx, rows, quad, (const double*)(ctx->partials.as<char>() +
                                                                       (size_t)(ctx->partials_parity ^ 1) * ctx->partials_half),
                                      true, it + 1 == iters);
                prev_rows = 0;
            }
        } else if (fused_path(ctx)) {
            int rows = 0, quad = 1;
            rc = launch_iterate_fused(ctx, &rows, &quad);
            if (!rc) rc = launch_sum_solve(ctx, rows, quad, nullptr, false, it + 1 == iters && poll == 0);
        } else if (ctx->cost == ICP_COST_POINT_TO_POINT) {
            (rc = launch_search(ctx)) || (rc = launch_reduce_p2p(ctx, true));
            ctx->iter_in_registration += 1;  // (the fused launch counts itself)
        } else {
            (rc = launch_search(ctx)) || (rc = launch_normals(ctx)) || (rc = launch_reduce_solve(ctx));
            ctx->iter_in_registration += 1;
        }
        if (rc) {
            ctx->in_registration = false;
            return rc;
        }
        if (poll > 0 && (it + 1) % poll == 0 && it + 1 < iters) {
            int done = 0;
            ICP_HIP(ctx, hipMemcpyAsync(&done, &reg_state(ctx)->done, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (done) break;
        }
    }
    return ICP_OK;
}

static int register_launch(icp_ctx* ctx, const float* xyz, int64_t n, int mem, int target_mode,
                           const float init_pose[16], bool from_last) {
    if (!ctx) return ICP_ERR_INVALID_ARGUMENT;
    RegisteringGuard leave_on_error(ctx);
    if (ctx->r_count >= 2) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "two results are already pending");
    int rc = continue_launch(ctx, -1);  // (an older launch still holding iterations back: they come first on the stream)
    if (rc) return rc;
    rc = register_begin(ctx, xyz, n, mem, target_mode, init_pose, from_last);
    if (rc) return rc;
    // forced iteration count (threshold <= 0): every iteration is enqueued.  Live threshold: a first chunk of as many
    // iterations as the last frame ran plus one (launches behind an early stop would see `done` on the device and return
    // at once — but each still costs its slot on the stream, twenty of them more than the three that do the work);
    // icp_register_end enqueues further chunks while the loop is still running
    const int iters = ctx->cfg.max_num_alignments;
    int first_chunk = iters;
    // (lead launches that end in a resident tail run to the end of the loop on the device: nothing to chunk)
    const bool tail = ctx->lead_latched && fused_path(ctx) && !ctx->exchange_on && fused_tail_planned(ctx, iters);
    if (ctx->cfg.threshold_delta_pose > 0.f && ctx->chunked_launch && !tail) {
        first_chunk = (ctx->last_iterations > 0 ? ctx->last_iterations : 3) + 1;
        if (first_chunk > iters) first_chunk = iters;
    }
    rc = result_fold_begin(ctx, false);  // (the last solving launch of the chunk delivers the result block itself)
    if (!rc) rc = enqueue_iterations(ctx, false, 0, first_chunk);
    ctx->launch_enqueued = first_chunk;
    ctx->launch_remaining = iters - first_chunk;
    if (!rc) rc = enqueue_result_copy(ctx);
    ctx->result_fold_to = nullptr;
    ctx->result_folded = false;
    ctx->in_registration = false;  // the result waits in its slot
    return rc;
}

int icp_register_launch(icp_ctx* ctx, const float* xyz, int64_t n, int mem, int target_mode, const float init_pose[16]) {
    DeviceGuard device_guard(ctx);
    return register_launch(ctx, xyz, n, mem, target_mode, init_pose, false);
}

int icp_register_launch_from_last(icp_ctx* ctx, const float* xyz, int64_t n, int mem, int target_mode) {
    DeviceGuard device_guard(ctx);
    return register_launch(ctx, xyz, n, mem, target_mode, nullptr, true);
}

int icp_register(icp_ctx* ctx, const float* xyz, int64_t n, int mem, int target_mode, const float init_pose[16],
                 icp_register_result* result, double* loss_per_iter_out, float* dx_per_iter_out) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !result) return ICP_ERR_INVALID_ARGUMENT;
    RegisteringGuard leave_on_error(ctx);
    int rc = icp_register_begin(ctx, xyz, n, mem, target_mode, init_pose);
    if (rc) return rc;
    if ((rc = enqueue_iterations(ctx, true))) return rc;
    return icp_register_end(ctx, result, loss_per_iter_out, dx_per_iter_out);
}

// ---- B sequences per launch (include/icp_mi355x.h: icp_batch_*) ----------------------------------------------------------
// The descriptor tables of ONE frame — iteration launch after iteration launch, [count] entries each, the summing / solving
// launches in between — are filled in pinned host memory while the frame's launches are PREPARED (all host bookkeeping of
// all iterations of all members first: nothing in it depends on a launch having been issued), travel to the device in one
// copy, and the launches follow.  Three slots in rotation: a slot is rewritten when the registration two frames back has been
// collected (at most two registrations are ever pending per member), so its launches have long finished.
struct icp_batch {
    std::vector<icp_ctx*> members;
    std::string error;
    int device = 0;
    static constexpr int SLOTS = 3;
    char* host[SLOTS] = {nullptr, nullptr, nullptr};  // pinned
    size_t host_bytes[SLOTS] = {0, 0, 0};
    DeviceBuffer dev[SLOTS];
    hipEvent_t copied[SLOTS] = {nullptr, nullptr, nullptr};  // the slot's copy has left the pinned buffer
    hipEvent_t done[2] = {nullptr, nullptr};                  // ONE event behind the results of a batched registration
    int slot = 0, done_next = 0;
    // the registration in progress: iterations [run_next, run_iters) are still held back (a live stop threshold: chunks)
    bool run_active = false, run_lead = false;
    int run_next = 0, run_iters = 0, run_done = 0;
    int run_prev_rows[ICP_BATCH_MAX_SEQUENCES] = {}, run_prev_quad[ICP_BATCH_MAX_SEQUENCES] = {};
    // ... and of the batched grid build behind a map update: [count] GridBuildDesc per slot
    GridBuildDesc* grid_host[SLOTS] = {nullptr, nullptr, nullptr};
    DeviceBuffer grid_dev[SLOTS];
    hipEvent_t grid_copied[SLOTS] = {nullptr, nullptr, nullptr};
    int grid_slot = 0;
};

static int batch_fail(icp_batch* b, int code, const std::string& msg) {
    if (b) b->error = msg;
    return code;
}

int icp_batch_create(icp_ctx* const* ctxs, int32_t count, icp_batch** out) {
    if (!ctxs || !out || count < 1 || count > ICP_BATCH_MAX_SEQUENCES) return ICP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    for (int i = 0; i < count; ++i) {
        if (!ctxs[i] || ctxs[i]->cfg.device != ctxs[0]->cfg.device) return ICP_ERR_INVALID_ARGUMENT;
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return ICP_ERR_INVALID_ARGUMENT;
    }
    icp_batch* b = new icp_batch();
    b->members.assign(ctxs, ctxs + count);
    b->device = ctxs[0]->cfg.device;
    *out = b;
    return ICP_OK;
}

void icp_batch_destroy(icp_batch* b) {
    if (!b) return;
    DeviceGuard device_guard(b->device);
    (void)hipDeviceSynchronize();  // (launches that read the tables; members whose result slots wait for the batch's events)
    for (icp_ctx* ctx : b->members) {
        ctx->batch_hold = false;  // (iterations the batch still held back are dropped: the members' results stand as they are)
        for (auto& r : ctx->rslot)
            if (r.wait == b->done[0] || r.wait == b->done[1]) r.wait = r.event;
    }
    for (int k = 0; k < icp_batch::SLOTS; ++k) {
        if (b->host[k]) (void)hipHostFree(b->host[k]);
        b->dev[k].release();
        if (b->copied[k]) (void)hipEventDestroy(b->copied[k]);
        if (b->grid_host[k]) (void)hipHostFree(b->grid_host[k]);
        b->grid_dev[k].release();
        if (b->grid_copied[k]) (void)hipEventDestroy(b->grid_copied[k]);
    }
    for (auto& e : b->done)
        if (e) (void)hipEventDestroy(e);
    delete b;
}

const char* icp_batch_last_error(const icp_batch* b) { return b ? b->error.c_str() : "null batch"; }

int icp_batch_set_stream(icp_batch* b, void* hip_stream) {
    if (!b) return ICP_ERR_INVALID_ARGUMENT;
    for (icp_ctx* ctx : b->members) {
        const int rc = icp_set_stream(ctx, hip_stream);
        if (rc) return batch_fail(b, rc, ctx->error);
    }
    return ICP_OK;
}

// Enqueues the next `chunk` iterations of the batch's registration in progress (all that are left when chunk < 0): their
// descriptor tables into a fresh pinned slot, one copy, the launches; behind the chunk's last iteration a summing / solving
// launch that also delivers every member's result block; ONE event behind it all.  `packs` (first chunk only): the target
// packing / state initialisation of the members, run in front.
static int batch_enqueue(icp_batch* b, int chunk, const PackDesc* packs, bool first_chunk) {
    const int count = (int)b->members.size();
    icp_ctx* const* ctxs = b->members.data();
    icp_ctx* first = ctxs[0];
    const int left = b->run_iters - b->run_next;
    if (chunk < 0 || chunk > left) chunk = left;
    if (chunk <= 0) return ICP_OK;
    const int it_begin = b->run_next, it_end = it_begin + chunk;
    const bool lead = b->run_lead;
    int rc = ICP_OK;
    const int slot = b->slot;
    b->slot = (slot + 1) % icp_batch::SLOTS;
    const size_t it_bytes = iterate_desc_bytes() * (size_t)count, ss_bytes = sum_solve_desc_bytes() * (size_t)count;
    const size_t pack_bytes = ((sizeof(PackDesc) * (size_t)count + 255) / 256) * 256;  // the packing launch's table leads the slot
    const size_t need = pack_bytes + (size_t)chunk * (it_bytes + ss_bytes);
    if (b->host_bytes[slot] < need) {
        if (b->copied[slot]) ICP_HIP(first, hipEventSynchronize(b->copied[slot]));
        if (b->host[slot]) (void)hipHostFree(b->host[slot]);
        b->host[slot] = nullptr;
        b->host_bytes[slot] = 0;
        ICP_HIP(first, hipHostMalloc((void**)&b->host[slot], need, hipHostMallocDefault));
        b->host_bytes[slot] = need;
    }
    ICP_HIP(first, b->dev[slot].reserve(need));
    if (!b->copied[slot]) ICP_HIP(first, hipEventCreateWithFlags(&b->copied[slot], hipEventDisableTiming));
    else ICP_HIP(first, hipEventSynchronize(b->copied[slot]));  // (three chunks ago: long past)
    for (int i = 0; i < count; ++i) {
        ctxs[i]->in_registration = true;
        if ((rc = result_fold_begin(ctxs[i], !first_chunk))) return batch_fail(b, rc, ctxs[i]->error);
    }
    struct Op {
        int kind;  // 0: fused iteration, 1: sum + solve
        size_t offset;
        BatchedIteration it;
    };
    std::vector<Op> ops;
    ops.reserve(2 * (size_t)chunk);
    if (packs) memcpy(b->host[slot], packs, sizeof(PackDesc) * (size_t)count);
    size_t used = pack_bytes;
    for (int it = it_begin; it < it_end; ++it) {
        Op op{0, used, BatchedIteration()};
        if ((rc = prepare_iterate_batch(ctxs, count, lead, b->run_prev_rows, b->run_prev_quad, b->host[slot] + used, &op.it))) {
            for (int i = 0; i < count; ++i)
                if (!ctxs[i]->error.empty()) b->error = ctxs[i]->error;
            return rc;
        }
        used += it_bytes;
        ops.push_back(op);
        bool solve = true;
        if (lead) {  // (as enqueue_iterations: the narrow launches solve the iteration before them themselves)
            for (int i = 0; i < count; ++i) {
                b->run_prev_rows[i] = op.it.rows[i];
                b->run_prev_quad[i] = op.it.quad[i];
            }
            solve = it + 1 == it_end || !next_fused_launch_is_narrow(first) || (op.it.quad[0] && !first->lead_after_dense);
        }
        if (solve) {
            Op so{1, used, BatchedIteration()};
            if ((rc = prepare_sum_solve_batch(ctxs, count, op.it.rows, op.it.quad, lead, it + 1 == it_end, b->host[slot] + used)))
                return batch_fail(b, rc, "batched registration: sum + solve");
            used += ss_bytes;
            ops.push_back(so);
            for (int i = 0; i < count; ++i) b->run_prev_rows[i] = 0;
        }
    }
    b->run_next = it_end;
    // ---- one copy, then the launches
    ICP_HIP(first, hipMemcpyAsync(b->dev[slot].ptr, b->host[slot], used, hipMemcpyHostToDevice, first->stream));
    ICP_HIP(first, hipEventRecord(b->copied[slot], first->stream));
    if (packs && (rc = launch_pack_targets_batch(first, packs, b->dev[slot].as<PackDesc>(), count))) return batch_fail(b, rc, first->error);
    for (const Op& op : ops) {
        const char* table = b->dev[slot].as<char>() + op.offset;
        rc = op.kind == 0 ? launch_iterate_batch(first, op.it, table) : launch_sum_solve_batch(first, count, table);
        if (rc) return batch_fail(b, rc, first->error);
    }
    // ---- the results: every member's block lands in its own pinned slot (written by the chunk's last solving launch), ONE
    // event; further chunks of the same registration record the same event again
    if (first_chunk) {
        b->run_done = b->done_next;
        b->done_next ^= 1;
    }
    hipEvent_t& done = b->done[b->run_done];
    if (!done) ICP_HIP(first, hipEventCreateWithFlags(&done, hipEventDisableTiming));
    for (int i = 0; i < count; ++i) {
        icp_ctx* ctx = ctxs[i];
        ctx->launch_enqueued = it_end;
        ctx->launch_remaining = 0;  // (what the batch still holds back is the batch's: run_next .. run_iters)
        ctx->batch_hold = it_end < b->run_iters;
        if ((rc = enqueue_result_copy(ctx, !first_chunk, done))) return batch_fail(b, rc, ctx->error);
        ctx->result_fold_to = nullptr;
        ctx->result_folded = false;
        ctx->in_registration = false;  // the result waits in its slot
    }
    ICP_HIP(first, hipEventRecord(done, first->stream));
    if (it_end >= b->run_iters) b->run_active = false;
    return ICP_OK;
}

// everything the batch still holds back goes onto the stream (a map update by the device-resident poses, another launch, the
// destruction of the batch follow the WHOLE registration)
static int batch_flush(icp_batch* b) {
    if (!b->run_active) return ICP_OK;
    const int rc = batch_enqueue(b, -1, nullptr, false);
    b->run_active = false;
    for (icp_ctx* ctx : b->members) ctx->batch_hold = false;
    return rc;
}

int icp_batch_register_launch(icp_batch* b, const float* const* xyz, const int64_t* n, int mem, int target_mode,
                              const float* init_poses, int from_last) {
    if (!b || !xyz || !n) return ICP_ERR_INVALID_ARGUMENT;
    DeviceGuard device_guard(b->device);
    const int count = (int)b->members.size();
    icp_ctx* const* ctxs = b->members.data();
    icp_ctx* first = ctxs[0];
    const int iters = first->cfg.max_num_alignments;
    for (int i = 0; i < count; ++i) {
        icp_ctx* ctx = ctxs[i];
        { DeviceGuard join_map_stream(ctx); }
        if (ctx->stream != first->stream)
            return batch_fail(b, ICP_ERR_INVALID_ARGUMENT, "batched registration: the members must enqueue on one stream (icp_batch_set_stream)");
        if (ctx->cfg.max_num_alignments != iters || ctx->cfg.scheme != first->cfg.scheme || ctx->cfg.sigma != first->cfg.sigma)
            return batch_fail(b, ICP_ERR_INVALID_ARGUMENT, "batched registration: the members must share max_num_alignments, scheme and sigma");
        if (ctx->r_count >= 2) return batch_fail(b, ICP_ERR_INVALID_ARGUMENT, "two results are already pending");
        if (ctx->exchange_on || ctx->prof.enabled || ctx->search_stats || ctx->cost != ICP_COST_POINT_TO_PLANE)
            return batch_fail(b, ICP_ERR_INVALID_ARGUMENT, "batched registration: point-to-plane registrations without exchange, "
                                                          "profiling or search statistics only");
    }
    struct Unwind {  // an error below leaves no member "in registration" (and none counted among the registering contexts)
        icp_batch* b;
        bool armed = true;
        ~Unwind() {
            if (!armed) return;
            for (icp_ctx* ctx : b->members) {
                ctx->in_registration = false;
                ctx->result_fold_to = nullptr;
                ctx->result_folded = false;
                if (ctx->r_count == 0) registering_leave(ctx);
            }
        }
    } unwind{b};
    int rc = ICP_OK;
    if ((rc = batch_flush(b))) return rc;  // (iterations of the previous batched registration still held back go first)
    PackDesc packs[ICP_BATCH_MAX_SEQUENCES];
    for (int i = 0; i < count; ++i) {
        icp_ctx* ctx = ctxs[i];
        if ((rc = continue_launch(ctx, -1))) return batch_fail(b, rc, ctx->error);  // (an older chunked launch of this member goes first)
        if ((rc = register_begin(ctx, xyz[i], n[i], mem, target_mode, (init_poses && !from_last) ? init_poses + 16 * i : nullptr,
                                 from_last != 0, &packs[i])))
            return batch_fail(b, rc, ctx->error);
        if (!fused_path(ctx) || ctx->lazy_now)
            return batch_fail(b, ICP_ERR_INVALID_ARGUMENT, "batched registration: the fused iteration path with eagerly estimated "
                                                          "normals only (see icp_set_option: fuse_iteration, lazy_fused, eager_normals_limit)");
    }
    // lead launches (every member's solve in the head of the next launch, by a lead workgroup of its own) unless a context
    // OUTSIDE the batch is registering on the device: the members themselves are counted, and are no strangers
    const int d = b->device >= 0 && b->device < 64 ? b->device : 0;
    int counted = 0;
    bool lead = true;
    for (int i = 0; i < count; ++i) {
        counted += ctxs[i]->counted_registering ? 1 : 0;
        lead = lead && ctxs[i]->lead_solve && !ctxs[i]->handoff_disabled;
    }
    lead = lead && g_registering[d].load() <= counted;
    for (int i = 0; i < count; ++i) ctxs[i]->lead_latched = lead;
    // ---- the iterations: all of them (forced count), or a first chunk — as many as the slowest member ran last time plus one —
    // with a live stop threshold (icp_batch_register_end enqueues more while a member is still running; launches behind
    // every member's stop would be no-ops that still cost their slot on the stream: a hundred of them with the default
    // max_num_alignments)
    int first_chunk = iters;
    if (first->cfg.threshold_delta_pose > 0.f && first->chunked_launch) {
        int most = 0;
        for (int i = 0; i < count; ++i) most = std::max(most, ctxs[i]->last_iterations > 0 ? ctxs[i]->last_iterations : 3);
        first_chunk = std::min(iters, most + 1);
    }
    b->run_active = true;
    b->run_lead = lead;
    b->run_next = 0;
    b->run_iters = iters;
    for (int i = 0; i < count; ++i) {
        b->run_prev_rows[i] = 0;
        b->run_prev_quad[i] = 1;
    }
    if ((rc = batch_enqueue(b, first_chunk, packs, true))) return rc;
    unwind.armed = false;
    return ICP_OK;
}

int icp_batch_project(icp_batch* b, const float* const* xyz, const int64_t* n, float* const* vmap_out) {
    if (!b || !xyz || !n || !vmap_out) return ICP_ERR_INVALID_ARGUMENT;
    DeviceGuard device_guard(b->device);
    const int count = (int)b->members.size();
    for (int i = 0; i < count; ++i)
        if (n[i] < 0 || (n[i] > 0 && !xyz[i]) || !vmap_out[i] || b->members[i]->stream != b->members[0]->stream)
            return batch_fail(b, ICP_ERR_INVALID_ARGUMENT, "batched projection: device pointers, one stream (icp_batch_set_stream)");
    const int rc = project_batch_device(b->members.data(), count, xyz, n, vmap_out);
    if (rc) return batch_fail(b, rc, b->members[0]->error);
    return ICP_OK;
}

int icp_batch_map_update(icp_batch* b) {
    if (!b) return ICP_ERR_INVALID_ARGUMENT;
    DeviceGuard device_guard(b->device);
    const int count = (int)b->members.size();
    icp_ctx* first = b->members[0];
    {   // the pose-only update reads the END of the registration: iterations the batch still holds back go first
        const int rc_flush = batch_flush(b);
        if (rc_flush) return rc_flush;
    }
    bool one_stream = true;
    for (icp_ctx* ctx : b->members) {
        { DeviceGuard join_map_stream(ctx); }
        one_stream = one_stream && ctx->stream == first->stream;
        if (!ctx->have_device_pose)
            return batch_fail(b, ICP_ERR_INVALID_ARGUMENT, "rel_pose = NULL needs a previous registration on every member");
    }
    if (!one_stream || count == 1) {  // members on streams of their own: one update each, as icp_map_update
        for (icp_ctx* ctx : b->members) {
            const int rc = icp_map_update(ctx, nullptr, nullptr, 0, ICP_MEM_DEVICE, ICP_TARGETS_ALL, nullptr);
            if (rc) return batch_fail(b, rc, ctx->error);
        }
        return ICP_OK;
    }
    // the host bookkeeping of every member's update (window, jobs of the rebuild, buffers), the launches of the grid builds
    // left out: their arguments travel to the device in one table, four launches build the B grids
    const int slot = b->grid_slot;
    b->grid_slot = (slot + 1) % icp_batch::SLOTS;
    if (!b->grid_host[slot]) {
        ICP_HIP(first, hipHostMalloc((void**)&b->grid_host[slot], sizeof(GridBuildDesc) * ICP_BATCH_MAX_SEQUENCES, hipHostMallocDefault));
        ICP_HIP(first, b->grid_dev[slot].reserve(sizeof(GridBuildDesc) * ICP_BATCH_MAX_SEQUENCES));
        ICP_HIP(first, hipEventCreateWithFlags(&b->grid_copied[slot], hipEventDisableTiming));
    } else {
        ICP_HIP(first, hipEventSynchronize(b->grid_copied[slot]));  // (three updates ago)
    }
    GridBuildDesc* table = b->grid_host[slot];
    int rc = ICP_OK;
    for (int i = 0; i < count; ++i) {
        icp_ctx* ctx = b->members[i];
        if ((rc = continue_launch(ctx, -1))) return batch_fail(b, rc, ctx->error);
        if ((rc = ensure_state(ctx))) return batch_fail(b, rc, ctx->error);
        if ((rc = map_update_body(ctx, nullptr, nullptr, nullptr, 0, false, nullptr, -1, &table[i])))
            return batch_fail(b, rc, ctx->error);
    }
    ICP_HIP(first, hipMemcpyAsync(b->grid_dev[slot].ptr, table, sizeof(GridBuildDesc) * (size_t)count, hipMemcpyHostToDevice,
                                  first->stream));
    ICP_HIP(first, hipEventRecord(b->grid_copied[slot], first->stream));
    if ((rc = launch_grid_build_batch(first, table, b->grid_dev[slot].as<GridBuildDesc>(), count)))
        return batch_fail(b, rc, first->error);
    for (int i = 0; i < count; ++i) {
        icp_ctx* ctx = b->members[i];
        if ((rc = build_grid_finish(ctx, table[i])) || (rc = map_update_finish(ctx))) return batch_fail(b, rc, ctx->error);
    }
    return ICP_OK;
}

int icp_batch_register_end(icp_batch* b, icp_register_result* results, double* loss_per_iter_out, float* dx_per_iter_out) {
    if (!b || !results) return ICP_ERR_INVALID_ARGUMENT;
    DeviceGuard device_guard(b->device);
    // a registration enqueued in chunks: wait for what is on the stream; while a member is still running and iterations are
    // held back, the next chunk (four iterations), and look again
    while (b->run_active) {
        icp_ctx* first = b->members[0];
        ICP_HIP(first, hipEventSynchronize(b->done[b->run_done]));
        bool running = false;
        for (icp_ctx* ctx : b->members) {
            if (ctx->r_count <= 0) continue;
            RegState st;
            memcpy(&st, ctx->rslot[(ctx->r_head + ctx->r_count - 1) & 1].host, sizeof(st));  // (the newest pending result: this registration's)
            running = running || (!st.done && st.status == ICP_OK && st.handoff_timeouts == 0);
        }
        if (!running) {
            b->run_active = false;  // (every member has stopped: the rest is never enqueued)
            for (icp_ctx* ctx : b->members) ctx->batch_hold = false;
            break;
        }
        const int rc_chunk = batch_enqueue(b, 4, nullptr, false);
        if (rc_chunk) return rc_chunk;
    }
    for (icp_ctx* ctx : b->members) ctx->batch_hold = false;
    int first_rc = ICP_OK;
    const size_t cap = (size_t)b->members[0]->cfg.max_num_alignments;
    for (size_t i = 0; i < b->members.size(); ++i) {
        icp_ctx* ctx = b->members[i];
        const int rc = icp_register_end(ctx, &results[i], loss_per_iter_out ? loss_per_iter_out + i * cap : nullptr,
                                        dx_per_iter_out ? dx_per_iter_out + i * cap * 6 : nullptr);
        if (rc && !first_rc) {
            first_rc = rc;
            b->error = ctx->error;
        }
    }
    return first_rc;
}

void* icp_normal_equations_ptr(icp_ctx* ctx) {
    DeviceGuard device_guard(ctx);
    if (!ctx) return nullptr;
    if (ensure_state(ctx) != ICP_OK) return nullptr;
    return ctx->neq;
}

int icp_set_normal_equations_buffer(icp_ctx* ctx, void* device_ptr) {
    DeviceGuard device_guard(ctx);
    if (!ctx) return ICP_ERR_INVALID_ARGUMENT;
    int rc = ensure_state(ctx);
    if (rc) return rc;
    ctx->neq = device_ptr ? (double*)device_ptr : ctx->neq_own.as<double>();
    return ICP_OK;
}

// ---- in-library exchange --------------------------------------------------------------------------------------------
static void exchange_release(icp_ctx* ctx) {
    for (int r = 0; r < EXCHANGE_MAX_RANKS; ++r) {
        if (ctx->x_peer[r]) (void)hipIpcCloseMemHandle(ctx->x_peer[r]);
        ctx->x_peer[r] = nullptr;
    }
    if (ctx->x_inbox) (void)hipFree(ctx->x_inbox);
    ctx->x_inbox = nullptr;
    ctx->exchange_on = false;
}

int icp_exchange_create(icp_ctx* ctx, int32_t rank, int32_t world, void* handle_out) {
    DeviceGuard device_guard(ctx);
    static_assert(sizeof(hipIpcMemHandle_t) == ICP_EXCHANGE_HANDLE_BYTES, "IPC handle size");
    if (!ctx || !handle_out || world < 1 || world > EXCHANGE_MAX_RANKS || rank < 0 || rank >= world)
        return ICP_ERR_INVALID_ARGUMENT;
    if (ctx->in_registration || ctx->result_pending()) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "registration in progress");
    int rc = ensure_state(ctx);
    if (rc) return rc;
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    exchange_release(ctx);
    const size_t bytes = (size_t)2 * world * sizeof(ExchangeSlot);
    ICP_HIP(ctx, hipExtMallocWithFlags(&ctx->x_inbox, bytes, hipDeviceMallocUncached));
    ICP_HIP(ctx, hipMemset(ctx->x_inbox, 0, bytes));
    ICP_HIP(ctx, ctx->x_seq.reserve(64));
    ICP_HIP(ctx, hipMemset(ctx->x_seq.ptr, 0, 64));
    ICP_HIP(ctx, hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    ICP_HIP(ctx, hipIpcGetMemHandle(&h, ctx->x_inbox));
    memcpy(handle_out, &h, sizeof(h));
    memset(&ctx->xview, 0, sizeof(ctx->xview));
    ctx->xview.rank = rank;
    ctx->xview.world = world;
    ctx->xview.seq = ctx->x_seq.as<unsigned long long>();
    ctx->xview.inbox[rank] = (ExchangeSlot*)ctx->x_inbox;
    return ICP_OK;
}

int icp_exchange_connect(icp_ctx* ctx, const void* handles) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !handles) return ICP_ERR_INVALID_ARGUMENT;
    if (!ctx->x_inbox) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "icp_exchange_create first");
    const int world = ctx->xview.world, rank = ctx->xview.rank;
    for (int r = 0; r < world; ++r) {
        if (r == rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + (size_t)r * ICP_EXCHANGE_HANDLE_BYTES, sizeof(h));
        void* p = nullptr;
        ICP_HIP(ctx, hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        ctx->x_peer[r] = p;
        ctx->xview.inbox[r] = (ExchangeSlot*)p;
    }
    ctx->exchange_on = true;
    return ICP_OK;
}

int icp_exchange_destroy(icp_ctx* ctx) {
    DeviceGuard device_guard(ctx);
    if (!ctx) return ICP_ERR_INVALID_ARGUMENT;
    if (ctx->in_registration || ctx->result_pending()) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "registration in progress");
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    exchange_release(ctx);
    return ICP_OK;
}

// ---- map-sharded normals --------------------------------------------------------------------------------------------
int icp_map_normals_owned(icp_ctx* ctx, int32_t rank, int32_t world, float* normals_by_index) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !normals_by_index || world < 1 || rank < 0 || rank >= world) return ICP_ERR_INVALID_ARGUMENT;
    {   // iterations a chunked launch still holds back run against the map / configuration they were launched with
        const int rc_held = continue_launch(ctx, -1);
        if (rc_held) return rc_held;
    }
    if (ctx->map_m <= 0 || !ctx->grid_valid) return fail(ctx, ICP_ERR_EMPTY_MAP, "the local map is empty");
    if (ctx->in_registration) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "registration in progress");
    ctx->sharded_normals = true;  // (the grid builds from now on keep the neighbourhood lists for it)
    return launch_normals_owned(ctx, rank, world, normals_by_index);
}

int icp_map_normals_install(icp_ctx* ctx, const float* normals_by_index) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !normals_by_index) return ICP_ERR_INVALID_ARGUMENT;
    {   // iterations a chunked launch still holds back run against the map / configuration they were launched with
        const int rc_held = continue_launch(ctx, -1);
        if (rc_held) return rc_held;
    }
    if (ctx->map_m <= 0 || !ctx->grid_valid) return fail(ctx, ICP_ERR_EMPTY_MAP, "the local map is empty");
    if (ctx->in_registration) return fail(ctx, ICP_ERR_INVALID_ARGUMENT, "registration in progress");
    return launch_normals_install(ctx, normals_by_index);
}

// ---- profiling ------------------------------------------------------------------------------------------------------
int icp_profile_enable(icp_ctx* ctx, int enable) {
    DeviceGuard device_guard(ctx);
    if (!ctx) return ICP_ERR_INVALID_ARGUMENT;
    ctx->prof.enabled = enable != 0;
    ctx->prof.mask = enable;
    ctx->prof.pending.clear();
    for (int k = 0; k < 3; ++k) {
        ctx->prof.ms[k] = 0;
        ctx->prof.launches[k] = 0;
    }
    for (int k = 0; k < Profile::ITER_SLOTS; ++k) {
        ctx->prof.ms_iter[k] = 0;
        ctx->prof.launches_iter[k] = 0;
    }
    ctx->prof.registrations = 0;
    return ICP_OK;
}

int icp_profile_read_iterations(icp_ctx* ctx, double* ms_out, int64_t* launches_out, int32_t cap) {
    DeviceGuard device_guard(ctx, false);  // (beside a map update on its own stream)
    if (!ctx || !ms_out || !launches_out || cap < 1) return ICP_ERR_INVALID_ARGUMENT;
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    for (int k = 0; k < cap; ++k) {
        ms_out[k] = k < Profile::ITER_SLOTS ? ctx->prof.ms_iter[k] : 0.0;
        launches_out[k] = k < Profile::ITER_SLOTS ? ctx->prof.launches_iter[k] : 0;
    }
    return ICP_OK;
}

int icp_profile_event_floor(icp_ctx* ctx, int32_t samples, double* median_us_out) {
    DeviceGuard device_guard(ctx);
    if (!ctx || !median_us_out || samples < 1 || samples > 4096) return ICP_ERR_INVALID_ARGUMENT;
    hipEvent_t a, b;
    ICP_HIP(ctx, hipEventCreateWithFlags(&a, hipEventDisableSystemFence));  // (the flavour prof_begin uses)
    ICP_HIP(ctx, hipEventCreateWithFlags(&b, hipEventDisableSystemFence));
    std::vector<float> us;
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < samples + 8; ++k) {
        // a busy predecessor and a successor on the stream, as the bracketed launch of a registration has them; the kernel
        // in the bracket spins for 20 us of the device's wall clock: what the pair measures beyond that is what it adds
        hipLaunchKernelGGL(k_event_floor, dim3(1), dim3(64), 0, ctx->stream, 1000ll);
        (void)hipEventRecord(a, ctx->stream);
        hipLaunchKernelGGL(k_event_floor, dim3(1), dim3(64), 0, ctx->stream, 2000ll);
        (void)hipEventRecord(b, ctx->stream);
        hipLaunchKernelGGL(k_event_floor, dim3(1), dim3(64), 0, ctx->stream, 1000ll);
        ICP_HIP(ctx, hipEventSynchronize(b));
        float ms = 0.f;
        if (k >= 8 && hipEventElapsedTime(&ms, a, b) == hipSuccess) us.push_back(ms * 1000.f - 20.0f);
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (us.empty()) return fail(ctx, ICP_ERR_HIP, "event timing failed");
    std::sort(us.begin(), us.end());
    *median_us_out = us[us.size() / 2];
    return ICP_OK;
}

int icp_profile_read(icp_ctx* ctx, double* search_ms_out, int64_t* search_launches_out, double* reduce_ms_out,
                     double* normals_ms_out) {
    DeviceGuard device_guard(ctx, false);  // (beside a map update on its own stream)
    if (!ctx) return ICP_ERR_INVALID_ARGUMENT;
    ICP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    if (search_ms_out) *search_ms_out = ctx->prof.ms[0];
    if (search_launches_out) *search_launches_out = ctx->prof.launches[0];
    if (reduce_ms_out) *reduce_ms_out = ctx->prof.ms[1];
    if (normals_ms_out) *normals_ms_out = ctx->prof.ms[2];
    return ICP_OK;
}

}  // extern "C"
