// capi_streams.hip -- the frame-stream scheduler of the device entry points: which internal streams share no hardware
// queue with each other or the caller's stream, whose turn it is, and the lease of a stream's resource set for one frame in flight.
#include "capi_ctx.h"

namespace mi355i {

// ---- which streams share a hardware queue (see mi355_ctx::cand_st) -------------------------------------------------
__global__ void k_probe_spin(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < (1 << 16) && wall_clock64() - t0 < ticks; i++) __builtin_amdgcn_s_sleep(16);      // (bounded either way)
}
__global__ void k_probe_touch() {}

// `a` spins for 200 us; every stream of `others` gets an empty kernel.  shared[j] = that kernel ended after the spin did,
// i.e. others[j] runs behind `a`: same hardware queue.  (Device time stamps: the host's scheduling does not enter.)
bool probe_queues(mi355_ctx *c, hipStream_t a, const hipStream_t *others, int n, bool *shared)
{
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
    hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, a, (unsigned long long)khz / 5ull);              // 200 us
    if (hipEventRecord(c->ev_probe[mi355_ctx::PIPE_CANDS], a) != hipSuccess) return false;
    for (int j = 0; j < n; j++) {
        hipLaunchKernelGGL(k_probe_touch, dim3(1), dim3(64), 0, others[j]);
        if (hipEventRecord(c->ev_probe[j], others[j]) != hipSuccess) return false;
    }
    if (hipEventSynchronize(c->ev_probe[mi355_ctx::PIPE_CANDS]) != hipSuccess) return false;
    for (int j = 0; j < n; j++) {
        float ms = 0.f;
        if (hipEventSynchronize(c->ev_probe[j]) != hipSuccess || hipEventElapsedTime(&ms, c->ev_probe[mi355_ctx::PIPE_CANDS], c->ev_probe[j]) != hipSuccess) return false;
        shared[j] = ms > -0.1f;           // (not shared: it ended ~190 us BEFORE the spin did)
    }
    return hipGetLastError() == hipSuccess;
}

// The queue classes of the candidate streams (once per context).
bool probe_classes(mi355_ctx *c)
{
    const int N = mi355_ctx::PIPE_CANDS;
    if (!c->cand_st[0]) return false;
    // (frames may still be running on the candidates: the probe must find them idle)
    for (int i = 0; i < N; i++) if (hipStreamSynchronize(c->cand_st[i]) != hipSuccess) return false;
    if (c->n_class >= 0) return true;
    // (a stream's first kernel may take milliseconds -- the runtime binds it to a hardware queue then: not inside a probe)
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_probe_touch, dim3(1), dim3(64), 0, c->cand_st[i]);
    for (int i = 0; i < N; i++) if (hipStreamSynchronize(c->cand_st[i]) != hipSuccess) return false;
    int n_class = 0;
    for (int i = 0; i < N; i++) c->cand_class[i] = -1;
    for (int i = 0; i < N; i++) {
        if (c->cand_class[i] >= 0) continue;
        c->cand_class[i] = n_class;
        hipStream_t others[N]; int idx[N], n = 0; bool shared[N];
        for (int j = i + 1; j < N; j++) if (c->cand_class[j] < 0) { others[n] = c->cand_st[j]; idx[n++] = j; }
        if (n > 0 && !probe_queues(c, c->cand_st[i], others, n, shared)) { for (int j = 0; j < N; j++) c->cand_class[j] = -1; return false; }
        for (int j = 0; j < n; j++) if (shared[j]) c->cand_class[idx[j]] = n_class;
        n_class++;
    }
    c->n_class = n_class;
    return true;
}

// The frame streams for raster frames the caller enqueues on `st`: one candidate of every queue class but st's own (at most
// PIPE_SETS).  Probed once per context (the classes) and once per caller's stream; nullptr = probing failed, fewer than
// two = the ordered pipeline is used instead.
const mi355_ctx::PipeChoice *pipe_streams_for(mi355_ctx *c, hipStream_t st)
{
    for (const auto &pc : c->pipe_choice) if (pc.caller == st) return &pc;
    const int N = mi355_ctx::PIPE_CANDS;
    if (!probe_classes(c)) return nullptr;
    hipStream_t reps[N]; int rep_class[N], n = 0; bool shared[N];
    for (int cl = 0; cl < c->n_class; cl++)
        for (int i = 0; i < N; i++) if (c->cand_class[i] == cl) { reps[n] = c->cand_st[i]; rep_class[n++] = i; break; }
    if (!probe_queues(c, st, reps, n, shared)) return nullptr;
    mi355_ctx::PipeChoice pc; pc.caller = st; pc.n = 0;
    // (how many frames in flight: as many as there are hardware queues besides the caller's -- three with the runtime's default of
    //  four queues, up to PIPE_SETS when the process was started with GPU_MAX_HW_QUEUES=8; MI355_PIPE_SETS caps it)
    static const int cap = [] { const char *v = getenv("MI355_PIPE_SETS"); const int k = v ? atoi(v) : 0; return k >= 1 && k <= (int)mi355_ctx::PIPE_SETS ? k : (int)mi355_ctx::PIPE_SETS; }();
    for (int j = 0; j < n && pc.n < cap; j++) if (!shared[j]) pc.cand[pc.n++] = rep_class[j];
    if (c->pipe_choice.size() >= 16) c->pipe_choice.erase(c->pipe_choice.begin());
    c->pipe_choice.push_back(pc);
    return &c->pipe_choice.back();
}

// The caller's stream sits on a hardware queue of its own (the frame streams were picked so), and all that stream carries for an
// overlapped frame is a wait and a copy: every (n + 1)-th RAYTRACED frame of a caller therefore runs ON the caller's stream itself --
// straight into the caller's buffer, no copy --, beside the n frames on the frame streams: four frames in flight on the runtime's
// four queues instead of three (4 spp 1080p: 966 -> 1 061 fps).  Stream order is the stream's own.
bool direct_turn(mi355_ctx *c, const mi355_ctx::PipeChoice *pc)
{
    c->direct_turn = (c->direct_turn + 1) % (pc->n + 1);
    return c->direct_turn == 0;
}

// One call in flight: resource set k (rasterizer scratch / control block, tile list, camera table), frame stream
// ps, frame buffer fb = pipe_fb[b].  lease_begin orders ps behind the set's last call, if that ran elsewhere (a call of another
// caller's stream, of the ordered pipeline, a counting frame, a batch), and behind the copy that last read the buffer;
// lease_done makes the caller's stream wait for the call's last kernel (`recorded`: that kernel carries ev_tile[k] itself).

// `st` behind `ev` -- unless the event has completed already: hipStreamWaitEvent costs the host ~5 us when it has to put a
// barrier packet into the queue and 0.06 us for a query (scripts/ubench/apicost.hip), and the events the frame streams wait for
// (the copy that last read a buffer two frames ago) have almost always completed
hipError_t wait_unless_done(hipStream_t st, hipEvent_t ev)
{
    const hipError_t q = hipEventQuery(ev);
    if (q == hipSuccess) return hipSuccess;
    (void)hipGetLastError();                  // (hipErrorNotReady is not an error here; it must not be taken for a failed launch later)
    return hipStreamWaitEvent(st, ev, 0);
}

int lease_begin(mi355_ctx *c, const mi355_ctx::PipeChoice *pc, size_t fb_bytes, FrameLease &L)
{
    L.k = c->pipe_turn % pc->n; c->pipe_turn = (L.k + 1) % pc->n;
    L.ps = c->cand_st[pc->cand[L.k]];
    L.b = 2 * L.k + c->fb_turn[L.k]; c->fb_turn[L.k] ^= 1;
    HIP_TRY(c->pipe_fb[L.b].ensure(fb_bytes), -31);
    L.fb = (uint32_t *)c->pipe_fb[L.b].p;
    if (c->ev_tile_set[L.k] && (c->ev_tile_ext[L.k] || c->pipe_st[L.k] != L.ps)) HIP_TRY(wait_unless_done(L.ps, c->ev_tile[L.k]), -40);
    c->pipe_st[L.k] = L.ps;
    if (c->ev_copy_set[L.b]) HIP_TRY(wait_unless_done(L.ps, c->ev_copy[L.b]), -40);
    if (c->ev_light_set) HIP_TRY(wait_unless_done(L.ps, c->ev_light), -40);        // (a shadow map redrawn by mi355_light_update)
    return 0;
}

int lease_done(mi355_ctx *c, const FrameLease &L, hipStream_t st, bool recorded)
{
    if (!recorded) HIP_TRY(hipEventRecord(c->ev_tile[L.k], L.ps), -40);
    c->ev_tile_set[L.k] = true; c->ev_tile_ext[L.k] = false;
    HIP_TRY(hipStreamWaitEvent(st, c->ev_tile[L.k], 0), -40);
    return 0;
}
} // namespace mi355i
