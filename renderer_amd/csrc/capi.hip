// capi.hip -- implementation of include/mi355_render.h: dispatch (options, frame parameters, one frame's launches) and the entry
// points.  The context and what the library's other translation units share: capi_ctx.h; trees: capi_tree.hip; the frame streams:
// capi_streams.hip; statistics, probes and known-answer tests: capi_diag.hip.
//
// There is no CPU rendering path in this library: every mode runs as HIP kernels and every
// entry point fails (negative return + mi355_last_error) when no HIP device is usable.
#include "capi_ctx.h"


// (for the launchers in the other translation units: laps of the calling thread's stopwatch; no-ops unless MI355_HOST_PROF is set)
static thread_local ProfMark t_prof;
extern "C" void mi355i_prof_start(void) { if (g_prof.on) t_prof = ProfMark(); }
extern "C" void mi355i_prof_lap(int i) { t_prof.lap(i); }


namespace mi355i {


int select_device(mi355_ctx *c)
{
    HIP_TRY(hipSetDevice(c->device), -10);
    return 0;
}


int count_rows(const mi355_opts &o)
{
    if (o.band_count <= 1 || o.band_rows <= 0) return o.height;
    int n = 0;
    for (int y = 0; y < o.height; y++)
        if ((y / o.band_rows) % o.band_count == o.band_index) n++;
    return n;
}

int validate_opts(const mi355_opts &o, int mode)
{
    if (o.width <= 0 || o.height <= 0 || o.width > 16384 || o.height > 16384) return fail(-20, "bad frame size %dx%d", o.width, o.height);
    if (o.screen_dist <= 0) return fail(-20, "bad screen_dist %d", o.screen_dist);
    if (mode >= MI355_MODE_RAYTRACE && (o.max_ray_depth < 1 || o.max_ray_depth > MI_MAX_DEPTH))
        return fail(-20, "max_ray_depth %d outside 1..%d", o.max_ray_depth, MI_MAX_DEPTH);
    if (o.band_count > 1 && (o.band_rows <= 0 || o.band_index < 0 || o.band_index >= o.band_count))
        return fail(-20, "bad band sharding rows=%d index=%d count=%d", o.band_rows, o.band_index, o.band_count);
    if (o.shadowmap_size <= 0 || o.shadowmap_size > 16384) return fail(-20, "bad shadowmap_size %d", o.shadowmap_size);
    if (mode >= MI355_MODE_RAYTRACE && o.ambient_occlusion && (o.ao_samples < 1 || o.ao_samples > 4096 || !(o.ao_range > 0.f)))
        return fail(-20, "ambient occlusion: ao_samples %d outside 1..4096 or ao_range %g not positive", o.ao_samples, (double)o.ao_range);
    return 0;
}

// Dispenser order of the 8x8 pixel tiles of a raytraced frame: nearest to the screen centre first.
// The benchmark camera (and any look-at camera) keeps the model around the centre, so the
// expensive tiles are handed out first and the cheap background tiles fill the tail of the launch.
int ensure_tile_order(mi355_ctx *c, const FrameParams &P, const uint32_t **out)
{
    const long long key[6] = {P.W, P.H, P.n_rows, P.band_rows, P.band_index, P.band_count};
    mi355_ctx::TileOrder *slot = nullptr;
    for (auto &o : c->orders)
        if (o.buf.p && !memcmp(key, o.key, sizeof key)) { o.used = ++c->order_clock; *out = (const uint32_t *)o.buf.p; return 0; }
    for (auto &o : c->orders) if (!o.buf.p) { slot = &o; break; }
    if (!slot) {
        // every slot holds another geometry: the least recently used one goes -- frames enqueued on ANY stream may still
        // read it (the device entry points are asynchronous), so the device is drained first; four geometries alternate
        // without ever coming here
        slot = &c->orders[0];
        for (auto &o : c->orders) if (o.used < slot->used) slot = &o;
        HIP_TRY(hipDeviceSynchronize(), -40);
    }
    const int tiles_x = (P.W + 7) >> 3, tiles_y = (P.n_rows + 7) >> 3;
    std::vector<std::pair<float, uint32_t>> t((size_t)tiles_x * tiles_y);
    for (int ty = 0; ty < tiles_y; ty++) {
        const int r = ty * 8 + 4 < P.n_rows ? ty * 8 + 4 : P.n_rows - 1;
        const float y = (float)band_row_to_y(r, P.band_rows, P.band_index, P.band_count) - 0.5f * (float)P.H;
        for (int tx = 0; tx < tiles_x; tx++) {
            const float x = (float)(tx * 8 + 4) - 0.5f * (float)P.W;
            t[(size_t)ty * tiles_x + tx] = {x * x + y * y, (uint32_t)(ty * tiles_x + tx)};
        }
    }
    std::sort(t.begin(), t.end());
    std::vector<uint32_t> order(t.size());
    for (size_t i = 0; i < t.size(); i++) order[i] = t[i].second;
    // (a fresh or drained buffer: nothing reads it, the blocking copy orders itself before every later launch)
    HIP_TRY(slot->buf.upload(order), -31);
    memcpy(slot->key, key, sizeof key);
    slot->used = ++c->order_clock;
    *out = (const uint32_t *)slot->buf.p;
    return 0;
}

int fill_params(mi355_ctx *c, int mode, const mi355_camera *cam, const mi355_light *lights, int n_lights,
                const mi355_opts *o, void *d_out, int pitch_bytes, void *d_outf, FrameParams &P, void *ctrl = nullptr)
{
    if (!ctrl) ctrl = c->ctrl.p;
    memset(&P, 0, sizeof P);
    if (n_lights < 0 || n_lights > MI355_MAX_LIGHTS) return fail(-21, "n_lights %d outside 0..%d", n_lights, MI355_MAX_LIGHTS);
    if (pitch_bytes < o->width * 4 || (pitch_bytes & 3)) return fail(-21, "bad pitch %d for width %d", pitch_bytes, o->width);
    memcpy(P.eye, cam->eye, sizeof P.eye);
    memcpy(P.mv, cam->mv, sizeof P.mv);
    P.n_lights = n_lights;
    for (int i = 0; i < n_lights; i++) {
        memcpy(P.light_pos[i], lights[i].pos, 12);
        memcpy(P.light_ics[i], lights[i].in_camera_space, 12);
        memcpy(P.light_c2l[i], lights[i].camera_to_light, 36);
        P.shadow_map[i] = (const float *)c->smap[i].p;
        if ((mode == MI355_MODE_PHONG_SHADOWMAPS || mode == MI355_MODE_PHONG_SOFTSHADOWMAPS) &&
            (!c->smap[i].p || c->smap_size[i] != o->shadowmap_size))
            return fail(-22, "light %d has no %d^2 shadow map: call mi355_shadowmap_render/_set first", i, o->shadowmap_size);
    }
    P.W = o->width; P.H = o->height; P.SD = o->screen_dist;
    P.max_depth = o->max_ray_depth; P.use_shadows = o->use_shadows; P.use_refl = o->use_reflections;
    P.use_refr = o->use_refractions ? 1 : 0; P.refr_rate = o->refract_rate;
    P.ao = o->ambient_occlusion ? 1 : 0; P.ao_samples = o->ao_samples; P.ao_range = o->ao_range;
    P.aa = mode == MI355_MODE_RAYTRACE_ANTIALIAS;
    P.sm_size = o->shadowmap_size;
    P.refl_rate = o->reflect_rate; P.nudge = o->nudge;
    P.ambient = o->ambient; P.diffuse = o->diffuse; P.specular = o->specular; P.clip_z = o->clip_z;
    P.band_rows = o->band_rows; P.band_index = o->band_index; P.band_count = o->band_count;
    P.compact = (o->band_count > 1 && o->compact_rows) ? 1 : 0;
    P.n_rows = count_rows(*o);
    P.out_rows = (o->band_count > 1 && !P.compact) ? o->height : P.n_rows;
    P.out = (uint32_t *)d_out;
    P.pitch_words = pitch_bytes / 4;
    P.outf = (float *)d_outf;
    P.work_counter = (uint32_t *)((char *)ctrl + MI_CTRL_DISPENSER_OFF);
    P.cams = nullptr; P.n_frames = 1;
    P.counters = (unsigned long long *)((char *)ctrl + 16);
    // tuning knobs (mi355_opts::tune, 0 = default)
    P.raster_stats = o->collect_stats ? 1 : 0;
    const int32_t *t = o->tune;
    const int flags = t[5];
    // defaults: serve transitions when every lane of the wave has finished its ray, take pixels when every lane is
    // free -- a wave then works through one tile in lockstep generations, with the fewest (expensive) transition and
    // refill phases; measured best for single frames and batches alike (profiles/r01_analysis.md)
    P.xmin = t[0] > 0 ? (t[0] > 64 ? 64 : t[0]) : 64;
    P.rmin = t[1] > 0 ? (t[1] > 64 ? 64 : t[1]) : 64;
    P.chunk = t[2] > 0 ? t[2] : 64;
    P.ref_order = (flags & 4) ? 1 : 0;
    P.prof_ordered = (flags & 8) ? 1 : 0;
    P.no_cull = (flags & 16) ? 1 : 0;
    P.no_pipe = (flags & 32) ? 1 : 0;
    // work sharing inside a wave (k_raytrace.hip): on by default with 16 idle lanes as the threshold; tune[6] sets the threshold,
    // flag 256 turns it off; it needs the wave in lockstep (xmin 64)
    P.steal_min = ((flags & 256) || P.xmin < 64) ? 0 : (t[6] > 0 ? (t[6] > 64 ? 64 : t[6]) : 16);
    // A profiler collecting hardware counters runs one kernel at a time (rocprofv3 --pmc): frames that wait for each other
    // across streams gain nothing there and were seen to stall for minutes.  MI355_NO_OVERLAP=1 asks for the same.
    static const int no_overlap = [] {
        auto on = [](const char *v) { return v && *v && strcmp(v, "0") && strcmp(v, "false") && strcmp(v, "False") && strcmp(v, "OFF") && strcmp(v, "off"); };
        return (on(getenv("MI355_NO_OVERLAP")) || on(getenv("ROCPROF_COUNTER_COLLECTION")) || getenv("ROCPROF_COUNTERS") || getenv("ROCPROF_COUNTER_GROUPS")) ? 1 : 0;
    }();
    if (no_overlap) P.no_pipe = 1;
    P.tile_sel = nullptr; P.tile_cnt = nullptr; P.tile_mask = nullptr;
    P.fill_counter = (uint32_t *)((char *)ctrl + MI_CTRL_FILL_OFF); P.fill_first = 0;
    P.blocks_per_cu = t[4] > 0 ? t[4] : 0;
    P.rs_threads = t[3];
    P.mlaa = o->mlaa ? 1 : 0;
    P.rt_keep_prev = nullptr; P.rt_keep_next = nullptr;
    if (P.mlaa) {
        if (o->band_count > 1) return fail(-20, "mlaa works on whole frames: no band sharding (mi355_mgpu_render filters the assembled frame)");
        if ((P.pitch_words & 3) || (o->height & 7) || P.pitch_words < 8 || o->height < 8)
            return fail(-20, "mlaa: pitch / 4 = %d must be a multiple of 4 and the height %d of 8 (MLAA.cc:395-396)", P.pitch_words, o->height);
        if (P.pitch_words > 16384) return fail(-20, "mlaa: surfaces up to 16384 words wide (pitch / 4 = %d)", P.pitch_words);
    }
    P.exact_box = (flags & 1) ? 1 : 0;
    if (!c->boxes_tame) P.exact_box = 1;     // box coordinates outside the filtered test's validated range
    P.wave_prof = nullptr;
    static const bool wavelog = getenv("MI355_WAVELOG") != nullptr;      // (measuring builds of k_raytrace: RT_WAVELOG)
    if (o->collect_stats || wavelog) {
        if (c->wave_prof.ensure((size_t)8 * c->n_cus * 4 * 16 * 8) == hipSuccess) P.wave_prof = (unsigned long long *)c->wave_prof.p;
    }
    P.tile_order = nullptr;
    if (mode >= MI355_MODE_RAYTRACE && !(flags & 2)) {
        if (int r = ensure_tile_order(c, P, &P.tile_order)) return r;
    }
    return 0;
}


int enqueue_frame(mi355_ctx *c, int mode, const FrameParams &P_in, int stats, hipStream_t st, void *ctrl = nullptr, RasterScratch *rs = nullptr,
                  DevBuf *mlaa_scratch = nullptr, uint32_t *const *frame_outs = nullptr, DevBuf *sel = nullptr)
{
    FrameParams P = P_in;
    const bool own_ctrl = ctrl != nullptr;
    {
        // The calls cannot be captured into a HIP graph: they size buffers, order themselves against earlier frames with events
        // recorded outside the capture, and the overlapped paths synchronise when they probe the streams.  (Tried: a captured
        // frame on the caller's stream alone replays a wrong picture.)  Refused rather than drawn wrong.
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs != hipStreamCaptureStatusNone) return fail(-47, "the render calls cannot be captured into a HIP graph (stream %p is capturing)", (void *)st);
    }
    if (!ctrl) { ctrl = c->ctrl.p; c->last_ctrl = ctrl; }
    if (!rs) rs = c->rscratch;
    // raytrace frames also reset the pixel dispenser behind the counters (same memset)
    const bool rt = mode == MI355_MODE_RAYTRACE || mode == MI355_MODE_RAYTRACE_ANTIALIAS;
    // Raytraced frames of the device entry points overlap like the raster frames below: a single 1080p frame ends on the
    // dependent chain of its most expensive tile (0.72 ms, most of the GPU idle; 0.24 ms per frame when eight share a
    // launch), so consecutive frames run on the frame streams, each with a control block and a tile list of its own, into
    // a buffer of the library's that the caller's stream copies out.  Not for counting frames, batches, float output,
    // bands in place (their other rows are not this frame's to write) and the debug profiles.
    if (rt && !own_ctrl && !sel && !stats && !P.cams && !P.no_pipe && !P.outf && !P.wave_prof && c->has_bvh && c->cand_st[0] && P.out_rows > 0 &&
        (P.band_count <= 1 || P.compact)) {
        const mi355_ctx::PipeChoice *pc = pipe_streams_for(c, st);
        if (pc && pc->n >= 2 && !direct_turn(c, pc)) {
            FrameLease fl;
            if (int r = lease_begin(c, pc, (size_t)P.pitch_words * (size_t)P.out_rows * 4, fl)) return r;
            const int k = fl.k;
            HIP_TRY(c->pipe_ctrl[k].ensure(MI_CTRL_BYTES), -31);
            FrameParams Q = P;
            Q.out = fl.fb;
            Q.mlaa = 0;                                   // (the filter runs on the caller's buffer, below)
            // (frames that share the GPU are a throughput problem: more waves per SIMD than a frame alone would take.  Round 2: the
            //  four-wave build, 2640 -> 3170 fps frame by frame; round 3: the three-wave build -- no scratch, two triangles per
            //  step -- 3 650 -> 3 850 fps at 1080p, 925 -> 965 at 4 spp)
            //  -- up to a 1080p frame's 32 400 tiles; a 3840 x 2160 frame's 129 600 still want the four-wave build: 1 500 against 1 360 fps)
            if (Q.blocks_per_cu == 0) Q.blocks_per_cu = ((long long)((P.W + 7) / 8) * ((P.n_rows + 7) / 8) > 65536ll) ? 4 : 3;
            Q.work_counter = (uint32_t *)((char *)c->pipe_ctrl[k].p + MI_CTRL_DISPENSER_OFF);
            Q.fill_counter = (uint32_t *)((char *)c->pipe_ctrl[k].p + MI_CTRL_FILL_OFF);
            Q.counters = (unsigned long long *)((char *)c->pipe_ctrl[k].p + 16);
            if (int r = enqueue_frame(c, mode, Q, 0, fl.ps, c->pipe_ctrl[k].p, nullptr, nullptr, nullptr, &c->pipe_sel[k])) return r;
            if (int r = lease_done(c, fl, st, false)) return r;
            hipError_t ce = mi355i_launch_frame_copy(P.out, Q.out, P.W, P.out_rows, P.pitch_words, st, c->ev_copy[fl.b]);
            if (ce != hipSuccess) return fail(-43, "kernel launch failed: %s", hipGetErrorString(ce));
            c->ev_copy_set[fl.b] = true;
            c->last_ctrl = c->pipe_ctrl[k].p;
            c->last_stats = false;
            if (P.mlaa) {
                HIP_TRY(c->mlaa.ensure((size_t)P.pitch_words * P.H * 4), -31);
                hipError_t me = mi355i_launch_mlaa(P.out, (uint32_t *)c->mlaa.p, P.pitch_words, P.H, st);
                if (me != hipSuccess) return fail(-43, "MLAA launch failed: %s", hipGetErrorString(me));
            }
            return 0;
        }
        if (pc && pc->n >= 2) {
            // this frame's turn on the caller's stream (direct_turn): it shares the GPU like the others -- the three-wave build --, and it
            // has a control block and a tile list of its own
            FrameParams Q = P;
            if (Q.blocks_per_cu == 0) Q.blocks_per_cu = ((long long)((P.W + 7) / 8) * ((P.n_rows + 7) / 8) > 65536ll) ? 4 : 3;
            HIP_TRY(c->direct_ctrl.ensure(MI_CTRL_BYTES), -31);
            if (!c->ev_direct) HIP_TRY(hipEventCreateWithFlags(&c->ev_direct, hipEventDisableTiming), -11);
            if (c->ev_direct_set) HIP_TRY(wait_unless_done(st, c->ev_direct), -40);
            Q.work_counter = (uint32_t *)((char *)c->direct_ctrl.p + MI_CTRL_DISPENSER_OFF);
            Q.fill_counter = (uint32_t *)((char *)c->direct_ctrl.p + MI_CTRL_FILL_OFF);
            Q.counters = (unsigned long long *)((char *)c->direct_ctrl.p + 16);
            if (int r = enqueue_frame(c, mode, Q, 0, st, c->direct_ctrl.p, nullptr, nullptr, nullptr, &c->direct_sel)) return r;
            HIP_TRY(hipEventRecord(c->ev_direct, st), -40);
            c->ev_direct_set = true;
            c->last_ctrl = c->direct_ctrl.p;
            c->last_stats = false;
            return 0;
        }
    }
    // (a raster frame that does not count zeroes its control block in its first kernel: one launch less per frame, ~4.7 us)
    const bool raster_self_clear = mode >= MI355_MODE_AMBIENT && mode <= MI355_MODE_PHONG_SOFTSHADOWMAPS && !stats && !P.cams;
    if (!raster_self_clear) HIP_TRY(hipMemsetAsync(ctrl, 0, rt ? MI_CTRL_BYTES : 16 + sizeof(unsigned long long) * CS_COUNT, st), -40);
    if (stats)   // the two "min" time stamps start at all-ones
        HIP_TRY(hipMemsetAsync((char *)ctrl + 16 + sizeof(unsigned long long) * CS_TIME0, 0xff, 2 * sizeof(unsigned long long), st), -40);
    c->last_stats = stats != 0;
    hipError_t e = hipSuccess;
    switch (mode) {
    case MI355_MODE_POINTS: e = mi355i_launch_points(&c->dev, &P, 0, st); break;
    case MI355_MODE_POINTS_FROM_TRIANGLES: e = mi355i_launch_points(&c->dev, &P, 1, st); break;
    case MI355_MODE_AMBIENT: case MI355_MODE_GOURAUD: case MI355_MODE_PHONG:
    case MI355_MODE_PHONG_SHADOWMAPS: case MI355_MODE_PHONG_SOFTSHADOWMAPS: {
        // (a shadow map being redrawn by mi355_light_update: the frame follows the redraw whatever stream it is on)
        if (c->ev_light_set && (mode == MI355_MODE_PHONG_SHADOWMAPS || mode == MI355_MODE_PHONG_SOFTSHADOWMAPS)) HIP_TRY(hipStreamWaitEvent(st, c->ev_light, 0), -40);
        const mi355_ctx::PipeChoice *pc = nullptr;
        if (raster_self_clear && rs == c->rscratch && c->pre && !P.no_pipe && c->cand_st[0] && P.out_rows > 0) pc = pipe_streams_for(c, st);
        // (no turn on the caller's stream for raster frames -- measured: 26.1 k -> 22.4 k fps.  Their kernels are short, and the copies
        //  of the frames behind a frame that occupies the caller's stream wait for it, and with them the frame streams' buffers.)
        if (pc && pc->n >= 2) {
            // overlapped: the whole frame on one of the frame streams, into a frame buffer of the library's; `st` waits for the
            // tile kernel and copies the frame to the caller's buffer
            FrameLease fl;
            if (int r = lease_begin(c, pc, (size_t)P.pitch_words * (size_t)P.out_rows * 4, fl)) return r;
            mi355i_prof_lap(1);
            FrameParams Q = P;
            Q.out = fl.fb;
            if (Q.wave_prof) { static unsigned log_seq = 0; Q.wave_prof += (size_t)(log_seq++ & 3u) * 2048u * 16u; c->last_blocks = 8192; }      // (RS_TILELOG builds: the last four frames side by side)
            e = mi355i_launch_raster_overlapped(&c->dev, &Q, mode, c->rs_pipe[fl.k], fl.ps, c->ev_tile[fl.k]);
            if (e != hipSuccess) {
                // (the turn counters have advanced and part of the frame may be on the frame stream: the set's event is recorded behind
                //  whatever got there, so that the next lease of this set waits for it like for any other frame)
                (void)hipGetLastError();
                (void)lease_done(c, fl, st, false);
                break;
            }
            if (int r = lease_done(c, fl, st, true)) return r;       // (ev_tile is the tile kernel's own completion signal)
            mi355i_prof_lap(5);
            e = mi355i_launch_frame_copy(P.out, Q.out, P.W, P.out_rows, P.pitch_words, st, c->ev_copy[fl.b]);
            mi355i_prof_lap(6);
            if (e == hipSuccess) c->ev_copy_set[fl.b] = true;
            break;
        }
        // (a frame outside the pipeline -- counting frames -- first lets the pipelined frames on other streams finish with
        //  the scratch it is about to use)
        if (rs == c->rscratch) for (int k = 0; k < mi355_ctx::PIPE_SETS; k++) if (c->ev_tile_set[k]) HIP_TRY(hipStreamWaitEvent(st, c->ev_tile[k], 0), -40);
        e = mi355i_launch_raster(&c->dev, &P, mode, rs, st);
        // (... and the next pipelined frame that takes this scratch set waits for this one)
        if (e == hipSuccess && rs == c->rscratch && c->pre) { HIP_TRY(hipEventRecord(c->ev_tile[0], st), -40); c->ev_tile_set[0] = true; c->ev_tile_ext[0] = true; }
        break;
    }
    case MI355_MODE_RAYTRACE: case MI355_MODE_RAYTRACE_ANTIALIAS: {
        if (!c->has_bvh) return fail(-41, "raytrace modes need mi355_scene_set_bvh first");
        int ordered = ((!stats || P.prof_ordered) && !P.ref_order && c->dev.ordered_ok) ? 1 : 0;
        // More waves per SIMD pay when the launch is long enough to be throughput bound (batches, 4K, 4 spp: three waves
        // +10-18 %, four another +6 %); a single 1080p frame is bound by its slowest tiles and runs fastest with two
        // (measured, profiles/).  A build is only used if that many blocks of it fit a CU (registers, LDS stack).
        const int batch = (P.n_frames > 1 && P.cams) ? 1 : 0;
        const int ext = (P.use_refr || P.ao) ? 1 : 0;         // the build with refractions and ray-cast ambient occlusion
        // (the request mapped onto a build that exists -- k_raytrace.hip: pick_kernel; fallbacks and measuring builds come with two
        //  waves per SIMD only)
        int waves = 2;
        mi355i_raytrace_variant(stats, &P.exact_box, &ordered, &waves, ext, batch);
        if (ext && (stats || batch)) return fail(-41, "refractions / ray-cast ambient occlusion: single frames without collect_stats only");
        if (batch && !(ordered && !stats)) return fail(-41, "batched frames need the ordered walk (checked tree, no collect_stats, no reference-order flag)");
        const long long work_tiles = (((long long)P.W + 7) / 8) * (((long long)P.n_rows + 7) / 8) * (P.aa ? 4 : 1) * P.n_frames;
        // (LDS rows the launch asks for: three colour rows per depth level, the tree's stack rows for the ordered walk, and a 4 spp
        //  frame's three rows of pixel sums behind the kernel's own)
        const int stack_rows = 3 * P.max_depth + (ordered ? (int)c->dev.stack_depth + (P.aa ? 3 : 0) : 0);
        if (ordered && !stats && !ext && !P.exact_box) {
            // (round 5: single frames queue their leaves' triangles -- k_raytrace.hip, DEFER -- and are never faster on the four-wave build:
            //  4K 0.912 ms on three waves against 0.935, 4 spp 1080p 1.85 against 2.25; batches keep it)
            // (a single frame takes the four-wave build only when it is asked for: frames that share the GPU -- the overlapped device path)
            for (int w = (batch || P.blocks_per_cu >= 4) ? 4 : 3; w >= 3; w--) {
                // (round 3, with work shared inside the waves: a single 1080p frame is 3 % faster on the three-wave build than on
                //  the two-wave one -- 0.649 against 0.668 ms -- so the bar for three waves is half of what it is for four)
                const bool wanted = P.blocks_per_cu == 0 ? work_tiles >= (w == 3 ? 10ll : 20ll) * w * c->n_cus * 4 : P.blocks_per_cu >= w;
                // (a block is one wave, so LDS bounds the waves of a CU one by one: a build is worth its registers while it holds at
                //  least two more waves per CU than the next smaller build would -- a tree two levels too deep for 16 waves of the
                //  four-wave build still runs 15 of them, not 12)
                if (wanted && mi355i_raytrace_waves_per_cu(0, 0, 1, w, batch, stack_rows, 0) >= 4 * (w - 1) + 2) { waves = w; break; }
            }
        }
        int per_cu = mi355i_raytrace_waves_per_cu(stats, P.exact_box, ordered, waves, batch, stack_rows, ext);      // waves
        if (per_cu > 4 * waves) per_cu = 4 * waves;
        if (P.blocks_per_cu > 0 && 4 * P.blocks_per_cu < per_cu) per_cu = 4 * P.blocks_per_cu;
        int n_blocks = per_cu * c->n_cus;                   // waves of the launch (mi355i_launch_raytrace: one per block)
        const long long lanes_needed = ((long long)P.W * P.n_rows * P.n_frames + 63) / 64;
        if (n_blocks > lanes_needed) n_blocks = (int)(lanes_needed > 0 ? lanes_needed : 1);
        c->last_blocks = n_blocks;
        if (P.wave_prof) HIP_TRY(hipMemsetAsync(c->wave_prof.p, 0, (size_t)n_blocks * 16 * 8, st), -40);
        // Tiles no camera ray can hit anything in are not handed out at all (they are most of the frame: a tile costs a
        // dispenser round trip, 64 primary rays and a shading phase even when it is background).  Not for counting frames
        // (they count the reference's rays), bands that cut through tile rows, and trees that failed the checks.
        const long long n_tiles = (((long long)P.W + 7) / 8) * (((long long)P.n_rows + 7) / 8);
        if (ordered && !stats && (P.band_count <= 1 || (P.band_rows > 0 && P.band_rows % 8 == 0)) && c->n_cull_boxes > 0 && !P.no_cull && n_tiles <= MI_CULL_MAX_TILES) {
            DevBuf *buf = sel ? sel : &c->tile_sel;
            // [512 B: the frames' counts][the frames' tile lists][the frames' masks]
            const size_t list_bytes = (size_t)P.n_frames * (size_t)n_tiles * 4, mask_words = (size_t)((n_tiles + 31) / 32);
            HIP_TRY(buf->ensure(512 + list_bytes + (size_t)P.n_frames * mask_words * 4 + 16), -31);
            P.tile_cnt = (const uint32_t *)buf->p;
            P.tile_sel = (const uint32_t *)((char *)buf->p + 512);
            // (the background of the tiles that are not traced: written by the selection kernel before anything is traced -- or, P.fill_first
            //  set: a frame that crosses PCIe as it is written, by waves of the tracing kernel while the others trace)
            uint32_t *gmask = P.fill_first > 0 ? (uint32_t *)((char *)buf->p + 512 + list_bytes) : nullptr;
            P.tile_mask = gmask;
            if ((e = mi355i_launch_tile_select(&P, (const float4 *)c->cull_boxes.p, c->n_cull_boxes, P.tile_order, (uint32_t *)((char *)buf->p + 512),
                                               (uint32_t *)buf->p, gmask, st)) != hipSuccess) {
                // (the selection could not be launched: the frame is traced without it -- every tile handed out, same pixels)
                (void)hipGetLastError();
                P.tile_cnt = nullptr; P.tile_sel = nullptr; P.tile_mask = nullptr;
                if (P.rt_keep_prev) return fail(-43, "kernel launch failed: %s (tile selection of a frame that keeps its canvas)", hipGetErrorString(e));
            }
        } else if (P.rt_keep_prev)
            return fail(-41, "keep_canvas: this raytraced frame cannot be culled tile by tile (checked before the call: a bug)");
        e = mi355i_launch_raytrace(&c->dev, &P, stats, P.exact_box, ordered, waves, batch, ext, stack_rows, n_blocks, st);
        break;
    }
    case MI355_MODE_LINES:
        // Scene::renderWireframe (Rasterizers.cc:117-187).  Synchronises the stream once inside (k_wire.hip).
        if (!mi355i_wireframe_fits(P.W, P.H, c->dev.n_tris))
            return fail(-42, "mode 3 (wireframe): frames up to 4095 x 4095 with at most 2^23 pixels and scenes up to 349525 triangles");
        HIP_TRY(hipDeviceSynchronize(), -40);          // (one set of key buffers per context: no two wireframe frames in flight)
        if (!c->wscratch) c->wscratch = mi355i_wire_scratch_create();
        e = mi355i_launch_wireframe(&c->dev, &P, c->wscratch, st);
        break;
    default:
        return fail(-42, "unknown render mode %d", mode);
    }
    if (e != hipSuccess) return fail(-43, "kernel launch failed: %s", hipGetErrorString(e));
    if (P.mlaa) {
        DevBuf &scratch = mlaa_scratch ? *mlaa_scratch : c->mlaa;
        HIP_TRY(scratch.ensure((size_t)P.pitch_words * P.H * 4), -31);
        const int nf = (P.n_frames > 1 && P.cams) ? P.n_frames : 1;
        for (int f = 0; f < nf; f++) {
            uint32_t *px = nf > 1 ? frame_outs[f] : P.out;
            if ((e = mi355i_launch_mlaa(px, (uint32_t *)scratch.p, P.pitch_words, P.H, st)) != hipSuccess)
                return fail(-43, "MLAA launch failed: %s", hipGetErrorString(e));
        }
    }
    return 0;
}

} // namespace mi355i

extern "C" {

int mi355_abi_version(void) { return MI355_ABI_VERSION; }

// (for the library's other translation units: mgpu.hip)
int mi355i_set_error(int code, const char *text) { g_err = text ? text : ""; return code; }

const char *mi355_last_error(void) { return g_err.c_str(); }

int mi355_init(int n_devices_requested, int *n_devices_out)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        if (n_devices_out) *n_devices_out = 0;
        return fail(-1, "no HIP device available (%s); this library has no CPU path", e != hipSuccess ? hipGetErrorString(e) : "0 devices");
    }
    if (n_devices_requested > 0 && n_devices_requested > n)
        return fail(-2, "%d devices requested, %d present", n_devices_requested, n);
    if (n_devices_out) *n_devices_out = n_devices_requested > 0 ? n_devices_requested : n;
    return 0;
}

void mi355_default_opts(mi355_opts *o, int width, int height)
{
    memset(o, 0, sizeof *o);
    o->width = width; o->height = height; o->screen_dist = 2 * height;       // Defines.h:26-28
    o->max_ray_depth = 3; o->use_shadows = 1; o->use_reflections = 1;         // Raytracer.cc:56,63,67
    o->use_refractions = 0; o->refract_rate = 0.58f;                          // Raytracer.cc:72-73
    o->ambient_occlusion = 0; o->ao_samples = 32; o->ao_range = 0.15f;        // Raytracer.cc:76-80
    o->shadowmap_size = 1024;                                                 // Defines.h:25
    o->reflect_rate = 0.375f; o->nudge = 1e-5f;                               // Raytracer.cc:68,59
    o->ambient = 96.f; o->diffuse = 128.f; o->specular = 192.f;               // Defines.h:30-32
    o->clip_z = 0.2f;                                                         // Rasterizers.cc:39
    o->band_rows = 8; o->band_index = 0; o->band_count = 1; o->compact_rows = 0;     // (8 = one row of the kernels' 8x8 tiles)
}

mi355_ctx *mi355_scene_create(const mi355_scene_desc *d, int device)
{
    if (!d || !d->vertex_pos || !d->vertex_normal || !d->vertex_ao || !d->tri_index || !d->tri_center ||
        !d->tri_normal || !d->tri_colorf || !d->tri_color32 || !d->tri_two_sided || !d->tri_d || !d->tri_e) {
        fail(-3, "mi355_scene_create: null array in scene descriptor");
        return nullptr;
    }
    int ndev = 0;
    if (mi355_init(0, &ndev) != 0) return nullptr;
    if (device < 0 || device >= ndev) { fail(-3, "device %d out of range (have %d)", device, ndev); return nullptr; }
    for (uint32_t i = 0; i < d->n_triangles * 3u; i++)
        if (d->tri_index[i] < 0 || (uint32_t)d->tri_index[i] >= d->n_vertices) {
            fail(-3, "triangle %u references vertex %d of %u", i / 3, d->tri_index[i], d->n_vertices);
            return nullptr;
        }
    mi355_ctx *c = new mi355_ctx;
    c->device = device;
    device_use(device, +1);
    { std::lock_guard<std::mutex> lk(g_dev_mu); g_ctx_list.push_back(c); }
    auto bail = [&](const char *what, hipError_t e) { fail(-4, "%s: %s", what, hipGetErrorString(e)); mi355_scene_destroy(c); return (mi355_ctx *)nullptr; };
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const uint32_t V = d->n_vertices, T = d->n_triangles;
    c->nV = V; c->nT = T;
    c->vpos.assign(d->vertex_pos, d->vertex_pos + 3 * (size_t)V);
    c->vnrm.assign(d->vertex_normal, d->vertex_normal + 3 * (size_t)V);
    c->vao.assign(d->vertex_ao, d->vertex_ao + V);
    c->tidx.assign(d->tri_index, d->tri_index + 3 * (size_t)T);
    c->tcenter.assign(d->tri_center, d->tri_center + 3 * (size_t)T);
    c->tnormal.assign(d->tri_normal, d->tri_normal + 3 * (size_t)T);
    c->tcolorf.assign(d->tri_colorf, d->tri_colorf + 3 * (size_t)T);
    c->tcolor32.assign(d->tri_color32, d->tri_color32 + T);
    c->ttwo.assign(d->tri_two_sided, d->tri_two_sided + T);
    c->td.assign(d->tri_d, d->tri_d + 4 * (size_t)T);
    c->te.assign(d->tri_e, d->tri_e + 9 * (size_t)T);
    // rasterizer streams, input order
    std::vector<float4> rs_tri((size_t)T * 2), rs_col(T), rs_vert((size_t)V * 2);
    std::vector<uint4> rs_idx(T);
    for (uint32_t t = 0; t < T; t++) {
        rs_tri[(size_t)t * 2] = make_float4(c->tcenter[3 * t], c->tcenter[3 * t + 1], c->tcenter[3 * t + 2], u2f(c->ttwo[t] ? 1u : 0u));
        rs_tri[(size_t)t * 2 + 1] = make_float4(c->tnormal[3 * t], c->tnormal[3 * t + 1], c->tnormal[3 * t + 2], u2f(c->tcolor32[t]));
        rs_col[t] = make_float4(c->tcolorf[3 * t], c->tcolorf[3 * t + 1], c->tcolorf[3 * t + 2], 0.f);
        rs_idx[t] = make_uint4((uint32_t)c->tidx[3 * t], (uint32_t)c->tidx[3 * t + 1], (uint32_t)c->tidx[3 * t + 2], 0u);
    }
    for (uint32_t v = 0; v < V; v++) {
        rs_vert[(size_t)v * 2] = make_float4(c->vpos[3 * v], c->vpos[3 * v + 1], c->vpos[3 * v + 2], (float)c->vao[v]);
        rs_vert[(size_t)v * 2 + 1] = make_float4(c->vnrm[3 * v], c->vnrm[3 * v + 1], c->vnrm[3 * v + 2], 0.f);
    }
    if ((e = c->rs_tri.upload(rs_tri)) != hipSuccess) return bail("upload", e);
    if ((e = c->rs_col.upload(rs_col)) != hipSuccess) return bail("upload", e);
    if ((e = c->rs_idx.upload(rs_idx)) != hipSuccess) return bail("upload", e);
    if ((e = c->rs_vert.upload(rs_vert)) != hipSuccess) return bail("upload", e);
    if ((e = c->ctrl.ensure(MI_CTRL_BYTES)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemset(c->ctrl.p, 0, c->ctrl.bytes)) != hipSuccess) return bail("hipMemset", e);
    if ((e = hipStreamCreate(&c->stream)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return bail("hipEventCreate", e);
    c->rscratch = mi355i_raster_scratch_create();
    // the overlapped frames of the device entry points: a rasterizer scratch set per frame stream and the event of each set's last
    // kernel (if any of this fails the frames simply are not overlapped)
    c->rs_pipe[0] = c->rscratch;
    for (int k = 1; k < mi355_ctx::PIPE_SETS; k++) c->rs_pipe[k] = mi355i_raster_scratch_create();
    c->pre = true;
    for (int k = 1; k < mi355_ctx::PIPE_SETS; k++) c->pre = c->pre && c->rs_pipe[k] != nullptr;      // (pipe_streams_for may pick any of the sets)
    for (int k = 0; c->pre && k < mi355_ctx::PIPE_SETS; k++) c->pre = hipEventCreateWithFlags(&c->ev_tile[k], hipEventDisableTiming) == hipSuccess;
    if (c->pre) {
        // the frames' own streams have a priority of their own: the runtime hands out hardware queues per priority, and a
        // frame stream that shares a queue with the caller's stream would sit behind that stream's waits (measured: every
        // third frame stalled for a whole frame time)
        bool ok = true;
        for (int k = 0; k < mi355_ctx::PIPE_CANDS; k++) ok = ok && hipStreamCreateWithFlags(&c->cand_st[k], hipStreamNonBlocking) == hipSuccess;
        for (int k = 0; k < 2 * mi355_ctx::PIPE_SETS; k++) ok = ok && hipEventCreateWithFlags(&c->ev_copy[k], hipEventDisableTiming) == hipSuccess;
        for (int k = 0; k <= mi355_ctx::PIPE_CANDS; k++) ok = ok && hipEventCreate(&c->ev_probe[k]) == hipSuccess;
        if (!ok) for (int k = 0; k < mi355_ctx::PIPE_CANDS; k++) { if (c->cand_st[k]) (void)hipStreamDestroy(c->cand_st[k]); c->cand_st[k] = nullptr; }
    }
    c->dev.rs_tri = (const float4 *)c->rs_tri.p;
    c->dev.rs_col = (const float4 *)c->rs_col.p;
    c->dev.rs_idx = (const uint4 *)c->rs_idx.p;
    c->dev.rs_vert = (const float4 *)c->rs_vert.p;
    c->dev.n_tris = T; c->dev.n_verts = V;
    return c;
}

void mi355_scene_destroy(mi355_ctx *c)
{
    if (!c) return;
    device_use(c->device, -1);
    { std::lock_guard<std::mutex> lk(g_dev_mu); for (size_t i = 0; i < g_ctx_list.size(); i++) if (g_ctx_list[i] == c) { g_ctx_list.erase(g_ctx_list.begin() + (long)i); break; } }
    (void)hipSetDevice(c->device);
    // (nothing of this context may be in flight on any stream -- its own, the internal frame streams, a caller's -- while its
    //  buffers go away)
    (void)hipDeviceSynchronize();
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : {&c->walk, &c->tri_edge, &c->tri_shade, &c->rs_tri, &c->rs_col, &c->rs_idx,
                      &c->rs_vert, &c->ctrl, &c->fb, &c->fbf, &c->mlaa, &c->cam_table, &c->wave_prof, &c->bvh_prim, &c->bvh_list[0], &c->bvh_list[1],
                      &c->bvh_lvl[0], &c->bvh_lvl[1], &c->bvh_tree, &c->bvh_cnt, &c->bvh_big[0], &c->bvh_big[1], &c->bvh_task[0], &c->bvh_task[1],
                      &c->bvh_gthr[0], &c->bvh_gthr[1], &c->bvh_gbin, &c->bvh_tcnt, &c->bvh_choff, &c->bvh_num[0], &c->bvh_num[1], &c->bvh_num[2],
                      &c->bvh_num[3], &c->bvh_num[4], &c->bvh_out, &c->bvh_in_td, &c->bvh_in_te, &c->cull_boxes, &c->tile_sel})
        b->release();
    // (every resource set of the frame streams, however many PIPE_SETS there are)
    for (DevBuf &b : c->pipe_fb) b.release();
    for (int k = 0; k < mi355_ctx::PIPE_SETS; k++) { c->pipe_ctrl[k].release(); c->pipe_sel[k].release(); c->pipe_cam[k].release(); }
    for (PinBuf *b : {&c->pin_walk, &c->pin_edge, &c->pin_shade, &c->pin_tree, &c->pin_list, &c->pin_ctl, &c->pin_counters}) b->release();
    for (auto &m : c->smap) m.release();
    for (auto &o : c->orders) o.buf.release();
    for (auto &a : c->slot) {
        if (a.st) (void)hipStreamSynchronize(a.st);
        a.ctrl.release(); a.fb.release(); a.mlaa.release(); a.sel.release(); a.pin.release(); a.pin_counters.release();
        if (a.rs) mi355i_raster_scratch_destroy(a.rs);
        if (a.ev0) (void)hipEventDestroy(a.ev0);
        if (a.ev1) (void)hipEventDestroy(a.ev1);
        if (a.st && a.st_owned) (void)hipStreamDestroy(a.st);
    }
    for (auto &h : c->host_reg) if (h.p) { const hipError_t ue = hipHostUnregister(h.p); host_trace("destroy ctx %p: unregister %p + %zu -> %d", (void *)c, (void *)h.p, h.bytes, (int)ue); }
    c->direct_ctrl.release(); c->direct_sel.release();
    for (auto &cv : c->canvas) { cv.mask[0].release(); cv.mask[1].release(); }     // (cv.ev is borrowed)
    if (c->ev_direct) (void)hipEventDestroy(c->ev_direct);
    if (c->rs_light) mi355i_raster_scratch_destroy(c->rs_light);
    if (c->ev_light) (void)hipEventDestroy(c->ev_light);
    if (c->rscratch) mi355i_raster_scratch_destroy(c->rscratch);
    if (c->wscratch) mi355i_wire_scratch_destroy(c->wscratch);
    for (int k = 1; k < mi355_ctx::PIPE_SETS; k++) if (c->rs_pipe[k]) mi355i_raster_scratch_destroy(c->rs_pipe[k]);
    for (int k = 0; k < mi355_ctx::PIPE_SETS; k++) if (c->ev_tile[k]) (void)hipEventDestroy(c->ev_tile[k]);
    for (int k = 0; k < mi355_ctx::PIPE_CANDS; k++) if (c->cand_st[k]) { (void)hipStreamSynchronize(c->cand_st[k]); (void)hipStreamDestroy(c->cand_st[k]); }
    for (int k = 0; k < 2 * mi355_ctx::PIPE_SETS; k++) if (c->ev_copy[k]) (void)hipEventDestroy(c->ev_copy[k]);
    for (int k = 0; k <= mi355_ctx::PIPE_CANDS; k++) if (c->ev_probe[k]) (void)hipEventDestroy(c->ev_probe[k]);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
int mi355_shadowmap_set(mi355_ctx *c, int slot, const float *map, int size)
{
    if (!c || !map) return fail(-3, "mi355_shadowmap_set: null argument");
    if (slot < 0 || slot >= MI355_MAX_LIGHTS || size <= 0) return fail(-3, "bad light slot %d / size %d", slot, size);
    if (int r = select_device(c)) return r;
    // frames enqueued on any stream (the device entry points do not synchronise) may still read this light's map
    HIP_TRY(hipDeviceSynchronize(), -40);
    HIP_TRY(c->smap[slot].ensure((size_t)size * size * 4), -31);
    HIP_TRY(hipMemcpy(c->smap[slot].p, map, (size_t)size * size * 4, hipMemcpyHostToDevice), -31);
    c->smap_size[slot] = size;
    return 0;
}

int mi355_shadowmap_render(mi355_ctx *c, int slot, const mi355_light *light, int size, float *out_map)
{
    if (!c || !light) return fail(-3, "mi355_shadowmap_render: null argument");
    if (slot < 0 || slot >= MI355_MAX_LIGHTS || size <= 0 || size > 16384) return fail(-3, "bad light slot %d / size %d", slot, size);
    if (int r = select_device(c)) return r;
    // frames enqueued on any stream (the device entry points do not synchronise) may still read this light's map
    HIP_TRY(hipDeviceSynchronize(), -40);
    HIP_TRY(c->smap[slot].ensure((size_t)size * size * 4), -31);
    for (int attempt = 0;; attempt++) {
        hipError_t e = mi355i_launch_shadowmap(&c->dev, light->pos, light->world_to_light, size, (float *)c->smap[slot].p,
                                               c->rscratch, c->stream);
        if (e != hipSuccess) return fail(-43, "shadow map launch failed: %s", hipGetErrorString(e));
        HIP_TRY(hipStreamSynchronize(c->stream), -40);
        const uint32_t dropped = mi355i_raster_overflow(c->rscratch);
        if (!dropped) break;
        if (attempt < 6 && mi355i_raster_grow(c->rscratch)) continue;      // twice the span buffer, draw again
        return fail(-44, "shadow map span buffer overflowed (%u rows dropped)", dropped);
    }
    c->smap_size[slot] = size;
    if (out_map) host_trace("shadowmap ctx %p: out %p + %zu", (void *)c, (void *)out_map, (size_t)size * size * 4);
    if (out_map) HIP_TRY(hipMemcpy(out_map, c->smap[slot].p, (size_t)size * size * 4, hipMemcpyDeviceToHost), -31);
    return 0;
}

// Light::RenderSceneIntoShadowBuffer for a light that MOVES while frames are being drawn (renderer.cc:410-431: the W / Q keys):
// the map of `slot` is redrawn for a light at `pos`, asynchronously and in the order of the calls on hip_stream -- frames enqueued
// before see the old map, frames enqueued after see the new one -- without synchronising anything.  The light's world-to-light
// basis (Light.cc:173-192: forward = towards the origin, right = forward x zenith, up = right x forward) is computed here from
// the position and returned in *light_out (pos and world_to_light filled in; the camera-space members are the caller's, per
// frame).  A row buffer that turns out too small is reported by the next mi355_fetch_stats (-44; it has grown by then).
int mi355_light_update(mi355_ctx *c, int slot, const float pos[3], int size, mi355_light *light_out, void *hip_stream)
{
    if (!c || !pos) return fail(-3, "mi355_light_update: null argument");
    if (slot < 0 || slot >= MI355_MAX_LIGHTS || size <= 0 || size > 16384) return fail(-3, "bad light slot %d / size %d", slot, size);
    if (int r = select_device(c)) return r;
    hipStream_t st = (hipStream_t)hip_stream;
    mi355_light l;
    memset(&l, 0, sizeof l);
    {
        // (the float operations of Light.cc:173-192 / Camera.cc:24-42, in their order)
        V3h f = {-pos[0], -pos[1], -pos[2]};
        const float fl = lenh(f);
        f = {f.x / fl, f.y / fl, f.z / fl};
        V3h right = crossh(f, V3h{0.f, 0.f, 1.f});
        const float rl = lenh(right);
        right = {right.x / rl, right.y / rl, right.z / rl};
        V3h up = crossh(right, f);
        const float ul = lenh(up);
        up = {up.x / ul, up.y / ul, up.z / ul};
        const float m[9] = {up.x, up.y, up.z, right.x, right.y, right.z, f.x, f.y, f.z};
        memcpy(l.pos, pos, 12);
        memcpy(l.world_to_light, m, 36);
    }
    if (light_out) { memcpy(light_out->pos, l.pos, 12); memcpy(light_out->world_to_light, l.world_to_light, 36); }
    if (c->smap_size[slot] != size || !c->smap[slot].p) {
        // a first map, or another size: nothing in flight may read the buffer that is replaced
        HIP_TRY(hipDeviceSynchronize(), -40);
        HIP_TRY(c->smap[slot].ensure((size_t)size * size * 4), -31);
    }
    if (!c->ev_light) HIP_TRY(hipEventCreateWithFlags(&c->ev_light, hipEventDisableTiming), -11);
    if (!c->rs_light) c->rs_light = mi355i_raster_scratch_create();
    if (!c->rs_light) return fail(-11, "out of memory");
    // behind every frame enqueued so far, wherever it runs (the frames' own streams carry ev_tile; the caller's stream orders itself)
    for (int k = 0; k < mi355_ctx::PIPE_SETS; k++) if (c->ev_tile_set[k]) HIP_TRY(hipStreamWaitEvent(st, c->ev_tile[k], 0), -40);
    for (auto &a : c->slot) if (a.busy && a.ev1) HIP_TRY(hipStreamWaitEvent(st, a.ev1, 0), -40);      // (frames of mi355_render_async still in flight)
    static const bool wavelog = getenv("MI355_WAVELOG") != nullptr;       // (measuring builds: RS_TILELOG)
    if (wavelog && c->wave_prof.ensure((size_t)8 * c->n_cus * 4 * 16 * 8) == hipSuccess) { mi355i_raster_set_log(c->rs_light, (unsigned long long *)c->wave_prof.p); c->last_blocks = 8192; }
    const hipError_t e = mi355i_launch_shadowmap(&c->dev, l.pos, l.world_to_light, size, (float *)c->smap[slot].p, c->rs_light, st);
    if (e != hipSuccess) return fail(-43, "shadow map launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipEventRecord(c->ev_light, st), -40);
    c->ev_light_set = true;
    c->smap_size[slot] = size;
    return 0;
}

int mi355_render_device(mi355_ctx *c, int mode, const mi355_camera *cam, const mi355_light *lights, int n_lights,
                        const mi355_opts *o, void *d_out, int pitch_bytes, void *d_outf, void *hip_stream)
{
    if (!c || !cam || !o || !d_out || (n_lights > 0 && !lights)) return fail(-3, "mi355_render_device: null argument");
    mi355i_prof_start();
    if (int r = validate_opts(*o, mode)) return r;
    if (int r = select_device(c)) return r;
    FrameParams P;
    if (int r = fill_params(c, mode, cam, lights, n_lights, o, d_out, pitch_bytes, d_outf, P)) return r;
    mi355i_prof_lap(0);
    const int r = enqueue_frame(c, mode, P, o->collect_stats, (hipStream_t)hip_stream);
    mi355i_prof_lap(8);
    return r;
}

int mi355_render_batch_device(mi355_ctx *c, int mode, int n_frames, const mi355_camera *cams, const mi355_light *lights,
                              int n_lights, const mi355_opts *o, void *const *d_out, int pitch_bytes, void *const *d_outf,
                              void *hip_stream)
{
    if (!c || !cams || !o || !d_out || (n_lights > 0 && !lights)) return fail(-3, "mi355_render_batch_device: null argument");
    const bool raster = mode >= MI355_MODE_AMBIENT && mode <= MI355_MODE_PHONG_SOFTSHADOWMAPS;
    if (raster) {
        // The tiles of all frames of the batch are work items of ONE set of launches (k_raster.hip); every frame is the
        // frame mi355_render_device produces.
        if (n_frames < 1 || n_frames > MI355_MAX_BATCH) return fail(-21, "n_frames %d outside 1..%d", n_frames, MI355_MAX_BATCH);
        if (o->collect_stats) return fail(-21, "batched frames cannot collect the counters");
        for (int f = 0; f < n_frames; f++) if (!d_out[f]) return fail(-3, "mi355_render_batch_device: frame %d has no output buffer", f);
        if (int r = validate_opts(*o, mode)) return r;
        if (int r = select_device(c)) return r;
        hipStream_t user = (hipStream_t)hip_stream;
        HIP_TRY(hipMemsetAsync(c->ctrl.p, 0, 16 + sizeof(unsigned long long) * CS_COUNT, user), -40);
        std::vector<FrameParams> frames((size_t)n_frames);
        for (int f = 0; f < n_frames; f++)
            if (int r = fill_params(c, mode, &cams[f], lights + (size_t)f * n_lights, n_lights, o, d_out[f], pitch_bytes, nullptr, frames[f])) return r;
        for (int k = 0; k < mi355_ctx::PIPE_SETS; k++) if (c->ev_tile_set[k]) HIP_TRY(hipStreamWaitEvent(user, c->ev_tile[k], 0), -40);   // (pipelined single frames still using the scratch)
        hipError_t e = mi355i_launch_raster_batch(&c->dev, frames.data(), n_frames, mode, c->rscratch, user);
        if (e != hipSuccess) return fail(-43, "kernel launch failed: %s", hipGetErrorString(e));
        if (c->pre) { HIP_TRY(hipEventRecord(c->ev_tile[0], user), -40); c->ev_tile_set[0] = true; c->ev_tile_ext[0] = true; }
        if (frames[0].mlaa) {
            HIP_TRY(c->mlaa.ensure((size_t)frames[0].pitch_words * frames[0].H * 4), -31);
            for (int f = 0; f < n_frames; f++)
                if ((e = mi355i_launch_mlaa((uint32_t *)d_out[f], (uint32_t *)c->mlaa.p, frames[0].pitch_words, frames[0].H, user)) != hipSuccess)
                    return fail(-43, "MLAA launch failed: %s", hipGetErrorString(e));
        }
        c->last_stats = false;
        return 0;
    }
    if (mode != MI355_MODE_RAYTRACE && mode != MI355_MODE_RAYTRACE_ANTIALIAS) return fail(-42, "batched frames: raster and raytrace modes only (got mode %d)", mode);
    if (n_frames < 1 || n_frames > MI355_MAX_BATCH) return fail(-21, "n_frames %d outside 1..%d", n_frames, MI355_MAX_BATCH);
    if (o->collect_stats) return fail(-21, "batched frames cannot collect the traversal counters");
    for (int f = 0; f < n_frames; f++) if (!d_out[f]) return fail(-3, "mi355_render_batch_device: frame %d has no output buffer", f);
    if (n_frames == 1) return mi355_render_device(c, mode, cams, lights, n_lights, o, d_out[0], pitch_bytes, d_outf ? d_outf[0] : nullptr, hip_stream);
    if (o->use_refractions || o->ambient_occlusion) {       // (these builds render single frames)
        for (int f = 0; f < n_frames; f++)
            if (int r = mi355_render_device(c, mode, &cams[f], lights + (size_t)f * n_lights, n_lights, o, d_out[f], pitch_bytes, d_outf ? d_outf[f] : nullptr, hip_stream)) return r;
        return 0;
    }
    if (int r = validate_opts(*o, mode)) return r;
    if (int r = select_device(c)) return r;
    FrameParams P;
    if (int r = fill_params(c, mode, &cams[0], lights, n_lights, o, d_out[0], pitch_bytes, d_outf ? d_outf[0] : nullptr, P)) return r;
    FrameCam tab[MI355_MAX_BATCH];
    memset(tab, 0, sizeof tab);
    for (int f = 0; f < n_frames; f++) {
        for (int k = 0; k < 3; k++) tab[f].eye[k] = cams[f].eye[k];
        for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) tab[f].mv[r][k] = cams[f].mv[3 * r + k];
        for (int i = 0; i < n_lights; i++) for (int k = 0; k < 3; k++) tab[f].light_pos[i][k] = lights[(size_t)f * n_lights + i].pos[k];
        tab[f].out = (uint32_t *)d_out[f];
        tab[f].outf = d_outf ? (float *)d_outf[f] : nullptr;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    // Consecutive batches overlap like consecutive frames (enqueue_frame): a launch of eight 1080p frames ends on ~0.3 ms of
    // a few waves finishing their tiles (11.5 Grays/s at 8 frames per launch, 13.5 at 32: measured), which the head of the
    // next batch fills when it runs on another stream.  The batch renders into a buffer of the library's; the caller's
    // stream copies the frames out in one launch.
    const size_t frame_words = (size_t)P.pitch_words * (size_t)P.out_rows;
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs != hipStreamCaptureStatusNone) return fail(-47, "the render calls cannot be captured into a HIP graph (stream %p is capturing)", (void *)st);
    }
    if (!d_outf && !P.mlaa && !P.no_pipe && !P.wave_prof && c->has_bvh && c->cand_st[0] && P.out_rows > 0 && (P.band_count <= 1 || P.compact) &&
        frame_words * 4 * (size_t)n_frames <= ((size_t)1 << 30) && n_frames <= 64) {
        const mi355_ctx::PipeChoice *pc = pipe_streams_for(c, st);
        if (pc && pc->n >= 2) {
            FrameLease fl;
            if (int r = lease_begin(c, pc, frame_words * 4 * (size_t)n_frames, fl)) return r;
            const int k = fl.k;
            HIP_TRY(c->pipe_ctrl[k].ensure(MI_CTRL_BYTES), -31);
            HIP_TRY(c->pipe_cam[k].ensure(sizeof tab), -31);
            for (int f = 0; f < n_frames; f++) tab[f].out = fl.fb + (size_t)f * frame_words;
            HIP_TRY(hipMemcpyAsync(c->pipe_cam[k].p, tab, sizeof(FrameCam) * (size_t)n_frames, hipMemcpyHostToDevice, fl.ps), -31);
            FrameParams Q = P;
            Q.cams = (const FrameCam *)c->pipe_cam[k].p;
            Q.n_frames = n_frames;
            Q.out = fl.fb;
            Q.work_counter = (uint32_t *)((char *)c->pipe_ctrl[k].p + MI_CTRL_DISPENSER_OFF);
            Q.fill_counter = (uint32_t *)((char *)c->pipe_ctrl[k].p + MI_CTRL_FILL_OFF);
            Q.counters = (unsigned long long *)((char *)c->pipe_ctrl[k].p + 16);
            if (int r = enqueue_frame(c, mode, Q, 0, fl.ps, c->pipe_ctrl[k].p, nullptr, nullptr, nullptr, &c->pipe_sel[k])) return r;
            if (int r = lease_done(c, fl, st, false)) return r;
            hipError_t ce = mi355i_launch_frames_copy(d_out, n_frames, fl.fb, P.W, P.out_rows, P.pitch_words, st, c->ev_copy[fl.b]);
            if (ce != hipSuccess) return fail(-43, "kernel launch failed: %s", hipGetErrorString(ce));
            c->ev_copy_set[fl.b] = true;
            c->last_ctrl = c->pipe_ctrl[k].p;
            c->last_stats = false;
            return 0;
        }
    }
    HIP_TRY(c->cam_table.ensure(sizeof tab), -31);
    // (pageable source: the runtime stages the bytes before returning, so `tab` may go out of scope; the copy is ordered
    //  after the previous launch on this stream, which may still be reading the table)
    HIP_TRY(hipMemcpyAsync(c->cam_table.p, tab, sizeof(FrameCam) * (size_t)n_frames, hipMemcpyHostToDevice, st), -31);
    P.cams = (const FrameCam *)c->cam_table.p;
    P.n_frames = n_frames;
    return enqueue_frame(c, mode, P, 0, st, nullptr, nullptr, nullptr, (uint32_t *const *)d_out);
}

int mi355_mlaa_device(mi355_ctx *c, void *d_xrgb, int pitch_bytes, int height, void *hip_stream)
{
    if (!c || !d_xrgb) return fail(-3, "mi355_mlaa_device: null argument");
    const int pw = pitch_bytes / 4;
    if (pitch_bytes <= 0 || (pitch_bytes & 3) || (pw & 3) || (height & 7) || pw < 8 || height < 8)
        return fail(-20, "mlaa: pitch / 4 = %d must be a multiple of 4 and the height %d of 8 (MLAA.cc:395-396)", pw, height);
    // (k_mlaa_scan keeps the line starts of a row -- or, in its vertical pass, of a column -- in an LDS list sized for 16384 pixels)
    if (pw > 16384 || height > 16384) return fail(-20, "mlaa: surfaces up to 16384 x 16384 words (got %d x %d)", pw, height);
    if (int r = select_device(c)) return r;
    HIP_TRY(c->mlaa.ensure((size_t)pw * height * 4), -31);
    const hipError_t e = mi355i_launch_mlaa((uint32_t *)d_xrgb, (uint32_t *)c->mlaa.p, pw, height, (hipStream_t)hip_stream);
    if (e != hipSuccess) return fail(-43, "MLAA launch failed: %s", hipGetErrorString(e));
    return 0;
}
// frame memory handed out by mi355_host_alloc (process-wide: page-locked for every context)
static std::mutex g_host_alloc_mu;
static std::vector<std::pair<char *, size_t>> g_host_alloc;

static bool host_range_registered(const mi355_ctx *c, const void *p, size_t bytes)
{
    for (const auto &h : c->host_reg)
        if (h.p && (const char *)p >= h.p && (const char *)p + bytes <= h.p + h.bytes) return true;
    std::lock_guard<std::mutex> lk(g_host_alloc_mu);
    for (const auto &h : g_host_alloc)
        if ((const char *)p >= h.first && (const char *)p + bytes <= h.first + h.second) return true;
    return false;
}

// (mgpu.hip: a frame assembled in host memory by something other than mi355_render)
void mi355i_canvases_written(const void *p, size_t bytes) { canvases_written(nullptr, p, bytes); }

void *mi355_host_alloc(size_t bytes)
{
    if (!bytes) { (void)fail(-3, "mi355_host_alloc: zero bytes"); return nullptr; }
    void *p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped);
    host_trace("host_alloc %p + %zu -> %d", p, bytes, (int)e);
    if (e != hipSuccess || !p) { (void)hipGetLastError(); (void)fail(-46, "mi355_host_alloc: hipHostMalloc of %zu bytes: %s", bytes, hipGetErrorString(e)); return nullptr; }
    memset(p, 0, bytes);
    std::lock_guard<std::mutex> lk(g_host_alloc_mu);
    g_host_alloc.emplace_back((char *)p, bytes);
    return p;
}

void mi355_host_free(void *p)
{
    if (!p) return;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(g_host_alloc_mu);
        for (size_t i = 0; i < g_host_alloc.size(); i++)
            if (g_host_alloc[i].first == (char *)p) { bytes = g_host_alloc[i].second; g_host_alloc.erase(g_host_alloc.begin() + (long)i); break; }
    }
    // No copy or kernel may still target the buffer.  What can be on its way into caller's host memory when a call has returned: the
    // frames of mi355_render_async that have not been waited for (mi355_render itself returns when its frame is there).  So: the
    // streams of exactly those slots, in whatever context they are -- not every queue of every device that holds a context (round 5:
    // a hipDeviceSynchronize per device, from every Screen destructor).  A pointer this library did not hand out is released the
    // careful way.
    {
        int cur = 0;
        const bool have_dev = hipGetDevice(&cur) == hipSuccess;
        std::vector<std::pair<int, hipStream_t>> waits;
        bool known = bytes != 0;
        {
            std::lock_guard<std::mutex> lk(g_dev_mu);
            for (mi355_ctx *c : g_ctx_list) {
                // (a canvas whose last frame was known lives in this buffer: whatever comes to lie at the address later is not that frame)
                for (auto &cv : c->canvas)
                    if (cv.host && (!known || ((const char *)cv.host >= (const char *)p && (const char *)cv.host < (const char *)p + bytes))) cv.valid = false;
                for (auto &a : c->slot)
                    if (a.busy && a.st && !a.staged && (!known || ((const char *)a.user >= (const char *)p && (const char *)a.user < (const char *)p + bytes)))
                        waits.emplace_back(c->device, a.st);
            }
        }
        for (auto &w : waits) if (hipSetDevice(w.first) == hipSuccess) (void)hipStreamSynchronize(w.second);
        if (!known && have_dev) {
            bool used[64];
            { std::lock_guard<std::mutex> lk(g_dev_mu); for (int d = 0; d < 64; d++) used[d] = g_dev_use[d] > 0; }
            for (int d = 0; d < 64; d++) if (used[d] && hipSetDevice(d) == hipSuccess) (void)hipDeviceSynchronize();
        }
        if (have_dev) (void)hipSetDevice(cur);
        (void)hipGetLastError();
    }
    host_trace("host_free %p", p);
    (void)hipHostFree(p);
}

int mi355_host_register(mi355_ctx *c, void *p, size_t bytes)
{
    if (!c || !p || !bytes) return fail(-3, "mi355_host_register: null argument");
    if (int r = select_device(c)) return r;
    for (auto &h : c->host_reg)
        if (!h.p) {
            const hipError_t re = hipHostRegister(p, bytes, hipHostRegisterDefault);
            host_trace("register ctx %p: %p + %zu -> %d", (void *)c, p, bytes, (int)re);
            HIP_TRY(re, -46);
            h.p = (char *)p; h.bytes = bytes;
            return 0;
        }
    return fail(-46, "mi355_host_register: all %d slots are in use", (int)(sizeof c->host_reg / sizeof c->host_reg[0]));
}

int mi355_host_unregister(mi355_ctx *c, void *p)
{
    if (!c || !p) return fail(-3, "mi355_host_unregister: null argument");
    if (int r = select_device(c)) return r;
    for (auto &h : c->host_reg)
        if (h.p == (char *)p) {
            for (auto &a : c->slot) if (a.busy && a.st) HIP_TRY(hipStreamSynchronize(a.st), -40);   // no copy may still target it
            HIP_TRY(hipStreamSynchronize(c->stream), -40);
            const hipError_t ue = hipHostUnregister(p);
            host_trace("unregister ctx %p: %p + %zu -> %d", (void *)c, p, h.bytes, (int)ue);
            HIP_TRY(ue, -46);
            h.p = nullptr; h.bytes = 0;
            for (auto &cv : c->canvas) cv.valid = false;
            return 0;
        }
    return fail(-46, "mi355_host_unregister: %p was not registered", p);
}

// ---- mi355_opts::keep_canvas (capi_ctx.h: mi355_ctx::Canvas) ----
} // extern "C"

// is this frame one that keeps its canvas?  (alias = the device's address of the caller's memory)
static bool canvas_wanted(mi355_ctx *c, int mode, const mi355_opts *o, uint32_t *out_xrgb, int pitch_bytes, int rows, void **alias)
{
    const bool raster_mode = mode >= MI355_MODE_AMBIENT && mode <= MI355_MODE_PHONG_SOFTSHADOWMAPS;
    // (a raytraced frame: where its tiles are culled against the boxes at the tree's top -- enqueue_frame's conditions; the mask of
    //  the tiles it traces is what the canvas remembers)
    const bool rt_mode = mode >= MI355_MODE_RAYTRACE && c->has_bvh && c->dev.ordered_ok && c->n_cull_boxes > 0 && !(o->tune[5] & (4 | 8 | 16)) &&
                         ((long long)((o->width + 7) / 8) * ((o->height + 7) / 8)) <= (long long)MI_CULL_MAX_TILES;
    if (!o->keep_canvas || !(raster_mode || rt_mode)) return false;
    const bool ok = !o->mlaa && o->band_count <= 1 && !o->collect_stats && pitch_bytes >= o->width * 4 && !(pitch_bytes & 3) && rows > 0 &&
                    host_range_registered(c, out_xrgb, (size_t)pitch_bytes * (size_t)(rows - 1) + (size_t)o->width * 4) &&
                    hipHostGetDevicePointer(alias, out_xrgb, 0) == hipSuccess && *alias;
    if (!ok) (void)hipGetLastError();
    return ok;
}

// The context's record of the canvas at out_xrgb, ready for a frame on stream `st`: masks that describe what the canvas holds -- every
// bin, if that is not known (keep_canvas = 2, a canvas not seen before, anything else written there since) --, ordered behind the
// kernels of the canvas's last frame.  The canvas counts as unknown until canvas_done.
static int canvas_begin(mi355_ctx *c, int mode, const mi355_opts *o, uint32_t *out_xrgb, int pitch_bytes, hipStream_t st, mi355_ctx::Canvas **out)
{
    const int W = o->width, H = o->height;
    // (what a mask is: the rasterizer's -- a word per 64x64-pixel bin -- or the raytracer's -- a bit per 8x8-pixel tile)
    const int kind = mode >= MI355_MODE_RAYTRACE ? 1 : 0;
    const size_t mask_bytes = kind ? (size_t)(((long long)((W + 7) / 8) * ((H + 7) / 8) + 31) / 32) * 4 : (size_t)mi355i_raster_coarse_bins(W, H) * 4;
    mi355_ctx::Canvas *cv = nullptr;
    for (auto &e : c->canvas) if (e.host == out_xrgb) { cv = &e; break; }
    if (!cv) { cv = &c->canvas[0]; for (auto &e : c->canvas) if (e.used < cv->used) cv = &e; }       // (the least recently used record goes)
    const bool known = o->keep_canvas == 1 && cv->valid && cv->host == out_xrgb && cv->W == W && cv->H == H && cv->pitch == pitch_bytes && cv->kind == kind &&
                       cv->mask[0].bytes >= mask_bytes && cv->mask[1].bytes >= mask_bytes;
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);            // (canvases_written of another thread's frame reads these)
        cv->valid = false;
        cv->host = out_xrgb; cv->W = W; cv->H = H; cv->pitch = pitch_bytes; cv->kind = kind;
    }
    cv->used = ++c->canvas_clock;
    // (the frame this record last saw -- of this canvas or of the one it has just forgotten -- may still be reading and writing the masks)
    if (cv->ev_set) HIP_TRY(wait_unless_done(st, cv->ev), -40);
    if (!known) {
        HIP_TRY(cv->mask[0].ensure(mask_bytes), -31);
        HIP_TRY(cv->mask[1].ensure(mask_bytes), -31);
        cv->cur = 0;
        HIP_TRY(hipMemsetAsync(cv->mask[0].p, 0xff, mask_bytes, st), -40);
    }
    *out = cv;
    return 0;
}

// the frame's kernels are on `st`: the canvas holds it (in stream order), its masks have changed places
// (`behind` = an event of the context's or of a slot's that has just been recorded behind those kernels -- an event record costs the
//  host 5 us, a tenth of a kept raster frame in flight, so the canvas borrows it: if its owner records it again for a later frame, the
//  canvas's next frame waits a little longer than it has to)
static int canvas_done(mi355_ctx *c, mi355_ctx::Canvas *cv, hipEvent_t behind)
{
    cv->ev = behind;
    cv->ev_set = behind != nullptr;
    cv->cur ^= 1;
    cv->valid = true;
    return 0;
}

extern "C" {

int mi355_render(mi355_ctx *c, int mode, const mi355_camera *cam, const mi355_light *lights, int n_lights,
                 const mi355_opts *o, uint32_t *out_xrgb, int pitch_bytes, float *out_rgb_f32, mi355_stats *stats)
{
    if (!c || !cam || !o || !out_xrgb || (n_lights > 0 && !lights)) return fail(-3, "mi355_render: null argument");
    if (int r = validate_opts(*o, mode)) return r;
    if (int r = select_device(c)) return r;
    const int W = o->width;
    const int rows = (o->band_count > 1 && o->compact_rows) ? count_rows(*o) : o->height;
    HIP_TRY(c->fb.ensure((size_t)W * o->height * 4), -31);
    const bool wantf = out_rgb_f32 && mode >= MI355_MODE_RAYTRACE;
    if (wantf) HIP_TRY(c->fbf.ensure((size_t)W * o->height * 12), -31);
    if (o->band_count > 1) {    // rows of other bands are never written: define them as black
        HIP_TRY(hipMemsetAsync(c->fb.p, 0, (size_t)W * o->height * 4, c->stream), -40);
        if (wantf) HIP_TRY(hipMemsetAsync(c->fbf.p, 0, (size_t)W * o->height * 12, c->stream), -40);
    }
    // A raytraced frame into page-locked memory of the caller's (mi355_host_register: Screen::_pixels of the host layer) is written
    // THERE by the kernels: the traced tiles a pixel at a time, the background -- most of the frame -- in whole cache lines by the
    // waves that have run out of pixels, while the others trace (k_raytrace).  No copy behind the frame: 0.54 ms of kernel + 0.17 ms
    // of DMA became the kernel's time alone.  Not for the rasterizer (its 80 us of kernels would wait for 8 MB to cross PCIe first),
    // the post filter (it works in place) and bands (whose other rows are defined as black: a memset of device memory).
    void *host_alias = nullptr;
    // (... and only where the tiles are culled: a frame traced tile by tile would cross PCIe 32 bytes at a time)
    const bool culled = c->n_cull_boxes > 0 && !(o->tune[5] & (4 | 8 | 16)) && ((long long)((W + 7) / 8) * ((o->height + 7) / 8)) <= (long long)MI_CULL_MAX_TILES;
    bool zero_copy = culled && mode >= MI355_MODE_RAYTRACE && !o->mlaa && o->band_count <= 1 && !o->collect_stats && pitch_bytes >= W * 4 && !(pitch_bytes & 3) &&
                           host_range_registered(c, out_xrgb, (size_t)pitch_bytes * (size_t)(rows - 1) + (size_t)W * 4) &&
                           hipHostGetDevicePointer(&host_alias, out_xrgb, 0) == hipSuccess && host_alias;
    if (!zero_copy) (void)hipGetLastError();
    // A raster frame into a canvas whose last frame is known (mi355_opts::keep_canvas): written THERE too, but only where it can differ
    // from the frame before -- the 64x64-pixel bins that hold triangles now (by their tiles' blocks) and black into those that held
    // some before and hold none now (k_rs_tile): a 1080p chessboard frame sends ~1.5 of its 8.3 MB across PCIe, and the 166 us of the
    // copy behind the kernels are gone.  The first such frame of a canvas (or after anything else was drawn into that memory) is
    // written in full: every bin counts as held.
    void *canvas_alias = nullptr;
    mi355_ctx::Canvas *cv = nullptr;
    if (!wantf && canvas_wanted(c, mode, o, out_xrgb, pitch_bytes, rows, &canvas_alias))      // (the float buffer is not a canvas: its background is written in full)
        if (int r = canvas_begin(c, mode, o, out_xrgb, pitch_bytes, c->stream, &cv)) return r;
    const bool keep = cv != nullptr;
    // (a frame into host memory: whoever keeps a canvas there knows nothing of it -- but the canvas this frame keeps)
    canvases_written(cv, out_xrgb, (size_t)pitch_bytes * (size_t)(rows > 0 ? rows - 1 : 0) + (size_t)W * 4);
    host_trace("render ctx %p mode %d %dx%d: out %p + %zu zero_copy %d (alias %p) registered %d f32 %p + %zu", (void *)c, mode, W, o->height, (void *)out_xrgb,
               (size_t)pitch_bytes * (size_t)(rows > 0 ? rows - 1 : 0) + (size_t)W * 4, (int)zero_copy, host_alias,
               (int)host_range_registered(c, out_xrgb, (size_t)pitch_bytes * (size_t)(rows > 0 ? rows - 1 : 0) + (size_t)W * 4), (void *)(wantf ? out_rgb_f32 : nullptr), wantf ? (size_t)W * rows * 12 : (size_t)0);
    FrameParams P;
    if (keep) zero_copy = false;             // (the kept frame is written there too, but not all of it)
    if (int r = fill_params(c, mode, cam, lights, n_lights, o, zero_copy ? host_alias : keep ? canvas_alias : c->fb.p, zero_copy || keep ? pitch_bytes : W * 4, wantf ? c->fbf.p : nullptr, P)) return r;
    if (keep && mode >= MI355_MODE_RAYTRACE) { P.rt_keep_prev = (const uint32_t *)cv->mask[cv->cur].p; P.rt_keep_next = (uint32_t *)cv->mask[cv->cur ^ 1].p; }
    else if (keep) { P.canvas_keep = 1; P.canvas_prev = (const uint32_t *)cv->mask[cv->cur].p; P.canvas_next = (uint32_t *)cv->mask[cv->cur ^ 1].p; }
    mi355_stats tmp;
    mi355_stats *st = stats ? stats : &tmp;
    P.no_pipe = 1;          // (a synchronous frame has nothing to overlap with: its kernels follow each other on the context's stream)
    // (waves that start with the background -- the rest of it is written by waves that have run out of pixels.  Measured, frames/s of the
    //  synchronous call: 4: 1 584, 8: 1 601, 16: 1 593, 32: 1 559, 96: 1 555, 256: 1 570, 1024: 1 531; profiles/r04_analysis.md 3)
    if (zero_copy) P.fill_first = 16;
    for (int attempt = 0;; attempt++) {
        HIP_TRY(hipEventRecord(c->ev0, c->stream), -40);
        if (int r = enqueue_frame(c, mode, P, o->collect_stats, c->stream)) return r;
        HIP_TRY(hipEventRecord(c->ev1, c->stream), -40);
        // (the counters travel behind the kernels: the control block is the context's own for a synchronous frame)
        HIP_TRY(c->pin_counters.ensure(sizeof(unsigned long long) * CS_COUNT), -31);
        const bool own_block = !c->last_ctrl || c->last_ctrl == c->ctrl.p;
        if (own_block) HIP_TRY(hipMemcpyAsync(c->pin_counters.p, (char *)c->ctrl.p + 16, sizeof(unsigned long long) * CS_COUNT, hipMemcpyDeviceToHost, c->stream), -31);
        HIP_TRY(hipStreamSynchronize(c->stream), -40);
        const int r = own_block ? stats_from_counters(c, st, (unsigned long long *)c->pin_counters.p)
                                : mi355_fetch_stats(c, st);  // also surfaces a rasterizer bin overflow ...
        if (r == -44 && attempt < 8) continue;               // ... after which the buffers have grown: draw the frame again
        if (r) return r;
        break;
    }
    if (zero_copy) {
        // (the frame is where it belongs; the stream has been synchronised)
    } else if (keep) {
        // (... and so is this one; the tile kernel has noted the bins it holds)
        if (int r = canvas_done(c, cv, c->ev1)) return r;
    } else if (pitch_bytes > 0 && host_range_registered(c, out_xrgb, (size_t)pitch_bytes * (size_t)(rows - 1) + (size_t)W * 4)) {
        // page-locked by the caller (mi355_host_register): one DMA transfer, no staging by the runtime
        HIP_TRY(hipMemcpy2DAsync(out_xrgb, (size_t)pitch_bytes, c->fb.p, (size_t)W * 4, (size_t)W * 4, (size_t)rows, hipMemcpyDeviceToHost, c->stream), -31);
        HIP_TRY(hipStreamSynchronize(c->stream), -40);
    } else
        HIP_TRY(hipMemcpy2D(out_xrgb, (size_t)pitch_bytes, c->fb.p, (size_t)W * 4, (size_t)W * 4, (size_t)rows, hipMemcpyDeviceToHost), -31);
    if (wantf) HIP_TRY(hipMemcpy(out_rgb_f32, c->fbf.p, (size_t)W * rows * 12, hipMemcpyDeviceToHost), -31);
    if (stats) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1), -40);
        stats->kernel_ms = ms;
    }
    return 0;
}

// Pipelined frames.  The reference's seam is one synchronous frame per call (renderer.cc:522-583); a front-end that
// draws frame k+1 while frame k is on its way to the host splits the call in two.  Up to MI355_MAX_IN_FLIGHT frames
// are in flight, each on its own stream with its own control block, framebuffer and rasterizer scratch: their kernels
// overlap (a single 1080p raytraced frame leaves most of the GPU idle while its slowest tiles finish) and so do the
// transfers.  The frame is in out_xrgb when mi355_render_wait(ticket) returns; the pixels are those of mi355_render.
int mi355_render_async(mi355_ctx *c, int mode, const mi355_camera *cam, const mi355_light *lights, int n_lights, const mi355_opts *o,
                       uint32_t *out_xrgb, int pitch_bytes, int *ticket)
{
    if (!c || !cam || !o || !out_xrgb || !ticket || (n_lights > 0 && !lights)) return fail(-3, "mi355_render_async: null argument");
    if (int r = validate_opts(*o, mode)) return r;
    if (n_lights < 0 || n_lights > MI355_MAX_LIGHTS) return fail(-21, "n_lights %d outside 0..%d", n_lights, MI355_MAX_LIGHTS);
    if (o->collect_stats) return fail(-21, "pipelined frames cannot collect the counters");
    // (the staged path copies row by row with the caller's pitch: checked here, fill_params below only sees the internal one)
    if (pitch_bytes < o->width * 4 || (pitch_bytes & 3)) return fail(-21, "bad pitch %d for width %d", pitch_bytes, o->width);
    if (int r = select_device(c)) return r;
    mi355_ctx::AsyncSlot *a = nullptr;
    for (auto &s : c->slot) if (!s.busy) { a = &s; break; }
    if (!a) return fail(-45, "%d frames are in flight: call mi355_render_wait first", MI355_MAX_IN_FLIGHT);
    const int W = o->width;
    const int rows = (o->band_count > 1 && o->compact_rows) ? count_rows(*o) : o->height;
    if (!a->ready) {
        // (slot i on a candidate stream of queue class i: frames in flight then never queue up behind one another)
        // Every piece is created if it is missing and the slot counts as usable only once all of them exist: a call that fails
        // half way leaves a slot the next call completes instead of one that looks initialised.
        const int si = (int)(a - c->slot);
        if (!a->st && probe_classes(c) && c->n_class >= 2)
            for (int i = 0; i < (int)mi355_ctx::PIPE_CANDS && !a->st; i++) if (c->cand_class[i] == si % c->n_class) a->st = c->cand_st[i];
        if (!a->st) { HIP_TRY(hipStreamCreateWithFlags(&a->st, hipStreamNonBlocking), -11); a->st_owned = true; }
        if (!a->ev0) HIP_TRY(hipEventCreate(&a->ev0), -11);
        if (!a->ev1) HIP_TRY(hipEventCreate(&a->ev1), -11);
        HIP_TRY(a->ctrl.ensure(MI_CTRL_BYTES), -31);
        HIP_TRY(hipMemsetAsync(a->ctrl.p, 0, MI_CTRL_BYTES, a->st), -40);
        if (!a->rs) a->rs = mi355i_raster_scratch_create();
        if (!a->rs) return fail(-11, "out of memory");
        a->ready = true;
    }
    HIP_TRY(a->fb.ensure((size_t)W * o->height * 4), -31);
    if (o->band_count > 1) HIP_TRY(hipMemsetAsync(a->fb.p, 0, (size_t)W * o->height * 4, a->st), -40);
    // (a raster frame into a canvas whose last frame is known, mi355_opts::keep_canvas: written there by the kernels, where it can differ)
    void *canvas_alias = nullptr;
    mi355_ctx::Canvas *cv = nullptr;
    if (canvas_wanted(c, mode, o, out_xrgb, pitch_bytes, rows, &canvas_alias))
        if (int r = canvas_begin(c, mode, o, out_xrgb, pitch_bytes, a->st, &cv)) return r;
    // (a frame on its way into host memory: any other kept canvas it touches is no longer what its masks say)
    canvases_written(cv, out_xrgb, (size_t)pitch_bytes * (size_t)(rows > 0 ? rows - 1 : 0) + (size_t)W * 4);
    FrameParams P;
    if (int r = fill_params(c, mode, cam, lights, n_lights, o, cv ? canvas_alias : a->fb.p, cv ? pitch_bytes : W * 4, nullptr, P, a->ctrl.p)) return r;
    if (cv && mode >= MI355_MODE_RAYTRACE) { P.rt_keep_prev = (const uint32_t *)cv->mask[cv->cur].p; P.rt_keep_next = (uint32_t *)cv->mask[cv->cur ^ 1].p; }
    else if (cv) { P.canvas_keep = 1; P.canvas_prev = (const uint32_t *)cv->mask[cv->cur].p; P.canvas_next = (uint32_t *)cv->mask[cv->cur ^ 1].p; }
    HIP_TRY(hipEventRecord(a->ev0, a->st), -40);
    if (int r = enqueue_frame(c, mode, P, 0, a->st, a->ctrl.p, a->rs, &a->mlaa, nullptr, &a->sel)) return r;
    HIP_TRY(hipEventRecord(a->ev1, a->st), -40);
    // (the counters travel behind the kernels, as for a synchronous frame: mi355_render_wait finds them in page-locked memory -- a
    //  blocking hipMemcpy of its own was 20 us of every frame's wait)
    HIP_TRY(a->pin_counters.ensure(sizeof(unsigned long long) * CS_COUNT), -31);
    HIP_TRY(hipMemcpyAsync(a->pin_counters.p, (char *)a->ctrl.p + 16, sizeof(unsigned long long) * CS_COUNT, hipMemcpyDeviceToHost, a->st), -31);
    a->kept = cv;
    if (cv) { if (int r = canvas_done(c, cv, a->ev1)) return r; }
    a->staged = !cv && !host_range_registered(c, out_xrgb, (size_t)pitch_bytes * (size_t)(rows - 1) + (size_t)W * 4);
    host_trace("render_async ctx %p mode %d %dx%d: out %p + %zu staged %d", (void *)c, mode, W, o->height, (void *)out_xrgb, (size_t)pitch_bytes * (size_t)(rows > 0 ? rows - 1 : 0) + (size_t)W * 4, (int)a->staged);
    if (cv) {
        // (nothing to copy: the frame is in the canvas when the stream gets there)
    } else if (a->staged) {
        HIP_TRY(a->pin.ensure((size_t)W * rows * 4), -31);
        HIP_TRY(hipMemcpyAsync(a->pin.p, a->fb.p, (size_t)W * rows * 4, hipMemcpyDeviceToHost, a->st), -31);
    } else
        HIP_TRY(hipMemcpy2DAsync(out_xrgb, (size_t)pitch_bytes, a->fb.p, (size_t)W * 4, (size_t)W * 4, (size_t)rows, hipMemcpyDeviceToHost, a->st), -31);
    a->busy = true; a->ticket = c->next_ticket++; a->mode = mode; a->n_lights = n_lights; a->pitch_bytes = pitch_bytes;
    a->user = out_xrgb; a->cam = *cam; a->opts = *o;
    for (int i = 0; i < n_lights; i++) a->lights[i] = lights[i];
    *ticket = a->ticket;
    return 0;
}

int mi355_render_wait(mi355_ctx *c, int ticket, mi355_stats *stats)
{
    if (!c) return fail(-3, "mi355_render_wait: null argument");
    if (int r = select_device(c)) return r;
    mi355_ctx::AsyncSlot *a = nullptr;
    for (auto &s : c->slot) if (s.busy && s.ticket == ticket) { a = &s; break; }
    if (!a) return fail(-45, "mi355_render_wait: no frame with ticket %d is in flight", ticket);
    HIP_TRY(hipStreamSynchronize(a->st), -40);
    a->busy = false;
    unsigned long long h[CS_COUNT];
    if (a->pin_counters.p) memcpy(h, a->pin_counters.p, sizeof h);                         // (copied behind the frame's kernels)
    else HIP_TRY(hipMemcpy(h, (char *)a->ctrl.p + 16, sizeof h, hipMemcpyDeviceToHost), -31);
    // (the frame has landed in host memory: see mi355_render_async.  A kept frame whose bins overflowed is drawn again below like any
    //  other -- into the slot's buffer, copied to the canvas in full --, and the canvas is no longer what its masks say)
    canvases_written(h[CS_OVERFLOW] ? nullptr : a->kept, a->user, (size_t)a->pitch_bytes * (size_t)(a->opts.height > 0 ? a->opts.height - 1 : 0) + (size_t)a->opts.width * 4);
    a->kept = nullptr;
    if (h[CS_OVERFLOW]) {
        // the rasterizer's bins were too small for this frame: this slot's grow, and the frame is drawn again (synchronously)
        for (int attempt = 0; attempt < 8; attempt++) {
            if (!mi355i_raster_grow(a->rs)) break;
            HIP_TRY(hipMemset((char *)a->ctrl.p + 16 + sizeof(unsigned long long) * CS_OVERFLOW, 0, sizeof(unsigned long long)), -31);
            FrameParams P;
            if (int r = fill_params(c, a->mode, &a->cam, a->lights, a->n_lights, &a->opts, a->fb.p, a->opts.width * 4, nullptr, P, a->ctrl.p)) return r;
            if (int r = enqueue_frame(c, a->mode, P, 0, a->st, a->ctrl.p, a->rs, &a->mlaa, nullptr, &a->sel)) return r;
            HIP_TRY(hipStreamSynchronize(a->st), -40);
            HIP_TRY(hipMemcpy(h, (char *)a->ctrl.p + 16, sizeof h, hipMemcpyDeviceToHost), -31);
            if (!h[CS_OVERFLOW]) break;
        }
        if (h[CS_OVERFLOW]) return fail(-44, "rasterizer triangle bins overflowed (%llu entries dropped)", h[CS_OVERFLOW]);
        a->staged = true;
        const int W = a->opts.width, rows = (a->opts.band_count > 1 && a->opts.compact_rows) ? count_rows(a->opts) : a->opts.height;
        HIP_TRY(a->pin.ensure((size_t)W * rows * 4), -31);
        HIP_TRY(hipMemcpy(a->pin.p, a->fb.p, (size_t)W * rows * 4, hipMemcpyDeviceToHost), -31);
    }
    if (a->staged) {
        const int W = a->opts.width, rows = (a->opts.band_count > 1 && a->opts.compact_rows) ? count_rows(a->opts) : a->opts.height;
        for (int r = 0; r < rows; r++) memcpy((char *)a->user + (size_t)r * a->pitch_bytes, (char *)a->pin.p + (size_t)r * W * 4, (size_t)W * 4);
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->normal_rays = h[CS_NORMAL_RAYS]; stats->shadow_rays = h[CS_SHADOW_RAYS];
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, a->ev0, a->ev1), -40);
        stats->kernel_ms = ms;
    }
    return 0;
}

} // extern "C"
