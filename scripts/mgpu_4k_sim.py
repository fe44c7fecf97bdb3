#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU playing every rank in turn: dragon 3840x2160, a step of 8 frames cut over N ranks by interleaved
8-scanline bands.  Per N: the slowest rank's batched launch (the step's render time on an N-GPU node before the exchange), what
rank 0 has to take in per step, and that transfer at the per-link rate the guide quotes -- the numbers DESIGN.md 5 puts side by
side.  Also the C library's own step (mi355_mgpu_render_batch, N virtual ranks on this one device, steps in flight)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import renderer_amd as R
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
W, H, F = 3840, 2160, 8
LINK_GBS = 50.0          # what one xGMI link delivers into rank 0 in practice (MI355X_MICROARCH.md: ~153 GB/s raw per link per direction; RCCL send/recv pairs reach a third of it)
s = R.Scene(R.assets.mesh_path("dragon_vis.ply")); s.bvh_create()
cams = [R.benchmark_frame(k) for k in range(200)]
def launch_ms(o, rows, reps=8):
    bufs = [torch.zeros((rows, W), dtype=torch.int32, device=dev) for _ in range(F)]
    def go(k):
        fs = [(k * F + j) % 200 for j in range(F)]
        s.render_batch_device(9, [cams[f][0] for f in fs], [cams[f][1] for f in fs], 1, o, [b.data_ptr() for b in bufs], W * 4, None, stream.cuda_stream)
    go(0); go(1); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(reps): go(k)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / reps * 1e3
for N in (1, 2, 4, 8):
    worst, rows_max = 0.0, 0
    for r in range(N):
        o = R.default_opts(W, H, band_rows=8, band_index=r, band_count=N, compact_rows=1) if N > 1 else R.default_opts(W, H)
        rows = sum(1 for y in range(H) if (y // 8) % N == r)
        rows_max = max(rows_max, rows)
        worst = max(worst, launch_ms(o, rows))
    ingest = (N - 1) * rows_max * W * 4 * F
    print(json.dumps({"ranks": N, "frames_per_step": F, "slowest_rank_render_ms": round(worst, 3), "rank0_ingest_MB_per_step": round(ingest / 1e6, 1),
                      "ingest_ms_at_%d_GBs_per_link" % int(LINK_GBS): round(ingest / max(1, N - 1) / (LINK_GBS * 1e9) * 1e3, 3) if N > 1 else 0.0}), flush=True)
# the C library's step with virtual ranks on this device (everything serialises on one GPU: a plumbing check, not a speed)
L = R.lib()
L.mi355_mgpu_create.restype = C.c_void_p
L.mi355_mgpu_create.argtypes = [C.POINTER(R.SceneDesc), C.POINTER(C.c_int), C.c_int]
L.mi355_mgpu_destroy.argtypes = [C.c_void_p]
L.mi355_mgpu_set_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
L.mi355_mgpu_render_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(R.Camera), C.POINTER(R.Light), C.c_int, C.POINTER(R.Opts), C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
L.mi355_mgpu_wait.argtypes = [C.c_void_p, C.c_int, C.POINTER(R.Stats)]
nodes, idx = s.bvh_arrays()
for N in (1, 8):
    m = L.mi355_mgpu_create(C.byref(s.desc), (C.c_int * N)(*([0] * N)), N)
    L.mi355_mgpu_set_bvh(m, nodes.ctypes.data, nodes.shape[0], idx.ctypes.data, idx.shape[0])
    o = R.default_opts(W, H)
    bufs = [[torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(F)] for _ in range(2)]
    def step(k):
        fs = [(k * F + j) % 200 for j in range(F)]
        ca = (R.Camera * F)(*[cams[f][0] for f in fs]); la = (R.Light * F)(*[cams[f][1][0] for f in fs])
        outs = (C.c_void_p * F)(*[b.data_ptr() for b in bufs[k & 1]])
        t = C.c_int(0)
        assert L.mi355_mgpu_render_batch(m, 9, F, ca, la, 1, C.byref(o), outs, W * 4, C.byref(t)) == 0, L.mi355_last_error()
        return t.value
    tk = [step(0), step(1)]
    for t in tk: L.mi355_mgpu_wait(m, t, None)
    t0 = time.perf_counter(); q = []
    for k in range(10):
        if len(q) == 2: L.mi355_mgpu_wait(m, q.pop(0), None)
        q.append(step(k))
    for t in q: L.mi355_mgpu_wait(m, t, None)
    print(json.dumps({"mi355_mgpu_render_batch": "%d virtual ranks on one device" % N, "ms_per_step_of_8_4k_frames": round((time.perf_counter() - t0) / 10 * 1e3, 3)}), flush=True)
    L.mi355_mgpu_destroy(m)
