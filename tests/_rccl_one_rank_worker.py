"""Worker of test_gpu_rccl.py: a one-rank RCCL group drives FrameGatherer's collective path (send buffers, dist.gather
into rank 0's receive block, de-interleave) on the GPU; the frames must equal plain single-GPU frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import renderer_amd as R
from renderer_amd import multigpu

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
W, H, B = 640, 360, 3
s = R.Scene(R.assets.mesh_path("dragon_vis.ply")); s.bvh_update()
stream = torch.cuda.current_stream(dev)
g = multigpu.FrameGatherer(W, H, dev, frames=B, collective_when_alone=True)
o = R.default_opts(W, H, band_rows=multigpu.BAND_ROWS, band_index=0, band_count=1, compact_rows=1)
got = []
for step in range(4):                                  # double-buffered: step k+2 reuses buffer k
    slot = step & 1
    fs = [(step * B + j) % 200 for j in range(B)]
    cl = [R.benchmark_frame(f) for f in fs]
    buf = g.send_buffer(slot)
    s.render_batch_device(9, [c[0] for c in cl], [c[1] for c in cl], 1, o, [buf[j].data_ptr() for j in range(B)], W * 4, None, stream.cuda_stream)
    g.gather(slot)
    got.append((fs, slot))
    if step >= 1:                                      # read the previous step's frames while this one is in flight
        pfs, pslot = got[step - 1]
        fr = g.frame(pslot).clone()
        for j, f in enumerate(pfs):
            want = torch.zeros((H, W), dtype=torch.int32, device=dev)
            cam, lights, n = R.benchmark_frame(f)
            s.render_device(9, cam, lights, n, R.default_opts(W, H), want.data_ptr(), W * 4, 0, stream.cuda_stream)
            torch.cuda.synchronize(dev)
            assert torch.equal(fr[j], want), "frame %d differs after the gather" % f
g.drain()
# throughput mode: whole frames of a batch through the same collective
bg = multigpu.BatchGatherer(W, H, dev, B, collective_when_alone=True)
for step in range(3):
    slot = step & 1
    fs = multigpu.frames_of_rank([(step * B + j) % 200 for j in range(B)], 1, 0)
    cl = [R.benchmark_frame(f) for f in fs]
    buf = bg.send_buffer(slot)
    s.render_batch_device(9, [c[0] for c in cl], [c[1] for c in cl], 1, R.default_opts(W, H), [buf[j].data_ptr() for j in range(B)], W * 4, None, stream.cuda_stream)
    bg.gather(slot)
    fr = bg.frame(slot).clone()
    assert fr.shape == (B, H, W)
    for j, f in enumerate(fs):
        want = torch.zeros((H, W), dtype=torch.int32, device=dev)
        cam, lights, n = R.benchmark_frame(f)
        s.render_device(9, cam, lights, n, R.default_opts(W, H), want.data_ptr(), W * 4, 0, stream.cuda_stream)
        torch.cuda.synchronize(dev)
        assert torch.equal(fr[j], want), "frame %d differs after the whole-frame gather" % f
bg.drain()
t = torch.ones(4, device=dev); dist.all_reduce(t); assert float(t.sum()) == 4.0
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
