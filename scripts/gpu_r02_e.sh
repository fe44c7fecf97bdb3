mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_extras.py -q -k banding > gpurun_out/r02e_band.log 2>&1; echo "banding rc=$?"
timeout 300 python -c "
import renderer_amd as R
print(R.device_count())
s = R.Scene(R.assets.mesh_path('chessboard.tri')); s.bvh_create()
cam, l, n = R.benchmark_frame(0)
s.render(9, cam, l, n, R.default_opts(64, 64))
import torch
x = torch.zeros(4, device='cuda'); torch.cuda.synchronize()
print('ok')
" > gpurun_out/r02e_late_torch.log 2>&1; echo "late torch rc=$?"
tail -3 gpurun_out/r02e_band.log gpurun_out/r02e_late_torch.log
