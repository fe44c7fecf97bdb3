cat > /tmp/smdbg.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import renderer_amd as R
for mesh in ("chessboard.tri", "dragon_vis.ply", "statue.ply"):
    s = R.Scene(R.assets.mesh_path(mesh))
    cam, lights, n = R.benchmark_frame(0)
    print(mesh, flush=True)
    for _ in range(2): s.shadowmap_render(0, lights[0])
PY
MI355_SM_DEBUG=1 timeout 100 python /tmp/smdbg.py 2>&1 | grep -v amdgpu
