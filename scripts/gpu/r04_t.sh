mkdir -p gpurun_out
(timeout 300 python -m pytest tests -m gpu -q -x --capture=sys -k "shadow or light_rotation or screen_filling" 2>&1 | tail -2)
cp renderer_amd/lib/libmi355render.so /tmp/orig.so
echo "== W256"; timeout 100 python scripts/shadowmap_time.py 2>&1 | grep "us per"
for w in 128 64; do cp renderer_amd/lib/variant_W$w.so renderer_amd/lib/libmi355render.so; echo "== W$w"; timeout 100 python scripts/shadowmap_time.py 2>&1 | grep "us per"; done
cp /tmp/orig.so renderer_amd/lib/libmi355render.so
