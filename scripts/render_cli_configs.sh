#!/bin/bash
# The reference's own benchmark procedure (`renderer -b -n N -m MODE FILE`, renderer.cc:243-341, 481-520; src/Makefile.am:25-26) for the
# five BASELINE.json configurations, through the C++ host layer's render_cli: three frames in flight (its default: the cameras of
# -b are known) and the reference's loop itself (-p 1: one synchronous Scene::render* per pass, rate = frames / time inside the calls).
CLI=renderer_amd/lib/render_cli
M=$(python -c "import renderer_amd as R; print(R.assets.mesh_path('chessboard.tri'))" 2>/dev/null | tail -1); D=$(dirname $M)
run() { # label, args...
  label=$1; shift
  a=$($CLI -b "$@" 2>&1 | grep Rendering | sed 's/.*(\(.*\) fps.*/\1/')
  b=$($CLI -b -p 1 "$@" 2>&1 | grep Rendering | sed 's/.*(\(.*\) fps.*/\1/')
  k=""
  case "$*" in *"-m 6"*|*"-m 8"*|*"-m 9"*)   # ... and with --keep-canvas (Screen::_keepCanvas: frames cross PCIe only where they differ from the canvas's last)
    c=$($CLI -b --keep-canvas "$@" 2>&1 | grep Rendering | sed 's/.*(\(.*\) fps.*/\1/')
    d=$($CLI -b -p 1 --keep-canvas "$@" 2>&1 | grep Rendering | sed 's/.*(\(.*\) fps.*/\1/')
    k=", \"fps_in_flight_keep_canvas\": $c, \"fps_reference_loop_keep_canvas\": $d";;
  esac
  echo "{\"config\": \"$label\", \"fps_3_in_flight\": $a, \"fps_reference_loop\": $b$k}"
}
run "1: chessboard.tri -m 2 (points from triangles) 640x480" -n 2000 -m 2 -W 640 -H 480 $D/chessboard.tri
run "2: chessboard.tri -m 6 (Phong) 1920x1080" -n 2000 -m 6 -W 1920 -H 1080 $D/chessboard.tri
run "2': chessboard.tri -m 8 (Phong + soft shadow maps) 1920x1080" -n 2000 -m 8 -W 1920 -H 1080 $D/chessboard.tri
run "3: statue.ply -m 9 --depth 1 (primary + shadow rays) 1920x1080" -n 1000 -m 9 --depth 1 -W 1920 -H 1080 $D/statue.ply
run "4: dragon_vis.ply -m 9 (shadows + 2 reflection bounces) 1920x1080" -n 1000 -m 9 -W 1920 -H 1080 $D/dragon_vis.ply
run "5: dragon_vis.ply -m 9 3840x2160 (one GPU)" -n 400 -m 9 -W 3840 -H 2160 $D/dragon_vis.ply
