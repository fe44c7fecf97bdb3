#!/bin/bash
# make_reference_pins.sh -- regenerate tests/golden/reference_pins.json's frame / .bvh pins from the REAL reference binary.
#
# PROVENANCE OF THE COMMITTED PINS.  The frame hashes, counters, camera probes and .bvh statistics in
# tests/golden/reference_pins.json were recorded by the survey from a strict single-thread build of the
# reference that it made with a headless SDL stub of its own.  That build cannot be repeated in this repository: the image
# has no SDL 1.2 development files, and writing a stand-in for them is not allowed here.  What CAN be rebuilt here from the
# reference's own sources -- Raytrace<>, the shadow map, camera / light bases, LightingEquation<>, the scalar BVH builder,
# MLAA -- is rebuilt by oracle/refcore and pinned bit for bit in tests/test_refcore_pins.py.  The survey-provenance pins
# cover the rest (Scene::load, the rasterizer's span walk and plotters, whole frames through main()).
#
# This script is the survey's recipe for a machine that HAS SDL 1.2 (`sdl-config` on PATH): it builds the
# reference unmodified except for the frame size, with the pinned strict flags, and reads the frames back through an
# LD_PRELOAD shim around the real SDL_Flip / SDL_UpdateRect (the reference has no option to write a frame to disk).
# It has not been run in this repository's container (no SDL there); exit code 3 = prerequisites missing, nothing done.
#
# usage: scripts/make_reference_pins.sh [reference-dir] [out-dir]
set -euo pipefail
R=${1:-/root/reference}
OUT=${2:-/tmp/reference_pins}
command -v sdl-config >/dev/null 2>&1 || { echo "no SDL 1.2 development files (sdl-config) on this machine: the committed pins stay survey-provenance" >&2; exit 3; }
[ -d "$R/src" ] || { echo "no reference tree at $R" >&2; exit 3; }
mkdir -p "$OUT"/{obj,models,frames} && cd "$OUT"

# lib3ds is C, not C++
for f in io vector matrix quat tcb ease chunk file background atmosphere shadow viewport material mesh camera light tracks node; do
    gcc -O2 -w -I"$R/lib3ds-1.3.0" -c "$R/lib3ds-1.3.0/lib3ds/$f.c" -o obj/$f.o
done
printf '#define HAVE_GETOPT_H 1\n' > config.h         # no USE_OPENMP / USE_TBB / SIMD_SSE / HANDLERAYTRACER / MLAA_ENABLED

# frame dump: every SDL_Flip (raster modes) / SDL_UpdateRect (raytracer, Screen.h:161-164) writes the raw R,G,B bytes
cat > dump_shim.c <<'EOF'
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <SDL.h>
static int frame_no;
static void dump(SDL_Surface *s)
{
    const char *prefix = getenv("PIN_DUMP");
    if (!prefix || !s || s->format->BytesPerPixel != 4) return;
    char name[512];
    snprintf(name, sizeof name, "%s_%04d.rgb", prefix, frame_no++);
    FILE *f = fopen(name, "wb");
    for (int y = 0; y < s->h; y++)
        for (int x = 0; x < s->w; x++) {
            const Uint32 p = ((Uint32 *)((char *)s->pixels + (size_t)y * s->pitch))[x];
            const unsigned char rgb[3] = {(unsigned char)(p >> 16), (unsigned char)(p >> 8), (unsigned char)p};
            fwrite(rgb, 1, 3, f);
        }
    fclose(f);
}
int SDL_Flip(SDL_Surface *s) { dump(s); return ((int (*)(SDL_Surface *))dlsym(RTLD_NEXT, "SDL_Flip"))(s); }
void SDL_UpdateRect(SDL_Surface *s, Sint32 x, Sint32 y, Uint32 w, Uint32 h)
{
    dump(s);
    ((void (*)(SDL_Surface *, Sint32, Sint32, Uint32, Uint32))dlsym(RTLD_NEXT, "SDL_UpdateRect"))(s, x, y, w, h);
}
EOF
gcc -O2 -shared -fPIC $(sdl-config --cflags) dump_shim.c -o dump_shim.so -ldl

build() {   # build <name> <width> <height> <max ray depth>
    rm -rf src_$1 && cp -r "$R/src" src_$1 && chmod -R u+w src_$1
    sed -i "s/^#define WIDTH.*/#define WIDTH $2/; s/^#define HEIGHT.*/#define HEIGHT $3/" src_$1/Defines.h     # Defines.h:26-27
    sed -i "s/^#define MAX_RAY_DEPTH.*/#define MAX_RAY_DEPTH $4/" src_$1/Raytracer.cc                             # Raytracer.cc:56
    g++ -O2 -ffp-contract=off -DNDEBUG -w -I. -Isrc_$1 -I"$R/lib3ds-1.3.0" $(sdl-config --cflags) \
        src_$1/{renderer,Base3d,BVH,Camera,Keyboard,Light,Loader,Rasterizers,Raytracer,Screen,Wu}.cc obj/*.o $(sdl-config --libs) -lm -o renderer_$1
}
build 480 640 480 3
build 1080 1920 1080 3
build 1080d1 1920 1080 1
cp "$R"/3D-Objects/{chessboard.tri,statue.ply,dragon_vis.ply} models/     # writable: <model>.bvh is written beside the model
export SDL_VIDEODRIVER=dummy LD_PRELOAD="$OUT/dump_shim.so"

run() {     # run <binary> <mode> <model> <tag>: frames f0, f50, f100, f150 are dump indices 1, 51, 101, 151 (index 0 = the blank
            # initial ShowScreen, renderer.cc:309)
    PIN_DUMP="frames/$4" ./renderer_$1 -b -n 151 -m $2 models/$3 > frames/$4.log 2>&1
    for k in 0001 0051 0101 0151; do sha256sum "frames/$4_$k.rgb"; done
    ls frames/$4_*.rgb | grep -v -e _0001 -e _0051 -e _0101 -e _0151 | xargs rm -f
}
{
    run 480 2 chessboard.tri cfg1_m2
    for m in 1 4 5 6 7 8 10; do run 1080 $m chessboard.tri chess_m$m; done
    run 1080d1 9 statue.ply cfg3_statue_d1
    run 1080 9 dragon_vis.ply cfg4_dragon
    run 1080 10 dragon_vis.ply dragon_m10
    sha256sum models/*.bvh          # full 64-hex hashes of the scalar builder's cache files
} | tee pins.txt
echo "pins written to $OUT/pins.txt: paste them into tests/golden/reference_pins.json (frames[].sha256, bvh.*.sha256)"
