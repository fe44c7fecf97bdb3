#!/bin/sh
# Regenerates tests/golden/legocar_3ds.r3ds.xz: the REAL lib3ds 1.3.0 (built from the reference tree by
# oracle/ref3ds/Makefile into oracle/_ref/, only possible where /root/reference exists) is run on the reference's
# 3D-Objects/legocar.3ds the way src/Loader.cc:276-353 drives it, and dumps what the loader pushes into the scene.
set -e
cd "$(dirname "$0")/.."
make -C oracle/ref3ds
oracle/_ref/dump3ds /root/reference/3D-Objects/legocar.3ds /tmp/legocar.r3ds
xz -9 -c /tmp/legocar.r3ds > tests/golden/legocar_3ds.r3ds.xz
sha256sum /tmp/legocar.r3ds
