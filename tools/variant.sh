#!/bin/bash
# Build a variant of libesr_hip.so with one source recompiled under extra flags (developer experiments):
#   bash tools/variant.sh NAME feat.hip -DFEAT_EXP_X   ->  tools/_variants/NAME.so   (git-ignored; travels to the GPU box)
# On the box: cp tools/_variants/NAME.so esr_nerf_amd/libesr_hip.so && bash tools/kstats.sh NAME
set -e
NAME=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/esr_nerf_amd/_obj
mkdir -p "$ROOT/tools/_variants"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wall -Wno-unused-function -fno-fast-math \
    -I"$ROOT/include" "$@" -c "$ROOT/esr_nerf_amd/csrc/$SRC" -o "$ROOT/tools/_variants/$NAME.o"
OBJS=$(ls "$OBJ"/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$ROOT/tools/_variants/$NAME.so" $OBJS "$ROOT/tools/_variants/$NAME.o"
rm "$ROOT/tools/_variants/$NAME.o"
echo "$ROOT/tools/_variants/$NAME.so"
