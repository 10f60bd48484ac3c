#!/bin/bash
# Developer aid: build the library of a given git revision (default HEAD) as ctrlv_amd/lib/libctrlv_old.so for same-session
# A/B runs on the GPU box (select with CTRLV_HIP_LIB=$PWD/ctrlv_amd/lib/libctrlv_old.so).
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/ctrlv_amd/csrc $T/include
git -C $ROOT show $REV:include/ctrlv_hip.h > $T/include/ctrlv_hip.h
for f in $(git -C $ROOT ls-tree --name-only $REV ctrlv_amd/csrc/); do git -C $ROOT show $REV:$f > $T/$f; done
cd $T/ctrlv_amd/csrc
for s in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $s -o ${s%.hip}.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ctrlv_amd/lib/libctrlv_old.so *.o
rm -rf $T
echo "built $ROOT/ctrlv_amd/lib/libctrlv_old.so from $REV"
