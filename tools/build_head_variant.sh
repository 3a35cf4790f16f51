#!/bin/bash
# The library as of a commit (default HEAD) into variants/libet_head.so, for A/B runs against the working tree:
#   tools/build_head_variant.sh [commit];  then on the GPU box: bash tools/ab_env.sh base "head:ET_LIB_PATH=$PWD/variants/libet_head.so"
set -e
cd "$(dirname "$0")/.."
rev=${1:-HEAD}
src=$(mktemp -d)
out=variants/build_head
rm -rf $out variants/libet_head.so
mkdir -p $out
git archive $rev entreepy_amd/csrc include | tar -x -C $src
HIPCC=/opt/rocm/bin/hipcc
SRCDIR=$src/entreepy_amd/csrc
CXX="-O3 -std=c++17 -fPIC -I$src/include -I$src/entreepy_amd/csrc"
for f in $(cd $SRCDIR && ls *.hip | sed s/.hip//); do $HIPCC $CXX --offload-arch=gfx950 -c $src/entreepy_amd/csrc/$f.hip -o $out/$f.o & done
for f in $(cd $SRCDIR && ls *.cpp | grep -v entreepy_cli | sed s/.cpp//); do $HIPCC $CXX -c $src/entreepy_amd/csrc/$f.cpp -o $out/$f.o & done
wait
$HIPCC -shared -fPIC --offload-arch=gfx950 -o variants/libet_head.so $out/*.o -lpthread -ldl
rm -rf $out $src
echo variants/libet_head.so
