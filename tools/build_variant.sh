#!/bin/bash
# A second build of the library with extra compiler flags, for A/B runs in one tree (ET_LIB_PATH picks it up):
#   tools/build_variant.sh <name> "<flags>"   ->  variants/libet_<name>.so     (variants/ is git-ignored scratch)
# The round-3 timing probes (profiles/r03_probes.txt): git apply tools/probe/r03_probes.patch first, then e.g.
#   tools/build_variant.sh d3s0 "-DET_PROBE_D3_NO_STORES"; on the GPU box: ET_BENCH_NO_VERIFY=1 bash tools/ab_env.sh base "d3s0:ET_LIB_PATH=$PWD/variants/libet_d3s0.so"
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
out=variants/build_$name
mkdir -p $out
HIPCC=/opt/rocm/bin/hipcc
SRCDIR=entreepy_amd/csrc
CXX="-O3 -std=c++17 -fPIC -Iinclude -Ientreepy_amd/csrc $flags"
for f in $(cd $SRCDIR && ls *.hip | sed s/.hip//); do $HIPCC $CXX --offload-arch=gfx950 -c entreepy_amd/csrc/$f.hip -o $out/$f.o & done
for f in $(cd $SRCDIR && ls *.cpp | grep -v entreepy_cli | sed s/.cpp//); do $HIPCC $CXX -c entreepy_amd/csrc/$f.cpp -o $out/$f.o & done
wait
$HIPCC -shared -fPIC --offload-arch=gfx950 -o variants/libet_$name.so $out/*.o -lpthread -ldl
echo "variants/libet_$name.so"
