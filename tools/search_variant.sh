#!/bin/bash
# Build a variant of libvbm25.so whose search.hip is compiled with extra flags (kernel experiments outside scan_win.hip):
#   tools/search_variant.sh <name> [hipcc flags ...]   ->  vectorchord-bm25_amd/csrc/libvbm25_<name>.so
set -e
cd "$(dirname "$0")/../vectorchord-bm25_amd/csrc"
name=$1; shift
make -s scan_win.o flush.o segment.o pages.o blake3.o >/dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -pthread -w "$@" -c search.hip -o search_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o libvbm25_$name.so search_$name.o scan_win.o flush.o segment.o pages.o blake3.o
echo built libvbm25_$name.so
