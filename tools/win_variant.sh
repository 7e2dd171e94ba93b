#!/bin/bash
# Build a variant of libvbm25.so whose scan_win.hip is compiled with extra flags (kernel experiments):
#   tools/win_variant.sh <name> [hipcc flags ...]   ->  vectorchord-bm25_amd/csrc/libvbm25_<name>.so
# Run it on the GPU box with VBM25_LIBRARY=$GRAFT_REPO_ROOT/vectorchord-bm25_amd/csrc/libvbm25_<name>.so python bench.py ...
set -e
cd "$(dirname "$0")/../vectorchord-bm25_amd/csrc"
name=$1; shift
make -s search.o flush.o segment.o pages.o blake3.o >/dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -pthread -w "$@" -c scan_win.hip -o scan_win_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o libvbm25_$name.so search.o scan_win_$name.o flush.o segment.o pages.o blake3.o
echo built libvbm25_$name.so
