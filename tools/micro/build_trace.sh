#!/bin/bash
# experiment build of the library with the phase stamps of mz_tree_dev.h compiled in
set -e
cd "$(dirname "$0")/../../muzero_amd/csrc"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Wno-unused-function -DMZ_TREE_TRACE $EXTRA -shared \
    -o ../../tools/micro/libmzhip_trace.so mz_api.hip mz_tree.hip mz_net.hip mz_search.hip mz_env.hip mz_replay.hip mz_history.hip
