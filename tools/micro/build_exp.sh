#!/bin/bash
# experiment builds of the library: tools/micro/build_exp.sh <name> <extra hipcc flags...> -> tools/micro/libmzhip_<name>.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/../../muzero_amd/csrc"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Wno-unused-function "$@" -shared \
    -o ../../tools/micro/libmzhip_$NAME.so mz_api.hip mz_tree.hip mz_net.hip mz_search.hip mz_env.hip mz_replay.hip mz_history.hip
