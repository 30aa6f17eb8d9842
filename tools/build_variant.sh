#!/bin/bash
# build/libv_<name>.so from the working tree, or from a git revision of the three library sources (for tools/ab_round.sh)
#   usage: bash tools/build_variant.sh <name> [revision|""] [extra hipcc flags, e.g. -DLCGP_EXP=3]
NAME=$1; REV=${2:-}; EXTRA=${3:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build
if [ -z "$REV" ]; then
  SRC=$ROOT/lcgp_amd/csrc/lcgp_hip.hip
else
  T=$(mktemp -d); mkdir -p $T/lcgp_amd/csrc $T/include
  git -C $ROOT show $REV:lcgp_amd/csrc/lcgp_hip.hip > $T/lcgp_amd/csrc/lcgp_hip.hip
  git -C $ROOT show $REV:lcgp_amd/csrc/fill_sched.h > $T/lcgp_amd/csrc/fill_sched.h
  git -C $ROOT show $REV:include/lcgp_hip.h > $T/include/lcgp_hip.h
  SRC=$T/lcgp_amd/csrc/lcgp_hip.hip
fi
HASH=$(cd $ROOT && python3 -c "from lcgp_amd import _hip; print(_hip.source_hash())")
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared $EXTRA "-DLCGP_SRC_HASH=\"LCGP_SRC_HASH=$HASH\"" -o $ROOT/build/libv_$NAME.so $SRC && echo built build/libv_$NAME.so
