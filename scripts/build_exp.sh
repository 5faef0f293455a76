#!/bin/bash
# build experiment variants of the library: scripts/build_exp.sh NAME "-DFLAG ..."  -> artspeech_amd/lib/exp_NAME.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
FLAGS="$*"
mkdir -p artspeech_amd/lib/exp_obj_$NAME
for f in artspeech_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I artspeech_amd/csrc -I include $FLAGS -c $f -o artspeech_amd/lib/exp_obj_$NAME/$(basename $f).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o artspeech_amd/lib/exp_$NAME.so artspeech_amd/lib/exp_obj_$NAME/*.o
rm -rf artspeech_amd/lib/exp_obj_$NAME
echo built artspeech_amd/lib/exp_$NAME.so
