#!/bin/bash
# builds the library of a git revision (default HEAD) as artspeech_amd/lib/exp_head.so: A/B runs on ONE box (boxes differ by +-3 %)
#   AS_LIB_PATH=artspeech_amd/lib/exp_head.so python bench.py ...   against   python bench.py ...
set -e
cd "$(dirname "$0")/.."
REV=${1:-HEAD}
rm -rf /tmp/headsrc /tmp/headobj; mkdir -p /tmp/headsrc /tmp/headobj
git archive $REV artspeech_amd/csrc include | tar -x -C /tmp/headsrc
for f in /tmp/headsrc/artspeech_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I /tmp/headsrc/artspeech_amd/csrc -I /tmp/headsrc/include -c $f -o /tmp/headobj/$(basename $f).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o artspeech_amd/lib/exp_head.so /tmp/headobj/*.o
echo built artspeech_amd/lib/exp_head.so from $REV
