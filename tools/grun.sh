#!/bin/bash
# tools/grun.sh [--timeout S] -- '<command>': rebuild the product (and the reference-linked test binaries) here, then hand the tree to gpurun
cd "$(dirname "$0")/.."
make -s -j8 -C minimd_amd/csrc all 2>&1 | grep -E "error|Error" && exit 1
[ -d /root/reference/ref ] && make -s -C oracle ref_hip 2>&1 | grep -E "error|Error"
exec /usr/local/graft/bin/gpurun "$@"
