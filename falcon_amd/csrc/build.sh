#!/bin/bash
# Builds the C-ABI shared library for gfx950 in-tree: falcon_amd/libfalcon_amd.so
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="${FA_OUT:-$here/../libfalcon_amd.so}"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Wno-pass-failed \
  ${FA_EXTRA_FLAGS:-} -o "$out" "$here/capi.hip"
echo "built $out"
