#!/bin/bash
# Builds the C-ABI shared library for gfx950 in-tree: falcon_amd/libfalcon_amd.so
# The device assembly of the same compilation is kept for one check (tools/check_scalar_loads.py): no scalar load with
# an unaligned immediate offset -- a code shape hipcc has produced for this library and gfx950 executes wrongly; and one
# more (tools/check_asm_loads.py): nothing names the target register of a load written as an instruction (verify_kernel.hpp
# prefetch_s2) between the load and its wait.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="${FA_OUT:-$here/../libfalcon_amd.so}"
tmp="$(mktemp -d)"
trap 'rm -rf "$tmp"' EXIT
(cd "$tmp" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Wno-pass-failed \
  -save-temps ${FA_EXTRA_FLAGS:-} -o "$out" "$here/capi.hip")
python3 "$here/../../tools/check_scalar_loads.py" "$tmp"/*gfx950*.s
python3 "$here/../../tools/check_asm_loads.py" "$tmp"/*gfx950*.s
echo "built $out"
