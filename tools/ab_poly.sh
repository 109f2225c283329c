#!/bin/bash
# usage: tools/ab_poly.sh <lib.so> [<lib.so> ...]   A/B of library builds on the bench legs the headline does not show:
# polymul (config 2) and the prepared-key path against the packed path of the same rounds. Times, not cycles: see
# tools/ab_cycles.sh for differences of a per cent or two.
for lib in "$@"; do
  FALCON_AMD_LIB="$PWD/$lib" timeout -k 10 400 python bench.py --no-cpu --no-pmc 2>>gpurun_out/bench_err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); e=d['extras']; print(sys.argv[1], '| polymul', round(e['polymul_n512']['polys_per_s']/1e6,1), 'M/s | prepared', round(e['falcon512_prepared_keys']['verifies_per_s']/1e6,1), 'packed', round(e['falcon512_prepared_keys']['packed_path_same_rounds_verifies_per_s']/1e6,1), '| head', round(d['value']/1e6,1))" "$lib" || exit 1
done
