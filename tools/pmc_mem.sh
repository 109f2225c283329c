#!/bin/bash
# usage: tools/pmc_mem.sh <tag>   -> memory-side counters (TCC/TCP), summary printed
set -uo pipefail
tag="${1:-m}"; out="gpurun_out/pmcm_$tag"; mkdir -p "$out"; export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
   "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_WRREQ_STALL_sum TCC_READ_sum TCC_WRITEBACK_sum TCC_EA0_WR_UNCACHED_32B_sum" \
   "TCP_TA_TCP_STATE_READ_sum TCP_TCC_NC_WRITE_REQ_sum TCP_TCC_UC_WRITE_REQ_sum TCP_TCC_CC_WRITE_REQ_sum" "TA_BUSY_avr TA_FLAT_WRITE_WAVEFRONTS_sum" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  name="p_$(echo "$set" | tr ' ' '+' | cut -c1-30)"
  timeout -s KILL 120 rocprofv3 --pmc $set -d "$out/$name" -o pmc --output-format csv -- python3 tools/prof_run.py 9 1048576 2 > "$out/$name.log" 2>&1 || { echo "pass [$set] failed"; tail -2 "$out/$name.log"; }
done
python3 tools/rocprof_summary.py pmc $out/p_*/ > $out/summary.json
