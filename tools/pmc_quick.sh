#!/bin/bash
# usage: tools/pmc_quick.sh <tag> [logn items]   -> SQ counter passes only, summary printed
set -uo pipefail
tag="${1:-q}"; logn="${2:-9}"; items="${3:-1048576}"
out="gpurun_out/pmcq_$tag"; mkdir -p "$out"
export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_IFETCH" \
           "GRBM_GUI_ACTIVE"; do
  name="p_$(echo "$set" | tr ' ' '+' | cut -c1-30)"
  timeout -s KILL 120 rocprofv3 --pmc $set -d "$out/$name" -o pmc --output-format csv -- python3 tools/prof_run.py $logn $items 2 > "$out/$name.log" 2>&1 || { echo "pass [$set] failed"; tail -3 "$out/$name.log"; }
done
python3 tools/rocprof_summary.py pmc $out/p_*/ > $out/summary.json
cat $out/summary.json
