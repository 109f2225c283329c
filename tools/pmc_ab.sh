#!/bin/bash
# usage: tools/pmc_ab.sh <tag> <driver.py> [driver args...]  -> SQ/GRBM counter passes + kernel-trace stats of one driver
set -uo pipefail
tag="$1"; shift; drv="$1"; shift
out="gpurun_out/pmc_$tag"; mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o t -- python3 "$drv" "$@" > "$out/stats.log" 2>&1 || { echo "stats failed"; tail -5 "$out/stats.log"; exit 1; }
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
           "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  name="p_$(echo "$set" | tr ' ' '+' | cut -c1-30)"
  timeout -s KILL 150 rocprofv3 --pmc $set -d "$out/$name" -o pmc --output-format csv -- python3 "$drv" "$@" > "$out/$name.log" 2>&1 || { echo "pass [$set] failed"; tail -3 "$out/$name.log"; }
done
python3 tools/rocprof_summary.py pmc $out/p_*/ > $out/pmc.json
python3 tools/rocprof_summary.py stats $out/stats > $out/stats.json
echo "pmc_ab $tag done"
