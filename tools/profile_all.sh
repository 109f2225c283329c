#!/bin/bash
# Runs on the GPU box: kernel-trace stats of bench.py and the PMC passes (one rocprofv3 run each).
# usage: tools/profile_all.sh <tag>     -> gpurun_out/prof_<tag>/{stats,pmc_*}
set -uo pipefail
tag="${1:-r01}"
out="gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- python3 bench.py --steps 10 --warmup 2 --no-extras --no-cpu --no-pmc \
  > "$out/bench_under_rocprof.json" 2> "$out/stats.log" || { echo "stats run failed"; tail -5 "$out/stats.log"; exit 1; }
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
           "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name="pmc_$(echo "$set" | tr ' ' '+' | cut -c1-40)"
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d "$out/$name" -o pmc -- python3 tools/prof_run.py 9 1048576 2 \
    > "$out/$name.log" 2>&1 || { echo "pmc pass [$set] failed"; tail -3 "$out/$name.log"; exit 1; }
  echo "pmc pass [$set] ok"
done
echo "profile_all done"
