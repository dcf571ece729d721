#!/bin/bash
# Profile bench.py on the GPU box with rocprofv3; summaries land in gpurun_out/<tag>/.
#   tools/profile.sh <tag> [bench args...]
# Pass 1: kernel trace + stats.  Passes 2..: PMC counters, each in its own run.
set -u
TAG=${1:-prof}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps ${PROF_STEPS:-50} --warmup 5 --no-cpu-baseline $*"
run() { # name, rocprof flags...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 "$REPO/bench.py" $ARGS > "$OUT/$name.stdout" 2> "$OUT/$name.stderr"
  echo "[$name] rc=$?"
}
run trace --kernel-trace --stats
if [ -z "${TRACE_ONLY:-}" ]; then
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
run pmc_sq1 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run pmc_sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM
run pmc_sq3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM
run pmc_tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum
run pmc_grbm --kernel-trace --pmc GRBM_GUI_ACTIVE
fi
# keep only the small summaries
find "$OUT" -name "*.csv" -size +2M -delete
ls -R "$OUT" | head -60
