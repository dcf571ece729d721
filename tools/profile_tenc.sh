#!/bin/bash
# Profile the TransformerEnc path (tools/bench_tenc.py --quick) on the GPU box with rocprofv3.
#   [TENC_ARGS=--precision=f16x3] tools/profile_tenc.sh <tag>      summaries land in gpurun_out/<tag>/
# Pass 1: kernel trace + stats.  Further passes: SQ counters, each in its own run.
set -u
TAG=${1:-prof_tenc}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, rocprof flags...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 "$REPO/tools/bench_tenc.py" --quick ${TENC_ARGS:-} > "$OUT/$name.stdout" 2> "$OUT/$name.stderr"
  echo "[$name] rc=$?"
}
run trace --kernel-trace --stats
if [ -z "${TRACE_ONLY:-}" ]; then
run pmc_sq1 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run pmc_sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM
run pmc_grbm --kernel-trace --pmc GRBM_GUI_ACTIVE
fi
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pmc_*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        print(f.split("/")[-2], k, {c: f"{v:.4g}" for c, v in d.items()})
PY
find "$OUT" -name "*.csv" -size +2M -delete
