#!/bin/bash
# Development: time the bf16 kernel with parts removed (GPU box).  tools/ablate.sh "0 1 2 4 ..."
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
for a in ${1:-0 1 2 4 3 6 7}; do
  B2H_ABLATE=$a python -m hand_pose_sl_amd.build --force > /dev/null 2>&1
  r=$(timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 --warmup 5 --seqs 65536 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.0f GB/s-equiv' % (d['roofline']['launch_ms'], d['roofline']['achieved']))")
  echo "ablate=$a : $r"
done
B2H_ABLATE= python -m hand_pose_sl_amd.build --force > /dev/null 2>&1
