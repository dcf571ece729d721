#!/bin/bash
# Development: time the TransformerEnc path with parts of the chain kernel removed (GPU box).
#   tools/ablate_tenc.sh "0 256 512 ..."   (bits: kernel_tenc.h; sums combine)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
# whatever happens below, the shipped (B2H_ABLATE = 0) build is what is left in place
trap 'B2H_ABLATE= python -m hand_pose_sl_amd.build --force > /dev/null 2>&1' EXIT
export B2H_ALLOW_ABLATE=1   # hand_pose_sl_amd._lib refuses ablation builds without it
for a in ${1:-0 256 512 1024 2048 4096 8192}; do
  B2H_ABLATE=$a python -m hand_pose_sl_amd.build --force > /dev/null 2>&1
  r=$(timeout -k 10 200 python tools/bench_tenc.py --quick 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms' % d['runs'][0]['ms'])")
  echo "ablate=$a : $r"
done
