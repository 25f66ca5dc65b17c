# usage: bash tools/run_ab.sh <tag> <rounds> <profile> "<bench args>" lib...   (GPU box)
set -u
cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/$tag
bash tools/ab_multi.sh "$@" > gpurun_out/$tag/ab.txt 2>&1 || { tail -20 gpurun_out/$tag/ab.txt; exit 1; }
grep -v "amdgpu.ids\|waves, \|per wave:" gpurun_out/$tag/ab.txt
