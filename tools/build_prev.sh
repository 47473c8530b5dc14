#!/bin/bash
# Builds the library of another revision next to the current one (animal2vec_amd/liba2v_hip_prev.so) for same-box A/B runs:
#   tools/build_prev.sh <git-rev>;  A2V_LIB_PATH=$PWD/animal2vec_amd/liba2v_hip_prev.so python tools/....py
# (only valid while the C ABI of the two revisions is the same)
set -e
rev=${1:-HEAD}
root=$(cd $(dirname $0)/.. && pwd)
tmp=/tmp/a2v_prev_$$
mkdir -p $tmp
git -C $root archive $rev animal2vec_amd/csrc | tar -x -C $tmp
cd $tmp/animal2vec_amd/csrc
objs=""
for f in *.hip; do
  extra=""; [ $f = a2v_attn.hip ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form $extra -c $f -o ${f%.hip}.o &
  objs="$objs ${f%.hip}.o"
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/animal2vec_amd/liba2v_hip_prev.so $objs
rm -rf $tmp
ls -la $root/animal2vec_amd/liba2v_hip_prev.so
