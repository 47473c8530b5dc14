#!/bin/bash
# same-box A/B of two library builds by kernel-trace stats (prev = tools/build_prev.sh <rev>)
bash tools/collect_profiles.sh abA large "--steps 2 --warmup 1 --global-batch 64" "" | tail -1
A2V_LIB_PATH=$GRAFT_REPO_ROOT/animal2vec_amd/liba2v_hip_prev.so bash tools/collect_profiles.sh abB large "--steps 2 --warmup 1 --global-batch 64" "" | tail -1
bash tools/collect_profiles.sh abC large "--steps 2 --warmup 1 --global-batch 64" "" | tail -1
