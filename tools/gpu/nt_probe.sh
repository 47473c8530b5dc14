#!/bin/bash
# Round-4 NT GEMM evidence on one GPU box: tile-order A/B with times AND fabric bytes, write-through output
# stores, in-kernel clock stamps, vendor ceilings at the step's shapes.  Raw logs -> gpurun_out/r04_nt_*; summaries are copied
# into profiles/ by hand.  usage: bash tools/gpu/nt_probe.sh
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd $root
export PYTHONUNBUFFERED=1
echo "== tile order A/B (product library), times" | tee $out/r04_nt_tile_ab.txt
timeout 600 python3 tools/nt_tile_ab.py 2>&1 | tee -a $out/r04_nt_tile_ab.txt
echo "== tile order A/B, write-through (sc1) output stores" | tee -a $out/r04_nt_tile_ab.txt
A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_sc1.so timeout 600 python3 tools/nt_tile_ab.py 2>&1 | tee -a $out/r04_nt_tile_ab.txt
echo "== in-kernel clock (diagnostic build)" | tee $out/r04_inkernel_clock.txt
A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip_stamp.so timeout 600 python3 tools/inkernel_clock.py 2>&1 | tee -a $out/r04_inkernel_clock.txt
echo "== hipBLASLt at the step's shapes" | tee $out/r04_blas_ceiling.txt
timeout 600 python3 tools/blas_ceiling.py 2>&1 | tee -a $out/r04_blas_ceiling.txt
# fabric bytes per launch, one PMC pass per (library, tile order, counter)
echo "== PMC" | tee $out/r04_nt_tile_pmc.txt
for libtag in "" _sc1; do
  for arm in 0 1; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      echo "-- lib${libtag:-_product} tile order $arm $ctr" | tee -a $out/r04_nt_tile_pmc.txt
      export A2V_LIB_PATH=$root/animal2vec_amd/liba2v_hip${libtag}.so
      bash tools/pmc_pass.sh gemm_nt_bf16_p8 $ctr -- $root/tools/nt_tile_ab.py --one $arm 2>&1 | tee -a $out/r04_nt_tile_pmc.txt
    done
  done
done
