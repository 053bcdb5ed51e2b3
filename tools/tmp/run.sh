for lib in "" neural-astar_amd/lib/libnastar_hip_old.so; do
  for thr in "" 99999999; do
    echo "== lib=$lib hybrid_from=$thr"
    env ${lib:+NASTAR_LIB=$lib} ${thr:+NASTAR_HYBRID_FROM_CELLS=$thr} timeout 300 python tools/tmp/repro.py tools/tmp/case_31_41.npz 2>&1 | grep -v amdgpu.ids
  done
done
