cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r04 > gpurun_out/profiles_r04.log 2>&1
bash tools/collect_mfma_lds.sh r04 >> gpurun_out/profiles_r04.log 2>&1
tail -60 gpurun_out/profiles_r04.log
