cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_final.log 2>&1; tail -4 gpurun_out/pytest_final.log
bash tools/collect_profiles.sh > gpurun_out/prof_bundle_log.txt 2>&1; tail -2 gpurun_out/prof_bundle_log.txt
