cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
timeout -k 10 120 tools/probes/memset_capture_probe > gpurun_out/r3l/memset_probe.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/r3l/memset_probe.txt
