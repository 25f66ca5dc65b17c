set -u
cd $GRAFT_REPO_ROOT
bash tools/prof_bench.sh r3p_bench > gpurun_out/r3p_bench.txt 2>&1 && echo "prof_bench ok" && \
bash tools/prof_prefill.sh r3p_ext --reps 100 > gpurun_out/r3p_ext.txt 2>&1 && echo "prof_prefill ok" && \
bash tools/pmc_extend.sh r3p_pmc > gpurun_out/r3p_pmc.txt 2>&1 && echo "pmc_extend ok" && \
bash tools/pmc_traffic.sh r3p_traffic > gpurun_out/r3p_traffic.txt 2>&1 && echo "pmc_traffic ok" && \
(cd $GRAFT_REPO_ROOT && timeout -k 10 600 python bench.py > gpurun_out/r3p_benchline.json 2> gpurun_out/r3p_benchline.err) && echo "bench ok"
tail -3 gpurun_out/r3p_bench.txt gpurun_out/r3p_ext.txt gpurun_out/r3p_pmc.txt gpurun_out/r3p_traffic.txt | cut -c1-400
cut -c1-3000 gpurun_out/r3p_benchline.json
