cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4p
bash tools/profile_set.sh r04h > gpurun_out/prof_r04h.log 2>&1; tail -2 gpurun_out/prof_r04h.log
echo "profile set done"
timeout -k 10 300 python bench.py --gpus 2 --ranks-share-gpu > gpurun_out/r4p/multi_rank_2.json 2> gpurun_out/r4p/multi_rank_2.err; echo "2 ranks rc=$?"
timeout -k 10 300 python bench.py --gpus 6 --ranks-share-gpu --only-c5 > gpurun_out/r4p/multi_rank_6_c5.json 2> gpurun_out/r4p/multi_rank_6_c5.err; echo "6 ranks rc=$?"
( timeout -k 10 500 python tools/frontend_fuzz.py 1500 2027 2>&1 | tail -1; timeout -k 10 500 python tools/ba_fuzz.py 1500 2027 2>&1 | tail -1; timeout -k 10 300 python tools/match_fuzz.py 300 2027 2>&1 | tail -1 ) | tee gpurun_out/r4p/fuzz_summary.txt
