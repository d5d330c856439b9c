cd $GRAFT_REPO_ROOT
bash tools/prof_train.sh r03_h_train 8 > /dev/null 2>&1
cat gpurun_out/r03_h_train/bench.json | cut -c1-300
