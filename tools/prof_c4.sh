#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel statistics of the SECOND generator pass of 16x upsampling (BASELINE configs[3]:
# 32 clouds of 1024 points -> 4096; the first pass is the headline step and has its own table).  Usage: tools/prof_c4.sh <tag>
TAG=${1:-r05_c4}
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/c4_pass2.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from dispu_amd import synth
from dispu_amd.generator import Generator
from dispu_amd.params import init_params
dev = torch.device("cuda:0")
gen = Generator(params=init_params(1234), device=dev)
gen.return_views = True
gen.branches = False                  # one stream: every kernel alone on the device
x = torch.from_numpy(synth.patches(32, 1024, seed=3000)).to(dev)      # what the first pass hands over: 32 clouds of 1024 points
for _ in range(3):
    gen(x)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    gen(x)
torch.cuda.synchronize()
print("second pass (32, 1024 -> 4096): %.3f ms per call" % ((time.perf_counter() - t) * 100))
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o prof -- python /tmp/c4_pass2.py > $OUT/log.txt 2>&1 || true
cd $GRAFT_REPO_ROOT
find $OUT/raw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/raw
tail -2 $OUT/log.txt
head -24 $OUT/kernel_stats.csv | cut -d, -f1-4 | cut -c1-170
