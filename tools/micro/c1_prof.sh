# rocprofv3 kernel statistics of the single-patch generator forward (BASELINE configs[0]); run on the GPU box
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r03_c1_prof
mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/c1.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from dispu_amd import synth
from dispu_amd.generator import Generator
from dispu_amd.params import init_params
dev = torch.device("cuda:0")
gen = Generator(params=init_params(1234), device=dev)
x = torch.from_numpy(synth.patches(1, 256, seed=1000)).to(dev)
for _ in range(30):
    gen(x)
torch.cuda.synchronize()
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o prof -- python /tmp/c1.py > $OUT/log.txt 2>&1 || true
cd $GRAFT_REPO_ROOT
find $OUT/raw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/raw
