for v in "" "DISPU_TRAIN_DEFER=0" "DISPU_TRAIN_DW_STREAMS=1" "DISPU_TRAIN_DW_STREAMS=4" "DISPU_TRAIN_OVERLAP=0" "DISPU_TRAIN_DEFER=0 DISPU_TRAIN_DW_STREAMS=1"; do
  echo -n "[$v] "; env $v python tools/train_bench.py --graph | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step_repeats']['median'],4))"
done
echo -n "[eager OVERLAP=0] "; DISPU_TRAIN_OVERLAP=0 python tools/train_bench.py | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step_repeats']['median'],4))"
