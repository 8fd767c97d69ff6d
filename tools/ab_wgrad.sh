run() { echo "$@"; env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-iou 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run SALT_WGRAD_TPW=8
run SALT_WGRAD_TPW=6
run SALT_WGRAD_TPW=10
run SALT_WGRAD_TPW=12
run SALT_WGRAD_TPW=8 SALT_WGRAD_WGS=128
run SALT_WGRAD_TPW=8 SALT_MAIN_PRIO=-1
run SALT_WGRAD_TPW=8 SALT_SIDE_PRIO=-1
run SALT_WGRAD_TPW=1 SALT_MAIN_PRIO=-1
run SALT_WGRAD_TPW=4 SALT_MAIN_PRIO=-1
python -c "import torch; print(torch.cuda.Stream.priority_range())"
