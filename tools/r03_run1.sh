#!/bin/bash
# round 3, GPU call 1: baseline of the unchanged kernels with the new bench line, CPU-thread sweep, convergence parity, per-op table,
# and the tests the ADVICE fixes touched
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
python bench.py --steps 20 --warmup 5 > $O/r03a_bench.json 2> $O/r03a_bench.err
tail -c 600 $O/r03a_bench.err
python tools/op_profile.py --top 400 > $O/r03a_ops.txt 2>/dev/null
timeout 900 python tools/convergence_parity.py --steps 60 --batch 8 --threads 32 > $O/r03_convergence.json 2> $O/r03_convergence.err
tail -3 $O/r03_convergence.err
timeout 900 python tools/cpu_sweep.py > $O/r03_cpu_sweep.json 2> $O/r03_cpu_sweep.err
timeout 900 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_trainer.py tests/test_gpu_inference.py -m gpu -x -q 2>&1 | tail -5
