#!/usr/bin/env python
"""Does librccl write its NCCL_DEBUG=INFO lines to NCCL_DEBUG_FILE on this image?  (bench.py's rccl_info stayed null in rounds 2-3.)"""
import os, sys, glob
os.environ['NCCL_DEBUG'] = 'INFO'
os.environ['NCCL_DEBUG_SUBSYS'] = 'INIT,GRAPH,TUNING'
os.environ['NCCL_DEBUG_FILE'] = '/tmp/probe_rccl_%p.log'
import torch
import torch.distributed as dist
dist.init_process_group('nccl', rank=0, world_size=1, init_method='tcp://127.0.0.1:29977')
t = torch.ones(1 << 20, device='cuda')
dist.all_reduce(t)
torch.cuda.synchronize()
print('pid', os.getpid(), 'files', glob.glob('/tmp/probe_rccl_*'))
for f in glob.glob('/tmp/probe_rccl_*'):
    print(open(f, errors='replace').read()[:3000])
