import ctypes, subprocess, sys, os
sys.path.insert(0, '/root/repo')
import importlib
import numpy as np
import torch
torch.cuda.init()
torch.zeros(1).cuda()
lib = ctypes.CDLL(os.path.abspath('wrfchem-arc-interactions_amd/librrtmg_hip.so'))
sys.argv = ['bench.py', '--mode', 'sw', '--steps', '10', '--warmup', '2', '--no-cpu']
z = (ctypes.c_ulonglong * 8)()
lib.rrtmg_hip_debug_phases(z, 1)
import bench
try:
    bench.main()
except SystemExit as e:
    print('exit', e)
sys.stdout.flush()
lib.rrtmg_hip_debug_phases(z, 0)
v = np.array(list(z), dtype=float)
names = ['prologue', 'gsetup', 'up(+mcica,setup)', 'down', 'barrier', 'epilogue', 'gasphase', 'gas barrier']
for n, x in zip(names, v):
    print(f'{n:12s} {x:.3e} {100 * x / v.sum():.1f}%')
