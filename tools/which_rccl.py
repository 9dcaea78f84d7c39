#!/usr/bin/env python3
"""Which RCCL library instance(s) the process holds before and after the engine's native communicator is created (ss_comm_init dlopens
RCCL and must reuse the copy PyTorch already loaded instead of bringing a second one)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechsplit_amd import hparams as HP
from speechsplit_amd.engine import Engine


def rccl_maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'rccl' in l})


print('before comm_init:', rccl_maps())
hp = HP.default_hparams(max_len_pad=128)
e = Engine('G3', hp, 4, 128)
e.comm_init(0, 1)
print('after  comm_init:', rccl_maps())
