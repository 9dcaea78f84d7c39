#!/usr/bin/env python3
"""Workgroup index -> XCD, measured (ss_debug_xcc_map): alone, and for a second launch made while the first still holds its CUs."""
import ctypes as C
import os
import sys
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speechsplit_amd import _capi                     # noqa: E402

lib = _capi.lib()


def run(n, threads, lds, hold, stream):
    out = torch.zeros(2 * n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    _capi.check(lib.ss_debug_xcc_map(n, threads, lds, hold, C.c_void_p(out.data_ptr()), C.c_void_p(stream.cuda_stream)))
    return out


def show(tag, out):
    v = out.cpu().numpy().astype('uint32').reshape(-1, 2)
    x = v[:, 0]
    print(f'{tag}: raw XCC_ID register values seen: {sorted(set(int(a) for a in x))}')
    print('   first 24 workgroups -> XCC_ID & 15:', [int(a) & 15 for a in x[:24]])
    same = all((int(a) & 15) == (i % 8) for i, a in enumerate(x))
    print(f'   XCC_ID & 15 == workgroup index % 8 for every workgroup: {same};  workgroups per XCD: {sorted(Counter(int(a) & 15 for a in x).items())}')
    cu = Counter((int(a) & 15, int(h)) for a, h in v)
    print(f'   distinct (XCD, HW_ID) pairs: {len(cu)} for {len(x)} workgroups')


def main():
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    show('256 workgroups x 640 threads, 100 KB LDS, alone', run(256, 640, 100 * 1024, 50, sa))
    torch.cuda.synchronize()
    show('128 workgroups x 640 threads alone', run(128, 640, 100 * 1024, 50, sa))
    torch.cuda.synchronize()
    # a launch whose workgroups with index % 8 >= 4 leave at once is what the recurrence at B = 32 looks like; here: hold 128 of 256 CUs with a
    # first launch of index-filtered residents, then see where a second launch lands meanwhile
    a = run(256, 640, 100 * 1024, 400, sa)
    b = run(256, 512, 128 * 1024, 20, sb)
    torch.cuda.synchronize()
    show('first launch (256 x 640 threads x 100 KB, held 400 us)', a)
    show('second launch beside it (256 x 512 threads x 128 KB)', b)


if __name__ == '__main__':
    main()
