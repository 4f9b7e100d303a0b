#!/usr/bin/env python3
"""hipHostRegister of a read-only file mapping by several threads at once, and
the cost of unregistering: GB/s over piece sizes and thread counts."""
import ctypes as C
import mmap
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

hip = C.CDLL('libamdhip64.so')
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipSetDevice.argtypes = [C.c_int]
size = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else 4 << 30
for d in ('/dev/shm', '/tmp'):
    fp = os.path.join(d, 'wk_reg_test.bin')
    with open(fp, 'wb') as f:
        blk = os.urandom(1 << 20) * 64
        for _ in range(size // len(blk)):
            f.write(blk)
    fd = os.open(fp, os.O_RDONLY)
    for piece in (64 << 20, 256 << 20):
        for threads in (1, 2, 4, 8):
            m = mmap.mmap(fd, size, flags=mmap.MAP_SHARED, prot=mmap.PROT_READ)
            arr = np.frombuffer(m, dtype=np.uint8)
            ptr = arr.ctypes.data
            offs = list(range(0, size, piece))

            def reg(off):
                hip.hipSetDevice(0)
                return hip.hipHostRegister(ptr + off, min(piece, size - off), 8)
            with ThreadPoolExecutor(threads) as pool:
                t0 = time.perf_counter()
                rcs = list(pool.map(reg, offs))
                t1 = time.perf_counter()
                list(pool.map(lambda off: hip.hipHostUnregister(ptr + off), offs))
                t2 = time.perf_counter()
            print(f'{d} piece {piece >> 20} MB, {threads} threads: register {size / (t1 - t0) / 1e9:.1f} GB/s '
                  f'(failures {sum(1 for r in rcs if r)}), unregister {size / (t2 - t1) / 1e9:.1f} GB/s', flush=True)
            del arr
            m.close()
    os.close(fd)
    os.unlink(fp)
