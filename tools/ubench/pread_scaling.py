#!/usr/bin/env python3
"""How fast can a tmpfs / disk-cache file be read into memory with N threads?
os.preadv of 8 MB pieces into (a) a numpy buffer, (b) pinned memory
(hipHostMalloc), (c) via the tokenizer's reader (wk_tok_read)."""
import ctypes as C
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from woltka_amd import _native as nat  # noqa: E402

size = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else 4 << 30
hip = C.CDLL('libamdhip64.so')
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
pin = C.c_void_p()
assert hip.hipHostMalloc(C.byref(pin), size, 0) == 0
pinned = np.frombuffer((C.c_char * size).from_address(pin.value), dtype=np.uint8)
plain = np.empty(size, dtype=np.uint8)
plain[:] = 0
pinned[:] = 0
piece = 8 << 20
for d in ('/dev/shm', '/tmp'):
    fp = os.path.join(d, 'wk_pread_test.bin')
    with open(fp, 'wb') as f:
        blk = os.urandom(1 << 20) * 64
        for _ in range(size // len(blk)):
            f.write(blk)
    fd = os.open(fp, os.O_RDONLY)
    for name, buf in (('numpy', plain), ('pinned', pinned)):
        for threads in (1, 2, 4, 8, 16, 32):
            mv = memoryview(buf)

            def rd(off):
                return os.preadv(fd, [mv[off:off + piece]], off)
            with ThreadPoolExecutor(threads) as pool:
                t0 = time.perf_counter()
                n = sum(pool.map(rd, range(0, size, piece)))
                t = time.perf_counter() - t0
            print(f'{d} -> {name}, {threads:2d} threads: {n / t / 1e9:6.1f} GB/s', flush=True)
    for threads in (4, 16, 32):
        tok = nat.Tokenizer(threads)
        t0 = time.perf_counter()
        got = 0
        for off in range(0, size, 64 << 20):
            got += tok.read_into(fd, off, memoryview(pinned)[off:off + (64 << 20)])
        t = time.perf_counter() - t0
        print(f'{d} -> pinned, wk_tok_read {threads:2d} threads, 64 MB calls: {got / t / 1e9:6.1f} GB/s', flush=True)
        tok.close()
    os.close(fd)
    os.unlink(fp)
