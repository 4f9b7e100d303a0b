#!/usr/bin/env python3
"""H2D copy rate of 64 MB blocks from hipHostMalloc memory (default flags,
non-coherent, write-combined), alone and with 16 threads writing into other
pinned buffers at the same time (what the reader does)."""
import ctypes as C
import os
import threading
import time

import numpy as np

hip = C.CDLL('libamdhip64.so')
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
n = 64 << 20
dev = C.c_void_p()
hip.hipMalloc(C.byref(dev), n)
stream = C.c_void_p()
hip.hipStreamCreateWithFlags(C.byref(stream), 1)
for flags, label in ((0, 'default'), (0x40000000, 'non-coherent'), (0x80000000, 'coherent'), (0x4, 'write-combined')):
    pin = C.c_void_p()
    rc = hip.hipHostMalloc(C.byref(pin), n, flags)
    if rc:
        print(label, 'alloc failed', rc)
        continue
    arr = np.frombuffer((C.c_char * n).from_address(pin.value), dtype=np.uint8)
    arr[:] = 1
    for busy in (False, True):
        stop = threading.Event()
        ths = []
        if busy:
            others = []
            for _ in range(8):
                p2 = C.c_void_p()
                hip.hipHostMalloc(C.byref(p2), n, flags)
                others.append(np.frombuffer((C.c_char * n).from_address(p2.value), dtype=np.uint8))
            src = np.ones(n, dtype=np.uint8)

            def work(k):
                while not stop.is_set():
                    others[k][:] = src
            ths = [threading.Thread(target=work, args=(k,)) for k in range(8)]
            for t in ths:
                t.start()
            time.sleep(0.05)
        hip.hipMemcpyAsync(dev, pin, n, 1, stream)
        hip.hipStreamSynchronize(stream)
        t0 = time.perf_counter()
        t_call = 0.0
        for _ in range(20):
            t1 = time.perf_counter()
            hip.hipMemcpyAsync(dev, pin, n, 1, stream)
            t_call += time.perf_counter() - t1
            hip.hipStreamSynchronize(stream)
        t = time.perf_counter() - t0
        stop.set()
        for th in ths:
            th.join()
        print(f'{label:15s} {"with 8 writer threads" if busy else "alone":22s}: {20 * n / t / 1e9:6.1f} GB/s, '
              f'{t / 20 * 1e3:.2f} ms per block, enqueue {t_call / 20 * 1e3:.3f} ms', flush=True)

# source not aligned (a block begins where the last run of the block before was cut)
pin = C.c_void_p()
hip.hipHostMalloc(C.byref(pin), n + 4096, 0)
for off in (0, 1, 4, 16, 64, 123, 256, 4096 - 7):
    for size in (n, n - 13):
        hip.hipMemcpyAsync(dev, C.c_void_p(pin.value + off), size, 1, stream)
        hip.hipStreamSynchronize(stream)
        t0 = time.perf_counter()
        for _ in range(10):
            hip.hipMemcpyAsync(dev, C.c_void_p(pin.value + off), size, 1, stream)
            hip.hipStreamSynchronize(stream)
        t = time.perf_counter() - t0
        print(f'source offset {off:5d}, {size} bytes: {10 * size / t / 1e9:6.1f} GB/s', flush=True)
