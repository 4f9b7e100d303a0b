#!/usr/bin/env python3
"""Can the page cache feed the GPU without a CPU copy?  mmap a file (tmpfs and
disk), hipHostRegister the mapping in pieces, copy to the device from it; next
to pread into pinned memory + copy.  Prints GB/s of each step."""
import ctypes as C
import mmap
import os
import sys
import time

hip = C.CDLL('libamdhip64.so')
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipDeviceSynchronize.argtypes = []
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

size = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else 2 << 30
dev = C.c_void_p()
assert hip.hipMalloc(C.byref(dev), size) == 0
for d in ('/dev/shm', '/tmp'):
    fp = os.path.join(d, 'wk_reg_test.bin')
    with open(fp, 'wb') as f:
        blk = os.urandom(1 << 20) * 64
        for _ in range(size // len(blk)):
            f.write(blk)
    fd = os.open(fp, os.O_RDONLY)
    for flags, prot, rflag, label in ((mmap.MAP_SHARED, mmap.PROT_READ, 8, 'shared ro, ReadOnly flag'),
                                      (mmap.MAP_PRIVATE, mmap.PROT_READ, 8, 'private ro, ReadOnly flag'),
                                      (mmap.MAP_PRIVATE, mmap.PROT_READ | mmap.PROT_WRITE, 0, 'private rw'),
                                      (mmap.MAP_PRIVATE | mmap.MAP_POPULATE, mmap.PROT_READ | mmap.PROT_WRITE, 0, 'private rw populate')):
        m = mmap.mmap(fd, size, flags=flags, prot=prot)
        arr = __import__('numpy').frombuffer(m, dtype='uint8')
        ptr = arr.ctypes.data
        piece = 256 << 20
        t0 = time.perf_counter()
        rc = 0
        for off in range(0, size, piece):
            rc = hip.hipHostRegister(ptr + off, min(piece, size - off), rflag)
            if rc:
                break
        t1 = time.perf_counter()
        if rc:
            print(f'{d} {label}: hipHostRegister failed rc={rc}')
        else:
            r1 = r2 = 0
            for off in range(0, size, piece):   # (a copy may not span registrations)
                r1 |= hip.hipMemcpyAsync(C.c_void_p(dev.value + off), ptr + off, min(piece, size - off), 1, None)
            r1 |= hip.hipDeviceSynchronize()
            t2 = time.perf_counter()
            for off in range(0, size, piece):
                r2 |= hip.hipMemcpyAsync(C.c_void_p(dev.value + off), ptr + off, min(piece, size - off), 1, None)
            r2 |= hip.hipDeviceSynchronize()
            t3 = time.perf_counter()
            back = (C.c_char * 4096)()
            r3 = hip.hipMemcpy(back, C.c_void_p(dev.value + size - 4096), 4096, 2)
            ok = bytes(back) == bytes(arr[size - 4096:size])
            print('   rc', r1, r2, r3, 'data ok', ok)
            print(f'{d} {label}: register {size / (t1 - t0) / 1e9:.1f} GB/s, first copy {size / (t2 - t1) / 1e9:.1f} GB/s, '
                  f'second copy {size / (t3 - t2) / 1e9:.1f} GB/s', flush=True)
            for off in range(0, size, piece):
                hip.hipHostUnregister(ptr + off)
        del arr
        m.close()
    # baseline: pread into pinned + copy
    pin = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(pin), 256 << 20, 0) == 0
    view = (C.c_char * (256 << 20)).from_address(pin.value)
    t0 = time.perf_counter()
    for off in range(0, size, 256 << 20):
        n = os.preadv(fd, [memoryview(view)], off)
        hip.hipMemcpyAsync(dev, pin, n, 1, None)
        hip.hipDeviceSynchronize()
    t1 = time.perf_counter()
    print(f'{d} pread (1 thread) + copy: {size / (t1 - t0) / 1e9:.1f} GB/s')
    os.close(fd)
    os.unlink(fp)
