echo "== cgroup"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null; cat /proc/self/cgroup | head -3
echo "== affinity"; python3 -c "import os; print(len(os.sched_getaffinity(0)))"; nproc
echo "== cpu.stat"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
cat > /tmp/spin.py <<'PY'
import sys, time, threading, ctypes, os
import numpy as np
# numpy work releases the GIL: each thread sums a private array repeatedly
def work(n, out, i):
    a = np.arange(1<<16, dtype=np.int64)
    t0=time.perf_counter(); c=0
    while time.perf_counter()-t0 < 2.0:
        a.sum(); c+=1
    out[i]=c
T=int(sys.argv[1]); out=[0]*T
th=[threading.Thread(target=work,args=(0,out,i)) for i in range(T)]
[t.start() for t in th]; [t.join() for t in th]
print(sum(out))
PY
for t in 8 16 32 64 128 256; do echo -n "threads $t: "; python3 /tmp/spin.py $t; done
echo "== 4 procs x 32"; for i in 1 2 3 4; do python3 /tmp/spin.py 32 & done; wait
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
