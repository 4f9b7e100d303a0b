# what the GPU box gives a process: memory, CPUs, scratch space
echo "== memory"; grep -E "MemTotal|MemAvailable" /proc/meminfo; cat /sys/fs/cgroup/memory.max 2>/dev/null
echo "== cpu"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
echo "== disks"; df -h /dev/shm /tmp . 2>/dev/null
echo "== numa"; lscpu | grep -i -E "numa|socket|model name" | head -12
