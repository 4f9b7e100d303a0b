#!/bin/bash
# host-only: does the tokenizer scale over processes on this box? (no GPU work)
out=gpurun_out/${1:-r03h}
mkdir -p $out
for mode in mmap pread; do
  for dir in /dev/shm /tmp; do
    echo "== mode $mode dir $dir" | tee -a $out/host_scaling.txt
    WOLTKA_TOK_TIMING=1 timeout 900 python tools/tok_scaling.py --reads 20000000 --threads 32 --procs 1,2,4,8 --mode $mode --dir $dir >> $out/host_scaling.txt 2>&1
  done
done
grep -v wk_tok $out/host_scaling.txt
