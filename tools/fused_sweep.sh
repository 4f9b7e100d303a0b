for seed in 21 22 23 24 25 26; do
  for shape in long_lines long_runs plain tiny open_end; do
    for block in 8192 32768 262144; do
      out=$(timeout 60 python tools/fused_vs_six_blocks.py $seed $block 800 $shape 2>&1 | tail -1)
      echo "$seed $shape $block: $out"
    done
  done
done
