D=/dev/shm/wk_e2e
python tools/e2e_once.py ordinal --dir $D --prepare > /dev/null 2>&1
python tools/e2e_once.py ordinal --dir $D --run --reps 5 2>&1 | grep '^{"kind"' | cut -c1-120
python tools/e2e_cprofile.py ordinal --dir $D --top 30 2>&1 | cut -c1-160 | grep "table_rows\|build_mapper\|set_genes\|write_profiles\|build_hierarchy\|timed run\|workflow.py:67\|_coords_table"
python tools/e2e_once.py lca --dir $D --prepare > /dev/null 2>&1
python tools/e2e_cprofile.py lca --dir $D --top 45 2>&1 | cut -c1-170 | tail -52
rm -rf $D
