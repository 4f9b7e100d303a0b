R=${GRAFT_REPO_ROOT:-$PWD}
D=/dev/shm/wk_ab
for k in lca twopass1; do python $R/tools/e2e_once.py $k --dir $D --prepare > /dev/null 2>&1; done
for rep in 1 2 3; do
  for k in lca twopass1; do
    echo -n "pin-ahead    $k: "; python $R/tools/e2e_once.py $k --dir $D --run --reps 3 2>/dev/null | tail -1 | cut -c1-70
    echo -n "no pin-ahead $k: "; WOLTKA_NO_PIN_AHEAD=1 python $R/tools/e2e_once.py $k --dir $D --run --reps 3 2>/dev/null | tail -1 | cut -c1-70
    echo -n "no warm      $k: "; WOLTKA_NO_WARM=1 WOLTKA_NO_PIN_AHEAD=1 python $R/tools/e2e_once.py $k --dir $D --run --reps 3 2>/dev/null | tail -1 | cut -c1-70
  done
done
rm -rf $D
