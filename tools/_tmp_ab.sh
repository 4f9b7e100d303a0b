D=/dev/shm/wk_e2e
python tools/e2e_once.py twopass2 --dir $D --prepare > /dev/null 2>&1
for round in 1 2; do
echo "== default"; python tools/e2e_once.py twopass2 --dir $D --run --reps 3 2>&1 | grep '^{"kind"' | cut -c1-80
echo "== no pin ahead"; WOLTKA_NO_PIN_AHEAD=1 python tools/e2e_once.py twopass2 --dir $D --run --reps 3 2>&1 | grep '^{"kind"' | cut -c1-80
done
rm -rf $D
