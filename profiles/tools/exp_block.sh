# headline leg at other block lengths (ms per block and per 2^20 samples)
P='import json,sys;d=json.loads(sys.stdin.read());b=int(sys.argv[1]);print("block", b, "steps", d["steps"], "ms/step %.4f" % d["ms_per_step"], "per 2^20: %.4f" % (d["ms_per_step"]*(1<<20)/b), "Msps %.0f" % d["value"])'
for b in ${BLOCKS:-1048576 524288 262144}; do
  k=$(( ${STEPS20:-200} * 1048576 / b ))
  timeout -k 10 100 python bench.py --no-cpu-baseline --no-spectrum --no-config4 --no-config5 --no-config1 --no-one-gpu --block $b --steps $k --warmup 20 2>/dev/null | python3 -c "$P" $b || exit 1
done
