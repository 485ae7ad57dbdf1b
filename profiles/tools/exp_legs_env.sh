# the slice legs under a list of environment settings: SETTINGS="A=1;B=2 C=3" (space-separated groups, ; inside a group)
P='import json,sys;d=json.loads(sys.stdin.read());print(sys.argv[1], "c2 %.4f" % d["ms_per_step"], " ".join("%s %.4f" % (k, d[k]["ms_per_step"]) for k in ("config4_slice","config5_slice","config4_one_gpu") if k in d))'
for g in $SETTINGS; do
  ( for kv in $(echo "$g" | tr ';' ' '); do export "$kv"; done; timeout -k 10 200 python bench.py --no-cpu-baseline --no-spectrum --no-config1 ${EXTRA:-} --steps 100 --warmup 10 2>/dev/null | python3 -c "$P" "$g" ) || exit 1
done
