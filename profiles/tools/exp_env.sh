# headline leg under a list of environment settings: SETTINGS="A=1;B=2 C=3" (space-separated groups, ; inside a group)
B="python bench.py --no-cpu-baseline --no-spectrum --no-config4 --no-config5 --no-config1 --no-one-gpu --steps 200 --warmup 20"
P='import json,sys;d=json.loads(sys.stdin.read());print(sys.argv[1], d["value"],d["ms_per_step"],d["roofline"]["avg_launch_ms"])'
for g in $SETTINGS; do
  ( for kv in $(echo "$g" | tr ';' ' '); do export "$kv"; done; timeout -k 10 100 $B 2>/dev/null | python3 -c "$P" "$g" ) || exit 1
done
