# headline period against the front end's span length (QK_FE_SPAN, samples of input per workgroup)
B="python bench.py --no-cpu-baseline --no-spectrum --no-config4 --no-config5 --no-config1 --no-one-gpu --steps 200 --warmup 20"
P='import json,sys;d=json.loads(sys.stdin.read());print(sys.argv[1], d["value"],d["ms_per_step"],d["roofline"]["avg_launch_ms"])'
for sp in ${SPANS:-1536 2304 3072}; do QK_DEBUG_SKIP=${SKIP:-0} QK_FE_SPAN=$sp timeout -k 10 100 $B 2>/dev/null | python3 -c "$P" "span$sp" || exit 1; done
