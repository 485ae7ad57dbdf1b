# headline period with parts of the block left out (QK_DEBUG_SKIP bits), to see what each part costs the front end
B="python bench.py --no-cpu-baseline --no-spectrum --no-config4 --no-config5 --no-config1 --no-one-gpu --steps 200 --warmup 20"
P='import json,sys;d=json.loads(sys.stdin.read());print(sys.argv[1], d["value"],d["ms_per_step"],d["roofline"]["avg_launch_ms"])'
for sk in ${SKIPS:-0 1 2 3}; do QK_DEBUG_SKIP=$sk timeout -k 10 100 $B 2>/dev/null | python3 -c "$P" "skip$sk" || exit 1; done
