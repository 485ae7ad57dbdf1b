# headline period with variant builds of the library (profiles/tools/libqk_<name>.bin through QK_ENGINE_LIB)
B="python bench.py --no-cpu-baseline --no-spectrum --no-config4 --no-config5 --no-config1 --no-one-gpu --steps 200 --warmup 20"
P='import json,sys;d=json.loads(sys.stdin.read());print(sys.argv[1], d["value"],d["ms_per_step"],d["roofline"]["avg_launch_ms"])'
for v in $VARIANTS; do
  if [ $v = base ]; then L=$GRAFT_REPO_ROOT/quisk_amd/libqkengine.so; else L=$GRAFT_REPO_ROOT/profiles/tools/libqk_$v.bin; fi
  QK_ENGINE_LIB=$L timeout -k 10 100 $B 2>/dev/null | python3 -c "$P" "$v" || exit 1
done
