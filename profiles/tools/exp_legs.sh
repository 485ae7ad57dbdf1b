# the bench's slice legs with variant builds of the library (profiles/tools/libqk_<name>.bin through QK_ENGINE_LIB)
P='import json,sys;d=json.loads(sys.stdin.read());print(sys.argv[1], "c2 %.4f" % d["ms_per_step"], " ".join("%s %.4f" % (k, d[k]["ms_per_step"]) for k in ("config4_slice","config5_slice","config4_one_gpu") if k in d))'
for v in $VARIANTS; do
  if [ $v = base ]; then L=$GRAFT_REPO_ROOT/quisk_amd/libqkengine.so; else L=$GRAFT_REPO_ROOT/profiles/tools/libqk_$v.bin; fi
  QK_ENGINE_LIB=$L timeout -k 10 200 python bench.py --no-cpu-baseline --no-spectrum --no-config1 --steps 200 --warmup 20 2>/dev/null | python3 -c "$P" "$v" || exit 1
done
