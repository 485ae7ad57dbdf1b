#!/bin/bash
# usage (GPU box, repo root): profiles/tools/pmcset.sh TAG -- rocprofv3 PMC passes (one counter block per pass) over
#   c2  : bench.py headline leg (64 USB receivers, 1.536 Msps)
#   c4  : the configs[3] per-GPU slice (128 mixed receivers, 61.44 Msps), fp64
#   c5  : the same in fp32 mode
#   fft : the spectrum-only loop (2^20-point frames, 8 per launch pair)
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-spectrum --no-config4 --no-config5 --no-config1 --no-one-gpu --steps 3 --warmup 1"
run() { # name counters... -- command
  name=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -s KILL 150 rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $R/gpurun_out/${tag}_${name} -- "$@" > $R/gpurun_out/${tag}_${name}.log 2>&1 || { echo "pass $name failed"; tail -5 $R/gpurun_out/${tag}_${name}.log; exit 1; }
}
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU"
for cfg in c2 c4 c5 fft; do
  case $cfg in
    c2) cmd="$B";;
    c4) cmd="$B --receivers 128 --rate 61440000 --mix mixed";;
    c5) cmd="$B --receivers 128 --rate 61440000 --mix mixed --fp32";;
    fft) cmd="python $R/profiles/tools/fftbench.py 8 4";;
  esac
  run ${cfg}_fetch FETCH_SIZE -- $cmd &&
  run ${cfg}_write WRITE_SIZE -- $cmd &&
  run ${cfg}_tcc TCC_HIT_sum TCC_MISS_sum -- $cmd &&
  run ${cfg}_sq $SQ -- $cmd || exit 1
done
echo pmcset done
