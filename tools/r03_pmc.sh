#!/bin/bash
# Round-3 counter passes (GPU box): matrix-pipe utilisation + wave-cycle split (+ LDS / co-execution counters where the device
# lists them) for the fp32 headline, the bf16 forward and the cfg-3 training step; FETCH_SIZE / WRITE_SIZE for the headline.
# usage: tools/r03_pmc.sh <tag> [f32|bf16|train|traffic ...]   (default: all)
set -u
tag=$1; shift
what=${*:-"f32 bf16 train traffic"}
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > $out/counters_available.txt 2>&1
set_="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
for c in SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT; do
  grep -qw $c $out/counters_available.txt && set_="$set_ $c"
done
echo "counters: $set_" | tee $out/pmc_set.txt
pass() {   # pass <name> <program args...>
  name=$1; shift
  timeout -k 10 500 rocprofv3 --pmc $set_ --kernel-trace --output-format csv -d $out/pmc_$name -o p -- python3 "$@" > $out/pmc_$name.log 2>&1
  python tools/pmc_mfma.py $out/pmc_$name $out/pmc_mfma_$name.json | head -16 | tee $out/pmc_mfma_$name.txt
  rm -rf $out/pmc_$name
}
for w in $what; do
  case $w in
    f32) pass f32 bench.py --no-graph --no-cpu-baseline --no-secondary --no-roofline --steps 3 ;;
    bf16) pass bf16 bench.py --dtype bf16 --no-graph --no-cpu-baseline --no-secondary --no-roofline --steps 3 ;;
    train) pass train tools/train_step.py --dtype bf16 --batch 64 --steps 2 --warmup 1 ;;
    traffic)
      timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --no-graph --no-cpu-baseline --no-secondary --no-roofline --steps 3 > /dev/null 2>&1
      timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --no-graph --no-cpu-baseline --no-secondary --no-roofline --steps 3 > /dev/null 2>&1
      python tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json > $out/pmc_traffic.txt 2>&1; head -6 $out/pmc_traffic.txt
      rm -rf $out/pmc_fetch $out/pmc_write ;;
  esac
done
