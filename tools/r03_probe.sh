#!/bin/bash
# Round-3 experiment pass (GPU box): per-kernel stats of the folded / un-folded forward, bench lines.
# usage: tools/r03_probe.sh <tag> [quick]
set -u
tag=$1
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
prof() {   # prof <name> [ENV=VAL ...]: rocprofv3 kernel stats of the default bench under the given environment
  name=$1; shift
  ( export "$@" DUMMY=1; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -o b -- python3 bench.py --no-cpu-baseline --no-secondary --no-roofline > $out/bench_prof_$name.json 2>/dev/null )
  cp $(find $out/prof_$name -name "*kernel_stats.csv" | head -1) $out/kernel_stats_$name.csv && rm -rf $out/prof_$name
  cut -c1-160 $out/bench_prof_$name.json
}
python -m pytest tests/test_lnfold_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -x -q > $out/tests_a.log 2>&1; tail -4 $out/tests_a.log
for v in "" "CONFORMER_AMD_LN_FOLD=0"; do
  ( [ -n "$v" ] && export $v; python bench.py --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | cut -c1-220 )
done
prof fold
prof nofold CONFORMER_AMD_LN_FOLD=0
python bench.py --dtype bf16 --no-cpu-baseline 2>/dev/null | cut -c1-220
python -m pytest tests -m gpu -q > $out/tests_all.log 2>&1; tail -4 $out/tests_all.log
