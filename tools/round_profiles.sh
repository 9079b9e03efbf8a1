#!/bin/bash
# Round-end measurement pass on the GPU box: writes everything under gpurun_out/$1/ (copied into profiles/ afterwards).
# usage: tools/round_profiles.sh <tag> <part: a|b>
set -u
tag=$1; part=$2
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(while true; do date >> $out/heartbeat.txt; sleep 30; done) & HB=$!
trap "kill $HB" EXIT
if [ "$part" = "a" ]; then
  python bench.py > $out/bench.json 2> $out/bench.err && cut -c1-300 $out/bench.json
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -o bench -- python3 bench.py --no-cpu-baseline --no-secondary > $out/bench_under_rocprof.json 2>/dev/null
  cp $(find $out/prof_bench -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --no-graph --no-cpu-baseline --no-secondary --no-roofline --steps 3 > /dev/null 2>&1
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --no-graph --no-cpu-baseline --no-secondary --no-roofline --steps 3 > /dev/null 2>&1
  python tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json > $out/pmc_traffic.txt 2>&1; head -4 $out/pmc_traffic.txt
  python bench.py --dtype bf16 --no-cpu-baseline > $out/bf16_bench.json 2>/dev/null && cut -c1-200 $out/bf16_bench.json
  python tools/kernel_table.py > $out/kernel_table.md 2>/dev/null; grep -c "|" $out/kernel_table.md
else
  for cfg in "f32 32" "bf16 32" "bf16 64" "f16 32"; do set -- $cfg; python tools/train_step.py --dtype $1 --batch $2 --steps 5 2>/dev/null >> $out/train_steps.jsonl; done; cut -c1-260 $out/train_steps.jsonl
  python bench.py --train --steps 5 --warmup 2 > $out/train_bench.json 2> $out/train_bench.err; cut -c1-300 $out/train_bench.json
  python tools/decoder_bench.py 32 > $out/decoder_bench.json 2>/dev/null; python tools/decoder_bench.py 64 >> $out/decoder_bench.json 2>/dev/null
  (python tools/gemm16_sites.py 7968; python tools/gemm16_sites.py 15936) > $out/gemm16_sites.txt 2>/dev/null
  python tools/dw16_probe.py 2>/dev/null | grep dW > $out/dw16_probe.txt
  python tools/lstm_probe.py 2>/dev/null | grep -v amdgpu > $out/lstm_probe.txt
  python tools/attn_probe.py trace 2>/dev/null | grep -v amdgpu > $out/attn_fwd_trace.txt
  python tools/attn_bwd_bench.py > $out/attn_bwd_bench.txt 2>/dev/null
  python tools/ln_bwd_probe.py 2>/dev/null | grep rows > $out/ln_bwd_probe.txt
  python tools/streaming_bench.py > $out/streaming_bench.json 2>/dev/null
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_train -o train -- python3 tools/train_step.py --dtype bf16 --batch 64 --steps 2 --warmup 1 > /dev/null 2>&1
  cp $(find $out/prof_train -name "*kernel_stats.csv" | head -1) $out/train_bf16_b64_kernel_stats.csv
  ls $out
fi
