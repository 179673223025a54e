#!/bin/bash
# Everything the round's profiles/ entries come from, one gpurun call:   tools/profile_round.sh <tag>   ->  gpurun_out/<tag>/
#   kernel_stats.txt          rocprofv3 --kernel-trace --stats of 4 training steps (bench.py --no-extras), summarised per kernel
#   pmc_step.txt, traffic.json   three --pmc passes over one step (tools/pmc_step.sh; counters never combined with traces)
#   sampler_kernel_stats_{bf16,f32,x3}.txt   kernel trace of two 50-step sampler calls per mode
#   sampler_parity.json       tests/test_sampler50.py: each mode against the reference's 50-step run at B=4, L=1115
#   bench.json                the bench line itself (run last, unprofiled)
tag=${1:-prof}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $out/kt -o res -- python3 bench.py --steps 4 --warmup 1 --no-extras > $out/kt.log 2>&1
python3 tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) > $out/kernel_stats.txt 2>&1
bash tools/pmc_step.sh $out/pmc
cp $out/pmc/pmc_step.txt $out/pmc_step.txt; cp $out/pmc/traffic.json $out/traffic.json
for mode in bf16 f32 x3; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/s_$mode -o res -- python3 tools/sampler_prof.py $mode > $out/s_$mode.log 2>&1
  python3 tools/rocpd_stats.py $(find $out/s_$mode -name "*.db" | head -1) > $out/sampler_kernel_stats_$mode.txt 2>&1
done
rm -rf $out/kt $out/pmc/p1 $out/pmc/p2 $out/pmc/p3 $out/s_bf16 $out/s_f32 $out/s_x3
# the 50-step sampler against the reference's own run (writes gpurun_out/sampler_parity.json, tagged with the library's source hash)
timeout 600 python3 -m pytest tests/test_sampler50.py -m gpu -q > $out/sampler50_test.txt 2>&1; cp gpurun_out/sampler_parity.json $out/sampler_parity.json
# the bench line quotes the counter / parity records of THIS build: put them where bench.py looks (copy the same files into the
# repository's profiles/ afterwards; a record whose kernel_src_sha differs from the library's is ignored by bench.py)
round=${2:-r04}
cp $out/traffic.json profiles/${round}_traffic.json; cp $out/sampler_parity.json profiles/${round}_sampler_parity.json
timeout 900 python3 bench.py > $out/bench.json 2> $out/bench.err
tail -c 600 $out/bench.json
