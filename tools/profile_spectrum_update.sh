#!/bin/bash
# A partial profile round after a change that only touches the 1440-point spectrum kernels: kernel traces of the spectrum legs
# and of the configs[4] job, and the two bench lines (no profiler attached) -> gpurun_out/spec_update/;
# `python profiles/merge_partial_round.py gpurun_out/spec_update r03` then replaces those runs' rows in the round's files.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/spec_update
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
T="rocprofv3 --kernel-trace --stats"
timeout 200 $T -d $O/trace_spectrum -o r1 -- $B --steps 10 --warmup 3 --legs spectrum --no-cpu --no-config5 > $O/trace_spectrum.json 2> /dev/null
timeout 200 $T -d $O/trace_spectrum_lat -o r1 -- $B --steps 10 --warmup 3 --legs spectrum --no-cpu --no-config5 --layout lat_fastest > $O/trace_spectrum_lat.json 2> /dev/null
timeout 200 $T -d $O/trace_config5 -o r1 -- $B --legs config5 --no-cpu --config5-inits 48 > $O/trace_config5.json 2> /dev/null
python $R/profiles/summarize_rocpd.py $O/trace_*/r1_results.db > $O/summary.txt 2> $O/summary.err
rm -rf $O/trace_spectrum $O/trace_spectrum_lat $O/trace_config5
timeout 400 $B --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 200 $B --steps 20 --warmup 5 --layout lat_fastest --no-cpu > $O/bench_n1_lat_fastest.json 2>> $O/bench_n1.err
( cd $R && python tools/kbench_det_spectrum.py ) > $O/kbench_det_spectrum.txt 2>&1
ls -la $O; tail -3 $O/summary.err; head -c 300 $O/bench_n1.json
