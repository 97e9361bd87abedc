#!/bin/bash
# Regenerates everything under profiles/ for one round on the GPU box (run through gpurun, then
# `python profiles/make_round_files.py gpurun_out/prof rNN` copies the summaries into profiles/rNN_*).
# Every profiler pass is its own bounded command; --pmc passes carry no trace domain besides --kernel-trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
# 2. kernel traces, one leg per trace; each traced process also prints its own JSON line, so its HIP-event durations and
# rocprofv3's durations of the SAME launches sit side by side in rNN_events_vs_rocprof.txt
T="rocprofv3 --kernel-trace --stats"
timeout 300 $T -d $O/trace_main -o r1 -- $B --steps 20 --warmup 5 --legs main --no-cpu --no-config5 > $O/trace_main.line 2> /dev/null; cp $R/bench_full.json $O/trace_main.json
timeout 300 $T -d $O/trace_main_lat -o r1 -- $B --steps 20 --warmup 5 --legs main --no-cpu --no-config5 --layout lat_fastest > $O/trace_main_lat.line 2> /dev/null; cp $R/bench_full.json $O/trace_main_lat.json
timeout 300 $T -d $O/trace_configs1 -o r1 -- $B --steps 10 --warmup 3 --legs configs1 --no-cpu --no-config5 > $O/trace_configs1.line 2> /dev/null; cp $R/bench_full.json $O/trace_configs1.json
timeout 300 $T -d $O/trace_configs1_lat -o r1 -- $B --steps 10 --warmup 3 --legs configs1 --no-cpu --no-config5 --layout lat_fastest > $O/trace_configs1_lat.line 2> /dev/null; cp $R/bench_full.json $O/trace_configs1_lat.json
timeout 300 $T -d $O/trace_ensemble -o r1 -- $B --steps 10 --warmup 3 --legs ensemble --no-cpu --no-config5 > $O/trace_ensemble.line 2> /dev/null; cp $R/bench_full.json $O/trace_ensemble.json
timeout 300 $T -d $O/trace_public_chunk -o r1 -- $B --steps 40 --warmup 5 --legs public_chunk --no-cpu --no-config5 > $O/trace_public_chunk.line 2> /dev/null; cp $R/bench_full.json $O/trace_public_chunk.json
timeout 300 $T -d $O/trace_public_chunk_lat -o r1 -- $B --steps 40 --warmup 5 --legs public_chunk --no-cpu --no-config5 --layout lat_fastest > $O/trace_public_chunk_lat.line 2> /dev/null; cp $R/bench_full.json $O/trace_public_chunk_lat.json
# (WBX_ALTERNATE_STREAMS=0: the chunk loop deals ensemble launches to two streams, so consecutive kernels overlap by a few tens of
#  microseconds and each one's traced duration is longer than its share of the GPU -- on one stream the trace shows the kernel alone)
# (one kind of launch per traced process: the launch with a mask coordinate writes twelve lanes instead of six and folds the mask
#  into the atom ids first -- its kernel of the same name is ~12 % longer, and one average over both says nothing)
WBX_ALTERNATE_STREAMS=0 timeout 300 $T -d $O/trace_public_chunk_ens -o r1 -- $B --steps 40 --warmup 5 --legs public_chunk_ens --pce-mask 0 --no-cpu --no-config5 > $O/trace_public_chunk_ens.line 2> /dev/null; cp $R/bench_full.json $O/trace_public_chunk_ens.json
WBX_ALTERNATE_STREAMS=0 timeout 300 $T -d $O/trace_public_chunk_ens_lat -o r1 -- $B --steps 40 --warmup 5 --legs public_chunk_ens --pce-mask 0 --no-cpu --no-config5 --layout lat_fastest > $O/trace_public_chunk_ens_lat.line 2> /dev/null; cp $R/bench_full.json $O/trace_public_chunk_ens_lat.json
WBX_ALTERNATE_STREAMS=0 timeout 300 $T -d $O/trace_public_chunk_ens_mask -o r1 -- $B --steps 40 --warmup 5 --legs public_chunk_ens --pce-mask 1 --no-cpu --no-config5 > $O/trace_public_chunk_ens_mask.line 2> /dev/null; cp $R/bench_full.json $O/trace_public_chunk_ens_mask.json
WBX_ALTERNATE_STREAMS=0 timeout 300 $T -d $O/trace_public_chunk_ens_mask_lat -o r1 -- $B --steps 40 --warmup 5 --legs public_chunk_ens --pce-mask 1 --no-cpu --no-config5 --layout lat_fastest > $O/trace_public_chunk_ens_mask_lat.line 2> /dev/null; cp $R/bench_full.json $O/trace_public_chunk_ens_mask_lat.json
WBX_ALTERNATE_STREAMS=0 timeout 300 $T -d $O/trace_public_chunk_ens_nanmask -o r1 -- $B --steps 40 --warmup 5 --legs public_chunk_ens --pce-mask nan --no-cpu --no-config5 > $O/trace_public_chunk_ens_nanmask.line 2> /dev/null; cp $R/bench_full.json $O/trace_public_chunk_ens_nanmask.json
WBX_ALTERNATE_STREAMS=0 timeout 300 $T -d $O/trace_public_chunk_ens_nanmask_lat -o r1 -- $B --steps 40 --warmup 5 --legs public_chunk_ens --pce-mask nan --no-cpu --no-config5 --layout lat_fastest > $O/trace_public_chunk_ens_nanmask_lat.line 2> /dev/null; cp $R/bench_full.json $O/trace_public_chunk_ens_nanmask_lat.json
timeout 300 $T -d $O/trace_spectrum -o r1 -- $B --steps 10 --warmup 3 --legs spectrum --no-cpu --no-config5 > $O/trace_spectrum.line 2> /dev/null; cp $R/bench_full.json $O/trace_spectrum.json
timeout 300 $T -d $O/trace_spectrum_lat -o r1 -- $B --steps 10 --warmup 3 --legs spectrum --no-cpu --no-config5 --layout lat_fastest > $O/trace_spectrum_lat.line 2> /dev/null; cp $R/bench_full.json $O/trace_spectrum_lat.json
timeout 300 $T -d $O/trace_config5 -o r1 -- $B --legs config5 --no-cpu --config5-inits 48 > $O/trace_config5.line 2> /dev/null; cp $R/bench_full.json $O/trace_config5.json
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
for leg in main configs1 ensemble public_chunk public_chunk_ens spectrum; do
  timeout 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_$leg -o r1 -- $B --steps 2 --warmup 1 --legs $leg --pce-mask 0 --no-cpu --no-config5 --prewarm-ms 0 > /dev/null 2>&1
  timeout 250 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_$leg -o r1 -- $B --steps 2 --warmup 1 --legs $leg --pce-mask 0 --no-cpu --no-config5 --prewarm-ms 0 > /dev/null 2>&1
done
for lay in lon_fastest lat_fastest; do
  sfx=$([ $lay = lat_fastest ] && echo _lat || echo "")
  timeout 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_public_chunk_ens_nanmask$sfx -o r1 -- $B --steps 2 --warmup 1 --legs public_chunk_ens --pce-mask nan --no-cpu --no-config5 --prewarm-ms 0 --layout $lay > /dev/null 2>&1
  timeout 250 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_public_chunk_ens_nanmask$sfx -o r1 -- $B --steps 2 --warmup 1 --legs public_chunk_ens --pce-mask nan --no-cpu --no-config5 --prewarm-ms 0 --layout $lay > /dev/null 2>&1
done
for leg in main configs1 public_chunk public_chunk_ens spectrum; do
  timeout 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_${leg}_lat -o r1 -- $B --steps 2 --warmup 1 --legs $leg --pce-mask 0 --no-cpu --no-config5 --prewarm-ms 0 --layout lat_fastest > /dev/null 2>&1
done
DBS=$(ls $O/*/r1_results.db 2>/dev/null)
python $R/profiles/summarize_rocpd.py $DBS > $O/summary.txt 2> $O/summary.err
python - <<PY
import glob, json, sqlite3
out = {}
for db in sorted(glob.glob('$O/pmc_*/r1_results.db')):
  cur = sqlite3.connect(db).cursor()
  try:
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                       "where kernel_name like '%wbx::%' group by 1, 2").fetchall()
  except sqlite3.Error as e:
    out.setdefault('_errors', []).append(f'{db}: {e}')
    continue
  run = db.split('/')[-2]
  leg = run.replace('pmc_fetch_', '').replace('pmc_write_', '')
  lat = leg.endswith('_lat')
  leg = leg[:-4] if lat else leg
  suffix = '@' + leg + ('@lat_fastest' if lat else '@lon_fastest')  # the same kernel has another launch size in another leg
  for k, c, n, v in rows:
    e = out.setdefault(k.replace('void ', '').split('(')[0] + suffix, {})
    if c == 'FETCH_SIZE':
      e.update({'FETCH_SIZE_KiB_avg': v, 'launches': n, 'hbm_read_bytes': v * 1024 * 2, 'fetch_run': run})
    elif c == 'WRITE_SIZE':
      e.update({'WRITE_SIZE_KiB_avg': v, 'hbm_write_bytes': v * 1024, 'write_run': run})
for k, e in out.items():
  if isinstance(e, dict) and 'hbm_read_bytes' in e:
    e['traffic_bytes_per_launch'] = e['hbm_read_bytes'] + e.get('hbm_write_bytes', 0.0)
json.dump(out, open('$O/pmc_raw.json', 'w'), indent=1)
PY
# 1. the bench lines, without any profiler attached (the driver's own settings: --steps 20 --warmup 5). They come AFTER the
# counter passes: bench.py replays profiles/rNN_pmc_traffic.json into roofline.traffic, so that file is written first.
python $R/profiles/make_round_files.py $O ${WBX_ROUND_TAG:-r05} --traffic-only > $O/make_round_files.log 2>&1
timeout 900 $B --steps 20 --warmup 5 > $O/bench_n1.line 2> $O/bench_n1.err; cp $R/bench_full.json $O/bench_n1.json
timeout 600 $B --steps 20 --warmup 5 --layout lat_fastest --no-cpu > $O/bench_n1_lat_fastest.line 2>> $O/bench_n1.err; cp $R/bench_full.json $O/bench_n1_lat_fastest.json
for d in $O/trace_*; do cp $d/r1_kernel_stats.csv $O/$(basename $d)_kernel_stats.csv 2>/dev/null; done
# 4. same-box A/B and ceilings: the ensemble kernel variants, the fused spectra + deterministic kernel, the read stream
( cd $R && python tools/kbench.py ens ) > $O/kbench_ens.txt 2>&1
( cd $R && for layout in lon_fastest lat_fastest; do for v in default $( [ -f $R/weatherbenchx_amd/libwbx_hip_stats64.so ] && echo stats64 ); do if [ $v = default ]; then unset WBX_LIBRARY_PATH; else export WBX_LIBRARY_PATH=$R/weatherbenchx_amd/libwbx_hip_$v.so; fi; for pipe in 1 0; do echo "== $layout library $v WBX_ENS_PIPE=$pipe"; WBX_ENS_PIPE=$pipe python bench.py --legs main --no-cpu --no-config5 --steps 20 --warmup 5 --layout $layout 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read()); print('ms_per_step', round(r['ms_per_step'], 4), 'kernel', r['roofline']['kernel'].split(' (')[0], 'kernel_ms', r['roofline']['kernel_ms'], r['roofline']['kernel_ms_min_max'], 'frac', r['roofline']['frac'])"; done; done; done; unset WBX_LIBRARY_PATH ) > $O/ens_pipe_ab.txt 2>&1
# (kernels these five exercise did not change since r03: WBX_ROUND_FULL=1 re-measures them, make ab-zdlds first)
if [ "${WBX_ROUND_FULL:-0}" = 1 ]; then
( cd $R/tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 flat_fetch.hip -o /tmp/flat_fetch 2>/dev/null && cd /tmp && for shift in 0 16; do /tmp/flat_fetch $shift; timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace -d $O/ff$shift -o r -- /tmp/flat_fetch $shift > /dev/null 2>&1; python -c "
import sqlite3
for k, c, n, v in sqlite3.connect('$O/ff$shift/r_results.db').execute(\"select substr(kernel_name, 1, 44), counter_name, count(*), avg(value) from counters_collection where kernel_name like '%_kernel<%' group by 1, 2\"): print('   shift $shift', k, c, n, '%.0f' % v)"; done; rm -rf $O/ff0 $O/ff16 ) > $O/flat_fetch.txt 2>&1
( cd $R/tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 column_walk.hip -o /tmp/column_walk 2>/dev/null && /tmp/column_walk ) > $O/column_walk.txt 2>&1
( cd $R && bash tools/gpu_r3_ragged_walk.sh ) > $O/ragged_walk.txt 2>&1
( cd $R && bash tools/gpu_r3_ragged_wpb.sh ) > $O/ragged_wpb.txt 2>&1
( cd $R && python tools/kbench_det_spectrum.py; echo "== climatology row through the LDS (make ab-zdlds)"; WBX_LIBRARY_PATH=$R/weatherbenchx_amd/libwbx_hip_zdlds.so python tools/kbench_det_spectrum.py ) > $O/kbench_det_spectrum.txt 2>&1
fi
( cd $R/tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 read_stream.hip -o read_stream 2>/dev/null; timeout 120 ./read_stream ) > $O/read_stream.json 2>&1
( cd $R && bash tools/pmc_ens.sh ) > $O/pmc_ens.txt 2>&1
( cd $R && for s in "10 3" "20 5" "50 20" "200 50"; do set -- $s; python bench.py --legs main --no-cpu --no-config5 --steps $1 --warmup $2 --prewarm-ms 0 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read()); print('no prewarm, steps $1 warmup $2: ms_per_step', round(r['ms_per_step'], 4), 'kernel_ms', r['roofline']['kernel_ms'])"; done; python bench.py --legs main --no-cpu --no-config5 --steps 20 --warmup 5 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read()); print('prewarm 150 ms, steps 20 warmup 5: ms_per_step', round(r['ms_per_step'], 4), 'kernel_ms', r['roofline']['kernel_ms'], r['config']['prewarm'])" ) > $O/steady_state.txt 2>&1
# 5. only the summaries travel back (gpurun merges at most 64 MiB): the raw rocprofv3 databases stay on the box
python $R/profiles/make_round_files.py $O ${WBX_ROUND_TAG:-r05} >> $O/make_round_files.log 2>&1
mkdir -p $R/gpurun_out/round && cp $R/profiles/${WBX_ROUND_TAG:-r05}_* $R/gpurun_out/round/
for f in steady_state ens_pipe_ab kbench_det_spectrum flat_fetch column_walk ragged_walk ragged_wpb; do [ -f $O/$f.txt ] && cp $O/$f.txt $R/gpurun_out/round/${WBX_ROUND_TAG:-r05}_$f.txt; done
rm -rf $O/trace_* $O/pmc_fetch_* $O/pmc_write_* $R/gpurun_out/pmc_ens $R/gpurun_out/pmc_spec
find $O -maxdepth 1 -type d -name "*" | sed -n 2,100p | xargs -r rm -rf
du -sh $R/gpurun_out
