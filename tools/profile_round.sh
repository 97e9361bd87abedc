#!/bin/bash
# Regenerates everything under profiles/ for one round on the GPU box (run through gpurun, then copy
# gpurun_out/prof/{summary.txt,pmc_traffic.json,bench_*.json,*_kernel_stats.csv} into profiles/rNN_*).
# Every profiler pass is its own bounded command; --pmc passes carry no trace domain besides --kernel-trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the bench lines, without any profiler attached
timeout 600 python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 300 python $R/bench.py --layout lat_fastest --no-cpu > $O/bench_n1_lat_fastest.json 2>> $O/bench_n1.err
# 2. kernel traces (durations that bench.py's HIP-event figures must agree with)
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_bench -o r1 -- python $R/bench.py --steps 5 --warmup 2 --legs main,ensemble,public_chunk,spectrum > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_rmse_crps_37L -o r1 -- python $R/bench.py --steps 5 --warmup 2 --legs rmse_crps_37L > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_config5 -o r1 -- python $R/bench.py --legs config5 --config5-inits 48 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_bench_lat -o r1 -- python $R/bench.py --layout lat_fastest --steps 5 --warmup 2 --legs main > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_public_chunk -o r1 -- python $R/tools/kbench_binned.py lat_fastest 5 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_ens_regions -o r1 -- python $R/tools/bench_ens_regions.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_spectrum_lat -o r1 -- python $R/bench.py --layout lat_fastest --steps 5 --warmup 2 --legs spectrum > /dev/null 2>&1
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
timeout 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o r1 -- python $R/bench.py --steps 2 --warmup 1 --legs main,ensemble,public_chunk,spectrum > /dev/null 2>&1
timeout 250 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o r1 -- python $R/bench.py --steps 2 --warmup 1 --legs main,ensemble,public_chunk,spectrum > /dev/null 2>&1
timeout 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_lat -o r1 -- python $R/bench.py --layout lat_fastest --steps 2 --warmup 1 --legs main > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_public_chunk -o r1 -- python $R/tools/kbench_binned.py lat_fastest 3 > /dev/null 2>&1
timeout 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_37L -o r1 -- python $R/bench.py --steps 2 --warmup 1 --legs rmse_crps_37L > /dev/null 2>&1
timeout 250 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_37L -o r1 -- python $R/bench.py --steps 2 --warmup 1 --legs rmse_crps_37L > /dev/null 2>&1
timeout 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_spectrum_lat -o r1 -- python $R/bench.py --layout lat_fastest --steps 2 --warmup 1 --legs spectrum > /dev/null 2>&1
timeout 250 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_spectrum_lat -o r1 -- python $R/bench.py --layout lat_fastest --steps 2 --warmup 1 --legs spectrum > /dev/null 2>&1
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
  for k, c, n, v in rows:
    run = db.split('/')[-2]
    suffix = '@37L' if run.endswith('_37L') else ''  # (the north_star field: same kernel, its own launch size)
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
for d in trace_bench trace_bench_lat trace_rmse_crps_37L trace_config5 trace_public_chunk trace_ens_regions trace_spectrum_lat; do cp $O/$d/r1_kernel_stats.csv $O/${d}_kernel_stats.csv 2>/dev/null; done
# 4. the same-box read ceiling, and the SQ / memory-side counters of the ensemble and binned kernels (own --pmc passes)
( cd $R/tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 read_stream.hip -o read_stream 2>/dev/null; timeout 120 ./read_stream ) > $O/read_stream.json 2>&1
( cd $R/tools/ubench && for u in valu_rates valu_ops lds_rates load_patterns clock_rate; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $u.hip -o $u 2>/dev/null; echo "== $u"; timeout 120 ./$u; done ) > $O/ubench.txt 2>&1
( cd $R && bash tools/pmc_ens.sh ) > $O/pmc_ens.txt 2>&1
( cd $R && bash tools/pmc_spectrum.sh ) > $O/pmc_spectrum.txt 2>&1
( cd $R && bash tools/pmc_spec_latfast.sh ) > $O/pmc_spectrum_lat_fastest.txt 2>&1
( cd $R && python tools/spec_phase_profile.py 2>&1 | grep -v amdgpu.ids ) > $O/spectrum_phase_profile.txt 2>&1
( cd $R && for l in lon_fastest lat_fastest; do python tools/kbench_spectrum_raw.py 8 $l sorted 2>&1 | grep -v amdgpu.ids | sed "s/^/$l /"; done ) > $O/spectrum_raw.txt 2>&1
( cd $R && python tools/kbench.py ens 2>&1 | grep -v amdgpu.ids ) > $O/kbench_ens.txt 2>&1
( cd $R && python tools/config5_host_split.py 150 2>&1 | grep -v amdgpu.ids | tail -7 ) > $O/config5_host_split.txt 2>&1
( cd $R && python tools/bench_new_labels.py 2>&1 | grep -v amdgpu.ids | tail -2 ) > $O/new_time_labels.txt 2>&1
( cd $R && bash tools/pmc_binned.sh lon_fastest | grep -v rocprofv3 ) > $O/pmc_binned_lon_fastest.txt 2>&1
( cd $R && bash tools/pmc_binned.sh lat_fastest | grep -v rocprofv3 ) > $O/pmc_binned_lat_fastest.txt 2>&1
rm -rf $R/gpurun_out/pmc_ens $R/gpurun_out/pmc_spec $R/gpurun_out/pmc_binned_lon_fastest $R/gpurun_out/pmc_binned_lat_fastest
rm -rf $O/*/  # the databases stay on the box; only the summaries travel back
ls -la $O
