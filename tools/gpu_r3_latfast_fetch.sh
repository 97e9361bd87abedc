#!/bin/bash
# Where do the extra 11 % of FETCH_SIZE of the flat one-point-per-lane ensemble sweep (latitude-fastest main line) come from?
# Counter passes (own runs, --kernel-trace only) and HIP-event timings of geometry / load-hint variants on ONE box.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/latfetch; rm -rf $O; mkdir -p $O
B="python $R/bench.py --legs main --no-cpu --no-config5"
one() {  # name, layout, env assignments...
  name=$1; layout=$2; shift 2
  env "$@" timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${name}_fetch -o r -- $B --layout $layout --steps 3 --warmup 1 --prewarm-ms 0 > /dev/null 2>&1
  env "$@" timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/${name}_ea -o r -- $B --layout $layout --steps 3 --warmup 1 --prewarm-ms 0 > /dev/null 2>&1
  env "$@" timeout 200 $B --layout $layout --steps 20 --warmup 5 > $O/$name.json 2> $O/$name.err
}
for v in ${VARIANTS:-lat_default lat_threads64 lat_threads128 lat_bigchunks lat_plainld lon_default}; do
  case $v in
    lat_default) one $v lat_fastest WBX_X=0;;
    lat_threads64) one $v lat_fastest WBX_FLAT1_THREADS=64;;
    lat_threads128) one $v lat_fastest WBX_FLAT1_THREADS=128;;
    lat_bigchunks) one $v lat_fastest WBX_FLAT1_MIN_ELEMENTS=30000;;
    lat_plainld) one $v lat_fastest WBX_LIBRARY_PATH=$R/weatherbenchx_amd/libwbx_hip_plainld.so;;
    lon_default) one $v lon_fastest WBX_X=0;;
  esac
done
python - <<PY
import glob, json, sqlite3
for name in '${VARIANTS:-lat_default lat_threads64 lat_threads128 lat_bigchunks lat_plainld lon_default}'.split():
  line = name + ':'
  try:
    r = json.loads([l for l in open('$O/%s.json' % name).read().split('\n') if l.startswith('{')][-1])
    line += ' ms_per_step %.4f kernel_ms %.4f (median %.4f) %s' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline'].get('kernel_ms_median', 0), r['roofline']['kernel'].split(' (')[0])
  except Exception as e:
    line += ' (no bench line: %r)' % (e,)
  print(line)
  for db in sorted(glob.glob('$O/%s_*/r_results.db' % name)):
    try:
      rows = sqlite3.connect(db).execute("select substr(kernel_name, 1, 50), counter_name, count(*), avg(value) from counters_collection where kernel_name like '%wbx::%ens%' or kernel_name like '%EnsOp%' group by 1, 2").fetchall()
    except sqlite3.Error as e:
      rows = [('error', str(e), 0, 0.0)]
    for k, c, n, v in rows:
      print('   ', k, c, n, '%.1f' % v)
PY
rm -rf $O/*_fetch $O/*_ea
