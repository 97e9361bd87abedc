set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5_bench_replay_e.txt
: > $OUT
for w in ens ens_mask ens_nan det; do
  timeout 300 python tools/bench_replay.py $w lon_fastest n=150 >> $OUT 2>&1
done
timeout 300 python tools/bench_replay.py ens lat_fastest n=150 >> $OUT 2>&1
timeout 300 python tools/bench_replay.py spec lon_fastest n=100 >> $OUT 2>&1
grep -v amdgpu.ids $OUT | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): print(l[:300]); continue
    d=json.loads(l)
    print('%-8s %-11s 1-stream ord %s rep %s | 2-stream ord %s rep %s | kernel %.4f %s %s'%(d['which'],d['layout'],d['ordinary_one_stream_ms_per_chunk'],d['replay_one_stream_ms_per_chunk'],d['ordinary_ms_per_chunk'],d['replay_ms_per_chunk'],d['kernel_ms_per_chunk'],d['stats'],d['refusals'][:1]))
"
