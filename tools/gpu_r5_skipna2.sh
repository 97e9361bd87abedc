#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q -k "skipna or ensemble or ens_ or golden or metrics or corners" 2>&1 | tail -8 ) > gpurun_out/r5_pytest_skipna2.log
tail -5 gpurun_out/r5_pytest_skipna2.log
for lay in lon_fastest lat_fastest; do
python bench.py --legs ensemble --no-cpu --no-config5 --steps 3 --warmup 1 --layout $lay > /dev/null 2>gpurun_out/ens_leg.err; python - <<PY
import json
d=json.load(open("bench_full.json"))
e=d["ensemble"]
print("$lay", "ensemble", e["roofline"]["kernel_ms"], e["roofline"]["frac"], "| skipna", e["skipna_ensemble"]["roofline"]["kernel_ms"], e["skipna_ensemble"]["roofline"]["frac"], e["skipna_ensemble"]["roofline"]["kernel"][:40], e["skipna_ensemble"]["crps"])
PY
done
