#!/bin/bash
# round 4: tapered row splits of the ensemble atom kernel (patch_taper) against uniform splits, same box; parity first
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ens_binned.py tests/test_gpu_round4.py -m gpu -x -q -k "not slabs and not latitude_fastest_chunks" 2>&1 | tail -3
for rep in 1 2; do
for t in 1 0; do
  for layout in lon_fastest lat_fastest; do
    WBX_ENS_ATOMS_TAPER=$t bash tools/trace_ens_binned.sh $layout 2>&1 | grep "ens_atoms\|ens_pipe" | sed "s/^/taper $t /"
  done
done
done
for rows in 12 24 32; do
  WBX_ENS_ATOMS_ROWS=$rows bash tools/trace_ens_binned.sh lon_fastest lat_fastest 2>&1 | grep "ens_atoms" | sed "s/^/taper 1 rows $rows /"
done
