export TMPDIR=/tmp
for rows in 16 15 22 11 45; do
  WBX_ENS_ATOMS_ROWS=$rows bash tools/trace_ens_binned.sh lon_fastest 2>&1 | grep "ens_atoms" | sed "s/^/rows $rows /"
done
for rows in 16 15 23 12 45; do
  WBX_ENS_ATOMS_ROWS=$rows bash tools/trace_ens_binned.sh lat_fastest 2>&1 | grep "ens_atoms" | sed "s/^/rows $rows /"
done
