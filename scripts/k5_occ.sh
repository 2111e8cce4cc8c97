#!/bin/bash
# K5 isolated: time + FETCH_SIZE against the number of resident workgroups per CU (limited by unused dynamic LDS)
for lds in 0 41000 54000 65000; do
echo "== dynamic LDS $lds bytes per workgroup"
PMC_K5_LDS=$lds bash scripts/k5_traffic.sh || exit 1
done
