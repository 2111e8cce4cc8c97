#!/bin/bash
# A/B of the SpMM workgroup size: variants are prepared on the build host as parelagmc_amd/lib/libpmc_kb<N>.so
cp parelagmc_amd/lib/libpmc.so /tmp/libpmc_base.so
for v in 512 1024; do
  cp parelagmc_amd/lib/libpmc_kb$v.so parelagmc_amd/lib/libpmc.so
  echo "== kBlock $v"
  bash scripts/k5_traffic.sh || { cp /tmp/libpmc_base.so parelagmc_amd/lib/libpmc.so; exit 1; }
done
cp /tmp/libpmc_base.so parelagmc_amd/lib/libpmc.so
