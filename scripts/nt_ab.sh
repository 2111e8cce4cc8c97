#!/bin/bash
# A/B of the streaming hints: builds are prepared on the build host as parelagmc_amd/lib/libpmc_<variant>.so
for v in stores loads; do
  cp parelagmc_amd/lib/libpmc.so /tmp/libpmc_base.so
  cp parelagmc_amd/lib/libpmc_$v.so parelagmc_amd/lib/libpmc.so
  echo "== variant $v"
  bash scripts/k5_traffic.sh || exit 1
  cp /tmp/libpmc_base.so parelagmc_amd/lib/libpmc.so
done
