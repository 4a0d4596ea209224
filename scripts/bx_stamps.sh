#!/bin/bash
# The bf16-exact loop on 8 waves with one cost removed per build (general scales: bx*, power-of-two scales: bu8*), 4096^3.
cd "$(dirname "$0")/ubench" || exit 1
for v in bx128x256 bx_NODMA bx_NOLDS bx_NODMA_NOLDS bx_NOCVT bx_NOFMA bx_NOCVT_NOFMA bx_ALL bu128x256w8 bu8_NODMA bu8_NOLDS bu8_NODMA_NOLDS bu8_NOCVT bu8_ALL; do
  echo "== $v"
  timeout -k 10 120 ./stamp_tile_$v 4096 4096 4096 1500 || exit 1
done
