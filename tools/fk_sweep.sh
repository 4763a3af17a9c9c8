#!/bin/bash
# fk positions-only kernel variants on the GPU box: GMR_AMD_FK_PARTS = 0 (grouped flush, fk_kernel<0>), 1 (whole tile image), 2 (two halves)
for p in 0 1 2; do echo "parts=$p"; GMR_AMD_FK_PARTS=$p python tools/fk_bench.py unitree_g1 ${1:-24576000} 2>/dev/null; done
