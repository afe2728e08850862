#!/bin/bash
# mid-size f16 GEMM: 128 x 128 kernel vs 256 x 256 kernels (WG_F16_TILE) vs the vendor GEMM, per shape
SH="f16:1024x1024x1024 f16:2048x2048x2048 f16:3072x3072x3072 f16:4096x4096x4096 f16:2048x2048x8192 f16:4096x4096x1024 f16:1536x1536x1536 f16:512x512x4096 f16:4096x1024x4096"
for t in 128 256; do echo "== WG_F16_TILE=$t"; WG_F16_TILE=$t python tools/tall_skinny_probe.py $SH 2>&1 | grep "^f16"; done
