#!/bin/bash
# standalone GEMM variants (no torch): VR2_BENCH_V7=1 runs the shipped structure against v7 on the LM / ViT shapes
cd "$(dirname "$0")"
[ -x ./gemm_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -Wno-inline-asm gemm_bench.hip -o gemm_bench
VR2_BENCH_V7=1 ./gemm_bench
