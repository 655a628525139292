#!/bin/bash
# builds tools/bin/gram_bench_k{0,1,2,3,4}: the Gram harness against kernels_gram_wave.hip compiled with -DSI_GW_KNOB=k
set -e
cd "$(dirname "$0")/.."
C=subspaceinference.jl_amd/csrc
mkdir -p tools/bin /tmp/gwk
for k in 0 1 2 3 4 16; do
  for p in 0 1 2; do
    [ -f /tmp/gwk/gw${p}_k$k.o ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSI_GW_PART=$p -DSI_GW_KNOB=$k -c $C/kernels_gram_wave.hip -o /tmp/gwk/gw${p}_k$k.o &
  done
  wait
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c tools/gram_bench.hip -o /tmp/gwk/bench.o
for k in 0 1 2 3 4 16; do
  hipcc --offload-arch=gfx950 /tmp/gwk/bench.o /tmp/gwk/gw0_k$k.o /tmp/gwk/gw1_k$k.o /tmp/gwk/gw2_k$k.o -o tools/bin/gram_bench_k$k
done
ls -la tools/bin/gram_bench_k*
