#!/bin/bash
# what kind of box is this?  (partition modes, clocks, power cap) -- to correlate with the fast / slow behaviour of the fused kernel
mkdir -p gpurun_out
{
echo "== host"; hostname; nproc; 
echo "== rocm-smi partitions"; rocm-smi --showmemorypartition --showcomputepartition 2>&1 | grep -v "^=\|^$" | head -10
echo "== clocks"; rocm-smi --showclocks 2>&1 | grep -v "^=\|^$" | head -12
echo "== power"; rocm-smi --showpower --showmaxpower 2>&1 | grep -v "^=\|^$" | head -8
echo "== perf level / temp"; rocm-smi --showperflevel --showtemp 2>&1 | grep -v "^=\|^$" | head -10
echo "== vbios / fw"; rocm-smi --showvbios --showdriverversion 2>&1 | grep -v "^=\|^$" | head -6
echo "== memory"; rocm-smi --showmeminfo vram 2>&1 | grep -v "^=\|^$" | head -4
echo "== rocminfo"; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock|Cacheline|L2|L3|Uuid" | head -20
} > gpurun_out/boxinfo.txt 2>&1
cat gpurun_out/boxinfo.txt
