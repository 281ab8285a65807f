#!/bin/bash
# State of the GPU a gpurun call landed on (clocks, perf level, power cap, partition modes): printed next to every
# measurement of this round so box-to-box differences can be attributed.
S=/opt/rocm/bin/rocm-smi
$S --showperflevel --showclocks --showpower --showmaxpower --showmemorypartition --showcomputepartition --showtemp 2>&1 | grep -v "^=\|^$" | head -60
$S --showuse --showmemuse 2>&1 | grep -v "^=\|^$" | head
cat /sys/class/drm/card*/device/power_dpm_force_performance_level 2>/dev/null | head -2
nproc; lscpu | grep -i "model name\|numa node(s)\|^CPU(s)" 
hostname; $S --showuniqueid --showserial --showvbios --showdriverversion --showfwinfo 2>&1 | grep -v "^=\|^$" | head -40
cat /proc/cmdline; uname -r
cat /sys/module/amdgpu/parameters/vm_fragment_size /sys/module/amdgpu/parameters/vm_block_size /sys/module/amdgpu/parameters/vm_size /sys/module/amdgpu/parameters/noretry 2>/dev/null | tr '\n' ' '; echo
cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null
