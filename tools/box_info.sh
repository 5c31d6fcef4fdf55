#!/bin/bash
# What distinguishes one box of the pool from another: driver / firmware versions, partition modes, clocks, memory-side settings.
# Read-only queries; output to stdout (gpu_session.sh stage `boxinfo` stores it as boxinfo.txt).
echo "## uname"; uname -r
echo "## amdgpu module"; cat /sys/module/amdgpu/version 2>/dev/null; cat /sys/module/amdgpu/srcversion 2>/dev/null
for p in vm_fragment_size vm_block_size noretry mtype_local sched_policy hws_max_conc_proc mes cwsr_enable; do
  f=/sys/module/amdgpu/parameters/$p; [ -r $f ] && echo "param $p = $(cat $f)"; done
echo "## rocm-smi partitions / fw / clocks"
rocm-smi --showcomputepartition --showmemorypartition 2>/dev/null | grep -v "^$" | head -20
rocm-smi --showfwinfo 2>/dev/null | grep -v "^$" | head -40
rocm-smi --showclocks --showperflevel --showpower 2>/dev/null | grep -v "^$" | head -30
rocm-smi --showmeminfo vram --showuse --showmemuse 2>/dev/null | grep -v "^$" | head -20
rocm-smi --showpids 2>/dev/null | grep -v "^$" | head -20
echo "## kfd topology (GPU nodes)"
for n in /sys/class/kfd/kfd/topology/nodes/*; do
  if grep -q "simd_count [1-9]" $n/properties 2>/dev/null; then
    echo "node $n"; grep -E "simd_count|cu_count|max_engine_clk|num_xcc|gfx_target|sdma|local_mem_size|unique_id|drm_render_minor|num_cp_queues|fw_version|device_id|location_id" $n/properties
    for c in $n/caches/*; do echo "cache $(grep -E '^level|^size |^type' $c/properties | tr '\n' ' ')"; done | sort | uniq -c
    for m in $n/mem_banks/*; do grep -E "size_in_bytes|width|mem_clk" $m/properties | tr '\n' ' '; echo; done
  fi
done
echo "## how many GPU nodes are visible to kfd (other tenants on the same host?)"
grep -l "simd_count [1-9]" /sys/class/kfd/kfd/topology/nodes/*/properties 2>/dev/null | wc -l
echo "## env"; env | grep -E "^HSA|^HIP|^ROC|^GPU|^AMD|^NCCL|^RCCL" | sort
echo "## host"; nproc; cat /proc/loadavg; grep -E "MemTotal|HugePages_Total|AnonHugePages" /proc/meminfo; cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null
