#!/bin/bash
# host only: what a 16 GiB output file costs on the GPU box (tools/ubench/bigfile_write.cpp), and the limits that may explain it
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/${1:-r5big}; mkdir -p $D; O=$D/bigfile_write.txt
g++ -O2 -std=c++17 -pthread tools/ubench/bigfile_write.cpp -o /tmp/bigfile_write || exit 1
{
echo "# $(date -u) $(hostname)"; df -h /tmp /dev/shm | sed 's/^/# /'; grep -E "MemTotal|MemAvailable|Dirty|Writeback:" /proc/meminfo | sed 's/^/# /'
echo "# cgroup memory.max $(cat /sys/fs/cgroup/memory.max 2>/dev/null) memory.high $(cat /sys/fs/cgroup/memory.high 2>/dev/null) cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
echo "# vm.dirty_ratio $(cat /proc/sys/vm/dirty_ratio) dirty_background_ratio $(cat /proc/sys/vm/dirty_background_ratio) dirty_bytes $(cat /proc/sys/vm/dirty_bytes) dirty_background_bytes $(cat /proc/sys/vm/dirty_background_bytes)"
mount | grep -E " / | /tmp " | sed 's/^/# /'
for m in plain sfr sfr_drop; do timeout 300 /tmp/bigfile_write /tmp 16 $m 2; done
timeout 300 /tmp/bigfile_write /dev/shm 8 plain 2
timeout 300 /tmp/bigfile_write /tmp 16 plain 1
} > $O 2>&1
cat $O
