R=$GRAFT_REPO_ROOT; cd /tmp
echo "nproc $(nproc)  python cpu_count $(python -c 'import os; print(os.cpu_count(), len(os.sched_getaffinity(0)))')"
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null
grep -i "cpus_allowed_list\|threads" /proc/self/status; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8; lscpu | grep -E "^CPU\(s\)|Model name|Socket|Thread|NUMA node\(s\)"
cat /proc/loadavg; free -g | head -2
