"""Ad-hoc: how much does the choice of cores move the measurement on this host? (run on the MI355X box)"""
import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
from nuts333_amd import workloads
from nuts333_amd.talker import REF_BINARY
print(subprocess.run("lscpu -e=CPU,CORE,SOCKET,NODE,CACHE | awk 'NR==1 || $1%8==0 || $1==1 || $1==129'", shell=True, capture_output=True, text=True).stdout)
print(open('/proc/loadavg').read())
all_cpus = sorted(os.sched_getaffinity(0))
t0 = time.time()
for base in (0, 8, 16, 24, 32, 64, 72, 128):
    os.sched_setaffinity(0, set(range(base, base + 8)) & set(all_cpus))
    r2 = workloads.config2(lines=20000, warmup=1000, binary=REF_BINARY)
    r4 = workloads.config4(lines=500, warmup=20, binary=REF_BINARY)
    p = subprocess.run([str(workloads.LOADGEN_BIN), "--probe-line", "67", "9", "20000", "1", "0", "4", str(base), ",".join(map(str, range(base + 1, base + 5)))], capture_output=True, text=True).stdout
    import json; p = json.loads(p)
    print(f"{time.time()-t0:6.1f}s cpus {base:3d}-{base+7:3d}: config2 {r2['delivered_lines_per_s']/1e3:6.0f}k {r2['servers'][0]['cpu_us_per_written_line']:.3f} us | "
          f"config4 {r4['delivered_lines_per_s']/1e3:6.0f}k {r4['servers'][0]['cpu_us_per_written_line']:.3f} us | probe k=9 full closed {p['cpu_ns_per_written_line']:.0f} ns/line", flush=True)
