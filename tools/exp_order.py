"""Ad-hoc: does run order / TIME_WAIT / idle time / core group move the numbers on this host? (run on the MI355X box;
output kept in profiles/hostnoise_r02_mi355xhost.log)"""
import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
from nuts333_amd import workloads
from nuts333_amd.talker import REF_BINARY
t0=time.time()
def tw():
    for l in open('/proc/net/sockstat'):
        if l.startswith('TCP:'): return l.strip()
def c2(tag, n=3):
    r=[workloads.config2(lines=20000, warmup=1000, binary=REF_BINARY) for _ in range(n)]
    print(f"{time.time()-t0:6.1f}s {tag:28s} config2", [round(x['delivered_lines_per_s']/1e3) for x in r], [round(x['servers'][0]['cpu_us_per_written_line'],3) for x in r], tw(), flush=True)
def c4(tag, n=2):
    r=[workloads.config4(lines=1000, warmup=20, binary=REF_BINARY) for _ in range(n)]
    print(f"{time.time()-t0:6.1f}s {tag:28s} config4", [round(x['delivered_lines_per_s']/1e3) for x in r], [round(x['servers'][0]['cpu_us_per_written_line'],3) for x in r], tw(), flush=True)
c2("fresh")
c4("first 1000-client runs")
c2("after config4")
for leg in ("0 0","1 0","1 1"):
    subprocess.run([str(workloads.LOADGEN_BIN),"--probe-line","69","999","300",*leg.split(),"4","0","1,2,3,4"],stdout=subprocess.DEVNULL)
c2("after probes")
c4("config4 again")
time.sleep(30)
c2("after 30 s idle")
c4("after 30 s idle")
# a different talker core: rotate the cpu list so that the talker lands on cpu 8 and the clients on 9-12
os.sched_setaffinity(0, set(range(8, 16)))
c2("cpus 8-15")
c4("cpus 8-15")
