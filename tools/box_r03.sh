#!/bin/bash
# Round-3 run on the MI355X box (launched through gpurun from the repo root).  Everything lands in gpurun_out/.
# gpurun allows 20 minutes per call: `tools/box_r03.sh tests` then `tools/box_r03.sh bench` then `tools/box_r03.sh more`.
set -o pipefail
O=gpurun_out
mkdir -p $O
STAGE=${1:-tests}
if [ "$STAGE" = tests ]; then
{ lscpu | grep -v Flags | head -24; nproc; uname -r; ulimit -n; cat /proc/loadavg; cat /sys/fs/cgroup/cpu.max; } > $O/host_r03.txt 2>&1
python -c 'import __graft_entry__ as g; g.build(); g.smoke()' > $O/entry_r03.log 2>&1 || { echo "entry failed"; tail -20 $O/entry_r03.log; exit 1; }
echo "[box] entry ok"; tail -4 $O/entry_r03.log
python -m pytest tests -q -m gpu -x > $O/pytest_gpu_r03.log 2>&1; rc=$?
tail -3 $O/pytest_gpu_r03.log
[ $rc -eq 0 ] || { tail -60 $O/pytest_gpu_r03.log; exit $rc; }
echo "[box] gpu tier ok"
fi
if [ "$STAGE" = bench ]; then
# the driver's own command line, three times: how far apart do whole lines sit on this allocation?
for k in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_r03_driverargs_run$k.json 2> $O/bench_r03_driverargs_run$k.err || { echo "bench $k failed rc=$?"; tail -20 $O/bench_r03_driverargs_run$k.err; exit 1; }
  tail -8 $O/bench_r03_driverargs_run$k.err
done
echo "[box] bench (driver args) ok"
python bench.py > $O/bench_r03_n1.json 2> $O/bench_r03_n1.err || { echo "bench failed"; tail -20 $O/bench_r03_n1.err; exit 1; }
echo "[box] bench default ok"
fi
if [ "$STAGE" = more ]; then
NUTS_BENCH_CPUS=first python bench.py > $O/bench_r03_n1_firstcpus.json 2> $O/bench_r03_n1_firstcpus.err || { echo "bench first failed"; tail -20 $O/bench_r03_n1_firstcpus.err; exit 1; }
echo "[box] bench with the old placement ok"
python tools/probe_roofline.py --reps 3 --out $O/probe_r03_mi355xhost.json > $O/probe_r03.log 2>&1 || { echo "probe failed"; tail $O/probe_r03.log; exit 1; }
cat $O/probe_r03.log
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof_r03 -o bench -- python3 $OLDPWD/bench.py --steps 5 --warmup 1 > $OLDPWD/$O/rocprof_bench_r03.json 2> $OLDPWD/$O/rocprof_bench_r03.err ) || { echo "rocprof failed"; tail -5 $O/rocprof_bench_r03.err; exit 1; }
echo "[box] rocprof ok"; find $O/prof_r03 -name '*stats*' | head
for n in 2 4; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --steps 10 --warmup 2 > $O/bench_r03_replicas$n.json 2> $O/bench_r03_replicas$n.err || { echo "replicas $n failed"; tail -5 $O/bench_r03_replicas$n.err; exit 1; }
  echo "[box] replicas $n: amdgpu lines in stderr: $(grep -c amdgpu $O/bench_r03_replicas$n.err)"
done
echo "[box] replicas ok"
fi
