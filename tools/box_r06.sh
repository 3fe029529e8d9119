#!/bin/bash
# Round-6 run on the MI355X box (launched through gpurun from the repo root).  Everything lands in gpurun_out/.
#   tools/box_r06.sh tests   build + smoke + the gpu tier
#   tools/box_r06.sh bench   the driver's own command line; checks that the line survives the driver's ~8 KB tail whole
#   tools/box_r06.sh more    rocprof kernel trace of the same command; 2 replicas
set -o pipefail
# the snapshot must carry the prebuilt reference: if oracle/_ref/ AND its marker were lost together, fail instead of skipping
export NUTS_REQUIRE_REFERENCE=1
O=gpurun_out
mkdir -p $O
STAGE=${1:-tests}
if [ "$STAGE" = tests ]; then
{ lscpu | grep -v Flags | head -24; nproc; uname -r; ulimit -n; cat /proc/loadavg; cat /sys/fs/cgroup/cpu.max; } > $O/host_r06.txt 2>&1
python -c 'import __graft_entry__ as g; g.build(); g.smoke()' > $O/entry_r06.log 2>&1 || { echo "entry failed"; tail -20 $O/entry_r06.log; exit 1; }
echo "[box] entry ok"; tail -4 $O/entry_r06.log
python -m pytest tests -q -m gpu -x > $O/pytest_gpu_r06.log 2>&1; rc=$?
tail -3 $O/pytest_gpu_r06.log
[ $rc -eq 0 ] || { tail -60 $O/pytest_gpu_r06.log; exit $rc; }
echo "[box] gpu tier ok"
fi
if [ "$STAGE" = bench ]; then
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_r06_driverargs.json 2> $O/bench_r06_driverargs.err || { echo "bench failed rc=$?"; tail -20 $O/bench_r06_driverargs.err; exit 1; }
cp $O/bench_full_n1.json $O/bench_r06_driverargs_full.json
tail -9 $O/bench_r06_driverargs.err
# what the driver's record would keep: the last 8000 bytes of stdout + "---- stderr ----" + stderr.  The line must be whole in it.
python - <<'PY'
import json
out = open("gpurun_out/bench_r06_driverargs.json").read(); err = open("gpurun_out/bench_r06_driverargs.err").read()
tail = (out + "\n---- stderr ----\n" + err)[-8000:]
line = next(l for l in tail.splitlines() if l.startswith("{"))
j = json.loads(line)
print(f"[box] line {len(out.strip())} bytes, whole inside an 8000-byte tail; value {j['value']:,.0f} frac {j['roofline']['frac']} "
      f"loadavg {j['host']['loadavg_before_run']} restatement x{j['cpu_baseline_port']['ratio_to_timed_run']} "
      f"O0 x{(j.get('cpu_baseline_O0') or {}).get('ratio_to_timed_run')} {(j.get('cpu_baseline_O0') or {}).get('rate_all_reps')} "
      f"probe legs {j['roofline']['probe_legs']} "
      f"configs {[c['name'] for c in j['configs']]} exact {j['configs_all_exact']} warnings {j['warnings']}")
PY
[ $? -eq 0 ] || { echo "[box] the line is not whole in the 8000-byte tail (or is not JSON)"; exit 1; }
echo "[box] bench (driver args) ok"
fi
if [ "$STAGE" = more ]; then
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof_r06 -o bench -- python3 $OLDPWD/bench.py --steps 5 --warmup 1 > $OLDPWD/$O/rocprof_bench_r06.json 2> $OLDPWD/$O/rocprof_bench_r06.err ) || { echo "rocprof failed"; tail -5 $O/rocprof_bench_r06.err; exit 1; }
echo "[box] rocprof ok"; find $O/prof_r06 -name '*stats*' | head
for n in 2; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --steps 10 --warmup 2 > $O/bench_r06_replicas$n.json 2> $O/bench_r06_replicas$n.err || { echo "replicas $n failed"; tail -5 $O/bench_r06_replicas$n.err; exit 1; }
  cp $O/bench_full_n$n.json $O/bench_r06_replicas${n}_full.json
  echo "[box] replicas $n: amdgpu lines in stderr: $(grep -c amdgpu $O/bench_r06_replicas$n.err)"
  python -c "import json,sys; j=json.load(open('$O/bench_r06_replicas$n.json')); print('[box]', j['n_gpus'], 'replicas', j['value'], 'lines/s; receiver threads', j['host']['receiver_threads_per_replica'], 'warnings', j['warnings'])"
done
echo "[box] replicas ok"
fi
