#!/bin/bash
# Round-2 run on the MI355X box (launched through gpurun from the repo root).  Everything lands in gpurun_out/.
set -o pipefail
O=gpurun_out
mkdir -p $O
{ lscpu | grep -v Flags | head -24; nproc; uname -r; ulimit -n; } > $O/host_r02.txt 2>&1
python -c 'import __graft_entry__ as g; g.build(); g.smoke()' > $O/entry_r02.log 2>&1 || { echo "entry failed"; tail -20 $O/entry_r02.log; exit 1; }
echo "[box] entry ok"
python -m pytest tests -q -m gpu -x > $O/pytest_gpu_r02.log 2>&1; rc=$?
tail -3 $O/pytest_gpu_r02.log
[ $rc -eq 0 ] || exit $rc
echo "[box] gpu tier ok"
python bench.py > $O/bench_r02_n1.json 2> $O/bench_r02_n1.err || { echo "bench failed"; tail -20 $O/bench_r02_n1.err; exit 1; }
tail -6 $O/bench_r02_n1.err
echo "[box] bench ok"
python tools/probe_roofline.py --reps 3 --out $O/probe_r02_mi355xhost.json > $O/probe_r02.log 2>&1 || { echo "probe failed"; tail $O/probe_r02.log; exit 1; }
cat $O/probe_r02.log
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof_r02 -o bench -- python3 $OLDPWD/bench.py --steps 5 --warmup 1 > $OLDPWD/$O/rocprof_bench_r02.json 2> $OLDPWD/$O/rocprof_bench_r02.err ) || { echo "rocprof failed"; tail -5 $O/rocprof_bench_r02.err; exit 1; }
echo "[box] rocprof ok"; find $O/prof_r02 -name '*stats*' | head
for n in 2 4; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --steps 10 --warmup 2 > $O/bench_r02_replicas$n.json 2> $O/bench_r02_replicas$n.err || { echo "replicas $n failed"; tail -5 $O/bench_r02_replicas$n.err; exit 1; }
done
echo "[box] replicas ok"
python -m nuts333_amd.baseline --binary reference --reps 3 --out $O/baseline_r02_mi355xhost_reference.json > $O/bl_r02_ref.log 2>&1 || { echo "baseline failed"; tail $O/bl_r02_ref.log; exit 1; }
echo "[box] baseline reference ok"
python -m nuts333_amd.baseline --binary port --reps 3 --out $O/baseline_r02_mi355xhost_port.json > $O/bl_r02_port.log 2>&1 || { echo "baseline port failed"; tail $O/bl_r02_port.log; exit 1; }
echo "[box] baseline port ok"
