#!/bin/bash
# Round-2 run on the MI355X box (launched through gpurun from the repo root).  Everything lands in gpurun_out/.
set -o pipefail
O=gpurun_out
mkdir -p $O
{ lscpu | head -20; nproc; uname -r; ulimit -n; } > $O/host_r02.txt 2>&1
python -c 'import __graft_entry__ as g; g.build(); g.smoke()' > $O/entry_r02.log 2>&1 || { echo "entry failed"; tail -20 $O/entry_r02.log; exit 1; }
echo "[box] entry ok"
python -m pytest tests -q -m gpu -x > $O/pytest_gpu_r02.log 2>&1; rc=$?
tail -3 $O/pytest_gpu_r02.log
[ $rc -eq 0 ] || exit $rc
echo "[box] gpu tier ok"
python bench.py > $O/bench_r02_n1.json 2> $O/bench_r02_n1.err || { echo "bench failed"; tail -20 $O/bench_r02_n1.err; exit 1; }
tail -6 $O/bench_r02_n1.err
echo "[box] bench ok"
python -m nuts333_amd.baseline --binary reference --reps 3 --out $O/baseline_r02_mi355xhost_reference.json > $O/bl_r02_ref.log 2>&1 || { echo "baseline failed"; tail $O/bl_r02_ref.log; exit 1; }
echo "[box] baseline ok"
