#!/usr/bin/env python3
"""Box experiment (round 3): which start-up step of a bench replica opens the GPUs?

libdrm prints ``/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory`` to stderr whenever a process opens an
amdgpu device on the MI355X box, which makes every step that initialises HIP visible.  A child process marks each
step on stderr; the output shows between which two marks the line appears.  Result, recorded in
``profiles/enum_r03_mi355xhost_step{1,2}.log``: only ``dist.barrier()`` does it (it asks for the current accelerator
even on a gloo group); ``*_VISIBLE_DEVICES=""`` does not prevent it.  bench.py therefore synchronises replicas with an
all-reduce of a CPU scalar (``bench.cpu_barrier``).
"""
import subprocess
import sys

STEPS = r'''
import os, sys
def mark(s): sys.stderr.write("MARK " + s + "\n"); sys.stderr.flush()
mark("start")
import torch
mark("after import torch")
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[1])
dist.init_process_group("gloo", rank=0, world_size=1)
mark("after init_process_group gloo")
t = torch.tensor([1.0], dtype=torch.float64); dist.all_reduce(t)
mark("after all_reduce (cpu tensor)")
box = [{"a": 1}]; dist.broadcast_object_list(box, src=0)
mark("after broadcast_object_list")
dist.monitored_barrier()
mark("after monitored_barrier")
dist.destroy_process_group()
mark("after destroy")
dist.init_process_group("gloo", rank=0, world_size=1)
dist.barrier()
mark("after barrier()")
dist.destroy_process_group()
'''

if __name__ == "__main__":
    p = subprocess.run([sys.executable, "-c", STEPS, "29655"], stderr=subprocess.PIPE, stdout=subprocess.PIPE, timeout=300)
    print("rc=" + str(p.returncode))
    print(p.stderr.decode(errors="replace")[-2500:])
