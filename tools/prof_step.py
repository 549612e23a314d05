"""cProfile of one bench workload's step() on the host side: python tools/prof_step.py <workload> [rows]"""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rayforce_amd.engine import Engine
name = sys.argv[1]
rows = int(float(sys.argv[2])) if len(sys.argv) > 2 else bench.WORKLOADS[name]["rows"]
eng = Engine(0)
job = bench.Job(name, eng, None, rows, 0)
job.step(); job.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
job.step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
