"""Cost of the multi-GPU slab schedule on ONE GPU: a single periodic rank exchanges its halos with
itself over RCCL (self seam), everything else as bench.py --gpus N does per rank."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from latticeboltzmann_amd.slabs import SlabSolver
from oracle import lbm_numpy as onp

def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    cols = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    halo = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    depth = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    torch.cuda.set_device(0)
    from latticeboltzmann_amd.slabs import init_process_group_rccl
    init_process_group_rccl(0, init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
    slab = SlabSolver(rows, cols, onp.OMEGA, 0, 1, boundary="periodic", halo=halo, device=0, dist=dist,
                      self_seam=True, tb_max_depth=depth)
    slab.set_surroundings(onp.ambient())
    slab.init_device_scene(100.0, (0.0, 0.1), amp=0.01, seed=0)
    slab.step(20); slab.sync()
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        slab.step(steps); slab.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"self-seam slab {rows}x{cols} halo={halo} depth={depth}: {dt/steps*1e3:.3f} ms/step  {rows*cols*steps/dt/1e9:.2f} GLUPS", flush=True)
    slab.close(); dist.destroy_process_group()
main()
