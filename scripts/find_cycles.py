#!/usr/bin/env python3
"""Reference cycles a training step leaves behind (they keep their tensors alive until Python's cyclic collector runs):
a few steps with the collector off, then one collection with DEBUG_SAVEALL and a census of what it found.
python scripts/find_cycles.py <bench.py flags ...>"""
import collections
import gc
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import crosslayer_transcoder_amd as cta  # noqa: E402

args = bench.parse_args()
cta.set_precision(args.precision)
dev = torch.device("cuda:0")
if args.shard_rehearsal:                       # (the sharded schedule through RCCL on one rank, as bench.py sets it up)
    import torch.distributed as dist
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29523", rank=0, world_size=1, device_id=dev)
mod = bench.build(args, dev)
mod.attach_optimizers()
A = bench.synth_target_map(args.layers, args.d_acts, dev)
batches = [bench.synth_batch(args.batch, args.layers, args.d_acts, dev, seed=i, A=A) for i in range(4)]
for t in range(3):
    mod.training_step(batches[t % 4], t)
torch.cuda.synchronize()
gc.collect()
gc.disable()
a0 = torch.cuda.memory_allocated()
for t in range(3, 7):
    mod.training_step(batches[t % 4], t)
    torch.cuda.synchronize()
    print(f"step {t}: allocated {torch.cuda.memory_allocated() / 2**20:.1f} MiB (+{(torch.cuda.memory_allocated() - a0) / 2**20:.1f})")
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
print("collected", n, "objects; allocated after:", f"{torch.cuda.memory_allocated() / 2**20:.1f} MiB")
census = collections.Counter(type(o).__name__ for o in gc.garbage)
print(census.most_common(12))
for o in gc.garbage:
    if isinstance(o, types.FunctionType):
        cells = []
        for name, cell in zip(o.__code__.co_freevars, o.__closure__ or ()):
            try:
                v = cell.cell_contents
            except ValueError:
                continue
            if isinstance(v, torch.Tensor):
                cells.append(f"{name}:Tensor{tuple(v.shape)}")
            elif not isinstance(v, (int, float, bool, str, type(None))):
                cells.append(f"{name}:{type(v).__name__}")
        print("function", o.__qualname__, "at line", o.__code__.co_firstlineno, "closes over", cells)
    elif isinstance(o, torch.Tensor):
        print("tensor", tuple(o.shape), o.dtype)
    elif not isinstance(o, (types.CellType, tuple, dict, list)):
        print("object", type(o).__name__, [k for k in getattr(o, "__dict__", {})][:12] or getattr(type(o), "__slots__", ""))
