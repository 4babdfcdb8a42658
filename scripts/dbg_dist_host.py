"""Where does the host spend its time in the block-column driver at world size 1?"""
import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from tinygp_amd import kernels, synthetic
from tinygp_amd.distributed import BlockCyclicCholesky, HipBlockOps, MAIN
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
X, y = synthetic.make_inputs(n, 1, "float64")
kern = 1.5**2 * kernels.ExpSquared(2.5)
s = BlockCyclicCholesky(kern, X, np.full(n, 0.01), nb=1024, ops=HipBlockOps(0), dist=dist)
for rep in range(3):
    s.log_probability(y)
torch.cuda.synchronize()
ops = s.ops
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
t_all = time.perf_counter()
r = np.ascontiguousarray(y, dtype=np.float64)
t0 = time.perf_counter(); ops.assemble(s.prog); ops.begin(r); ops.first_panel(); tick("assemble+begin+first", t0)
t0 = time.perf_counter(); work = s._bcast_panel(0); tick("bcast", t0)
for k in range(s.nblk):
    t0 = time.perf_counter()
    with ops.stream(MAIN):
        work.wait()
    tick("wait", t0)
    t0 = time.perf_counter(); ops.after_recv(k); tick("after_recv", t0)
    if k + 1 < s.nblk:
        t0 = time.perf_counter(); work = s._bcast_panel(k + 1); tick("bcast", t0)
    t0 = time.perf_counter(); ops.fwd_step(k); tick("fwd_step", t0)
    t0 = time.perf_counter(); ops.rest(k); tick("rest", t0)
t_enq = time.perf_counter() - t_all
t0 = time.perf_counter(); ops.end(); tick("end (sync)", t0)
t_tot = time.perf_counter() - t_all
print(f"self_broadcast={s.self_broadcast}  host enqueue {t_enq*1e3:.2f} ms, total {t_tot*1e3:.2f} ms")
for k, v in T.items():
    print(f"  {k:24s} {v*1e3:8.2f} ms")
dist.destroy_process_group()
