"""One-off: nb independent compress! chains -- qil_compress_batch, and the same chains on nb separate contexts driven by Python
threads -- against one chain alone.  gpurun -- python tools/_compress_concurrent.py [nb] [chi]"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qilaplace_jl_amd as qil
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = qil.default_context()
def sat(L, chi, base=2): return [int(min(base ** (i + 1), base ** (L - 1 - i), chi)) for i in range(L - 1)]
same = len(sys.argv) > 3 and sys.argv[3] == "same"          # identical chains: the lock-step groups never fall out of step
def make(i, c=None): return qil.SignalMPS.alloc(sat(24, chi), dtype=np.float64, ctx=c).fill_random(5 if same else 5 + i)
for rep in range(3):
    psi = make(0); ctx.synchronize(); t0 = time.perf_counter(); qil.compress(psi, maxdim=chi // 2, tol=1e-10); ctx.synchronize(); t1 = time.perf_counter() - t0
for rep in range(3):
    items = [make(i) for i in range(nb)]
    ctx.synchronize(); t0 = time.perf_counter(); qil.compress_batch(items, maxdim=chi // 2, tol=1e-10); ctx.synchronize(); tb = time.perf_counter() - t0
ctxs = [qil.Context(0) for _ in range(nb)]
def run(p): qil.compress(p, maxdim=chi // 2, tol=1e-10); p.ctx.synchronize()
for rep in range(3):
    its = [make(i, ctxs[i]) for i in range(nb)]
    for c in ctxs: c.synchronize()
    th = [threading.Thread(target=run, args=(p,)) for p in its]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    tp = time.perf_counter() - t0
print(f"compress f64 chi {chi}->{chi//2}, 24 sites: single {t1*1e3:.1f} ms, batch of {nb} {tb*1e3:.1f} ms = {tb/t1:.2f} x single; "
      f"python threads on {nb} contexts {tp*1e3:.1f} ms = {tp/t1:.2f} x; host cpus {os.cpu_count()} affinity {len(os.sched_getaffinity(0))}", flush=True)
