"""Does a workload run slower behind another one in the same process?  python scripts/bench_sequence_probe.py synth3,stock_held,sleep,stock_held"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
c = bench.Comm(None, 1, 0)
def show(name, **kw):
    t0=time.time(); r = bench.run_workload(name, c, 0, want_cpu=False, **kw); print(name, kw, "%.3e" % r["value"], round(r["roofline"]["kernel_ms_avg"],3), "wall %.1fs" % (time.time()-t0), flush=True)
order = sys.argv[1].split(",")
for n in order:
    if n == "sleep": time.sleep(15); print("slept", flush=True); continue
    show(n, warmup=1, ring_n=2)
