import sys, time, argparse
sys.path.insert(0, ".")
import bench
a = argparse.Namespace(nz=40, hill=1000.0, adv="mpdata", mp="thompson")
t=time.time(); r = bench.cpu_baseline(a, 9); print(r, time.time()-t)
