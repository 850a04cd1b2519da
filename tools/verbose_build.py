import os, sys, time
sys.path.insert(0, "/root/repo")
import amg_amd as AMG
t0=time.time(); A = AMG.poisson((256,256,256)); ml = AMG.ruge_stuben(A); print("setup", time.time()-t0, flush=True)
t0=time.time(); dev = ml.device(); print("upload", time.time()-t0, flush=True)
