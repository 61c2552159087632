import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccv_amd import nnc
from ccv_amd.comm import ProcessComm
class Solo:
    def broadcast_object_list(self, objs, src=0): return None
L = nnc.load()
t0 = time.time(); comm = ProcessComm(L, Solo(), 0, 1); print("comm init %.2f s  env=%s" % (time.time() - t0, {k: v for k, v in os.environ.items() if k.startswith("NCCL") or k.startswith("RCCL")}), flush=True)
