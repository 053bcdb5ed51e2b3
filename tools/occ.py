import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import torch
from neural_astar import _native
lib = _native.load()
torch.zeros(1, device="cuda")
for (H, W) in [(32, 32), (64, 64), (16, 16), (64, 128), (20, 45)]:
    n = ctypes.c_int(0)
    print(H, W, "blocks/CU", lib.nastar_debug_occupancy(H, W, ctypes.byref(n)), "lds", n.value)
